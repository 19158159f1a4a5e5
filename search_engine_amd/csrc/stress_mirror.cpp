// Searches through the host mirror from several threads while an indexing thread keeps adding
// documents with ever larger docids, so that the stores grow (vs_store_reserve) and are sealed again
// and again under the searchers. All searchers hand over the SAME universe object: its device copy is
// shared between them and must be replaced, not pulled away, when a store outgrows it. In the
// reference this is readers under RoTxns beside the one indexing RwTxn
// (src/search/ranking/vector_sort.rs:55-78, src/update/index_documents/typed_chunk.rs:562-710).
//
//   stress_mirror [threads=6] [rounds=40]      prints "ok ..." and exits 0, or says what broke
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <thread>
#include <vector>

#include "../../include/vs_abi.h"
#include "host_engine.h"

using namespace search_engine;

static void fill(std::vector<float>& v, uint32_t seed) {
    uint32_t s = seed * 2654435761u | 1u;
    for (float& x : v) {
        s ^= s << 13;
        s ^= s >> 17;
        s ^= s << 5;
        x = (float)(int32_t)s * (1.0f / 2147483648.0f);
    }
}

int main(int argc, char** argv) {
    const int n_threads = argc > 1 ? std::atoi(argv[1]) : 6;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 40;
    const uint32_t dim = 32, per_round = 700;
    const uint8_t category = 41;
    const DocumentId last = (DocumentId)(1000 + rounds * per_round);  // the planted document, added in the last round
    vs_ctx* ctx = nullptr;
    if (vs_init(0, &ctx) != VS_OK) return std::fprintf(stderr, "vs_init failed\n"), 1;
    int rc = 0;
    {
        GpuWriters writers(ctx, category, dim, 1024);
        auto add_docs = [&](DocumentId from, DocumentId to, const std::vector<float>* planted) {
            VectorPoints p;
            for (DocumentId d = from; d < to; ++d) {
                std::vector<std::vector<float>> vecs(d % 4 == 0 ? 2 : 1, std::vector<float>(dim));
                for (size_t j = 0; j < vecs.size(); ++j) fill(vecs[j], d * 8 + (uint32_t)j);
                p.embeddings.emplace_back(d, std::move(vecs));
            }
            if (planted) p.embeddings.emplace_back(last, std::vector<std::vector<float>>{*planted});
            write_vector_points(writers, p);
            build_vector_stores(writers);
        };
        add_docs(0, 1000, nullptr);
        GpuContext gctx(ctx, {{"default", category}}, DocSet::from_range((uint64_t)last + 1));
        // two docids in three pass the filter; docids that do not exist yet are in it already
        std::vector<DocumentId> u;
        for (DocumentId d = 0; d <= last; ++d)
            if (d % 3 != 1 || d == last) u.push_back(d);
        const DocSet universe = DocSet::from_ids(u);
        std::vector<float> query(dim);
        fill(query, 0xBEEF);
        std::atomic<DocumentId> written{1000};  // docids below this may be answered
        std::atomic<bool> stop{false};
        std::atomic<uint64_t> searches{0};
        std::atomic<int> failures{0};
        auto check = [&](const SearchResult& r, DocumentId bound, const char* who) {
            std::set<DocumentId> seen;
            bool ok = r.documents_ids.size() <= 10 && r.documents_ids.size() == r.document_scores.size();
            for (DocumentId d : r.documents_ids) ok = ok && d < bound && (d % 3 != 1 || d == last) && seen.insert(d).second;
            if (!ok && failures.fetch_add(1) < 5) std::fprintf(stderr, "%s: malformed result\n", who);
            return ok;
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t)
            pool.emplace_back([&, t] {
                std::vector<float> q(dim);
                try {
                    for (uint32_t i = 0; !stop.load(); ++i) {
                        fill(q, 0x1000u * (uint32_t)t + i);
                        // (the bound is read AFTER the search: everything the search may have seen is below it)
                        const SearchResult r = execute_vector_search(gctx, i % 2 ? q : query, universe, 0, 10, "default",
                                                                     std::nullopt);
                        check(r, (DocumentId)(written.load() + per_round + 1) > last ? last + 1 : written.load() + per_round + 1, "searcher");
                        searches.fetch_add(1);
                    }
                } catch (const std::exception& e) {
                    failures.fetch_add(1);
                    std::fprintf(stderr, "searcher %d: %s\n", t, e.what());
                }
            });
        try {
            for (int r = 0; r < rounds; ++r) {
                const DocumentId from = (DocumentId)(1000 + r * per_round);
                add_docs(from, from + per_round, r + 1 == rounds ? &query : nullptr);
                written.store(from + per_round);
            }
        } catch (const std::exception& e) {
            failures.fetch_add(1);
            std::fprintf(stderr, "indexer: %s\n", e.what());
        }
        stop.store(true);
        for (auto& th : pool) th.join();
        // after the last build: the planted document (the query itself) is the first hit, and two
        // searches agree
        const SearchResult a = execute_vector_search(gctx, query, universe, 0, 10, "default", std::nullopt);
        const SearchResult b = execute_vector_search(gctx, query, universe, 0, 10, "default", std::nullopt);
        if (!check(a, last + 1, "final") || a.documents_ids.empty() || a.documents_ids[0] != last ||
            a.documents_ids != b.documents_ids) {
            failures.fetch_add(1);
            std::fprintf(stderr, "final search: first hit %u, wanted %u\n",
                         a.documents_ids.empty() ? 0u : (unsigned)a.documents_ids[0], (unsigned)last);
        }
        rc = failures.load() ? 1 : 0;
        std::printf("%s: %d searching threads, %llu searches beside %d indexing rounds up to docid %u, %d failures\n",
                    rc ? "FAILED" : "ok", n_threads, (unsigned long long)searches.load(), rounds, (unsigned)last,
                    failures.load());
    }
    for (int j = 0; j < 4; ++j) {
        vs_store* s = nullptr;
        if (vs_store_open(ctx, (uint16_t)((category << 8) | j), &s) == VS_OK) vs_store_destroy(s);
    }
    vs_shutdown(ctx);
    return rc;
}
