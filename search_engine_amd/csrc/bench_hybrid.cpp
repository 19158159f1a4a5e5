// Times the host mirror of the reference's search entry points end to end over the C ABI:
// execute_vector_search (search.rs:98-135) and execute_hybrid (hybrid.rs:138-199) with a
// stubbed keyword list, semanticRatio 0.5, facet filters of 1 / 10 / 50 % pass rate, on
// synthetic 768-d f32 vectors. Every call is synchronous from host buffers: it includes
// the candidate-mask upload over PCIe, the scan, the result download and the CPU-side
// ranking-rule / merge logic. Prints one JSON line per pass rate.
//
//   bench_hybrid [docs=1000000] [dim=768] [queries=200] [stores=1]
//
// stores > 1: documents with several vectors. Store j >= 1 of the embedder holds a vector for every
// 16^j-th document ("vectors packed in the first writers", typed_chunk.rs:597-607), and the host
// mirror's fill_buffer searches all stores in one vs_search_stores call (VS_MIRROR_STORE_LOOP=1 in
// the environment: reader by reader, as the reference's loop does). The bare vs_search / vs_search_mask
// columns stay single calls on store 0.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../include/vs_abi.h"
#include "host_engine.h"

using namespace search_engine;

static uint32_t hash32(uint32_t x) {  // the synthetic facet: hash32(docid) % 100 < pass rate
    x = (x ^ (x >> 16)) * 0x7FEB352Du;
    x = (x ^ (x >> 15)) * 0x846CA68Bu;
    return x ^ (x >> 16);
}

static void fill_rows(float* dst, uint64_t first, uint64_t n, uint32_t dim) {
    // cheap reproducible values in [-1, 1): the timing does not depend on the distribution
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t s = hash32((uint32_t)(first + i) * 2654435761u + 12345u) | 1u;
        for (uint32_t j = 0; j < dim; ++j) {
            s ^= s << 13;
            s ^= s >> 17;
            s ^= s << 5;
            dst[i * dim + j] = (float)(int32_t)s * (1.0f / 2147483648.0f);
        }
    }
}

#define CHECK(call)                                                        \
    do {                                                                   \
        const int32_t rc_ = (call);                                        \
        if (rc_ != VS_OK) {                                                \
            std::fprintf(stderr, "%s failed: %d\n", #call, (int)rc_);      \
            return 1;                                                      \
        }                                                                  \
    } while (0)

int main(int argc, char** argv) {
    const uint64_t docs = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1000000;
    const uint32_t dim = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 768;
    const int n_queries = argc > 3 ? std::atoi(argv[3]) : 200;
    const int n_stores = argc > 4 ? std::atoi(argv[4]) : 1;
    const uint8_t category = 1;
    if (n_stores < 1 || n_stores > 8) {
        std::fprintf(stderr, "stores must be 1..8\n");
        return 1;
    }

    vs_ctx* ctx = nullptr;
    CHECK(vs_init(0, &ctx));
    std::vector<vs_store*> stores((size_t)n_stores, nullptr);
    for (int j = 0; j < n_stores; ++j) {
        uint64_t every = 1;
        for (int i = 0; i < j; ++i) every *= 16;
        CHECK(vs_store_create(ctx, (uint16_t)((category << 8) | j), dim, VS_F32, VS_ANGULAR, 0, docs, &stores[j]));
        const uint64_t chunk = 1 << 16;
        const unsigned n_threads = 16;
        std::vector<float> rows(chunk * n_threads * dim);
        std::vector<uint32_t> ids(chunk * n_threads);
        const uint64_t items = (docs + every - 1) / every;  // item i is document i * every
        for (uint64_t lo = 0; lo < items; lo += chunk * n_threads) {
            const uint64_t n = std::min<uint64_t>(chunk * n_threads, items - lo);
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < n_threads; ++t) {
                const uint64_t a = std::min<uint64_t>(t * chunk, n), b = std::min<uint64_t>((t + 1) * chunk, n);
                if (a < b) pool.emplace_back(fill_rows, rows.data() + a * dim, lo + a + (uint64_t)j * 0x01000000u, b - a, dim);
            }
            for (auto& th : pool) th.join();
            for (uint64_t i = 0; i < n; ++i) ids[i] = (uint32_t)((lo + i) * every);
            CHECK(vs_store_upload(stores[j], ids.data(), rows.data(), n));
        }
        CHECK(vs_store_seal(stores[j]));
    }
    vs_store* store = stores[0];

    std::vector<DocumentId> all(docs);
    for (uint64_t i = 0; i < docs; ++i) all[i] = (DocumentId)i;
    GpuContext gctx(ctx, {{"default", category}}, DocSet::from_ids(all));

    std::vector<std::vector<float>> queries(64, std::vector<float>(dim));
    for (size_t q = 0; q < queries.size(); ++q) fill_rows(queries[q].data(), 0xABCD0000u + q, 1, dim);

    const uint64_t limit = 10, offset = 0;
    for (int pass : {50, 10, 1}) {
        std::vector<DocumentId> u;
        for (uint64_t i = 0; i < docs; ++i)
            if (hash32((uint32_t)i) % 100 < (uint32_t)pass) u.push_back((DocumentId)i);
        const DocSet universe = DocSet::from_ids(u);
        // keyword side: 20 documents of the universe with middling ranking-rule scores
        SearchResult keyword;
        for (size_t i = 0; i < 20 && i < u.size(); ++i) {
            const DocumentId d = u[i * (u.size() / 20)];
            keyword.documents_ids.push_back(d);
            keyword.candidates.insert(d);
            keyword.document_scores.push_back({ScoreDetails::ranked(ScoreDetails::Words, 2, 3),
                                               ScoreDetails::ranked(ScoreDetails::Typo, (uint32_t)(i % 3), 2)});
        }
        auto vector_search = [&](const std::vector<float>& v) {
            return execute_vector_search(gctx, v, universe, 0, limit + offset, "default",
                                         DistributionShift::create(0.5f, 0.15f));
        };
        double secs[5] = {0, 0, 0, 0, 0};
        uint64_t hits = 0;
        vs_mask* kept = nullptr;  // the same filter, uploaded once
        CHECK(vs_mask_create(store, universe.words().data(), universe.words().size(), &kept));
        // 0: vector search alone, 1: hybrid, 2: the bare vs_search call, 3: vs_search_mask with the kept filter
        // (0 and 1 hand over the SAME universe object every time: a host that keeps the bitmap of a facet
        // filter it has evaluated before), 4: hybrid with a universe that was never seen before (a filter
        // evaluated for this request: its cardinality is counted and its words are uploaded inside the call)
        for (int mode = 0; mode < 5; ++mode) {
            std::vector<DocSet> fresh;
            if (mode == 4)
                for (int i = -5; i < n_queries; ++i) fresh.push_back(DocSet::from_words(universe.words()));
            for (int i = -5; i < n_queries; ++i) {  // 5 untimed warm-up calls
                const auto t0 = std::chrono::steady_clock::now();
                const std::vector<float>& q = queries[(size_t)(i + 5) % queries.size()];
                if (mode == 4) {
                    HybridInput in;
                    in.query = std::string("query");
                    in.has_semantic = true;
                    in.vector = q;
                    in.offset = offset;
                    in.limit = limit;
                    in.keyword_search = [&]() { return keyword; };
                    in.vector_search = [&](const std::vector<float>& v) {
                        return execute_vector_search(gctx, v, std::move(fresh[(size_t)(i + 5)]), 0, limit + offset,
                                                     "default", DistributionShift::create(0.5f, 0.15f));
                    };
                    hits += execute_hybrid(in, 0.5f).first.documents_ids.size();
                } else if (mode == 0) {
                    hits += vector_search(q).documents_ids.size();
                } else if (mode == 2) {
                    uint32_t ids[16], n = 0;
                    float dist[16];
                    const std::vector<uint64_t>& w = universe.words();
                    CHECK(vs_search(store, q.data(), dim, (uint32_t)limit, w.data(), w.size(), ids, dist, &n));
                    hits += n;
                } else if (mode == 3) {
                    uint32_t ids[16], n = 0;
                    float dist[16];
                    CHECK(vs_search_mask(store, q.data(), dim, (uint32_t)limit, kept, ids, dist, &n));
                    hits += n;
                } else {
                    HybridInput in;
                    in.query = std::string("query");
                    in.has_semantic = true;
                    in.vector = q;
                    in.offset = offset;
                    in.limit = limit;
                    in.keyword_search = [&]() { return keyword; };
                    in.vector_search = vector_search;
                    hits += execute_hybrid(in, 0.5f).first.documents_ids.size();
                }
                if (i >= 0) secs[mode] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
        }
        std::printf("{\"workload\": \"hybrid, semanticRatio 0.5, %llu docs x %u f32, top-%llu, facet pass rate %d%%\", \"stores\": %d, \"store_loop\": \"%s\", "
                    "\"vector_search_qps\": %.1f, \"vector_search_ms\": %.4f, \"hybrid_qps\": %.1f, \"hybrid_ms\": %.4f, "
                    "\"bare_vs_search_ms\": %.4f, \"vs_search_mask_ms\": %.4f, \"hybrid_fresh_universe_ms\": %.4f, "
                    "\"hybrid_over_vs_search_mask\": %.3f, \"hybrid_fresh_over_bare_vs_search\": %.3f, "
                    "\"queries\": %d, \"universe\": %zu, \"hits\": %llu}\n",
                    (unsigned long long)docs, dim, (unsigned long long)limit, pass, n_stores,
                    std::getenv("VS_MIRROR_STORE_LOOP") ? "reader by reader" : "vs_search_stores", n_queries / secs[0],
                    1e3 * secs[0] / n_queries, n_queries / secs[1], 1e3 * secs[1] / n_queries, 1e3 * secs[2] / n_queries, 1e3 * secs[3] / n_queries, 1e3 * secs[4] / n_queries, secs[1] / secs[3],
                    secs[4] / secs[2], n_queries, u.size(),
                    (unsigned long long)hits);
        std::fflush(stdout);
        CHECK(vs_mask_destroy(kept));
    }
    for (vs_store* st : stores) CHECK(vs_store_destroy(st));
    CHECK(vs_shutdown(ctx));
    return 0;
}
