// Host side above the C ABI: a C++ mirror of the parts of the reference crate
// that sit on the vector-search path, with the reference's names, argument
// meaning and error behaviour. (The reference is Rust; no Rust toolchain exists
// where this is built, so the host logic a maintainer would keep in Rust is
// restated here in C++ and driven through the same C ABI the Rust crate would
// bind.) Everything in this file is CPU control flow; distances come only from
// vs_search through a `Reader`.
//
//   DistributionShift                 src/vector/mod.rs:317-330, 395-433
//   ScoreDetails, Rank, ScoreValue    src/score_details.rs
//   RankingRule, RankingRuleOutput    src/search/ranking/ranking_rule.rs:17-27
//   VectorSort                        src/search/ranking/vector_sort.rs
//   bucket_sort / recursive_sort      src/search/ranking/bucket_sort.rs:17-91
//   Context (vector methods)          src/search/context.rs:28,59,60
//   execute_vector_search             src/search/search.rs:98-135
//   hybrid merge                      src/search/hybrid.rs
//   embedder category ids             src/update/settings.rs:775-785
//   vector write path                 src/update/index_documents/typed_chunk.rs:562-710,
//                                     src/update/index_documents/mod.rs:544-566
#pragma once
#include <stdint.h>

#include <atomic>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

struct vs_ctx;
struct vs_store;

namespace search_engine {

using DocumentId = uint32_t;  // src/lib.rs:79

// ---- errors (src/error.rs) ---------------------------------------------------

struct Error : std::runtime_error {
    enum Kind {
        InvalidEmbedder,          // UserError::InvalidEmbedder (context.rs:183-186)
        InvalidVectorDimensions,  // UserError::InvalidVectorDimensions (error.rs:262-264)
        ArroyError,               // InternalError::ArroyError (error.rs:266-272)
        TooManyVectors,           // UserError::TooManyVectors (typed_chunk.rs:644-647)
        TooManyEmbedders          // UserError::TooManyEmbedders (update/settings.rs:783)
    };
    Kind kind;
    uint32_t expected = 0, found = 0;
    Error(Kind k, const std::string& msg, uint32_t e = 0, uint32_t f = 0)
        : std::runtime_error(msg), kind(k), expected(e), found(f) {}
};

// ---- docid sets ----------------------------------------------------------------

// The handful of RoaringBitmap operations the path uses. Like a roaring bitmap the set picks
// its representation: a sorted array while it is small (a one-document bucket, the documents
// already returned), a range [0, n) when it holds every docid below n (the documents of an index
// without holes: roaring's run container), and a dense bitmap otherwise -- bit (d & 63) of word
// d >> 6, the layout vs_search takes as its candidate mask. Operations cost the size of the
// smaller operand wherever a roaring bitmap's would; a range intersected with a bitmap IS that
// bitmap. Copies of a dense set share its words (and the cardinality once it has been counted)
// until one of them is modified. One DocSet object is for one thread at a time; copies of it may
// be used from different threads.
class DocSet {
   public:
    DocSet() = default;
    static DocSet from_ids(const std::vector<DocumentId>& ids);
    static DocSet from_words(std::vector<uint64_t> words);
    static DocSet from_range(uint64_t n);  // the docids [0, n)
    bool contains(DocumentId d) const;
    bool insert(DocumentId d);  // true if newly inserted
    bool remove(DocumentId d);
    uint64_t len() const;
    bool is_empty() const;
    DocSet operator&(const DocSet& o) const;
    // *this & o, and its cardinality
    DocSet intersect(const DocSet& o, uint64_t* len) const;
    DocSet operator|(const DocSet& o) const;
    DocSet& operator|=(const DocSet& o);
    DocSet& operator-=(const DocSet& o);
    std::vector<DocumentId> to_ids() const;  // ascending
    // dense words (trailing zero words may be missing); materialised on demand for an array or a range
    const std::vector<uint64_t>& words() const;
    // What a data-access layer derived from words() and wants to find again for as long as these
    // words live unchanged (the device copy of a candidate mask), one slot per owner. Copies that
    // share the words share the slots; a modification drops them.
    std::shared_ptr<void> attachment(const void* owner) const;
    void attach(const void* owner, std::shared_ptr<void> what) const;
    const char* representation() const;  // "array", "range" or "bitmap"

   private:
    struct Block {
        std::vector<uint64_t> w;
        std::atomic<int64_t> len{-1};  // bits set, -1 = not counted yet
        std::mutex mu;                 // guards `attached`
        std::vector<std::pair<const void*, std::shared_ptr<void>>> attached;
        Block() = default;
        explicit Block(std::vector<uint64_t> words) : w(std::move(words)) {}
    };
    enum Kind { Array, Range, Bitmap };
    static constexpr size_t SMALL_MAX = 4096;  // roaring's array-container limit
    void to_bitmap();                          // array or range -> bitmap
    std::vector<uint64_t>& own();              // the bitmap's words, unshared, attachments dropped
    const Block& word_block() const;           // the block words() hands out
    Kind kind_ = Array;
    std::vector<DocumentId> ids_;           // Array: ascending, unique
    uint64_t range_n_ = 0;                  // Range
    std::shared_ptr<Block> dense_;          // Bitmap: shared between copies
    mutable std::shared_ptr<Block> cache_;  // Array, Range: what words() last built
};

// ---- DistributionShift -----------------------------------------------------------

struct DistributionShift {
    float current_mean;
    float current_sigma;
    // `None` if sigma <= 0 (vector/mod.rs:397-403)
    static std::optional<DistributionShift> create(float mean, float sigma);
    float shift(float score) const;
};

// ---- score details ------------------------------------------------------------------

struct Rank {
    uint32_t rank;
    uint32_t max_rank;
    double local_score() const { return (double)rank / (double)max_rank; }
    static Rank merge(Rank outer, Rank inner);
    static double global_score(const std::vector<Rank>& details);
};

// serde_json::Value as far as Sort compares it
struct SortValue {
    enum Kind { Null, Number, String, Other } kind = Null;
    double number = 0.0;
    std::string string;
};

struct Sort {
    std::string field_name;
    bool ascending = true;
    SortValue value;
    // -1 / 0 / +1, nullopt where the reference returns None
    std::optional<int> partial_cmp(const Sort& other) const;
};

struct ScoreDetails {
    enum Kind { Words, Typo, Proximity, Exactness, Attribute, SortKind, Vector } kind = Vector;
    // Words{matching_words,max_matching_words}, Typo{typo_count,max_typo_count},
    // Proximity{proximity,max_proximity}, Exactness{exact_words,max_exact_words},
    // Attribute{attribute,max_attribute}
    uint32_t value = 0, max_value = 0;
    Sort sort;
    std::optional<float> similarity;  // Vector

    static ScoreDetails vector(std::optional<float> similarity);
    static ScoreDetails ranked(Kind kind, uint32_t value, uint32_t max_value);
    static ScoreDetails sorted(Sort sort);
    std::optional<Rank> rank() const;
};

struct ScoreValue {
    bool is_score = true;
    double score = 0.0;
    const Sort* sort = nullptr;
};

std::vector<ScoreValue> score_values(const std::vector<ScoreDetails>& details);
double global_score(const std::vector<ScoreDetails>& details);
// serde_json rendering of ScoreDetails::to_json_map, keys in insertion order
std::string to_json_map(const std::vector<ScoreDetails>& details);

// ---- ranking rules ----------------------------------------------------------------------

struct RankingRuleOutput {
    ScoreDetails score;
    DocSet candidates;
};

struct RankingRule {
    virtual ~RankingRule() = default;
    virtual void start_iteration(DocSet candidates) = 0;
    virtual std::optional<RankingRuleOutput> next_bucket() = 0;
};

struct BucketSortOutput {
    DocSet candidates;
    std::vector<DocumentId> docids;
    std::vector<std::vector<ScoreDetails>> scores;
};

BucketSortOutput bucket_sort(std::vector<std::unique_ptr<RankingRule>>& ranking_rules, uint64_t limit,
                             uint64_t skip, DocSet candidates);

// ---- the data-access seam --------------------------------------------------------------------

// arroy::Reader<Angular> as VectorSort uses it.
struct Reader {
    virtual ~Reader() = default;
    // nns_by_vector(txn, target, count, None, Some(candidates)): (item, distance)
    // ascending by (distance, item). Throws Error.
    virtual std::vector<std::pair<DocumentId, float>> nns_by_vector(const std::vector<float>& target,
                                                                    size_t count,
                                                                    const DocSet& candidates) = 0;
    // The same for several targets that share `candidates`; result i belongs to targets[i].
    // No counterpart in arroy: the default is the loop a caller would write, a reader that can
    // do better (one pass over the store for all targets) overrides it.
    virtual std::vector<std::vector<std::pair<DocumentId, float>>> nns_by_vectors(
        const std::vector<std::vector<float>>& targets, size_t count, const DocSet& candidates);
    // ... and with a candidate set PER target (several requests, each with the universe of its own
    // filter, src/search/search.rs:222-228): result i = nns_by_vector(targets[i], count, *candidates[i]).
    virtual std::vector<std::vector<std::pair<DocumentId, float>>> nns_by_vectors_each(
        const std::vector<std::vector<float>>& targets, size_t count, const std::vector<const DocSet*>& candidates);
};

// The methods of `trait Context` the vector path calls.
struct Context {
    virtual ~Context() = default;
    virtual DocSet all_docids() = 0;                                         // context.rs:28
    virtual uint8_t embedder_category_id(const std::string& name) = 0;       // context.rs:59
    // nullptr plays arroy::Error::MissingMetadata ("store k does not exist")
    virtual std::unique_ptr<Reader> vector_reader(uint16_t index) = 0;       // context.rs:60
    // The whole store loop of VectorSort::fill_buffer (vector_sort.rs:55-78) as one call: result k =
    // nns_by_vector(target, count, candidates) of the reader (embedder_index << 8) | k, for k = 0.. up
    // to the first missing store. No counterpart in the reference's trait: a context that has no such
    // call returns nullopt and the caller opens the readers one by one.
    virtual std::optional<std::vector<std::vector<std::pair<DocumentId, float>>>> nns_all_stores(
        uint8_t /*embedder_index*/, const std::vector<float>& /*target*/, size_t /*count*/,
        const DocSet& /*candidates*/) {
        return std::nullopt;
    }
};

// Context over libvs_amd.so: vector_reader = vs_store_open, nns_by_vector = vs_search.
class GpuContext : public Context {
   public:
    GpuContext(vs_ctx* ctx, std::map<std::string, uint8_t> embedder_category_ids, DocSet documents_ids)
        : ctx_(ctx), ids_(std::move(embedder_category_ids)), docs_(std::move(documents_ids)) {}
    DocSet all_docids() override { return docs_; }
    uint8_t embedder_category_id(const std::string& name) override;
    std::unique_ptr<Reader> vector_reader(uint16_t index) override;
    // vs_search_stores: the filter crosses to the device once, one wait for all stores
    std::optional<std::vector<std::vector<std::pair<DocumentId, float>>>> nns_all_stores(
        uint8_t embedder_index, const std::vector<float>& target, size_t count, const DocSet& candidates) override;

   private:
    vs_ctx* ctx_;
    std::map<std::string, uint8_t> ids_;
    DocSet docs_;
};

// ---- VectorSort -------------------------------------------------------------------------------------

class VectorSort : public RankingRule {
   public:
    VectorSort(Context& ctx, std::vector<float> target, DocSet vector_candidates, size_t limit,
               const std::string& embedder_name, std::optional<DistributionShift> distribution);
    void start_iteration(DocSet candidates) override;
    std::optional<RankingRuleOutput> next_bucket() override;
    // Hands the rule the per-store results of its FIRST fill_buffer (computed elsewhere, for
    // vector_candidates & the candidates it will be started with), store k at index k.
    void preload(std::vector<std::vector<std::pair<DocumentId, float>>> per_store_results);
    // every store (embedder_index << 8) | k of the embedder, k = 0.. until one is missing
    static std::vector<std::unique_ptr<Reader>> open_readers(Context& ctx, uint8_t embedder_index);

   private:
    void fill_buffer();
    void set_cache(std::vector<std::vector<std::pair<DocumentId, float>>> per_store_results);
    std::optional<std::vector<std::vector<std::pair<DocumentId, float>>>> preloaded_;
    Context& ctx_;
    std::vector<float> target_;
    DocSet vector_candidates_;
    // The reference keeps `candidates`, removes every emitted document from it and recomputes
    // vector_candidates & candidates in every next_bucket (vector_sort.rs:100,118). Only that
    // intersection and whether either set is used up are ever looked at, so what is kept is:
    // candidates_ and live_ as of the last (re)fill -- untouched in between, so that they can share
    // their words with the caller's universe and keep its device copy -- and the documents emitted
    // since. The two cardinalities are counted when the cache runs dry, not before.
    DocSet candidates_, live_, emitted_;
    uint64_t emitted_from_live_ = 0;  // |emitted_|: how far live_ and candidates_ have been used up
    bool finished_ = false;
    std::vector<std::pair<DocumentId, float>> cached_sorted_docids_;
    size_t cursor_ = 0;
    size_t limit_;
    std::optional<DistributionShift> distribution_shift_;
    uint8_t embedder_index_;
};

// ---- search ----------------------------------------------------------------------------------------------

struct SearchResult {
    DocSet candidates;
    std::vector<DocumentId> documents_ids;
    std::vector<std::vector<ScoreDetails>> document_scores;
};

// Rules that follow VectorSort (the index's criteria on the placeholder graph in
// the reference, ranking_rule.rs:98-130); built by the caller.
using ExtraRules = std::function<std::vector<std::unique_ptr<RankingRule>>()>;

SearchResult execute_vector_search(Context& ctx, const std::vector<float>& vector, DocSet universe,
                                   uint64_t skip, uint64_t limit, const std::string& embedder_name,
                                   std::optional<DistributionShift> distribution,
                                   const ExtraRules& extra_rules = nullptr);

// execute_vector_search for several vectors under one universe: result i is what
// execute_vector_search(ctx, vectors[i], universe, ...) returns. The reference has no
// multi-query Search (src/search/mod.rs:132-179 takes one vector); this is the entry point
// such a Search would call. Each store is asked once for all vectors (Reader::nns_by_vectors).
std::vector<SearchResult> execute_vector_search_batch(Context& ctx, const std::vector<std::vector<float>>& vectors,
                                                      DocSet universe, uint64_t skip, uint64_t limit,
                                                      const std::string& embedder_name,
                                                      std::optional<DistributionShift> distribution,
                                                      const ExtraRules& extra_rules = nullptr);
// ... for several REQUESTS: vector i is searched under universes[i] (one per vector). Each store is
// still asked once (Reader::nns_by_vectors_each); requests that hand over the same universe object
// share passes over the rows.
std::vector<SearchResult> execute_vector_search_batch(Context& ctx, const std::vector<std::vector<float>>& vectors,
                                                      const std::vector<DocSet>& universes, uint64_t skip,
                                                      uint64_t limit, const std::string& embedder_name,
                                                      std::optional<DistributionShift> distribution,
                                                      const ExtraRules& extra_rules = nullptr);

// ---- hybrid ------------------------------------------------------------------------------------------------

using ScoreWithRatio = std::pair<std::vector<ScoreDetails>, float>;
// -1 / 0 / +1 = Less / Equal / Greater (hybrid.rs:19-56)
int compare_scores(const ScoreWithRatio& left, const ScoreWithRatio& right);
bool results_good_enough(const SearchResult& keyword_results, float semantic_ratio, uint64_t limit,
                         uint64_t offset);
// ScoreWithRatioResult::merge (hybrid.rs:74-135): (merged, semantic_hit_count)
std::pair<SearchResult, uint32_t> merge(const SearchResult& vector_results, float vector_ratio,
                                        const SearchResult& keyword_results, float keyword_ratio,
                                        uint64_t from, uint64_t length);

struct HybridInput {
    std::optional<std::string> query;
    bool has_semantic = false;
    std::optional<std::vector<float>> vector;
    uint64_t offset = 0, limit = 20;
    // search.execute() without / with the semantic part, both with offset 0 and
    // limit = limit + offset (hybrid.rs:140-156, 186-190)
    std::function<SearchResult()> keyword_search;
    std::function<SearchResult(const std::vector<float>&)> vector_search;
    // embedder.embed_one(query): nullopt = embedding failed (hybrid.rs:176-183)
    std::function<std::optional<std::vector<float>>(const std::string&)> embed_one;
};
// Search::execute_hybrid (hybrid.rs:138-199)
std::pair<SearchResult, std::optional<uint32_t>> execute_hybrid(const HybridInput& in,
                                                                float semantic_ratio);

// ---- write path --------------------------------------------------------------------------------------------

// The ids `update_embedding_configs` writes into `embedder_category_id` (update/settings.rs:775-785):
// the database is cleared and the embedders that remain get 0, 1, 2, ... in BTreeMap (byte-wise
// name) order. More than 256 embedders: Error::TooManyEmbedders.
std::map<std::string, uint8_t> assign_embedder_category_ids(const std::vector<std::string>& embedder_names);

// arroy::Writer<Angular> as the indexer uses it (typed_chunk.rs:601-606,618,650,668,673,691,702;
// index_documents/mod.rs:558-562).
struct Writer {
    virtual ~Writer() = default;
    virtual bool del_item(DocumentId docid) = 0;  // true if the item existed
    virtual void add_item(DocumentId docid, const std::vector<float>& vector) = 0;
    virtual std::optional<std::vector<float>> item_vector(DocumentId docid) = 0;
    virtual bool contains_item(DocumentId docid) = 0;
    virtual bool is_empty() = 0;
    virtual void build() = 0;
    // Called by the build loop for the first EMPTY writer, which the reference leaves unbuilt
    // (index_documents/mod.rs:559-561), and for the writers behind it for as long as one answers true
    // (a document's vectors are packed in the first writers, so those are empty as well, and a
    // document that lost its last three vectors empties three writers in one run). No counterpart
    // in arroy. A writer over a store that would otherwise keep answering with the items of its
    // last build publishes its emptiness here; the answer is whether such a store exists at all.
    virtual bool publish_if_emptied() { return false; }
};

// The 256 writers `(embedder_index << 8) | k` of one embedder.
struct Writers {
    virtual ~Writers() = default;
    virtual Writer& writer(uint8_t k) = 0;
};

// Writers over libvs_amd.so: add_item = vs_store_upload, del_item = vs_store_remove, build =
// vs_store_seal. A store is created (room for `capacity` docids) when its first item arrives.
class GpuWriters : public Writers {
   public:
    GpuWriters(vs_ctx* ctx, uint8_t embedder_index, uint32_t dimension, uint64_t capacity);
    ~GpuWriters() override;
    Writer& writer(uint8_t k) override;

   private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

// TypedChunk::VectorPoints after its chunks were merged (typed_chunk.rs:82-88, 566-589).
struct VectorPoints {
    std::vector<DocumentId> remove_vectors;
    // generated embeddings: every vector of a document (Embeddings<f32>)
    std::vector<std::pair<DocumentId, std::vector<std::vector<float>>>> embeddings;
    // user-provided vectors as a diff, one entry per (document, vector)
    struct ManualDiff {
        DocumentId docid;
        std::optional<std::vector<float>> deletion, addition;
    };
    std::vector<ManualDiff> manual_vectors;
};
// The VectorPoints arm of write_typed_chunk_into_index (typed_chunk.rs:562-710): removals,
// generated embeddings, then the manual diff — all three keep the documents' vectors packed in
// the first writers.
void write_vector_points(Writers& writers, const VectorPoints& points);
// The build loop that ends an indexing run (index_documents/mod.rs:554-563): writers 0.. are built
// until the first empty one. Returns the number of writers built.
uint32_t build_vector_stores(Writers& writers);

}  // namespace search_engine
