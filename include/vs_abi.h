/*
 * vs_abi.h — C ABI of the MI355X vector-scan library (libvs_amd.so).
 *
 * This is the drop-in boundary for the filtered nearest-neighbour scan of the
 * reference engine. The reference has no FFI of its own; the narrowest seam is
 *   trait Context::vector_reader / embedder_category_id   src/search/context.rs:59-60
 * consumed only by
 *   VectorSort::fill_buffer                               src/search/ranking/vector_sort.rs:50-82
 * which calls arroy 0.2.0 `Reader<Angular>::nns_by_vector(rtxn, target, limit,
 * None, Some(&candidates))` (vector_sort.rs:73-77). Every entry point below
 * names the reference call it replaces.
 *
 * Conventions
 *  - every function returns an int32_t status (VS_OK == 0); nothing throws or
 *    aborts across the boundary;
 *  - the caller owns all host buffers it passes in; the library owns all device
 *    memory and the opaque handles;
 *  - document ids are the engine's internal `DocumentId = u32` (src/lib.rs:79);
 *  - results are ordered ascending by (distance, docid), at most `k` of them,
 *    exactly as arroy returns them from nns_by_vector;
 *  - `vs_search*` are thread-safe and re-entrant on a sealed store, also while
 *    the writer works; upload / remove / reserve / seal are serialised against
 *    each other by the library (the engine has one LMDB writer anyway);
 *  - a search sees the items AND the rows of the last `vs_store_seal` that
 *    completed before it started (the RoTxn snapshot rule): a row uploaded for a
 *    docid that searches can still see is parked beside the slab and moved in by
 *    the seal, which for that move alone waits for the searches in flight and
 *    holds new ones back. Device-API searches (vs_search_device and friends) are
 *    covered through an event on the stream they were given;
 *    vs_search_device_noalloc is not (see there).
 *
 * There is no CPU fallback: without a usable HIP device `vs_init` fails with
 * VS_E_DEVICE and no other entry point can be reached.
 */
#ifndef VS_ABI_H
#define VS_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vs_ctx vs_ctx;     /* one per process and GPU */
typedef struct vs_store vs_store; /* one arroy "index" = (embedder_id << 8) | k */

/* Status codes. Mapping onto the reference's errors (src/error.rs:257-274):
 *   VS_E_DIM_MISMATCH -> arroy::Error::InvalidVecDimension -> UserError::InvalidVectorDimensions
 *   VS_E_NO_STORE     -> arroy::Error::MissingMetadata (loop terminator, vector_sort.rs:60-63)
 *   everything else   -> InternalError::ArroyError */
enum {
    VS_OK = 0,
    VS_E_INVALID_ARG = 1,
    VS_E_DIM_MISMATCH = 2,
    VS_E_NO_STORE = 3,
    VS_E_DEVICE = 4,
    VS_E_NOT_SEALED = 5,
    VS_E_OOM = 6,
    VS_E_CAPACITY = 7,
    VS_E_FORMAT = 8
};

/* Element type of a store's rows in HBM. */
enum { VS_F32 = 0, VS_BF16 = 1, VS_BIN1 = 2 };
/* Distance. VS_ANGULAR is arroy::distances::Angular (src/index.rs:155), the only
 * metric of the reference. The others have no counterpart there:
 *   VS_HAMMING    binary-quantised stores (VS_BIN1): popcount(row XOR query);
 *   VS_EUCLIDEAN  f32/bf16 stores, as arroy's Euclidean: results ordered by the
 *                 squared distance, reported distance = its square root;
 *   VS_DOT        f32/bf16 stores, as arroy's DotProduct on a built index: results
 *                 ordered by descending <row, query>, reported "distance" = that
 *                 product. Ties break by ascending docid under every metric. */
enum { VS_ANGULAR = 0, VS_HAMMING = 1, VS_EUCLIDEAN = 2, VS_DOT = 3 };

/* Sentinel key that pads device-side key lists (see vs_search_device). */
#define VS_KEY_NONE 0xFFFFFFFFFFFFFFFFull
/* Largest `k` any search accepts. */
#define VS_MAX_K 1024u
/* Largest `k` of the batched matrix-core path (vs_search_batch_device). */
#define VS_BATCH_MAX_K 32u
/* Queries one batched launch takes. */
#define VS_BATCH_MAX_QUERIES 256u

/* ---- context ------------------------------------------------------------ */

/* Binds the calling process to GPU `device_ordinal`. Replaces nothing in the
 * reference (arroy needs no set-up); corresponds to opening the LMDB env. */
int32_t vs_init(int32_t device_ordinal, vs_ctx** out);
int32_t vs_shutdown(vs_ctx* ctx);

/* Copies the calling thread's last error message (NUL terminated) into buf. */
int32_t vs_last_error(vs_ctx* ctx, char* buf, size_t cap);
/* After VS_E_DIM_MISMATCH: the two fields of arroy::Error::InvalidVecDimension
 * {expected, received} (src/error.rs:262-264). */
int32_t vs_last_dim_mismatch(vs_ctx* ctx, uint32_t* expected, uint32_t* received);

/* Scan schedule knobs, for measurement only (no reference counterpart; results
 * do not depend on them). blocks: grid size, 0 = default; nt: non-zero =
 * non-temporal row loads (the default), 0 = plain loads. Also read once from
 * the environment (VS_SCAN_BLOCKS, VS_SCAN_NT) by vs_init. */
int32_t vs_set_scan_tuning(vs_ctx* ctx, int32_t blocks, int32_t nt);

/* ---- stores ------------------------------------------------------------- */

/* Creates the store `store_index` covering docids [docid_base, docid_base +
 * capacity). Replaces arroy::Writer::new(db, index, dimensions)
 * (src/update/index_documents/typed_chunk.rs:601-606). docid_base must be a
 * multiple of 64. */
int32_t vs_store_create(vs_ctx* ctx, uint16_t store_index, uint32_t dim, int32_t dtype,
                        int32_t metric, uint32_t docid_base, uint64_t capacity, vs_store** out);

/* Grows a store to `capacity` slots (no-op when it is not larger); contents, the
 * sealed state and the dirty state are kept. milli hands out ever larger internal
 * docids as an index grows (src/update/available_documents_ids.rs); the indexer
 * calls this before `add_item`s beyond the current capacity
 * (src/update/index_documents/typed_chunk.rs:650,691). Searches may run
 * meanwhile (they wait while the buffers are swapped); masks made for the store
 * before the call are refused afterwards and must be created again, and a
 * device-API caller must size its mask words for the new capacity. */
int32_t vs_store_reserve(vs_store* s, uint64_t capacity);

/* Looks a store up by index. Replaces Context::vector_reader(index) =
 * arroy::Reader::open (src/search/context.rs:188-190): a store that does not
 * exist or was never sealed yields VS_E_NO_STORE, as Reader::open yields
 * MissingMetadata for a store that was never built. */
int32_t vs_store_open(vs_ctx* ctx, uint16_t store_index, vs_store** out);

/* Adds or overwrites `n` items. `rows` is host memory, row-major f32 [n][dim];
 * the library converts to the store's element type (bf16: round to nearest
 * even; bin1: bit = (x > 0), bit i of a row in word i/64 at position i%64).
 * Of several rows for one docid the last wins, as with repeated add_item calls.
 * Replaces arroy::Writer::add_item (typed_chunk.rs:650,691). */
int32_t vs_store_upload(vs_store* s, const uint32_t* docids, const float* rows, uint64_t n);

/* Adds `n` items with consecutive docids first_docid.. from DEVICE memory
 * already in the store's native element type and row layout
 * (vs_store_row_bytes per row, densely packed). */
int32_t vs_store_upload_device(vs_store* s, uint32_t first_docid, const void* device_rows,
                               uint64_t n, void* hip_stream);

/* Replaces arroy::Writer::del_item (typed_chunk.rs:618,668,701). Unknown ids are ignored. */
int32_t vs_store_remove(vs_store* s, const uint32_t* docids, uint64_t n);

/* Makes every upload/remove since the previous seal visible to searches and
 * precomputes the per-row norms (arroy stores ‖v‖ in each leaf header at
 * add_item). Replaces arroy::Writer::build (src/update/index_documents/mod.rs:562). */
int32_t vs_store_seal(vs_store* s);

int32_t vs_store_destroy(vs_store* s);

int32_t vs_store_len(vs_store* s, uint64_t* n_items);      /* arroy Reader::n_items */
int32_t vs_store_dim(vs_store* s, uint32_t* dim);          /* arroy Reader::dimensions */
int32_t vs_store_row_bytes(vs_store* s, uint64_t* bytes);  /* bytes of one row in HBM */
/* arroy Writer::contains_item (typed_chunk.rs:673) */
int32_t vs_store_contains(vs_store* s, uint32_t docid, int32_t* out);
/* arroy Writer::item_vector (typed_chunk.rs:702): copies the row back as f32 [dim]
 * (bf16 widened; bin1 as 0.0 / 1.0). VS_E_INVALID_ARG if the item is absent. */
int32_t vs_store_item_vector(vs_store* s, uint32_t docid, float* out);

/* arroy Writer::is_empty / the item count of the WRITER's view (uploads and
 * removals since the last seal included), which the build loop tests before it
 * builds a store (src/update/index_documents/mod.rs:559-561). */
int32_t vs_store_writer_len(vs_store* s, uint64_t* n_items);

/* ---- filling stores from the reference's vector database -------------------- */

/* Takes `n` raw (key, value) pairs of the reference's `vector-arroy` LMDB
 * database (src/index.rs:152-155,208) exactly as a cursor yields them, and
 * uploads every ITEM LEAF into the store whose index is the key's
 * `(embedder_id << 8) | k` (stores must have been created; the write path's
 * "vectors packed in the first writers" layout, typed_chunk.rs:597-607, is
 * thereby taken over as it is on disk). Tree nodes, metadata and keys of a mode
 * this layout does not know are skipped and counted in *n_other_nodes: the
 * exhaustive scan needs no forest. Leaves of an index without a store are counted
 * in *n_unrouted. A leaf whose dimension differs from its store's is
 * VS_E_DIM_MISMATCH, anything malformed VS_E_FORMAT; every pair is checked before
 * the first row is uploaded, so a failed call leaves the stores as they were.
 * Follow with vs_store_seal on the stores touched (arroy: build). arroy 0.2.0's
 * node layout is restated from the published crate and unverified here (the
 * crate is not in the reference tree):
 *   key   = index u16 BE | mode u8 (0 item, 1 tree, 2 metadata) | item u32 BE
 *   leaf  = tag u8 (0) | norm f32 | dim x f32 little endian
 * Replaces the item reads of arroy::Reader::open / nns_by_vector
 * (src/search/context.rs:188-190) for an index that already exists on disk. */
int32_t vs_ingest_arroy_items(vs_ctx* ctx, const uint8_t* const* keys, const size_t* key_lens,
                              const uint8_t* const* values, const size_t* value_lens, uint64_t n,
                              uint64_t* n_items, uint64_t* n_other_nodes, uint64_t* n_unrouted);
/* The key layout alone (mode: 0 item, 1 tree, 2 metadata; other values are handed back as they are). */
int32_t vs_arroy_parse_key(const uint8_t* key, size_t len, uint16_t* index, uint8_t* mode,
                           uint32_t* item);
/* Writes the 7-byte key and the leaf bytes of one item in the layout above
 * (value_out == NULL: only *value_len). For hosts and tests that have vectors
 * but no arroy at hand. */
int32_t vs_arroy_encode_item(uint16_t index, uint32_t item, const float* vector, uint32_t dim,
                             uint8_t* key_out /* 7 bytes */, uint8_t* value_out, size_t value_cap,
                             size_t* value_len);

/* ---- search, host buffers ------------------------------------------------ */

/* Replaces reader.nns_by_vector(txn, target, limit, None, Some(candidates))
 * (vector_sort.rs:73-77) in its exhaustive regime: the `k` nearest items among
 * candidates ∩ items, ascending by (distance, docid).
 *   mask_words: dense candidate bitmask over docids, bit (d & 63) of word d >> 6;
 *               NULL means "every item" (the None case of arroy's API);
 *               docids >= 64 * n_mask_words are not candidates.
 *   out_ids / out_dist: room for k entries; *out_n <= k entries are written.
 * k == 0 returns VS_OK with *out_n = 0 (arroy: count == 0 -> empty). */
int32_t vs_search(vs_store* s, const float* query, uint32_t dim, uint32_t k,
                  const uint64_t* mask_words, uint64_t n_mask_words, uint32_t* out_ids,
                  float* out_dist, uint32_t* out_n);

/* Same, the candidates given as a serialised RoaringBitmap — the bytes
 * RoaringBitmap::serialize_into writes (src/heed_codec/roaring_bitmap/
 * roaring_bitmap_codec.rs:29-33). Expanded to the dense mask on the device. */
int32_t vs_search_roaring(vs_store* s, const float* query, uint32_t dim, uint32_t k,
                          const uint8_t* roaring_bytes, size_t len, uint32_t* out_ids,
                          float* out_dist, uint32_t* out_n);

/* A candidate filter kept in device memory for reuse. The reference evaluates
 * its filter into a RoaringBitmap once per request (`filtered_universe`,
 * src/search/search.rs:222-228) and passes it to every nns_by_vector call of the
 * request (src/search/ranking/vector_sort.rs:73-77); a host that sees the same
 * facet filter again (the next page, the next query under the same facets) can
 * keep its mask on the device instead of sending N/8 bytes with every search.
 * A mask belongs to the store it was created for (it is aligned with that
 * store's docid range) and is immutable: any number of threads may search with
 * it. Destroy it before the store. Documents removed or added later need no
 * new mask: a search intersects the mask with the store's sealed state. */
/* vs_mask_create: docids >= 64 * n_mask_words are not candidates, so
 * (NULL, 0) makes the EMPTY filter — unlike vs_search, where mask_words == NULL
 * means "every item"; the mask argument of vs_search_mask expresses that by
 * being NULL itself. */
typedef struct vs_mask vs_mask;
int32_t vs_mask_create(vs_store* s, const uint64_t* mask_words, uint64_t n_mask_words, vs_mask** out);
int32_t vs_mask_create_roaring(vs_store* s, const uint8_t* roaring_bytes, size_t len, vs_mask** out);
int32_t vs_mask_destroy(vs_mask* m);
/* 1 if `m` was created for `s` and still matches its size (vs_store_reserve
 * outdates a mask), else 0. Never fails; reads nothing of the store the mask
 * was created for, which may be gone. */
int32_t vs_mask_fits(vs_store* s, const vs_mask* m);
/* vs_search with the candidates of `mask` (NULL = every item). */
int32_t vs_search_mask(vs_store* s, const float* query, uint32_t dim, uint32_t k, const vs_mask* mask,
                       uint32_t* out_ids, float* out_dist, uint32_t* out_n);

/* One query against EVERY vector store of an embedder in one call: replaces the
 * whole store loop of VectorSort::fill_buffer
 * (src/search/ranking/vector_sort.rs:55-78), which opens the readers
 * `(embedder_index << 8) | k` for k = 0, 1, .. until one is missing
 * (MissingMetadata), calls nns_by_vector(target, limit, None, candidates) on each
 * and sorts the concatenated lists by distance. Here the stores
 * store_index_base | 0, | 1, .. (low byte of store_index_base 0) are taken up to
 * the first that does not exist or was never sealed; the filter crosses to the
 * device ONCE (the reference passes the same `candidates` to every reader), every
 * store's scan is queued on one stream, and there is one wait.
 *   filter: at most one of mask_words (as vs_search), roaring_bytes (as
 *           vs_search_roaring), mask (as vs_search_mask: made for one of these
 *           stores; the others must start at the same docid and be no larger);
 *           all three NULL = every item.
 *   out_*:  out_cap entries each. Every store contributes its own top-k exactly
 *           as vs_search returns it (at most k entries, duplicates of a docid
 *           across stores included: a document has a vector in several stores);
 *           the n_stores lists go out merged, ascending by (distance, docid,
 *           store), NaN last. out_store[i] = k of the store entry i came from.
 *           The reference sorts by distance alone with an unstable sort, so its
 *           order of exact ties is unspecified; this is one of its orders.
 *   *out_n_stores is set as soon as the stores are counted; if out_cap <
 *   n_stores * k the call returns VS_E_CAPACITY and the caller may retry with
 *   room for *out_n_stores * k entries. No store at all: VS_OK, *out_n = 0.
 * Errors as vs_search (VS_E_DIM_MISMATCH if any store's dimension differs). */
int32_t vs_search_stores(vs_ctx* ctx, uint16_t store_index_base, const float* query, uint32_t dim,
                         uint32_t k, const uint64_t* mask_words, uint64_t n_mask_words,
                         const uint8_t* roaring_bytes, size_t roaring_len, const vs_mask* mask,
                         uint32_t* out_ids, float* out_dist, uint8_t* out_store, uint32_t out_cap,
                         uint32_t* out_n, uint32_t* out_n_stores);

/* `n_q` queries against one shared candidate mask. out_ids / out_dist hold
 * n_q * k entries (query i at offset i * k), out_n holds n_q counts. On a
 * VS_BF16 angular store with k <= VS_BATCH_MAX_K this runs the batched
 * matrix-core scan (queries rounded to bf16). On an f32 angular or dot-product
 * store of up to 768 dimensions with k <= 32, batches of 17 or more queries go
 * 32 at a time through the f32 matrix cores (exact f32 products, another
 * summation order: distances equal vs_search's up to rounding). Otherwise groups
 * of eight (f32 rows of up to 768 dimensions) or four queries (f32 up to 1536,
 * bf16 up to 1536 dimensions) share each pass over the rows when k <= 128, and
 * every query's result is bit-identical to its own vs_search; on binary stores
 * (rows of up to 2048 bits, k <= 128) four queries share a pass, results equal
 * to vs_search's; anything else is one scan per query. On the matrix-core path
 * only the filter's candidates are read and multiplied: rows of more than 128
 * dimensions are fetched where they lie; shorter rows, under a filter that keeps
 * at most a quarter of the items, are first copied into a dense slab inside the
 * call's own workspace. The result is the same either way. No reference
 * counterpart (the reference searches one vector per request). */
int32_t vs_search_batch(vs_store* s, const float* queries, uint32_t n_q, uint32_t dim,
                        uint32_t k, const uint64_t* mask_words, uint64_t n_mask_words,
                        uint32_t* out_ids, float* out_dist, uint32_t* out_n);

/* vs_search_batch with one filter PER QUERY: query i is searched under
 * masks[i] if `masks` and masks[i] are non-NULL, else under mask_words[i]
 * (n_mask_words[i] words) if non-NULL, else over every item. Either array may
 * be NULL as a whole. The reference evaluates one filter per request
 * (`filtered_universe`, src/search/search.rs:222-228), so a batch of requests
 * has a universe each. Queries that name the same filter (the same vs_mask or
 * the same host pointer) share passes exactly as in vs_search_batch; every
 * other query is scanned under its own mask and reads only its own candidates.
 * Result i equals vs_search_batch of its group alone. */
int32_t vs_search_batch_masks(vs_store* s, const float* queries, uint32_t n_q, uint32_t dim,
                              uint32_t k, const uint64_t* const* mask_words,
                              const uint64_t* n_mask_words, const vs_mask* const* masks,
                              uint32_t* out_ids, float* out_dist, uint32_t* out_n);

/* ---- search, device buffers (asynchronous; multi-GPU and benchmark path) -- */

/* Bytes of device workspace vs_search_device needs for this store and k. */
int32_t vs_search_workspace_bytes(vs_store* s, uint32_t k, uint64_t* bytes);

/* Enqueues one query on `hip_stream` and returns immediately.
 *   d_query:      `dim` f32 in device memory; for VS_BIN1 stores the packed query
 *                 bits instead (bit i of the query in word i/64 at position i%64),
 *                 vs_store_row_bytes bytes, zero padded
 *   d_mask_words: candidate words ALIGNED TO THE STORE: word j covers docids
 *                 docid_base + 64 j .. ; capacity/64 words; NULL = every item
 *   d_workspace:  `workspace_bytes` bytes, private to this call until the stream
 *                 has passed it. The call checks the size against what the store
 *                 needs NOW and returns VS_E_CAPACITY if it is short: sizes
 *                 grow with the store, so ask vs_search_workspace_bytes again
 *                 after a vs_store_reserve
 *   d_out_keys:   k u64 keys, ascending; key = (orderable(distance) << 32) |
 *                 docid, padded with VS_KEY_NONE. This is the shard-local
 *                 top-k a multi-GPU caller all-gathers. */
int32_t vs_search_device(vs_store* s, const void* d_query, uint32_t dim, uint32_t k,
                         const uint64_t* d_mask_words, void* d_workspace, uint64_t workspace_bytes,
                         uint64_t* d_out_keys, void* hip_stream);

/* No counterpart in the reference. vs_search_device for callers that cannot
 * allocate device memory or pass a workspace: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc. Same arguments and the same
 * d_out_keys as vs_search_device, minus d_workspace: the scan's partial lists go
 * to a static array of VS_NOALLOC_SCRATCH_BYTES in device memory (one per GPU,
 * held until the process ends). The call allocates nothing and frees nothing.
 * Limit: (workgroups of the scan: by default at most 512, 1024 over float rows of
 * 1 KiB or less) * k * 8 bytes must fit, i.e. k <= 256 (128) on a large store;
 * beyond that VS_E_CAPACITY. Every call on one
 * GPU shares the array: calls whose kernels can overlap on that GPU (different
 * streams, or several threads) race; issue them on one stream. The call records
 * no event either (the stream may be under capture), so unlike the other
 * searches it is not tracked by the snapshot rule: have these searches complete
 * before a vs_store_seal that replaces or removes items. A CAPTURED call holds
 * the store's buffers as they were when it was captured (the present bitmap of
 * that seal, the slab's address, its size): a graph that contains it is valid
 * until the next vs_store_seal or vs_store_reserve on the store and must be
 * captured again after either; the query and the mask are read at their
 * addresses on every replay, so their contents may change between replays. */
#define VS_NOALLOC_SCRATCH_BYTES (1u << 20)
int32_t vs_search_device_noalloc(vs_store* s, const void* d_query, uint32_t dim, uint32_t k,
                                 const uint64_t* d_mask_words, uint64_t* d_out_keys,
                                 void* hip_stream);

/* Batched form: `n_q` <= VS_BATCH_MAX_QUERIES queries (f32 [n_q][dim] in device
 * memory) under one candidate mask; d_out_keys receives n_q lists of k keys.
 *   - angular VS_BF16 stores, k <= VS_BATCH_MAX_K: ONE pass over the rows on the
 *     matrix cores (queries rounded to bf16 on the device, so the distances are
 *     those of the rounded queries); with a mask and rows of more than 128
 *     dimensions, over the candidates only;
 *   - f32 angular / dot-product stores of up to 768 dimensions, k <= 32, 17 or
 *     more queries: 32 per pass on the f32 matrix cores (distances equal
 *     vs_search_device's up to rounding);
 *   - otherwise, f32 rows of up to 1536 dimensions or bf16 rows of up to 1536,
 *     k <= 128: groups of eight (f32, up to 768 dimensions) or four queries share
 *     a pass, each list bit-identical to vs_search_device's;
 *   - VS_BIN1 stores (rows of up to 2048 bits, k <= 128): d_queries holds n_q
 *     PACKED queries of vs_store_row_bytes bytes each (as vs_search_device takes
 *     one); four share a pass, lists equal to vs_search_device's;
 *   - anything else: VS_E_INVALID_ARG (use vs_search_device per query).
 * `workspace_bytes` is checked against vs_search_batch_workspace_bytes for the
 * store as it is now (VS_E_CAPACITY if short; the need grows with the store).
 * No reference counterpart (the reference searches one vector per request). */
int32_t vs_search_batch_workspace_bytes(vs_store* s, uint32_t k, uint64_t* bytes);
int32_t vs_search_batch_device(vs_store* s, const float* d_queries, uint32_t n_q, uint32_t dim,
                               uint32_t k, const uint64_t* d_mask_words, void* d_workspace,
                               uint64_t workspace_bytes, uint64_t* d_out_keys, void* hip_stream);

/* Merges `n_lists` ascending key lists of `k` entries each (list i at
 * d_keys + i * k; e.g. the all-gathered shard results) into the global top-k
 * and decodes it: d_out_ids[k], d_out_dist[k], *d_out_n. `metric` selects how a
 * key's distance field is decoded (VS_ANGULAR: f32; VS_HAMMING: integer as f32;
 * VS_EUCLIDEAN: square root of the f32; VS_DOT: the f32 negated). */
int32_t vs_merge_topk_device(vs_ctx* ctx, const uint64_t* d_keys, uint32_t n_lists, uint32_t k,
                             int32_t metric, uint32_t* d_out_ids, float* d_out_dist,
                             uint32_t* d_out_n, void* hip_stream);

/* `n_queries` such merges in one launch (one batch of queries after ONE
 * all-gather): list i of query q starts at d_keys + q * query_stride + i *
 * list_stride (in keys); outputs of query q at d_out_ids + q * k, d_out_dist +
 * q * k, d_out_n + q. For the layout [rank][query][k] an all-gather of
 * [query][k] produces: list_stride = n_queries * k, query_stride = k. */
int32_t vs_merge_topk_batch_device(vs_ctx* ctx, const uint64_t* d_keys, uint32_t n_lists,
                                   uint64_t list_stride, uint32_t n_queries, uint64_t query_stride,
                                   uint32_t k, int32_t metric, uint32_t* d_out_ids,
                                   float* d_out_dist, uint32_t* d_out_n, void* hip_stream);

/* ---- one store sharded over the GPUs of a node --------------------------------- */

/* No counterpart in the reference, which is a single-process library: this is
 * the multi-GPU form of VectorSort::fill_buffer's loop over stores
 * (src/search/ranking/vector_sort.rs:73-78) — shards play the role of stores,
 * their (distance, docid)-ordered lists are merged into the list arroy would
 * return for one store holding everything. One process per GPU; rank r holds
 * the docids [r * per, (r + 1) * per), per a multiple of 64, in a store created
 * with docid_base = r * per. The library owns the RCCL communicator (librccl is
 * loaded on first use), so the host needs no collective library of its own:
 *   rank 0:      vs_group_unique_id(id); hand the VS_GROUP_ID_BYTES bytes to
 *                every rank by whatever channel the host has;
 *   every rank:  vs_group_create(ctx, rank, world, id, &group)   (collective). */
typedef struct vs_group vs_group;
#define VS_GROUP_ID_BYTES 128
int32_t vs_group_unique_id(uint8_t* out /* VS_GROUP_ID_BYTES */);
int32_t vs_group_create(vs_ctx* ctx, int32_t rank, int32_t world, const uint8_t* unique_id,
                        vs_group** out);
int32_t vs_group_destroy(vs_group* g);

/* Searches `n_q` <= VS_BATCH_MAX_QUERIES queries on every rank's shard and
 * leaves the GLOBAL top-k of each on every rank: local scans, ONE all-gather of
 * the shard-local keys (8 * k * n_q bytes per rank) over xGMI, one merge launch.
 * Collective: every rank of the group calls it with the same n_q, k and flags,
 * in the same order. Asynchronous on `hip_stream`. A group's buffers are reused
 * from call to call, so its searches run one after the other on the device as
 * well: a call on another stream than the group's previous one first makes its
 * stream wait for that one's end. The buffers are grown (an allocation) the
 * first time a larger n_q * k comes along.
 *   d_queries:    f32 [n_q][dim] in device memory (VS_BIN1: n_q packed queries
 *                 of vs_store_row_bytes bytes), the same on every rank
 *   d_mask_words: this rank's candidate words, aligned to its store as in
 *                 vs_search_device; NULL = every item
 *   flags:        0 = one scan per query, every list bit-identical to what
 *                 vs_search on one store holding all shards returns;
 *                 VS_SHARDED_SHARE_PASSES = queries share passes over the rows
 *                 where vs_search_batch_device serves the store (same caveats);
 *                 VS_SHARDED_ALWAYS_GATHER = run the collective even in a group
 *                 of one (a rehearsal of the multi-rank path on one GPU)
 *   outputs:      d_out_ids / d_out_dist [n_q][k], d_out_n [n_q], device memory. */
#define VS_SHARDED_SHARE_PASSES 1u
#define VS_SHARDED_ALWAYS_GATHER 2u
int32_t vs_search_sharded_device(vs_group* g, vs_store* s, const void* d_queries, uint32_t n_q,
                                 uint32_t dim, uint32_t k, const uint64_t* d_mask_words,
                                 uint32_t flags, uint32_t* d_out_ids, float* d_out_dist,
                                 uint32_t* d_out_n, void* hip_stream);

/* ---- candidate bitmaps ---------------------------------------------------- */

/* Host-side expansion of a serialised RoaringBitmap into a dense mask of
 * n_words words (docids beyond are dropped); *cardinality = bits set. Accepts
 * both cookies of the portable format (with and without run containers). */
int32_t vs_roaring_to_mask(const uint8_t* bytes, size_t len, uint64_t* words, uint64_t n_words,
                           uint64_t* cardinality);

/* Serialises a dense mask as the portable RoaringBitmap format without run
 * containers (what roaring-rs writes). Call with out == NULL to get the size. */
int32_t vs_mask_to_roaring(const uint64_t* words, uint64_t n_words, uint8_t* out, size_t cap,
                           size_t* len);

/* CboRoaringBitmapCodec::deserialize_from (src/heed_codec/roaring_bitmap/
 * cbo_roaring_bitmap_codec.rs:44-58): <= 7 native-endian u32 when len <= 28,
 * otherwise a serialised RoaringBitmap. */
int32_t vs_cbo_roaring_to_mask(const uint8_t* bytes, size_t len, uint64_t* words,
                               uint64_t n_words, uint64_t* cardinality);
/* CboRoaringBitmapCodec::serialize_into (cbo_roaring_bitmap_codec.rs:30-42). */
int32_t vs_mask_to_cbo_roaring(const uint64_t* words, uint64_t n_words, uint8_t* out, size_t cap,
                               size_t* len);
/* BoRoaringBitmapCodec (bo_roaring_bitmap_codec.rs:13-31): raw native-endian u32 list. */
int32_t vs_bo_roaring_to_mask(const uint8_t* bytes, size_t len, uint64_t* words,
                              uint64_t n_words, uint64_t* cardinality);

#ifdef __cplusplus
}
#endif
#endif /* VS_ABI_H */
