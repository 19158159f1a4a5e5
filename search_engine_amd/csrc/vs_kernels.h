// Host-callable launchers of the HIP kernels (implemented in vs_kernels.hip).
// Everything here enqueues work on the given stream and returns; nothing
// allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {

// Geometry of a row slab in HBM. Rows are stored row-major, one row per
// docid slot (slot = docid - docid_base), each row `row_vecs` 16-byte vectors
// long and zero padded; `n_slots` is a multiple of 64.
constexpr uint32_t SLAB_PAD_ROWS = 192;  // readable rows behind the last slot (contents irrelevant)

struct SlabView {
    const void* rows;         // (n_slots + SLAB_PAD_ROWS) * row_vecs * 16 bytes
    const float* norms;       // n_slots, ‖row‖₂ (angular stores; null otherwise)
    const uint64_t* present;  // n_slots / 64 words, bit set = slot holds an item
    uint64_t n_slots;
    uint32_t row_vecs;    // 16-byte vectors per row
    uint32_t dim;         // logical dimension (elements, or bits for bin1)
    uint32_t docid_base;  // multiple of 64
    int32_t dtype;        // VS_F32 / VS_BF16 / VS_BIN1
    int32_t metric = 0;   // VS_ANGULAR / VS_EUCLIDEAN / VS_DOT (f32, bf16 rows), VS_HAMMING (bin1)
};

// Scan tuning knobs (0 = library default); exposed so the benchmark can A/B
// schedules in one process.
struct ScanTuning {
    int32_t blocks;  // grid size, 0 = default
    int32_t nt;      // non-zero: non-temporal row loads (the default)
    int32_t rows_in_flight = 0;  // rows one wave loads before it scores them, 0 = default
    int32_t query_lanes = 0;     // many-query launches: queries in flight per share of the mask words, 0 = default (1)
};

// Compute units the default grids are sized for (256 until vs_init has asked the device). One value
// per process: every GPU of a node is the same part.
void set_device_cus(uint32_t n);
uint32_t device_cus();

uint32_t scan_grid_blocks(const SlabView& s, const ScanTuning& t, bool masked);
// device bytes of the partial-list buffer one scan of this slab writes
uint64_t scan_partial_bytes(const SlabView& s, uint32_t k, const ScanTuning& t);
// the same for the largest grid any tuning may ask for: what a caller-provided workspace is
// sized with, so that a tuning change between sizing and launch cannot overrun it
constexpr int32_t MAX_TUNED_BLOCKS = 2048;
uint64_t scan_workspace_bytes(const SlabView& s, uint32_t k);

// Scan of one query over f32 or bf16 rows under s.metric (angular, Euclidean or
// dot product): writes the per-workgroup top-k lists to `partial` and then the
// merged ascending top-k keys to `out_keys[k]` (nullable) and, decoded, to
// out_ids / out_dist / out_n (nullable as a group). With out_keys and out_ids both
// null only the partial lists are written (rank-major: entry r of workgroup b at
// partial[r * blocks + b], blocks = scan_grid_blocks(s, t, cand != nullptr)) and the
// caller merges them, e.g. those of several queries in one launch_merge_batch.
// `cand` (nullable) is aligned with `present`.
hipError_t launch_scan_float(const SlabView& s, const float* d_query, const uint64_t* cand,
                               uint32_t k, uint64_t* partial, uint64_t* out_keys,
                               uint32_t* out_ids, float* out_dist, uint32_t* out_n,
                               const ScanTuning& t, hipStream_t stream);

// One query of a many-query launch that is not searched with the launch's common arguments: its own
// candidate mask, or another slab altogether (the stores of one embedder: same row shape, element
// type and metric). rows / norms / present / n_slots / docid_base as in SlabView; cand nullable;
// query nullable = the launch's d_queries + y * query_stride.
struct ScanJob {
    const void* rows;
    const float* norms;
    const uint64_t* present;
    const uint64_t* cand;
    const void* query;
    uint64_t n_slots;
    uint32_t docid_base;
    uint32_t reserved;
};

// `n_q` independent single-query scans in ONE launch: a workgroup takes its share of the mask words
// for one query after the other, so the queries follow each other without a launch's ramp and tail
// in between. No row read is shared; query q's partial lists (at partial + q * blocks * k keys, blocks =
// scan_grid_blocks(s, t, masked), rank-major as above) are bit for bit what launch_scan_float writes
// for it alone. Query q starts at d_queries + q * query_stride floats and is searched under `cand`
// (nullable) over `s` -- or, if `d_jobs` (a device array of n_q entries) is given, as d_jobs[q] says;
// `s` must then be the LARGEST of the jobs' slabs (it sizes the grid). masked = cand or d_jobs given.
// `partial` needs n_q times scan_partial_bytes. The caller merges, e.g. all queries in one
// launch_merge_batch.
hipError_t launch_scan_float_queries(const SlabView& s, const float* d_queries, uint32_t query_stride,
                                     uint32_t n_q, const uint64_t* cand, const ScanJob* d_jobs,
                                     uint32_t k, uint64_t* partial, const ScanTuning& t, hipStream_t stream);
// The same over a VS_BIN1 slab; query_stride in 16-byte vectors.
hipError_t launch_scan_hamming_queries(const SlabView& s, const void* d_queries, uint32_t query_stride,
                                       uint32_t n_q, const uint64_t* cand, const ScanJob* d_jobs,
                                       uint32_t k, uint64_t* partial, const ScanTuning& t, hipStream_t stream);

// Up to scan_multi_queries(s) queries (eight for f32 rows of up to 768 dimensions, else four) in
// one pass over the rows. Serves f32 rows of up to 1536 dimensions and bf16 rows of up to 1536,
// k <= 128 (scan_multi_serves). Query q starts at d_queries + q * query_stride floats; `partial`
// needs scan_multi_queries(s) times scan_partial_bytes; outputs are n_q consecutive blocks of k
// entries (out_keys nullable; out_ids / out_dist / out_n nullable as a group) and n_q counts. Results are bit-identical to n_q launch_scan_float calls.
bool scan_multi_serves(const SlabView& s, uint32_t k);
uint32_t scan_multi_queries(const SlabView& s);
hipError_t launch_scan_multi(const SlabView& s, const float* d_queries, uint32_t query_stride,
                             uint32_t n_q, const uint64_t* cand, uint32_t k, uint64_t* partial,
                             uint64_t* out_keys, uint32_t* out_ids, float* out_dist, uint32_t* out_n,
                             const ScanTuning& t, hipStream_t stream);

// Up to 32 queries in one pass over an f32 slab on the f32 matrix cores (angular or dot product,
// rows of up to 768 dimensions, k <= 32: scan_mq32_serves). `partial` needs
// scan_mq32_partial_bytes. Arguments and outputs as launch_scan_multi; distances agree with
// launch_scan_float within rounding (a different summation order), not bit for bit.
bool scan_mq32_serves(const SlabView& s, uint32_t k);
uint64_t scan_mq32_partial_bytes(const SlabView& s, uint32_t k);
hipError_t launch_scan_mq32(const SlabView& s, const float* d_queries, uint32_t query_stride,
                            uint32_t n_q, const uint64_t* cand, uint32_t k, uint64_t* partial,
                            uint64_t* out_keys, uint32_t* out_ids, float* out_dist, uint32_t* out_n,
                            hipStream_t stream);

// Hamming scan of one query against a VS_BIN1 slab. `d_query` holds the packed
// query bits, row_vecs * 16 bytes, zero padded. Outputs as launch_scan_float;
// a key's distance field is the integer Hamming distance.
hipError_t launch_scan_hamming(const SlabView& s, const void* d_query, const uint64_t* cand,
                               uint32_t k, uint64_t* partial, uint64_t* out_keys,
                               uint32_t* out_ids, float* out_dist, uint32_t* out_n,
                               const ScanTuning& t, hipStream_t stream);

// Up to four packed queries (query_stride 16-byte vectors apart) in one pass over a VS_BIN1 slab
// of rows of up to 2048 bits, k <= 128 (scan_hamming_multi_serves). `partial` needs four times
// scan_partial_bytes. Outputs as launch_scan_multi; results equal launch_scan_hamming's exactly.
bool scan_hamming_multi_serves(const SlabView& s, uint32_t k);
hipError_t launch_scan_hamming_multi(const SlabView& s, const void* d_queries, uint32_t query_stride,
                                     uint32_t n_q, const uint64_t* cand, uint32_t k, uint64_t* partial,
                                     uint64_t* out_keys, uint32_t* out_ids, float* out_dist,
                                     uint32_t* out_n, const ScanTuning& t, hipStream_t stream);

// Batched angular scan of a VS_BF16 slab on the matrix cores (vs_batch.hip): up to
// 256 queries (f32 [n_q][dim] in device memory, rounded to bf16 by a prologue
// kernel) share one pass over the rows. k <= VS_BATCH_MAX_K. `workspace` needs
// batch_workspace_bytes bytes. Outputs as n_q consecutive blocks of k entries:
// out_keys (nullable), out_ids / out_dist / out_n (nullable as a group).
uint64_t batch_workspace_bytes(const SlabView& s, uint32_t k);
// whole_tiles: under a mask, multiply the store's tiles anyway instead of fetching the candidates
// where they lie (pays when the mask excludes next to nothing). *d_candidates (nullable) receives
// the device address of the number of candidates the call counts on the stream, or null if it
// counts none (no mask, or rows too short to be fetched in place).
hipError_t launch_scan_batch_bf16(const SlabView& s, const float* d_queries, uint32_t n_q,
                                  const uint64_t* cand, uint32_t k, void* workspace,
                                  uint64_t* out_keys, uint32_t* out_ids, float* out_dist,
                                  uint32_t* out_n, hipStream_t stream, bool whole_tiles = false,
                                  const uint32_t** d_candidates = nullptr);
// True if launch_scan_batch_bf16 itself reads only the candidates of a filter (rows of more than 128
// elements: the streaming pass fetches them where they lie); otherwise it multiplies whole 256-row
// tiles, and the gathered form below pays under selective filters.
bool batch_filters_in_place(const SlabView& s);
// The same search under a selective filter: the candidates (present & cand, at most `max_candidates`
// of them: any upper bound; rows beyond it would be dropped) are first gathered into a dense slab inside
// `workspace` (batch_gather_workspace_bytes), in slot order, so that the scan reads and multiplies only
// them. Same outputs; docids are those of the store.
uint64_t batch_gather_workspace_bytes(const SlabView& s, uint64_t max_candidates, uint32_t k);
hipError_t launch_scan_batch_bf16_gathered(const SlabView& s, const float* d_queries, uint32_t n_q,
                                           const uint64_t* cand, uint64_t max_candidates, uint32_t k,
                                           void* workspace, uint64_t* out_keys, uint32_t* out_ids, float* out_dist,
                                           uint32_t* out_n, hipStream_t stream);
// *d_count = number of candidates (present & cand); d_block_sums: count_candidates_workspace_bytes bytes.
uint64_t count_candidates_workspace_bytes(const SlabView& s);
hipError_t launch_count_candidates(const SlabView& s, const uint64_t* cand, uint32_t* d_block_sums,
                                   uint32_t* d_count, hipStream_t stream);

// ‖row‖₂ for slots [slot0, slot0 + n) of an angular slab, same summation order
// as the scan uses for ⟨row, query⟩.
hipError_t launch_norms(const SlabView& s, float* norms, uint64_t slot0, uint64_t n,
                        hipStream_t stream);

// Scatter `n` staged f32 rows [n][dim], converting to the slab's element type. dest[i] names
// where row i goes: a slot of `rows_out`; INGEST_SIDE | j = row j of the densely packed side
// buffer `side_out` (rows that replace an item searches can still see; applied at the next seal);
// INGEST_SKIP = nowhere (an earlier duplicate of a docid within one upload).
constexpr uint64_t INGEST_SKIP = 1ull << 63, INGEST_SIDE = 1ull << 62;
hipError_t launch_ingest_f32(const SlabView& s, void* rows_out, void* side_out, const float* staged,
                             const uint64_t* dest, uint64_t n, hipStream_t stream);
// Copies `n` densely packed rows in the slab's element type to the slots d_slots[i].
hipError_t launch_scatter_rows(const SlabView& s, void* rows_out, const void* staged, const uint64_t* d_slots,
                               uint64_t n, hipStream_t stream);
// launch_norms for a list of slots.
hipError_t launch_norms_slots(const SlabView& s, float* norms, const uint64_t* d_slots, uint64_t n,
                              hipStream_t stream);

// Merge `n_lists` ascending key lists (entry r of list l at keys[l * list_stride
// + r * rank_stride]) into the ascending global top-k `out_keys[k]`; when
// out_ids is non-null also decode it into (ids, distances, count).
hipError_t launch_merge(const uint64_t* keys, uint32_t n_lists, uint64_t list_stride,
                        uint64_t rank_stride, uint32_t k, int32_t metric, uint64_t* out_keys,
                        uint32_t* out_ids, float* out_dist, uint32_t* out_n, hipStream_t stream);

// The same for `n_queries` independent merges in one launch: query q reads its
// lists at keys + q * query_stride and writes out_*[q * k ..], out_n[q].
hipError_t launch_merge_batch(const uint64_t* keys, uint32_t n_lists, uint64_t list_stride,
                              uint64_t rank_stride, uint32_t n_queries, uint64_t query_stride,
                              uint32_t k, int32_t metric, uint64_t* out_keys, uint32_t* out_ids,
                              float* out_dist, uint32_t* out_n, hipStream_t stream,
                              const uint32_t* gate = nullptr);  // non-null: the launch does nothing unless *gate != 0

// Expand a serialised RoaringBitmap already in device memory into the dense
// mask `words[n_words]` covering docids [docid_base, docid_base + 64 n_words).
// `desc` is the host-parsed container table copied to the device.
struct RoaringContainerDesc {
    uint32_t key;       // high 16 bits of the docids
    uint32_t kind;      // 0 array, 1 bitmap, 2 run
    uint32_t count;     // array: values, run: runs, bitmap: unused
    uint32_t offset;    // byte offset of the payload inside the serialised blob
};
hipError_t launch_roaring_expand(const uint8_t* d_bytes, const RoaringContainerDesc* d_desc,
                                 uint32_t n_containers, uint32_t docid_base, uint64_t* words,
                                 uint64_t n_words, hipStream_t stream);

}  // namespace vs
