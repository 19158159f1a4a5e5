// Internals shared by the translation units behind include/vs_abi.h: the opaque handle types,
// the workspace pool, and the snapshot protocol that lets searches run while a writer changes
// the store. Nothing here is part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/vs_abi.h"
#include "vs_kernels.h"

namespace vsrt {

// Records the calling thread's error message and returns `code`.
int32_t fail(int32_t code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void set_dim_mismatch(uint32_t expected, uint32_t received);

#define HIP_TRY(expr)                                                                                \
    do {                                                                                             \
        const hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                        \
            return vsrt::fail(e_ == hipErrorOutOfMemory ? VS_E_OOM : VS_E_DEVICE, "%s failed: %s",   \
                              #expr, hipGetErrorString(e_));                                         \
    } while (0)

constexpr uint32_t MAX_DIM_VECS = 16 * 64;  // row vectors the angular scan kernels are built for
constexpr uint32_t MAX_BIN_VECS = 32;       // and the Hamming scan (4096 bits)
constexpr size_t STAGE_BYTES = 32u << 20;
constexpr size_t QUERY_BYTES = (size_t)MAX_DIM_VECS * 16 * 2;  // bf16 rows: 8 f32 per vector
constexpr size_t RESULT_BYTES = (size_t)VS_MAX_K * 8 + 64;
constexpr size_t POOL_KEEP = 32;  // idle workspaces kept for reuse; more concurrent callers than this pay set-up again

// Everything one in-flight host-buffer search needs; handed out by the pool so
// that concurrent vs_search calls never share a stream or a buffer.
struct Workspace {
    hipStream_t stream = nullptr;
    float* d_query = nullptr;
    uint64_t* d_mask = nullptr;
    size_t mask_words = 0;
    uint64_t* d_partial = nullptr;
    size_t partial_bytes = 0;
    uint64_t* d_keys = nullptr;  // VS_MAX_K keys, ids, distances and the count, one allocation
    uint32_t* d_ids = nullptr;
    float* d_dist = nullptr;
    uint32_t* d_n = nullptr;
    uint8_t* d_blob = nullptr;
    size_t blob_bytes = 0;
    unsigned char* d_batch = nullptr;  // batched path: queries, kernel workspace, results
    size_t batch_bytes = 0;
    unsigned char* h_pinned = nullptr;  // query, then results
    size_t pinned_bytes = 0;
    ~Workspace();
};

// Grows a device buffer to `want` bytes (contents are not kept). The old buffer is released
// only once the new one exists, so a failed growth leaves the workspace usable.
template <typename T>
int32_t grow(T** p, size_t* have, size_t want) {
    if (*have >= want) return VS_OK;
    T* fresh = nullptr;
    HIP_TRY(hipMalloc(&fresh, want));
    if (*p) (void)hipFree(*p);
    *p = fresh;
    *have = want;
    return VS_OK;
}

}  // namespace vsrt

struct vs_ctx {
    int device = 0;
    std::mutex mu;  // stores map, pool
    std::map<uint16_t, vs_store*> stores;
    std::vector<std::unique_ptr<vsrt::Workspace>> pool;
    // scan schedule knobs; read as one snapshot per call (vsrt::tuning_of)
    std::atomic<int32_t> tune_blocks{0}, tune_nt{1}, tune_rows{0}, tune_qlanes{0};
    unsigned char* d_scratch = nullptr;  // this GPU's copy of the static scratch array (vs_search_device_noalloc)
    // writer side: upload staging (guarded by upload_mu; the engine has one writer)
    std::mutex upload_mu;
    unsigned char* h_stage = nullptr;
    unsigned char* d_stage = nullptr;
    unsigned char* h_small = nullptr;  // 64 KiB pinned: one row read back, small lists
    hipStream_t upload_stream = nullptr;
};

struct vs_mask {
    vs_store* store;
    uint64_t* d_words;  // store->n_slots / 64 words, aligned with the store
    uint64_t n_slots;   // of the store when the mask was made (vs_store_reserve outdates it)
    uint64_t store_serial;  // of that store: a later store at the same address is another store
    uint32_t docid_base;    // of that store: word 0 covers docids docid_base ..
};

// One arroy index as a row slab.
//
// Snapshot rule: a search sees the item set AND the rows of the last seal that completed before
// it started, like a reader holding an LMDB RoTxn. Three things make that hold while the single
// writer uploads, removes and seals:
//   * the present bitmap has two device buffers; a seal fills the inactive one and flips. Every
//     search registers on the buffer it reads (host-buffer searches by a count, device-API
//     searches by an event on the caller's stream), and a seal waits for the stragglers of the
//     buffer it is about to overwrite — they are two generations old, so it rarely waits;
//   * a row written to a slot that no search can see (not present in either buffer) goes straight
//     into the slab. A row that replaces an item searches can still see is parked in a side buffer
//     (`d_pending`) and copied into the slab by the seal, together with its norm;
//   * only that copy needs the store to itself: the seal stops new searches, waits for the ones in
//     flight, scatters the parked rows, flips, and lets searches go on. A seal without replaced
//     items never makes a search wait.
struct vs_store {
    vs_ctx* ctx = nullptr;
    uint64_t serial = 0;  // unique in the process, never reused
    uint16_t index = 0;
    uint32_t dim = 0;
    int32_t dtype = VS_F32, metric = VS_ANGULAR;
    uint32_t docid_base = 0;
    uint64_t capacity = 0, n_slots = 0;
    uint32_t row_vecs = 0;
    void* d_rows = nullptr;
    float* d_norms = nullptr;
    uint64_t* d_present[2] = {nullptr, nullptr};

    // ---- reader / writer hand-over, guarded by snap_mu
    std::mutex snap_mu;
    std::condition_variable snap_cv;
    int active = 0;                 // the buffer searches read
    uint32_t readers[2] = {0, 0};   // searches between snapshot_begin and snapshot_end, per buffer
    std::map<hipStream_t, hipEvent_t> stream_events[2];  // device-API searches: last use per caller stream
    bool writer_blocks = false;     // rows searches can see are being rewritten: new searches wait

    // ---- writer's view, guarded by ctx->upload_mu
    std::vector<uint64_t> h_present;  // published by seal
    std::vector<uint64_t> h_sealed;   // what d_present[active] holds
    std::vector<uint64_t> h_visible;  // OR of both device buffers: slots some search may still read
    uint64_t n_items = 0;             // of h_present
    uint64_t dirty_lo = 0, dirty_hi = 0;  // slot range written in place since the last seal
    bool dirty = false;
    unsigned char* d_pending = nullptr;  // parked rows, slab element type, densely packed
    size_t pending_cap_rows = 0;
    std::vector<uint64_t> pending_slots;                   // slot of parked row j
    std::unordered_map<uint64_t, uint64_t> pending_of_slot;  // slot -> j
    uint64_t* d_pending_slots = nullptr;
    size_t pending_slots_bytes = 0;
    std::map<hipStream_t, hipEvent_t> upload_events;  // vs_store_upload_device copies on caller streams

    std::atomic<uint64_t> sealed_items{0};
    std::atomic<bool> sealed{false};

    // ---- batched device searches under a mask: what the last one counted (guarded by hint_mu).
    // The batched matrix-core scan fetches a filter's candidates where they lie, which costs a few
    // small launches more than multiplying whole tiles; under a mask that excludes next to nothing
    // the whole tiles win. A device-API call cannot wait for a count, so it uses the one its
    // predecessor under the same mask address left behind. The result never depends on it.
    std::mutex hint_mu;
    const void* hint_mask = nullptr;   // mask of the newest count requested
    uint32_t* hint_count = nullptr;    // pinned host word the counts are copied to
    hipEvent_t hint_event = nullptr;   // behind the newest copy
    const void* known_mask = nullptr;  // newest count seen to have arrived (calls queued back to back never see their
    uint64_t known_count = 0;          // predecessor's: they go by the last one that had, e.g. a warm-up call's)

    // geometry plus the given present buffer
    vs::SlabView view(int buf) const {
        vs::SlabView v;
        v.rows = d_rows;
        v.norms = d_norms;
        v.present = d_present[buf];
        v.n_slots = n_slots;
        v.row_vecs = row_vecs;
        v.dim = dim;
        v.docid_base = docid_base;
        v.dtype = dtype;
        v.metric = metric;
        return v;
    }
    uint64_t row_bytes() const { return (uint64_t)row_vecs * 16; }
};

namespace vsrt {

int32_t ws_acquire(vs_ctx* ctx, std::unique_ptr<Workspace>* out);
void ws_release(vs_ctx* ctx, std::unique_ptr<Workspace> ws);
vs::ScanTuning tuning_of(const vs_ctx* ctx);
int32_t check_store(vs_store* s);  // non-null, makes the store's device current

// A search's hold on one present buffer.
struct Snapshot {
    vs::SlabView v;
    int buf = -1;
};
// Waits while a seal rewrites visible rows, then registers on the active buffer.
void snapshot_begin(vs_store* s, Snapshot* out);
// Host-buffer searches: call after the search's stream has been waited for.
void snapshot_end(vs_store* s, Snapshot* snap);
// Device-API searches: call after the launches; records the caller stream's event so that a
// later seal waits for the kernels, not for the host call. track = false (callers that may be
// under graph capture) only drops the registration: ordering against seals is then the caller's.
int32_t snapshot_end_async(vs_store* s, Snapshot* snap, hipStream_t stream, bool track);

// Checks shared by the device-API searches (store sealed, dimension, k).
int32_t check_device_search(vs_store* s, uint32_t dim, uint32_t k);
// One scan of a sealed store under its metric; out_keys and out_ids both null = partial lists only.
int32_t launch_scan(const vs::SlabView& v, const vs::ScanTuning& t, const void* d_query, const uint64_t* d_cand,
                    uint32_t k, uint64_t* d_partial, uint64_t* d_keys, uint32_t* d_ids, float* d_dist,
                    uint32_t* d_n, hipStream_t stream);
// `n_q` independent single-query scans in one launch (partial lists only; the caller merges):
// query q at d_queries + q * query_bytes, under d_cand (nullable) over `v` or as d_jobs[q] says
// (device array, nullable; `v` is then the largest of the jobs' slabs).
int32_t launch_scan_queries(const vs::SlabView& v, const vs::ScanTuning& t, const void* d_queries,
                            size_t query_bytes, uint32_t n_q, const uint64_t* d_cand,
                            const vs::ScanJob* d_jobs, uint32_t k, uint64_t* d_partial, hipStream_t stream);
// The batched device search on an already registered snapshot (vs_search_batch_device's body).
int32_t launch_batch(vs_store* s, const vs::SlabView& v, const vs::ScanTuning& t, const void* d_queries,
                     uint32_t n_q, uint32_t k, const uint64_t* d_mask_words, void* d_workspace,
                     uint64_t* d_out_keys, hipStream_t stream);
uint64_t batch_workspace_bytes_for(vs_store* s, const vs::SlabView& v, uint32_t k, bool* served);

}  // namespace vsrt
