// Host side of libvs_amd.so: contexts, stores (row slabs in HBM), the workspace
// pool that makes searches re-entrant, the snapshot protocol between searches and
// the writer, and the C-ABI entry points declared in include/vs_abi.h. No
// computation happens here; every distance is produced by the kernels in
// vs_kernels.hip, and there is no CPU fallback.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>

#include "vs_internal.h"
#include "vs_roaring.h"

using vsrt::fail;
using vsrt::Snapshot;
using vsrt::Workspace;

namespace {

thread_local std::string g_err;
thread_local uint32_t g_dim_expected = 0, g_dim_received = 0;

// Scratch of the entry point that allocates nothing (vs_search_device_noalloc): the scan's
// per-workgroup partial lists. One copy per GPU, held until the process ends. Host code never
// uses this name as an address: vs_init resolves the device address of the current GPU's copy
// with hipGetSymbolAddress and keeps it in the context.
constexpr size_t NOALLOC_SCRATCH_BYTES = VS_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

}  // namespace

namespace vsrt {

int32_t fail(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

void set_dim_mismatch(uint32_t expected, uint32_t received) {
    g_dim_expected = expected;
    g_dim_received = received;
}

Workspace::~Workspace() {
    if (d_query) (void)hipFree(d_query);
    if (d_mask) (void)hipFree(d_mask);
    if (d_partial) (void)hipFree(d_partial);
    if (d_keys) (void)hipFree(d_keys);
    if (d_blob) (void)hipFree(d_blob);
    if (d_batch) (void)hipFree(d_batch);
    if (h_pinned) (void)hipHostFree(h_pinned);
    if (stream) (void)hipStreamDestroy(stream);
}

int32_t ws_acquire(vs_ctx* ctx, std::unique_ptr<Workspace>* out) {
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        if (!ctx->pool.empty()) {
            *out = std::move(ctx->pool.back());
            ctx->pool.pop_back();
            return VS_OK;
        }
    }
    std::unique_ptr<Workspace> ws(new Workspace());
    HIP_TRY(hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&ws->d_query, QUERY_BYTES));
    unsigned char* out_block = nullptr;
    HIP_TRY(hipMalloc(&out_block, (size_t)VS_MAX_K * 16 + 64));
    ws->d_keys = reinterpret_cast<uint64_t*>(out_block);
    ws->d_ids = reinterpret_cast<uint32_t*>(out_block + (size_t)VS_MAX_K * 8);
    ws->d_dist = reinterpret_cast<float*>(out_block + (size_t)VS_MAX_K * 12);
    ws->d_n = reinterpret_cast<uint32_t*>(out_block + (size_t)VS_MAX_K * 16);
    ws->pinned_bytes = QUERY_BYTES + RESULT_BYTES;
    HIP_TRY(hipHostMalloc(&ws->h_pinned, ws->pinned_bytes, hipHostMallocDefault));
    *out = std::move(ws);
    return VS_OK;
}

void ws_release(vs_ctx* ctx, std::unique_ptr<Workspace> ws) {
    std::unique_lock<std::mutex> g(ctx->mu);
    if (ctx->pool.size() < POOL_KEEP) {
        ctx->pool.push_back(std::move(ws));
        return;
    }
    g.unlock();
    ws.reset();  // beyond what the pool keeps: released outside the lock
}

vs::ScanTuning tuning_of(const vs_ctx* ctx) {
    vs::ScanTuning t{0, 1};
    t.blocks = ctx->tune_blocks.load(std::memory_order_relaxed);
    t.nt = ctx->tune_nt.load(std::memory_order_relaxed);
    t.rows_in_flight = ctx->tune_rows.load(std::memory_order_relaxed);
    t.query_lanes = ctx->tune_qlanes.load(std::memory_order_relaxed);
    return t;
}

int32_t check_store(vs_store* s) {
    if (!s) return fail(VS_E_INVALID_ARG, "null store handle");
    HIP_TRY(hipSetDevice(s->ctx->device));
    return VS_OK;
}

void snapshot_begin(vs_store* s, Snapshot* out) {
    std::unique_lock<std::mutex> lk(s->snap_mu);
    s->snap_cv.wait(lk, [&] { return !s->writer_blocks; });
    out->buf = s->active;
    ++s->readers[out->buf];
    out->v = s->view(out->buf);
}

void snapshot_end(vs_store* s, Snapshot* snap) {
    if (snap->buf < 0) return;
    std::lock_guard<std::mutex> lk(s->snap_mu);
    if (--s->readers[snap->buf] == 0) s->snap_cv.notify_all();
    snap->buf = -1;
}

int32_t snapshot_end_async(vs_store* s, Snapshot* snap, hipStream_t stream, bool track) {
    if (snap->buf < 0) return VS_OK;
    int32_t rc = VS_OK;
    std::lock_guard<std::mutex> lk(s->snap_mu);
    if (track) {
        // one event per (buffer, caller stream), re-recorded by every search: waiting for it
        // waits for the stream's latest search on this buffer
        hipEvent_t& ev = s->stream_events[snap->buf][stream];
        hipError_t e = hipSuccess;
        if (!ev) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(ev, stream);
        if (e != hipSuccess) rc = fail(VS_E_DEVICE, "cannot record the search's event: %s", hipGetErrorString(e));
    }
    if (--s->readers[snap->buf] == 0) s->snap_cv.notify_all();
    snap->buf = -1;
    return rc;
}

int32_t check_device_search(vs_store* s, uint32_t dim, uint32_t k) {
    if (!s->sealed.load()) return fail(VS_E_NOT_SEALED, "store %u was never sealed", s->index);
    if (dim != s->dim) {
        set_dim_mismatch(s->dim, dim);
        return fail(VS_E_DIM_MISMATCH, "invalid vector dimensions: expected %u, received %u", s->dim, dim);
    }
    if (k == 0 || k > VS_MAX_K) return fail(VS_E_INVALID_ARG, "k = %u out of range", k);
    return VS_OK;
}

int32_t launch_scan(const vs::SlabView& v, const vs::ScanTuning& t, const void* d_query, const uint64_t* d_cand,
                    uint32_t k, uint64_t* d_partial, uint64_t* d_keys, uint32_t* d_ids, float* d_dist,
                    uint32_t* d_n, hipStream_t stream) {
    if (v.metric == VS_HAMMING)
        HIP_TRY(vs::launch_scan_hamming(v, d_query, d_cand, k, d_partial, d_keys, d_ids, d_dist, d_n, t, stream));
    else
        HIP_TRY(vs::launch_scan_float(v, static_cast<const float*>(d_query), d_cand, k, d_partial, d_keys, d_ids,
                                      d_dist, d_n, t, stream));
    return VS_OK;
}

int32_t launch_scan_queries(const vs::SlabView& v, const vs::ScanTuning& t, const void* d_queries,
                            size_t query_bytes, uint32_t n_q, const uint64_t* d_cand,
                            const vs::ScanJob* d_jobs, uint32_t k, uint64_t* d_partial, hipStream_t stream) {
    if (v.metric == VS_HAMMING)
        HIP_TRY(vs::launch_scan_hamming_queries(v, d_queries, (uint32_t)(query_bytes / 16), n_q, d_cand, d_jobs, k,
                                                d_partial, t, stream));
    else
        HIP_TRY(vs::launch_scan_float_queries(v, static_cast<const float*>(d_queries), (uint32_t)(query_bytes / 4),
                                              n_q, d_cand, d_jobs, k, d_partial, t, stream));
    return VS_OK;
}

}  // namespace vsrt

namespace {

using namespace vsrt;

// Writer side: waits until no search registered on present buffer `buf` is left: the host-side
// count drains, then the events of device-API searches complete. Called with snap_mu held through
// `lk`; the lock is dropped while events are waited for (no new search can register on an inactive
// buffer, nor on any buffer while writer_blocks is set).
int32_t wait_readers(vs_store* s, int buf, std::unique_lock<std::mutex>& lk) {
    s->snap_cv.wait(lk, [&] { return s->readers[buf] == 0; });
    std::vector<hipEvent_t> evs;
    for (auto& kv : s->stream_events[buf]) evs.push_back(kv.second);
    lk.unlock();
    hipError_t bad = hipSuccess;
    for (hipEvent_t ev : evs) {
        const hipError_t e = hipEventSynchronize(ev);
        if (e != hipSuccess) bad = e;
    }
    lk.lock();
    if (bad != hipSuccess) return fail(VS_E_DEVICE, "waiting for searches in flight failed: %s", hipGetErrorString(bad));
    return VS_OK;
}

// Stops new searches and waits for every search in flight on either buffer. The caller clears
// writer_blocks (resume) when the store may be read again.
int32_t quiesce(vs_store* s) {
    std::unique_lock<std::mutex> lk(s->snap_mu);
    s->writer_blocks = true;
    if (int32_t rc = wait_readers(s, 0, lk)) return rc;
    return wait_readers(s, 1, lk);
}

void resume(vs_store* s) {
    std::lock_guard<std::mutex> lk(s->snap_mu);
    s->writer_blocks = false;
    s->snap_cv.notify_all();
}

uint32_t vecs_for(uint32_t dim, int32_t dtype) {
    if (dtype == VS_F32) return (dim + 3) / 4;
    if (dtype == VS_BF16) return (dim + 7) / 8;
    return (dim + 127) / 128;
}

// Common front half of the host-buffer searches: argument checks.
int32_t begin_search(vs_store* s, const float* query, uint32_t dim, uint32_t k, uint32_t* out_ids,
                     float* out_dist, uint32_t* out_n) {
    if (int32_t rc = check_store(s)) return rc;
    if (!query || !out_n || (k && (!out_ids || !out_dist)))
        return fail(VS_E_INVALID_ARG, "null query or output buffer");
    if (!s->sealed.load()) return fail(VS_E_NOT_SEALED, "store %u was never sealed", s->index);
    if (dim != s->dim) {
        set_dim_mismatch(s->dim, dim);
        return fail(VS_E_DIM_MISMATCH, "invalid vector dimensions: expected %u, received %u", s->dim,
                    dim);
    }
    if (k > VS_MAX_K) return fail(VS_E_INVALID_ARG, "k = %u exceeds VS_MAX_K = %u", k, VS_MAX_K);
    return VS_OK;
}

// Whether a batch on this store runs the matrix-core scan.
bool on_matrix_cores(const vs_store* s, uint32_t k) {
    return s->dtype == VS_BF16 && s->metric == VS_ANGULAR && k >= 1 && k <= VS_BATCH_MAX_K;
}

// Batches outside the bf16 matrix-core path. Queues the scans of `n` queries (device memory,
// `stride` floats apart) under one candidate mask: 32 at a time on the f32 matrix cores while
// more than 16 are left and that scan serves the store, then 8 or 4 at a time sharing plain
// passes. Outputs per query: k keys at keys + i * k (nullable) and/or ids, distances, counts.
// The caller has checked shared_pass_serves().
bool shared_pass_serves(const vs::SlabView& v, uint32_t k) { return vs::scan_multi_serves(v, k); }
constexpr uint32_t MQ32_MIN_QUERIES = 17;
// sized for the largest grid a tuning may ask for (the workspace may outlive a tuning change)
uint64_t shared_pass_partial_bytes(const vs::SlabView& v, uint32_t k) {
    const uint64_t a = vs::scan_workspace_bytes(v, k) * vs::scan_multi_queries(v);
    const uint64_t b = vs::scan_mq32_serves(v, k) ? vs::scan_mq32_partial_bytes(v, k) : 0;
    return a > b ? a : b;
}
int32_t queue_shared_passes(const vs::SlabView& v, const vs::ScanTuning& t, const float* d_q, uint32_t stride,
                            uint32_t n, const uint64_t* d_cand, uint32_t k, uint64_t* d_partial,
                            uint64_t* keys, uint32_t* ids, float* dist, uint32_t* counts,
                            hipStream_t stream) {
    const uint32_t small = vs::scan_multi_queries(v);
    const bool mfma = vs::scan_mq32_serves(v, k);
    for (uint32_t i = 0; i < n;) {
        const uint32_t left = n - i;
        const float* q = d_q + (size_t)i * stride;
        uint64_t* ko = keys ? keys + (size_t)i * k : nullptr;
        uint32_t* io = ids ? ids + (size_t)i * k : nullptr;
        float* fo = ids ? dist + (size_t)i * k : nullptr;
        uint32_t* no = ids ? counts + i : nullptr;
        if (mfma && left >= MQ32_MIN_QUERIES) {
            const uint32_t m = left < 32 ? left : 32;
            HIP_TRY(vs::launch_scan_mq32(v, q, stride, m, d_cand, k, d_partial, ko, io, fo, no, stream));
            i += m;
        } else {
            const uint32_t m = left < small ? left : small;
            HIP_TRY(vs::launch_scan_multi(v, q, stride, m, d_cand, k, d_partial, ko, io, fo, no, t, stream));
            i += m;
        }
    }
    return VS_OK;
}

// Runs one query whose candidate mask (aligned to the store, or null) is already
// on the device, and copies the decoded result back.
int32_t run_and_fetch(vs_store* s, const vs::SlabView& v, Workspace* ws, const float* h_query, uint32_t k,
                      const uint64_t* d_cand, uint32_t* out_ids, float* out_dist, uint32_t* out_n) {
    const vs::ScanTuning t = tuning_of(s->ctx);
    if (int32_t rc = grow(&ws->d_partial, &ws->partial_bytes, (size_t)vs::scan_partial_bytes(v, k, t)))
        return rc;
    size_t query_bytes;
    if (s->dtype == VS_BIN1) {
        // the query is quantised like the rows: bit i = (q[i] > 0), zero padded to whole vectors
        uint64_t* hq = reinterpret_cast<uint64_t*>(ws->h_pinned);
        query_bytes = (size_t)s->row_vecs * 16;
        memset(hq, 0, query_bytes);
        for (uint32_t i = 0; i < s->dim; ++i)
            if (h_query[i] > 0.0f) hq[i >> 6] |= 1ull << (i & 63);
    } else {
        query_bytes = (size_t)s->dim * 4;
        memcpy(ws->h_pinned, h_query, query_bytes);
    }
    HIP_TRY(hipMemcpyAsync(ws->d_query, ws->h_pinned, query_bytes, hipMemcpyHostToDevice, ws->stream));
    unsigned char* hr = ws->h_pinned + QUERY_BYTES;
    // The merge writes the k results straight into the pinned host block (it is mapped into
    // the device's address space): no copy back, one wait.
    if (int32_t rc = launch_scan(v, t, ws->d_query, d_cand, k, ws->d_partial, nullptr,
                                 reinterpret_cast<uint32_t*>(hr),
                                 reinterpret_cast<float*>(hr + (size_t)VS_MAX_K * 4),
                                 reinterpret_cast<uint32_t*>(hr + (size_t)VS_MAX_K * 8), ws->stream))
        return rc;
    HIP_TRY(hipStreamSynchronize(ws->stream));
    uint32_t n;
    memcpy(&n, hr + (size_t)VS_MAX_K * 8, 4);
    if (n > k) return fail(VS_E_DEVICE, "device returned %u results for k = %u", n, k);
    memcpy(out_ids, hr, (size_t)n * 4);
    memcpy(out_dist, hr + (size_t)VS_MAX_K * 4, (size_t)n * 4);
    *out_n = n;
    return VS_OK;
}

// Batched matrix-core path for bf16 stores: chunks of up to 256 queries, each one
// pass over the rows.
int32_t run_batch_and_fetch(vs_store* s, const vs::SlabView& v, Workspace* ws, const float* h_queries,
                            uint32_t n_q, uint32_t k, const uint64_t* d_cand, uint32_t* out_ids,
                            float* out_dist, uint32_t* out_n) {
    const size_t chunk_q = VS_BATCH_MAX_QUERIES;
    const size_t q_bytes = chunk_q * s->dim * 4;
    const size_t res_bytes = chunk_q * k * 8 + chunk_q * 4;  // ids, distances, counts
    // Rows of up to 128 elements (two steps per tile): the scan multiplies whole 256-row tiles, so
    // under a filter that keeps at most a quarter of the rows the candidates are gathered into a
    // dense slab first (it then reads the candidate bytes three times instead of every row once).
    // Their number decides, so it is counted on the device: one small launch and one wait per
    // call. Longer rows need none of this: the scan fetches the candidates where they lie.
    uint64_t n_cand = 0;
    static const bool gather_allowed = getenv("VS_BATCH_NO_GATHER") == nullptr;  // measurement switch
    if (d_cand && gather_allowed && !vs::batch_filters_in_place(v)) {
        if (int32_t rc = grow(&ws->d_partial, &ws->partial_bytes, (size_t)vs::count_candidates_workspace_bytes(v)))
            return rc;
        HIP_TRY(vs::launch_count_candidates(v, d_cand, reinterpret_cast<uint32_t*>(ws->d_partial), ws->d_n, ws->stream));
        HIP_TRY(hipMemcpyAsync(ws->h_pinned, ws->d_n, 4, hipMemcpyDeviceToHost, ws->stream));
        HIP_TRY(hipStreamSynchronize(ws->stream));
        uint32_t c;
        memcpy(&c, ws->h_pinned, 4);
        n_cand = c;
    }
    const bool gather = n_cand > 0 && n_cand * 4 <= v.n_slots;
    const size_t ws_bytes = gather ? (size_t)vs::batch_gather_workspace_bytes(v, n_cand, k)
                                   : (size_t)vs::batch_workspace_bytes(v, k);
    // one device block: queries | kernel workspace | results
    const size_t off_ws = (q_bytes + 255) / 256 * 256, off_res = off_ws + (ws_bytes + 255) / 256 * 256;
    if (int32_t rc = grow(&ws->d_batch, &ws->batch_bytes, off_res + res_bytes)) return rc;
    std::vector<unsigned char> host(res_bytes);
    for (uint32_t at = 0; at < n_q; at += (uint32_t)chunk_q) {
        const uint32_t m = n_q - at < chunk_q ? n_q - at : (uint32_t)chunk_q;
        float* d_q = reinterpret_cast<float*>(ws->d_batch);
        uint32_t* d_ids = reinterpret_cast<uint32_t*>(ws->d_batch + off_res);
        float* d_dist = reinterpret_cast<float*>(ws->d_batch + off_res + chunk_q * k * 4);
        uint32_t* d_n = reinterpret_cast<uint32_t*>(ws->d_batch + off_res + chunk_q * k * 8);
        HIP_TRY(hipMemcpyAsync(d_q, h_queries + (size_t)at * s->dim, (size_t)m * s->dim * 4,
                               hipMemcpyHostToDevice, ws->stream));
        if (gather)
            HIP_TRY(vs::launch_scan_batch_bf16_gathered(v, d_q, m, d_cand, n_cand, k, ws->d_batch + off_ws, nullptr,
                                                        d_ids, d_dist, d_n, ws->stream));
        else
            HIP_TRY(vs::launch_scan_batch_bf16(v, d_q, m, d_cand, k, ws->d_batch + off_ws, nullptr, d_ids,
                                               d_dist, d_n, ws->stream));
        HIP_TRY(hipMemcpyAsync(host.data(), ws->d_batch + off_res, res_bytes, hipMemcpyDeviceToHost,
                               ws->stream));
        HIP_TRY(hipStreamSynchronize(ws->stream));
        const uint32_t* h_ids = reinterpret_cast<const uint32_t*>(host.data());
        const float* h_dist = reinterpret_cast<const float*>(host.data() + chunk_q * k * 4);
        const uint32_t* h_n = reinterpret_cast<const uint32_t*>(host.data() + chunk_q * k * 8);
        for (uint32_t i = 0; i < m; ++i) {
            const uint32_t n = h_n[i];
            if (n > k) return fail(VS_E_DEVICE, "device returned %u results for k = %u", n, k);
            memcpy(out_ids + (size_t)(at + i) * k, h_ids + (size_t)i * k, (size_t)n * 4);
            memcpy(out_dist + (size_t)(at + i) * k, h_dist + (size_t)i * k, (size_t)n * 4);
            out_n[at + i] = n;
        }
    }
    return VS_OK;
}

// Makes the workspace's pinned host block at least `bytes` long (contents are not kept).
int32_t grow_pinned(Workspace* ws, size_t bytes) {
    if (ws->pinned_bytes >= bytes) return VS_OK;
    unsigned char* fresh = nullptr;
    HIP_TRY(hipHostMalloc(&fresh, bytes, hipHostMallocDefault));
    if (ws->h_pinned) (void)hipHostFree(ws->h_pinned);
    ws->h_pinned = fresh;
    ws->pinned_bytes = bytes;
    return VS_OK;
}

// Packs one host query the way the device takes it: f32 as it is, binary stores quantised like the
// rows (bit j = (q[j] > 0), zero padded to whole vectors).
void pack_query(const vs_store* s, const float* q, unsigned char* dst, size_t q_bytes) {
    if (s->dtype == VS_BIN1) {
        uint64_t* w = reinterpret_cast<uint64_t*>(dst);
        memset(w, 0, q_bytes);
        for (uint32_t j = 0; j < s->dim; ++j)
            if (q[j] > 0.0f) w[j >> 6] |= 1ull << (j & 63);
    } else {
        memcpy(dst, q, q_bytes);
    }
}

// `n_q` independent single-query searches, host queries and host results, with ONE scan launch, one
// merge launch and one wait per chunk of queries: query i runs under d_cand_of[i] (device words
// aligned with the store, nullable; the array itself is host memory) if `d_cand_of` is given, else
// under d_cand (nullable). No row read is shared, so result i is bit for bit vs_search's.
int32_t run_queries_and_fetch(vs_store* s, const vs::SlabView& v, Workspace* ws, const float* h_queries,
                              uint32_t n_q, uint32_t k, const uint64_t* d_cand,
                              const uint64_t* const* d_cand_of, uint32_t* out_ids, float* out_dist,
                              uint32_t* out_n) {
    const vs::ScanTuning t = tuning_of(s->ctx);
    const bool masked = d_cand != nullptr || d_cand_of != nullptr;
    const uint32_t blocks = vs::scan_grid_blocks(v, t, masked);
    const size_t per_query = (size_t)blocks * k * 8;
    // queries per launch: the partial lists of a chunk stay within 64 MiB
    size_t chunk_q = (64u << 20) / per_query;
    chunk_q = chunk_q < 1 ? 1 : chunk_q > 256 ? 256 : chunk_q;
    if (chunk_q > n_q) chunk_q = n_q;
    if (int32_t rc = grow(&ws->d_partial, &ws->partial_bytes, chunk_q * per_query)) return rc;
    const size_t q_bytes = s->dtype == VS_BIN1 ? (size_t)s->row_vecs * 16 : (size_t)s->dim * 4;
    const size_t q_stride = (q_bytes + 255) / 256 * 256;
    // one device block and one host block: queries | the queries' jobs | ids, distances, counts
    const size_t off_tab = chunk_q * q_stride,
                 off_res = off_tab + (chunk_q * sizeof(vs::ScanJob) + 255) / 256 * 256;
    const size_t res_bytes = chunk_q * k * 8 + chunk_q * 4;
    if (int32_t rc = grow(&ws->d_batch, &ws->batch_bytes, off_res + res_bytes)) return rc;
    std::vector<unsigned char> up(off_res), down(res_bytes);
    for (uint32_t at = 0; at < n_q; at += (uint32_t)chunk_q) {
        const uint32_t m = n_q - at < chunk_q ? n_q - at : (uint32_t)chunk_q;
        for (uint32_t i = 0; i < m; ++i)
            pack_query(s, h_queries + (size_t)(at + i) * s->dim, up.data() + i * q_stride, q_bytes);
        if (d_cand_of) {  // every query over this store, each under its own mask
            vs::ScanJob* jobs = reinterpret_cast<vs::ScanJob*>(up.data() + off_tab);
            for (uint32_t i = 0; i < m; ++i)
                jobs[i] = vs::ScanJob{v.rows, v.norms, v.present, d_cand_of[at + i], nullptr, v.n_slots, v.docid_base, 0};
        }
        unsigned char* d_res = ws->d_batch + off_res;
        uint32_t* d_ids = reinterpret_cast<uint32_t*>(d_res);
        float* d_dist = reinterpret_cast<float*>(d_res + chunk_q * k * 4);
        uint32_t* d_n = reinterpret_cast<uint32_t*>(d_res + chunk_q * k * 8);
        HIP_TRY(hipMemcpyAsync(ws->d_batch, up.data(), off_tab + (d_cand_of ? (size_t)m * sizeof(vs::ScanJob) : 0),
                               hipMemcpyHostToDevice, ws->stream));
        if (int32_t rc = launch_scan_queries(
                v, t, ws->d_batch, q_stride, m, d_cand,
                d_cand_of ? reinterpret_cast<const vs::ScanJob*>(ws->d_batch + off_tab) : nullptr, k, ws->d_partial,
                ws->stream))
            return rc;
        HIP_TRY(vs::launch_merge_batch(ws->d_partial, blocks, 1, blocks, m, (uint64_t)blocks * k, k, v.metric, nullptr,
                                       d_ids, d_dist, d_n, ws->stream));
        HIP_TRY(hipMemcpyAsync(down.data(), d_res, res_bytes, hipMemcpyDeviceToHost, ws->stream));
        HIP_TRY(hipStreamSynchronize(ws->stream));
        const uint32_t* h_ids = reinterpret_cast<const uint32_t*>(down.data());
        const float* h_dist = reinterpret_cast<const float*>(down.data() + chunk_q * k * 4);
        const uint32_t* h_n = reinterpret_cast<const uint32_t*>(down.data() + chunk_q * k * 8);
        for (uint32_t i = 0; i < m; ++i) {
            const uint32_t n = h_n[i];
            if (n > k) return fail(VS_E_DEVICE, "device returned %u results for k = %u", n, k);
            memcpy(out_ids + (size_t)(at + i) * k, h_ids + (size_t)i * k, (size_t)n * 4);
            memcpy(out_dist + (size_t)(at + i) * k, h_dist + (size_t)i * k, (size_t)n * 4);
            out_n[at + i] = n;
        }
    }
    return VS_OK;
}

// Several queries under one candidate mask outside the matrix-core path (f32 and binary stores,
// other metrics, larger k): 32, eight or four queries per pass over the rows where a multi-query
// scan serves the store, else one scan each; the launches of a chunk are queued back to back on the
// workspace's stream, with ONE upload of the queries, ONE download of the results and ONE wait
// per chunk.
int32_t run_many_and_fetch(vs_store* s, const vs::SlabView& v, Workspace* ws, const float* h_queries,
                           uint32_t n_q, uint32_t k, const uint64_t* d_cand, uint32_t* out_ids,
                           float* out_dist, uint32_t* out_n) {
    const vs::ScanTuning t = tuning_of(s->ctx);
    // several queries share every pass over the rows where one of the multi-query scans serves the store
    const bool shared = shared_pass_serves(v, k);
    const bool shared_bits = vs::scan_hamming_multi_serves(v, k);  // binary rows: four queries per pass
    if (!shared && !shared_bits)  // (larger k, longer rows): one scan per query, all in one launch
        return run_queries_and_fetch(s, v, ws, h_queries, n_q, k, d_cand, nullptr, out_ids, out_dist, out_n);
    if (int32_t rc = grow(&ws->d_partial, &ws->partial_bytes,
                          shared ? (size_t)shared_pass_partial_bytes(v, k) : (size_t)vs::scan_workspace_bytes(v, k) * 4))
        return rc;
    const size_t chunk_q = 64;
    const bool bin = s->dtype == VS_BIN1;
    const size_t q_bytes = bin ? (size_t)s->row_vecs * 16 : (size_t)s->dim * 4;  // per query, on the device
    const size_t q_stride = (q_bytes + 255) / 256 * 256;
    const size_t res_bytes = chunk_q * k * 8 + chunk_q * 4;  // ids, distances, counts
    if (int32_t rc = grow(&ws->d_batch, &ws->batch_bytes, chunk_q * q_stride + res_bytes)) return rc;
    std::vector<unsigned char> hq(chunk_q * q_stride), hr(res_bytes);
    for (uint32_t at = 0; at < n_q; at += (uint32_t)chunk_q) {
        const uint32_t m = n_q - at < chunk_q ? n_q - at : (uint32_t)chunk_q;
        for (uint32_t i = 0; i < m; ++i) {
            pack_query(s, h_queries + (size_t)(at + i) * s->dim, hq.data() + i * q_stride, q_bytes);
        }
        unsigned char* d_res = ws->d_batch + chunk_q * q_stride;
        uint32_t* d_ids = reinterpret_cast<uint32_t*>(d_res);
        float* d_dist = reinterpret_cast<float*>(d_res + chunk_q * k * 4);
        uint32_t* d_n = reinterpret_cast<uint32_t*>(d_res + chunk_q * k * 8);
        HIP_TRY(hipMemcpyAsync(ws->d_batch, hq.data(), (size_t)m * q_stride, hipMemcpyHostToDevice, ws->stream));
        if (shared) {
            if (int32_t rc = queue_shared_passes(v, t, reinterpret_cast<const float*>(ws->d_batch),
                                                 (uint32_t)(q_stride / 4), m, d_cand, k, ws->d_partial, nullptr,
                                                 d_ids, d_dist, d_n, ws->stream))
                return rc;
        } else {
            for (uint32_t i = 0; i < m; i += 4)
                HIP_TRY(vs::launch_scan_hamming_multi(v, ws->d_batch + i * q_stride, (uint32_t)(q_stride / 16),
                                                      m - i < 4 ? m - i : 4, d_cand, k, ws->d_partial, nullptr,
                                                      d_ids + (size_t)i * k, d_dist + (size_t)i * k, d_n + i, t,
                                                      ws->stream));
        }
        HIP_TRY(hipMemcpyAsync(hr.data(), d_res, res_bytes, hipMemcpyDeviceToHost, ws->stream));
        HIP_TRY(hipStreamSynchronize(ws->stream));
        const uint32_t* h_ids = reinterpret_cast<const uint32_t*>(hr.data());
        const float* h_dist = reinterpret_cast<const float*>(hr.data() + chunk_q * k * 4);
        const uint32_t* h_n = reinterpret_cast<const uint32_t*>(hr.data() + chunk_q * k * 8);
        for (uint32_t i = 0; i < m; ++i) {
            const uint32_t n = h_n[i];
            if (n > k) return fail(VS_E_DEVICE, "device returned %u results for k = %u", n, k);
            memcpy(out_ids + (size_t)(at + i) * k, h_ids + (size_t)i * k, (size_t)n * 4);
            memcpy(out_dist + (size_t)(at + i) * k, h_dist + (size_t)i * k, (size_t)n * 4);
            out_n[at + i] = n;
        }
    }
    return VS_OK;
}

// Copies the caller's docid-indexed mask into `d_words`: n_words words covering the docids from
// docid_base (a multiple of 64) on.
int32_t copy_dense_window(hipStream_t stream, uint64_t* d_words, uint32_t docid_base, uint64_t n_words,
                          const uint64_t* mask_words, uint64_t n_mask_words) {
    const uint64_t w0 = docid_base / 64;
    const uint64_t avail = n_mask_words > w0 ? n_mask_words - w0 : 0;
    const uint64_t ncopy = avail < n_words ? avail : n_words;
    if (ncopy)
        HIP_TRY(hipMemcpyAsync(d_words, mask_words + w0, (size_t)ncopy * 8, hipMemcpyHostToDevice, stream));
    if (ncopy < n_words) HIP_TRY(hipMemsetAsync(d_words + ncopy, 0, (size_t)(n_words - ncopy) * 8, stream));
    return VS_OK;
}
// ... aligned to the store (n_slots / 64 words)
int32_t copy_dense_mask(vs_store* s, hipStream_t stream, uint64_t* d_words, const uint64_t* mask_words,
                        uint64_t n_mask_words) {
    return copy_dense_window(stream, d_words, s->docid_base, s->n_slots / 64, mask_words, n_mask_words);
}
// ... into ws->d_mask
int32_t stage_dense_mask(vs_store* s, Workspace* ws, const uint64_t* mask_words,
                         uint64_t n_mask_words) {
    if (int32_t rc = grow(&ws->d_mask, &ws->mask_words, (size_t)(s->n_slots / 64) * 8)) return rc;
    return copy_dense_mask(s, ws->stream, ws->d_mask, mask_words, n_mask_words);
}

// Expands a parsed serialised RoaringBitmap into `d_words`: n_words words covering the docids from
// docid_base on.
int32_t expand_roaring_window(Workspace* ws, const uint8_t* roaring_bytes, size_t len,
                              const std::vector<vs::RoaringContainerDesc>& desc, uint32_t docid_base,
                              uint64_t n_words, uint64_t* d_words) {
    HIP_TRY(hipMemsetAsync(d_words, 0, (size_t)n_words * 8, ws->stream));
    if (desc.empty()) return VS_OK;
    const size_t blob = (len + 15) / 16 * 16;
    const size_t desc_b = desc.size() * sizeof(vs::RoaringContainerDesc);
    if (int32_t rc = grow(&ws->d_blob, &ws->blob_bytes, blob + desc_b)) return rc;
    HIP_TRY(hipMemcpyAsync(ws->d_blob, roaring_bytes, len, hipMemcpyHostToDevice, ws->stream));
    HIP_TRY(hipMemcpyAsync(ws->d_blob + blob, desc.data(), desc_b, hipMemcpyHostToDevice, ws->stream));
    HIP_TRY(vs::launch_roaring_expand(ws->d_blob,
                                      reinterpret_cast<const vs::RoaringContainerDesc*>(ws->d_blob + blob),
                                      (uint32_t)desc.size(), docid_base, d_words, n_words, ws->stream));
    return VS_OK;
}
// ... aligned with the store (n_slots / 64 words)
int32_t expand_roaring(vs_store* s, Workspace* ws, const uint8_t* roaring_bytes, size_t len,
                       const std::vector<vs::RoaringContainerDesc>& desc, uint64_t* d_words) {
    return expand_roaring_window(ws, roaring_bytes, len, desc, s->docid_base, s->n_slots / 64, d_words);
}

// A host-buffer search from begin to end: workspace, snapshot, body, release. `body` gets the
// workspace and the registered view and has waited for the workspace's stream when it returns
// VS_OK.
template <typename F>
int32_t with_snapshot(vs_store* s, F&& body) {
    std::unique_ptr<Workspace> ws;
    if (int32_t rc = ws_acquire(s->ctx, &ws)) return rc;
    Snapshot snap;
    snapshot_begin(s, &snap);
    const int32_t rc = body(ws.get(), snap.v);
    if (rc != VS_OK) (void)hipStreamSynchronize(ws->stream);  // nothing of a failed call stays in flight
    snapshot_end(s, &snap);
    ws_release(s->ctx, std::move(ws));
    return rc;
}

}  // namespace

// ---- context -------------------------------------------------------------------

extern "C" int32_t vs_init(int32_t device_ordinal, vs_ctx** out) {
    if (!out) return fail(VS_E_INVALID_ARG, "null out pointer");
    *out = nullptr;
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(VS_E_DEVICE, "no HIP device: %s (this library has no CPU path)",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device_ordinal < 0 || device_ordinal >= n)
        return fail(VS_E_INVALID_ARG, "device %d out of range (%d devices)", device_ordinal, n);
    HIP_TRY(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(VS_E_DEVICE, "device %d is %s; the kernels are built for gfx950 only",
                    device_ordinal, prop.gcnArchName);
    vs::set_device_cus((uint32_t)prop.multiProcessorCount);  // grids go by the CUs this process can use
    std::unique_ptr<vs_ctx> ctx(new vs_ctx());
    ctx->device = device_ordinal;
    auto env_int = [](const char* name, int32_t lo, int32_t hi, std::atomic<int32_t>* dst) {
        if (const char* v = getenv(name)) {
            const int32_t x = atoi(v);
            if (x >= lo && x <= hi) dst->store(x);
        }
    };
    env_int("VS_SCAN_BLOCKS", 0, vs::MAX_TUNED_BLOCKS, &ctx->tune_blocks);
    env_int("VS_SCAN_NT", 0, 1, &ctx->tune_nt);
    env_int("VS_SCAN_ROWS", 0, 16, &ctx->tune_rows);
    env_int("VS_SCAN_QLANES", 0, 16, &ctx->tune_qlanes);
    HIP_TRY(hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    HIP_TRY(hipHostMalloc(&ctx->h_small, 64 << 10, hipHostMallocDefault));
    // device `device_ordinal` is current: this is its copy of the array
    HIP_TRY(hipGetSymbolAddress(reinterpret_cast<void**>(&ctx->d_scratch), HIP_SYMBOL(scratch)));
    if (!ctx->d_scratch) return fail(VS_E_DEVICE, "no device address for the static scratch array");
    *out = ctx.release();
    return VS_OK;
}

extern "C" int32_t vs_shutdown(vs_ctx* ctx) {
    if (!ctx) return VS_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    std::vector<vs_store*> stores;
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        for (auto& kv : ctx->stores) stores.push_back(kv.second);
    }
    for (vs_store* s : stores) vs_store_destroy(s);
    ctx->pool.clear();
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    if (ctx->upload_stream) (void)hipStreamDestroy(ctx->upload_stream);
    delete ctx;
    return VS_OK;
}

extern "C" int32_t vs_last_error(vs_ctx*, char* buf, size_t cap) {
    if (!buf || !cap) return VS_E_INVALID_ARG;
    snprintf(buf, cap, "%s", g_err.c_str());
    return VS_OK;
}

extern "C" int32_t vs_last_dim_mismatch(vs_ctx*, uint32_t* expected, uint32_t* received) {
    if (expected) *expected = g_dim_expected;
    if (received) *received = g_dim_received;
    return VS_OK;
}

extern "C" int32_t vs_set_scan_tuning(vs_ctx* ctx, int32_t blocks, int32_t nt) {
    if (!ctx) return fail(VS_E_INVALID_ARG, "null context");
    if (blocks < 0 || blocks > vs::MAX_TUNED_BLOCKS)
        return fail(VS_E_INVALID_ARG, "blocks = %d out of range (0 .. %d)", blocks, vs::MAX_TUNED_BLOCKS);
    ctx->tune_blocks.store(blocks);
    ctx->tune_nt.store(nt ? 1 : 0);
    return VS_OK;
}

// ---- stores ------------------------------------------------------------------------

extern "C" int32_t vs_store_create(vs_ctx* ctx, uint16_t store_index, uint32_t dim, int32_t dtype,
                                   int32_t metric, uint32_t docid_base, uint64_t capacity,
                                   vs_store** out) {
    if (!ctx || !out) return fail(VS_E_INVALID_ARG, "null context or out pointer");
    *out = nullptr;
    HIP_TRY(hipSetDevice(ctx->device));
    if (dim == 0) return fail(VS_E_INVALID_ARG, "dimension 0");
    if (dtype != VS_F32 && dtype != VS_BF16 && dtype != VS_BIN1)
        return fail(VS_E_INVALID_ARG, "unknown element type %d", dtype);
    if (metric != VS_ANGULAR && metric != VS_HAMMING && metric != VS_EUCLIDEAN && metric != VS_DOT)
        return fail(VS_E_INVALID_ARG, "unknown metric %d", metric);
    if ((dtype == VS_BIN1) != (metric == VS_HAMMING))
        return fail(VS_E_INVALID_ARG,
                    "bin1 stores use VS_HAMMING; f32/bf16 stores VS_ANGULAR, VS_EUCLIDEAN or VS_DOT");
    if (docid_base % 64) return fail(VS_E_INVALID_ARG, "docid_base must be a multiple of 64");
    if (capacity == 0 || (uint64_t)docid_base + capacity > (1ull << 32))
        return fail(VS_E_INVALID_ARG, "docid range [%u, %u + %llu) does not fit u32 docids", docid_base,
                    docid_base, (unsigned long long)capacity);
    const uint32_t row_vecs = vecs_for(dim, dtype);
    if (row_vecs > (dtype == VS_BIN1 ? MAX_BIN_VECS : MAX_DIM_VECS))
        return fail(VS_E_INVALID_ARG, "dimension %u too large (row of %u bytes, limit %u)", dim,
                    row_vecs * 16, (dtype == VS_BIN1 ? MAX_BIN_VECS : MAX_DIM_VECS) * 16);
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        if (ctx->stores.count(store_index))
            return fail(VS_E_INVALID_ARG, "store %u already exists", store_index);
    }
    std::unique_ptr<vs_store> s(new vs_store());
    static std::atomic<uint64_t> next_serial{1};
    s->serial = next_serial.fetch_add(1);
    s->ctx = ctx;
    s->index = store_index;
    s->dim = dim;
    s->dtype = dtype;
    s->metric = metric;
    s->docid_base = docid_base;
    s->capacity = capacity;
    s->n_slots = (capacity + 63) / 64 * 64;
    s->row_vecs = row_vecs;
    const size_t n_words = s->n_slots / 64;
    auto body = [&]() -> int32_t {
        // 192 zeroed rows of padding: the batched scan reads whole 256-row tiles
        HIP_TRY(hipMalloc(&s->d_rows, (size_t)(s->n_slots + vs::SLAB_PAD_ROWS) * s->row_bytes()));
        HIP_TRY(hipMemset(static_cast<unsigned char*>(s->d_rows) + (size_t)s->n_slots * s->row_bytes(), 0,
                          (size_t)vs::SLAB_PAD_ROWS * s->row_bytes()));
        HIP_TRY(hipMalloc(&s->d_present[0], n_words * 8));
        HIP_TRY(hipMalloc(&s->d_present[1], n_words * 8));
        HIP_TRY(hipMemset(s->d_present[0], 0, n_words * 8));
        HIP_TRY(hipMemset(s->d_present[1], 0, n_words * 8));
        if (metric == VS_ANGULAR) {
            HIP_TRY(hipMalloc(&s->d_norms, (size_t)s->n_slots * 4));
            HIP_TRY(hipMemset(s->d_norms, 0, (size_t)s->n_slots * 4));
        }
        return VS_OK;
    };
    if (int32_t rc = body()) {
        if (s->d_rows) (void)hipFree(s->d_rows);
        if (s->d_norms) (void)hipFree(s->d_norms);
        for (int i = 0; i < 2; ++i)
            if (s->d_present[i]) (void)hipFree(s->d_present[i]);
        return rc;
    }
    s->h_present.assign(n_words, 0);
    s->h_sealed.assign(n_words, 0);
    s->h_visible.assign(n_words, 0);
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        ctx->stores[store_index] = s.get();
    }
    *out = s.release();
    return VS_OK;
}

extern "C" int32_t vs_store_reserve(vs_store* s, uint64_t capacity) {
    if (int32_t rc = check_store(s)) return rc;
    std::lock_guard<std::mutex> g(s->ctx->upload_mu);
    if (capacity <= s->capacity) return VS_OK;
    if ((uint64_t)s->docid_base + capacity > (1ull << 32))
        return fail(VS_E_INVALID_ARG, "docid range [%u, %u + %llu) does not fit u32 docids", s->docid_base,
                    s->docid_base, (unsigned long long)capacity);
    for (auto& kv : s->upload_events) HIP_TRY(hipEventSynchronize(kv.second));
    const uint64_t old_slots = s->n_slots, new_slots = (capacity + 63) / 64 * 64;
    const size_t rb = s->row_bytes(), old_words = old_slots / 64, new_words = new_slots / 64;
    void* rows = nullptr;
    float* norms = nullptr;
    uint64_t* present[2] = {nullptr, nullptr};
    hipStream_t st = s->ctx->upload_stream;
    // the copies read rows and bitmaps no writer is changing (this thread is the writer);
    // searches go on until the pointers are swapped
    auto body = [&]() -> int32_t {
        HIP_TRY(hipMalloc(&rows, (size_t)(new_slots + vs::SLAB_PAD_ROWS) * rb));
        HIP_TRY(hipMemcpyAsync(rows, s->d_rows, (size_t)old_slots * rb, hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemsetAsync(static_cast<unsigned char*>(rows) + (size_t)old_slots * rb, 0,
                               (size_t)(new_slots - old_slots + vs::SLAB_PAD_ROWS) * rb, st));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipMalloc(&present[i], new_words * 8));
            HIP_TRY(hipMemcpyAsync(present[i], s->d_present[i], old_words * 8, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemsetAsync(present[i] + old_words, 0, (new_words - old_words) * 8, st));
        }
        if (s->d_norms) {
            HIP_TRY(hipMalloc(&norms, (size_t)new_slots * 4));
            HIP_TRY(hipMemcpyAsync(norms, s->d_norms, (size_t)old_slots * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemsetAsync(norms + old_slots, 0, (size_t)(new_slots - old_slots) * 4, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
        return VS_OK;
    };
    int32_t rc = body();
    if (rc == VS_OK) rc = quiesce(s);  // searches in flight still read the old buffers
    if (rc != VS_OK) {
        resume(s);
        if (rows) (void)hipFree(rows);
        if (norms) (void)hipFree(norms);
        for (int i = 0; i < 2; ++i)
            if (present[i]) (void)hipFree(present[i]);
        return rc;
    }
    void* old_rows = s->d_rows;
    float* old_norms = s->d_norms;
    uint64_t* old_present[2] = {s->d_present[0], s->d_present[1]};
    s->d_rows = rows;
    if (s->d_norms) s->d_norms = norms;
    for (int i = 0; i < 2; ++i) s->d_present[i] = present[i];
    s->h_present.resize(new_words, 0);
    s->h_sealed.resize(new_words, 0);
    s->h_visible.resize(new_words, 0);
    s->capacity = capacity;
    s->n_slots = new_slots;
    resume(s);
    (void)hipFree(old_rows);
    if (old_norms) (void)hipFree(old_norms);
    for (int i = 0; i < 2; ++i) (void)hipFree(old_present[i]);
    return VS_OK;
}

extern "C" int32_t vs_store_open(vs_ctx* ctx, uint16_t store_index, vs_store** out) {
    if (!ctx || !out) return fail(VS_E_INVALID_ARG, "null context or out pointer");
    *out = nullptr;
    std::lock_guard<std::mutex> g(ctx->mu);
    auto it = ctx->stores.find(store_index);
    if (it == ctx->stores.end() || !it->second->sealed.load())
        return fail(VS_E_NO_STORE, "missing metadata for store %u", store_index);
    *out = it->second;
    return VS_OK;
}

extern "C" int32_t vs_store_destroy(vs_store* s) {
    if (!s) return VS_OK;
    (void)hipSetDevice(s->ctx->device);
    {
        std::lock_guard<std::mutex> g(s->ctx->mu);
        s->ctx->stores.erase(s->index);
    }
    {
        std::lock_guard<std::mutex> g(s->ctx->upload_mu);
        (void)quiesce(s);  // searches in flight finish; nobody may start one on a destroyed store
        for (auto& kv : s->upload_events) {
            (void)hipEventSynchronize(kv.second);
            (void)hipEventDestroy(kv.second);
        }
    }
    for (int i = 0; i < 2; ++i)
        for (auto& kv : s->stream_events[i]) (void)hipEventDestroy(kv.second);
    if (s->hint_event) (void)hipEventDestroy(s->hint_event);
    if (s->hint_count) (void)hipHostFree(s->hint_count);
    if (s->d_rows) (void)hipFree(s->d_rows);
    if (s->d_norms) (void)hipFree(s->d_norms);
    if (s->d_present[0]) (void)hipFree(s->d_present[0]);
    if (s->d_present[1]) (void)hipFree(s->d_present[1]);
    if (s->d_pending) (void)hipFree(s->d_pending);
    if (s->d_pending_slots) (void)hipFree(s->d_pending_slots);
    delete s;
    return VS_OK;
}

namespace {

void mark_present(vs_store* s, uint64_t slot) {
    uint64_t& w = s->h_present[slot >> 6];
    const uint64_t bit = 1ull << (slot & 63);
    if (!(w & bit)) {
        w |= bit;
        ++s->n_items;
    }
}

void mark_dirty(vs_store* s, uint64_t slot) {
    if (!s->dirty) {
        s->dirty = true;
        s->dirty_lo = s->dirty_hi = slot;
    } else {
        if (slot < s->dirty_lo) s->dirty_lo = slot;
        if (slot > s->dirty_hi) s->dirty_hi = slot;
    }
}

bool visible(const vs_store* s, uint64_t slot) { return (s->h_visible[slot >> 6] >> (slot & 63)) & 1; }

// Makes room for `rows` parked rows (contents kept). Earlier vs_store_upload_device calls may still
// be copying parked rows into the old buffer on their callers' streams: those copies are waited for
// before the buffer is copied and freed.
int32_t reserve_pending(vs_store* s, size_t rows) {
    if (rows <= s->pending_cap_rows) return VS_OK;
    size_t cap = s->pending_cap_rows ? s->pending_cap_rows * 2 : 1024;
    while (cap < rows) cap *= 2;
    unsigned char* fresh = nullptr;
    HIP_TRY(hipMalloc(&fresh, cap * s->row_bytes()));
    if (s->d_pending) {
        hipError_t e = hipSuccess;
        for (auto& kv : s->upload_events) {
            const hipError_t w = hipEventSynchronize(kv.second);
            if (w != hipSuccess) e = w;
        }
        if (e == hipSuccess)
            e = hipMemcpy(fresh, s->d_pending, s->pending_cap_rows * s->row_bytes(), hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            (void)hipFree(fresh);
            return fail(VS_E_DEVICE, "copying parked rows failed: %s", hipGetErrorString(e));
        }
        (void)hipFree(s->d_pending);
    }
    s->d_pending = fresh;
    s->pending_cap_rows = cap;
    return VS_OK;
}

// Forgets the rows parked since there were `n_before` of them (a failed upload: its parked rows
// have no room, so the next seal must not scatter them).
void unpark_from(vs_store* s, size_t n_before) {
    while (s->pending_slots.size() > n_before) {
        s->pending_of_slot.erase(s->pending_slots.back());
        s->pending_slots.pop_back();
    }
}

// Where the writer puts the row of `slot`: the slab itself while no search can see the slot,
// else a parked row (a new one, or the one an earlier upload of this transaction took).
uint64_t route_slot(vs_store* s, uint64_t slot) {
    if (!visible(s, slot)) {
        mark_dirty(s, slot);
        return slot;
    }
    auto it = s->pending_of_slot.find(slot);
    if (it != s->pending_of_slot.end()) return vs::INGEST_SIDE | it->second;
    const uint64_t j = s->pending_slots.size();
    s->pending_slots.push_back(slot);
    s->pending_of_slot.emplace(slot, j);
    return vs::INGEST_SIDE | j;
}

}  // namespace

extern "C" int32_t vs_store_upload(vs_store* s, const uint32_t* docids, const float* rows,
                                   uint64_t n) {
    if (int32_t rc = check_store(s)) return rc;
    if (n == 0) return VS_OK;
    if (!docids || !rows) return fail(VS_E_INVALID_ARG, "null docids or rows");
    vs_ctx* ctx = s->ctx;
    std::lock_guard<std::mutex> g(ctx->upload_mu);
    for (uint64_t i = 0; i < n; ++i)
        if (docids[i] < s->docid_base || (uint64_t)docids[i] - s->docid_base >= s->capacity)
            return fail(VS_E_CAPACITY, "docid %u outside the store's range [%u, %llu)", docids[i],
                        s->docid_base, (unsigned long long)(s->docid_base + s->capacity));
    if (!ctx->h_stage) HIP_TRY(hipHostMalloc(&ctx->h_stage, STAGE_BYTES, hipHostMallocDefault));
    if (!ctx->d_stage) HIP_TRY(hipMalloc(&ctx->d_stage, STAGE_BYTES));
    const size_t per_row = (size_t)s->dim * 4 + 8;
    const uint64_t chunk = (STAGE_BYTES - 8) / per_row;
    const vs::SlabView v = s->view(0);  // geometry only
    // arroy's add_item is last-write-wins: of several rows for one docid in this call only the
    // last is written (two waves writing one slot would have no defined order)
    std::unordered_map<uint32_t, uint64_t> last;
    last.reserve(n);
    for (uint64_t i = 0; i < n; ++i) last[docids[i]] = i;
    const bool dups = last.size() != n;
    for (uint64_t at = 0; at < n; at += chunk) {
        const uint64_t m = n - at < chunk ? n - at : chunk;
        const size_t row_b = ((size_t)m * s->dim * 4 + 7) / 8 * 8;
        memcpy(ctx->h_stage, rows + at * s->dim, (size_t)m * s->dim * 4);
        uint64_t* dest = reinterpret_cast<uint64_t*>(ctx->h_stage + row_b);
        const size_t parked_before = s->pending_slots.size();
        for (uint64_t i = 0; i < m; ++i) {
            const uint32_t d = docids[at + i];
            dest[i] = dups && last[d] != at + i ? vs::INGEST_SKIP : route_slot(s, d - s->docid_base);
        }
        if (int32_t rc = reserve_pending(s, s->pending_slots.size())) {
            unpark_from(s, parked_before);
            return rc;
        }
        HIP_TRY(hipMemcpyAsync(ctx->d_stage, ctx->h_stage, row_b + (size_t)m * 8,
                               hipMemcpyHostToDevice, ctx->upload_stream));
        HIP_TRY(vs::launch_ingest_f32(v, s->d_rows, s->d_pending, reinterpret_cast<const float*>(ctx->d_stage),
                                      reinterpret_cast<const uint64_t*>(ctx->d_stage + row_b), m,
                                      ctx->upload_stream));
        HIP_TRY(hipStreamSynchronize(ctx->upload_stream));
    }
    for (uint64_t i = 0; i < n; ++i) mark_present(s, docids[i] - s->docid_base);
    return VS_OK;
}

extern "C" int32_t vs_store_upload_device(vs_store* s, uint32_t first_docid,
                                          const void* device_rows, uint64_t n, void* hip_stream) {
    if (int32_t rc = check_store(s)) return rc;
    if (n == 0) return VS_OK;
    if (!device_rows) return fail(VS_E_INVALID_ARG, "null device rows");
    if (first_docid < s->docid_base || (uint64_t)first_docid - s->docid_base + n > s->capacity)
        return fail(VS_E_CAPACITY, "docids [%u, %u + %llu) outside the store's range", first_docid,
                    first_docid, (unsigned long long)n);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    std::lock_guard<std::mutex> g(s->ctx->upload_mu);
    const uint64_t slot0 = first_docid - s->docid_base;
    const size_t rb = s->row_bytes();
    const unsigned char* src = static_cast<const unsigned char*>(device_rows);
    bool any_visible = false;
    for (uint64_t w = slot0 >> 6; w <= (slot0 + n - 1) >> 6 && !any_visible; ++w) any_visible = s->h_visible[w] != 0;
    if (any_visible) {  // a visible slot in the covering words: look slot by slot
        any_visible = false;
        for (uint64_t i = 0; i < n && !any_visible; ++i) any_visible = visible(s, slot0 + i);
    }
    if (!any_visible) {
        // the bulk-load case: nothing a search can see is touched, one copy
        HIP_TRY(hipMemcpyAsync(static_cast<unsigned char*>(s->d_rows) + slot0 * rb, src, (size_t)n * rb,
                               hipMemcpyDeviceToDevice, stream));
        mark_dirty(s, slot0);
        mark_dirty(s, slot0 + n - 1);
    } else {
        // rows that replace visible items are parked; copies go run by run
        std::vector<uint64_t> dest(n);
        const size_t parked_before = s->pending_slots.size();
        for (uint64_t i = 0; i < n; ++i) dest[i] = route_slot(s, slot0 + i);
        if (int32_t rc = reserve_pending(s, s->pending_slots.size())) {
            unpark_from(s, parked_before);
            return rc;
        }
        for (uint64_t i = 0; i < n;) {
            uint64_t j = i + 1;
            while (j < n && dest[j] == dest[j - 1] + 1) ++j;  // consecutive slots, or consecutive parked rows
            unsigned char* dst = (dest[i] & vs::INGEST_SIDE)
                                     ? s->d_pending + (dest[i] & ~vs::INGEST_SIDE) * rb
                                     : static_cast<unsigned char*>(s->d_rows) + dest[i] * rb;
            HIP_TRY(hipMemcpyAsync(dst, src + i * rb, (size_t)(j - i) * rb, hipMemcpyDeviceToDevice, stream));
            i = j;
        }
    }
    // the seal waits for this copy through the stream's event, not through a device-wide wait
    hipEvent_t& ev = s->upload_events[stream];
    if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev, stream));
    for (uint64_t i = 0; i < n; ++i) mark_present(s, slot0 + i);
    return VS_OK;
}

extern "C" int32_t vs_store_remove(vs_store* s, const uint32_t* docids, uint64_t n) {
    if (int32_t rc = check_store(s)) return rc;
    if (n && !docids) return fail(VS_E_INVALID_ARG, "null docids");
    std::lock_guard<std::mutex> g(s->ctx->upload_mu);
    for (uint64_t i = 0; i < n; ++i) {
        if (docids[i] < s->docid_base || (uint64_t)docids[i] - s->docid_base >= s->capacity) continue;
        const uint64_t slot = docids[i] - s->docid_base;
        uint64_t& w = s->h_present[slot >> 6];
        const uint64_t bit = 1ull << (slot & 63);
        if (w & bit) {
            w &= ~bit;
            --s->n_items;
        }
    }
    return VS_OK;
}

extern "C" int32_t vs_store_seal(vs_store* s) {
    if (int32_t rc = check_store(s)) return rc;
    vs_ctx* ctx = s->ctx;
    std::lock_guard<std::mutex> g(ctx->upload_mu);
    hipStream_t st = ctx->upload_stream;
    // rows copied on caller streams (vs_store_upload_device) have arrived
    for (auto& kv : s->upload_events) HIP_TRY(hipEventSynchronize(kv.second));
    // norms of the rows written in place: slots no search can see yet (for the visible, unchanged
    // slots inside the range the kernel rewrites the value that is already there)
    if (s->dirty && s->metric == VS_ANGULAR)
        HIP_TRY(vs::launch_norms(s->view(0), s->d_norms, s->dirty_lo, s->dirty_hi - s->dirty_lo + 1, st));
    int next;
    {
        std::unique_lock<std::mutex> lk(s->snap_mu);
        next = 1 - s->active;
        // stragglers of the generation before the current one still read the buffer filled next
        if (int32_t rc = wait_readers(s, next, lk)) return rc;
    }
    const size_t n_parked = s->pending_slots.size();
    auto body = [&]() -> int32_t {
        if (n_parked) {
            // parked rows replace rows searches can see: the slab is rewritten with no search in flight
            if (int32_t rc = vsrt::grow(&s->d_pending_slots, &s->pending_slots_bytes, n_parked * 8)) return rc;
            HIP_TRY(hipMemcpyAsync(s->d_pending_slots, s->pending_slots.data(), n_parked * 8,
                                   hipMemcpyHostToDevice, st));
            if (int32_t rc = quiesce(s)) return rc;
            const vs::SlabView v = s->view(0);
            HIP_TRY(vs::launch_scatter_rows(v, s->d_rows, s->d_pending, s->d_pending_slots, n_parked, st));
            if (s->metric == VS_ANGULAR)
                HIP_TRY(vs::launch_norms_slots(v, s->d_norms, s->d_pending_slots, n_parked, st));
        }
        HIP_TRY(hipMemcpyAsync(s->d_present[next], s->h_present.data(), s->h_present.size() * 8,
                               hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        return VS_OK;
    };
    const int32_t rc = body();
    {
        std::lock_guard<std::mutex> lk(s->snap_mu);
        if (rc == VS_OK) s->active = next;
        s->writer_blocks = false;
        s->snap_cv.notify_all();
    }
    if (rc != VS_OK) return rc;
    // the buffer that was active keeps the previous generation for its stragglers
    for (size_t w = 0; w < s->h_present.size(); ++w) s->h_visible[w] = s->h_present[w] | s->h_sealed[w];
    s->h_sealed = s->h_present;
    s->pending_slots.clear();
    s->pending_of_slot.clear();
    s->sealed_items.store(s->n_items);
    s->dirty = false;
    s->sealed.store(true);
    return VS_OK;
}

extern "C" int32_t vs_store_len(vs_store* s, uint64_t* n_items) {
    if (!s || !n_items) return fail(VS_E_INVALID_ARG, "null argument");
    *n_items = s->sealed_items.load();
    return VS_OK;
}

extern "C" int32_t vs_store_dim(vs_store* s, uint32_t* dim) {
    if (!s || !dim) return fail(VS_E_INVALID_ARG, "null argument");
    *dim = s->dim;
    return VS_OK;
}

extern "C" int32_t vs_store_row_bytes(vs_store* s, uint64_t* bytes) {
    if (!s || !bytes) return fail(VS_E_INVALID_ARG, "null argument");
    *bytes = s->row_bytes();
    return VS_OK;
}

extern "C" int32_t vs_store_contains(vs_store* s, uint32_t docid, int32_t* out) {
    if (!s || !out) return fail(VS_E_INVALID_ARG, "null argument");
    std::lock_guard<std::mutex> g(s->ctx->upload_mu);
    *out = 0;
    if (docid >= s->docid_base && (uint64_t)docid - s->docid_base < s->capacity) {
        const uint64_t slot = docid - s->docid_base;
        *out = (int32_t)((s->h_present[slot >> 6] >> (slot & 63)) & 1);
    }
    return VS_OK;
}

extern "C" int32_t vs_store_item_vector(vs_store* s, uint32_t docid, float* out) {
    if (int32_t rc = check_store(s)) return rc;
    if (!out) return fail(VS_E_INVALID_ARG, "null output");
    vs_ctx* ctx = s->ctx;
    std::lock_guard<std::mutex> g(ctx->upload_mu);
    const bool in_range = docid >= s->docid_base && (uint64_t)docid - s->docid_base < s->capacity;
    const uint64_t slot = in_range ? docid - s->docid_base : 0;
    if (!in_range || !((s->h_present[slot >> 6] >> (slot & 63)) & 1))
        return fail(VS_E_INVALID_ARG, "item %u is not in store %u", docid, s->index);
    // the writer's view: a row parked by this transaction wins over the slab's
    for (auto& kv : s->upload_events) HIP_TRY(hipEventSynchronize(kv.second));
    const size_t rb = s->row_bytes();
    auto it = s->pending_of_slot.find(slot);
    const unsigned char* src = it != s->pending_of_slot.end()
                                   ? s->d_pending + it->second * rb
                                   : static_cast<const unsigned char*>(s->d_rows) + slot * rb;
    unsigned char* row = ctx->h_small;
    HIP_TRY(hipMemcpyAsync(row, src, rb, hipMemcpyDeviceToHost, ctx->upload_stream));
    HIP_TRY(hipStreamSynchronize(ctx->upload_stream));
    for (uint32_t e = 0; e < s->dim; ++e) {
        if (s->dtype == VS_F32) {
            memcpy(&out[e], row + 4 * e, 4);
        } else if (s->dtype == VS_BF16) {
            uint16_t h;
            memcpy(&h, row + 2 * e, 2);
            const uint32_t u = (uint32_t)h << 16;
            memcpy(&out[e], &u, 4);
        } else {
            uint64_t w;
            memcpy(&w, row + 8 * (e / 64), 8);
            out[e] = (float)((w >> (e % 64)) & 1);
        }
    }
    return VS_OK;
}

// ---- search, host buffers ----------------------------------------------------------------

extern "C" int32_t vs_search(vs_store* s, const float* query, uint32_t dim, uint32_t k,
                             const uint64_t* mask_words, uint64_t n_mask_words, uint32_t* out_ids,
                             float* out_dist, uint32_t* out_n) {
    if (int32_t rc = begin_search(s, query, dim, k, out_ids, out_dist, out_n)) return rc;
    *out_n = 0;
    if (k == 0) return VS_OK;
    return with_snapshot(s, [&](Workspace* ws, const vs::SlabView& v) -> int32_t {
        const uint64_t* d_cand = nullptr;
        if (mask_words) {
            if (int32_t rc = stage_dense_mask(s, ws, mask_words, n_mask_words)) return rc;
            d_cand = ws->d_mask;
        }
        return run_and_fetch(s, v, ws, query, k, d_cand, out_ids, out_dist, out_n);
    });
}

extern "C" int32_t vs_search_roaring(vs_store* s, const float* query, uint32_t dim, uint32_t k,
                                     const uint8_t* roaring_bytes, size_t len, uint32_t* out_ids,
                                     float* out_dist, uint32_t* out_n) {
    if (int32_t rc = begin_search(s, query, dim, k, out_ids, out_dist, out_n)) return rc;
    *out_n = 0;
    if (!roaring_bytes) return fail(VS_E_INVALID_ARG, "null roaring bytes");
    std::vector<vs::RoaringContainerDesc> desc;
    if (!vs::roaring_parse(roaring_bytes, len, &desc))
        return fail(VS_E_FORMAT, "malformed serialised RoaringBitmap (%zu bytes)", len);
    if (k == 0) return VS_OK;
    return with_snapshot(s, [&](Workspace* ws, const vs::SlabView& v) -> int32_t {
        const uint64_t n_words = s->n_slots / 64;
        if (int32_t rc = grow(&ws->d_mask, &ws->mask_words, (size_t)n_words * 8)) return rc;
        if (int32_t rc = expand_roaring(s, ws, roaring_bytes, len, desc, ws->d_mask)) return rc;
        return run_and_fetch(s, v, ws, query, k, ws->d_mask, out_ids, out_dist, out_n);
    });
}

// ---- candidate masks kept on the device ------------------------------------------------------

namespace {
int32_t mask_new(vs_store* s, vs_mask** out, std::unique_ptr<vs_mask>* m) {
    if (!out) return fail(VS_E_INVALID_ARG, "null out pointer");
    *out = nullptr;
    if (int32_t rc = check_store(s)) return rc;
    m->reset(new vs_mask{s, nullptr, s->n_slots, s->serial, s->docid_base});
    if (hipMalloc(&(*m)->d_words, (size_t)(s->n_slots / 64) * 8) != hipSuccess)
        return fail(VS_E_OOM, "cannot allocate a %llu-byte mask", (unsigned long long)(s->n_slots / 8));
    return VS_OK;
}
}  // namespace

extern "C" int32_t vs_mask_create(vs_store* s, const uint64_t* mask_words, uint64_t n_mask_words,
                                  vs_mask** out) {
    if (out) *out = nullptr;
    if (!mask_words && n_mask_words) return fail(VS_E_INVALID_ARG, "null mask words");
    std::unique_ptr<vs_mask> m;
    if (int32_t rc = mask_new(s, out, &m)) return rc;
    std::unique_ptr<Workspace> ws;
    int32_t rc = ws_acquire(s->ctx, &ws);
    if (rc == VS_OK) {
        rc = copy_dense_mask(s, ws->stream, m->d_words, mask_words, n_mask_words);
        if (rc == VS_OK && hipStreamSynchronize(ws->stream) != hipSuccess)
            rc = fail(VS_E_DEVICE, "mask upload failed");
        ws_release(s->ctx, std::move(ws));
    }
    if (rc != VS_OK) {
        (void)hipFree(m->d_words);
        return rc;
    }
    *out = m.release();
    return VS_OK;
}

extern "C" int32_t vs_mask_create_roaring(vs_store* s, const uint8_t* roaring_bytes, size_t len,
                                          vs_mask** out) {
    if (out) *out = nullptr;
    if (!roaring_bytes) return fail(VS_E_INVALID_ARG, "null roaring bytes");
    std::vector<vs::RoaringContainerDesc> desc;
    if (!vs::roaring_parse(roaring_bytes, len, &desc))
        return fail(VS_E_FORMAT, "malformed serialised RoaringBitmap (%zu bytes)", len);
    std::unique_ptr<vs_mask> m;
    if (int32_t rc = mask_new(s, out, &m)) return rc;
    std::unique_ptr<Workspace> ws;
    int32_t rc = ws_acquire(s->ctx, &ws);
    if (rc == VS_OK) {
        rc = expand_roaring(s, ws.get(), roaring_bytes, len, desc, m->d_words);
        if (rc == VS_OK && hipStreamSynchronize(ws->stream) != hipSuccess)
            rc = fail(VS_E_DEVICE, "mask expansion failed");
        ws_release(s->ctx, std::move(ws));
    }
    if (rc != VS_OK) {
        (void)hipFree(m->d_words);
        return rc;
    }
    *out = m.release();
    return VS_OK;
}

extern "C" int32_t vs_mask_destroy(vs_mask* m) {
    if (!m) return VS_OK;
    (void)hipFree(m->d_words);
    delete m;
    return VS_OK;
}

namespace {
int32_t check_mask(vs_store* s, const vs_mask* mask) {
    if (mask && (mask->store != s || mask->store_serial != s->serial))
        return fail(VS_E_INVALID_ARG, "the mask was created for another store");
    if (mask && mask->n_slots != s->n_slots)
        return fail(VS_E_INVALID_ARG, "the mask was created before the store grew: create it again");
    return VS_OK;
}
}  // namespace

extern "C" int32_t vs_mask_fits(vs_store* s, const vs_mask* mask) {
    return s && mask && mask->store == s && mask->store_serial == s->serial && mask->n_slots == s->n_slots;
}

extern "C" int32_t vs_search_mask(vs_store* s, const float* query, uint32_t dim, uint32_t k,
                                  const vs_mask* mask, uint32_t* out_ids, float* out_dist,
                                  uint32_t* out_n) {
    if (int32_t rc = begin_search(s, query, dim, k, out_ids, out_dist, out_n)) return rc;
    *out_n = 0;
    if (int32_t rc = check_mask(s, mask)) return rc;
    if (k == 0) return VS_OK;
    return with_snapshot(s, [&](Workspace* ws, const vs::SlabView& v) -> int32_t {
        return run_and_fetch(s, v, ws, query, k, mask ? mask->d_words : nullptr, out_ids, out_dist, out_n);
    });
}

namespace {
// n_q queries under ONE device-resident candidate mask (null = every item), host results.
int32_t batch_under_mask(vs_store* s, const vs::SlabView& v, Workspace* ws, const float* queries, uint32_t n_q,
                         uint32_t k, const uint64_t* d_cand, uint32_t* out_ids, float* out_dist, uint32_t* out_n) {
    if (on_matrix_cores(s, k) && n_q > 1)
        return run_batch_and_fetch(s, v, ws, queries, n_q, k, d_cand, out_ids, out_dist, out_n);
    return run_many_and_fetch(s, v, ws, queries, n_q, k, d_cand, out_ids, out_dist, out_n);
}
}  // namespace

extern "C" int32_t vs_search_batch(vs_store* s, const float* queries, uint32_t n_q, uint32_t dim,
                                   uint32_t k, const uint64_t* mask_words, uint64_t n_mask_words,
                                   uint32_t* out_ids, float* out_dist, uint32_t* out_n) {
    if (n_q == 0) return VS_OK;
    if (int32_t rc = begin_search(s, queries, dim, k, out_ids, out_dist, out_n)) return rc;
    for (uint32_t i = 0; i < n_q; ++i) out_n[i] = 0;
    if (k == 0) return VS_OK;
    return with_snapshot(s, [&](Workspace* ws, const vs::SlabView& v) -> int32_t {
        const uint64_t* d_cand = nullptr;
        if (mask_words) {
            if (int32_t rc = stage_dense_mask(s, ws, mask_words, n_mask_words)) return rc;
            d_cand = ws->d_mask;
        }
        return batch_under_mask(s, v, ws, queries, n_q, k, d_cand, out_ids, out_dist, out_n);
    });
}

extern "C" int32_t vs_search_batch_masks(vs_store* s, const float* queries, uint32_t n_q, uint32_t dim,
                                         uint32_t k, const uint64_t* const* mask_words,
                                         const uint64_t* n_mask_words, const vs_mask* const* masks,
                                         uint32_t* out_ids, float* out_dist, uint32_t* out_n) {
    if (n_q == 0) return VS_OK;
    if (int32_t rc = begin_search(s, queries, dim, k, out_ids, out_dist, out_n)) return rc;
    if (mask_words && !n_mask_words) return fail(VS_E_INVALID_ARG, "mask words without their lengths");
    for (uint32_t i = 0; i < n_q; ++i) {
        out_n[i] = 0;
        if (masks && masks[i])
            if (int32_t rc = check_mask(s, masks[i])) return rc;
    }
    if (k == 0) return VS_OK;
    // Queries with the same filter (the same vs_mask, the same host words, or none) form a group
    // that shares passes exactly as vs_search_batch would; every group is searched under its own
    // mask, so a scan reads only that filter's candidates.
    struct Filter {
        const vs_mask* mask;
        const uint64_t* words;
        uint64_t n_words;
        bool operator<(const Filter& o) const {
            if (mask != o.mask) return mask < o.mask;
            if (words != o.words) return words < o.words;
            return n_words < o.n_words;
        }
    };
    std::map<Filter, std::vector<uint32_t>> groups;
    for (uint32_t i = 0; i < n_q; ++i) {
        Filter f{masks ? masks[i] : nullptr, nullptr, 0};
        if (!f.mask && mask_words && mask_words[i]) {
            f.words = mask_words[i];
            f.n_words = n_mask_words[i];
        }
        groups[f].push_back(i);
    }
    return with_snapshot(s, [&](Workspace* ws, const vs::SlabView& v) -> int32_t {
        std::vector<float> q;
        std::vector<uint32_t> ids, cnt;
        std::vector<float> dist;
        auto scatter = [&](const std::vector<uint32_t>& members, size_t from, size_t m) {
            for (size_t j = 0; j < m; ++j) {
                const uint32_t i = members[from + j];
                memcpy(out_ids + (size_t)i * k, ids.data() + j * k, (size_t)cnt[j] * 4);
                memcpy(out_dist + (size_t)i * k, dist.data() + j * k, (size_t)cnt[j] * 4);
                out_n[i] = cnt[j];
            }
        };
        auto gather_queries = [&](const std::vector<uint32_t>& members, size_t from, size_t m) {
            q.resize(m * dim);
            for (size_t j = 0; j < m; ++j)
                memcpy(q.data() + j * dim, queries + (size_t)members[from + j] * dim, (size_t)dim * 4);
            ids.assign(m * k, 0);
            dist.assign(m * k, 0.0f);
            cnt.assign(m, 0);
        };
        // Queries alone under their filter (a batch of requests with a universe each): ONE scan launch per
        // chunk takes them all, every query under its own mask. Host words are staged side by side, at most
        // 256 MiB of them at a time.
        std::vector<uint32_t> solo;
        std::vector<Filter> solo_filter;
        for (auto& kv : groups)
            if (kv.second.size() == 1) {
                solo.push_back(kv.second[0]);
                solo_filter.push_back(kv.first);
            }
        const size_t mask_bytes = (size_t)(s->n_slots / 64) * 8;
        size_t chunk = (256u << 20) / mask_bytes;
        chunk = chunk < 1 ? 1 : chunk > 256 ? 256 : chunk;
        for (size_t at = 0; at < solo.size(); at += chunk) {
            const size_t m = solo.size() - at < chunk ? solo.size() - at : chunk;
            size_t staged = 0;
            for (size_t j = 0; j < m; ++j) staged += solo_filter[at + j].words ? 1 : 0;
            if (staged)
                if (int32_t rc = grow(&ws->d_mask, &ws->mask_words, staged * mask_bytes)) return rc;
            std::vector<const uint64_t*> cand_of(m, nullptr);
            staged = 0;
            for (size_t j = 0; j < m; ++j) {
                const Filter& f = solo_filter[at + j];
                if (f.mask) {
                    cand_of[j] = f.mask->d_words;
                } else if (f.words) {
                    uint64_t* d_words = ws->d_mask + staged++ * (mask_bytes / 8);
                    if (int32_t rc = copy_dense_mask(s, ws->stream, d_words, f.words, f.n_words)) return rc;
                    cand_of[j] = d_words;
                }
            }
            gather_queries(solo, at, m);
            if (int32_t rc = run_queries_and_fetch(s, v, ws, q.data(), (uint32_t)m, k, nullptr, cand_of.data(),
                                                   ids.data(), dist.data(), cnt.data()))
                return rc;
            scatter(solo, at, m);
        }
        for (auto& kv : groups) {
            const Filter& f = kv.first;
            const std::vector<uint32_t>& members = kv.second;
            if (members.size() == 1) continue;
            const uint64_t* d_cand = nullptr;
            if (f.mask) {
                d_cand = f.mask->d_words;
            } else if (f.words) {
                if (int32_t rc = stage_dense_mask(s, ws, f.words, f.n_words)) return rc;
                d_cand = ws->d_mask;
            }
            gather_queries(members, 0, members.size());
            if (int32_t rc = batch_under_mask(s, v, ws, q.data(), (uint32_t)members.size(), k, d_cand, ids.data(),
                                              dist.data(), cnt.data()))
                return rc;
            scatter(members, 0, members.size());
        }
        return VS_OK;
    });
}

// ---- one query against every vector store of an embedder --------------------------------------

extern "C" int32_t vs_search_stores(vs_ctx* ctx, uint16_t store_index_base, const float* query, uint32_t dim,
                                    uint32_t k, const uint64_t* mask_words, uint64_t n_mask_words,
                                    const uint8_t* roaring_bytes, size_t roaring_len, const vs_mask* mask,
                                    uint32_t* out_ids, float* out_dist, uint8_t* out_store, uint32_t out_cap,
                                    uint32_t* out_n, uint32_t* out_n_stores) {
    if (!ctx || !query || !out_n || !out_n_stores) return fail(VS_E_INVALID_ARG, "null argument");
    *out_n = 0;
    *out_n_stores = 0;
    if (store_index_base & 0xFF)
        return fail(VS_E_INVALID_ARG, "store_index_base %u: the low byte numbers a document's vectors and must be 0",
                    store_index_base);
    if ((mask_words != nullptr) + (roaring_bytes != nullptr) + (mask != nullptr) > 1)
        return fail(VS_E_INVALID_ARG, "one filter at most: dense words, a serialised RoaringBitmap or a vs_mask");
    if (k > VS_MAX_K) return fail(VS_E_INVALID_ARG, "k = %u exceeds VS_MAX_K = %u", k, VS_MAX_K);
    HIP_TRY(hipSetDevice(ctx->device));
    // the stores (base | 0), (base | 1), ... up to the first that is missing or was never sealed
    std::vector<vs_store*> stores;
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        for (uint32_t j = 0; j <= 0xFF; ++j) {
            auto it = ctx->stores.find((uint16_t)(store_index_base | j));
            if (it == ctx->stores.end() || !it->second->sealed.load()) break;
            stores.push_back(it->second);
        }
    }
    const uint32_t n_stores = (uint32_t)stores.size();
    *out_n_stores = n_stores;
    if (n_stores == 0 || k == 0) return VS_OK;  // no reader opens: nothing to search
    if ((uint64_t)n_stores * k > out_cap)
        return fail(VS_E_CAPACITY, "%u stores x k = %u need %llu output entries, %u given", n_stores, k,
                    (unsigned long long)n_stores * k, out_cap);
    if (!out_ids || !out_dist || !out_store) return fail(VS_E_INVALID_ARG, "null output buffer");
    std::vector<vs::RoaringContainerDesc> desc;
    if (roaring_bytes && !vs::roaring_parse(roaring_bytes, roaring_len, &desc))
        return fail(VS_E_FORMAT, "malformed serialised RoaringBitmap (%zu bytes)", roaring_len);
    for (vs_store* s : stores) {
        if (s->dim != dim) {
            set_dim_mismatch(s->dim, dim);
            return fail(VS_E_DIM_MISMATCH, "invalid vector dimensions: expected %u, received %u", s->dim, dim);
        }
        if (s->dtype != stores[0]->dtype || s->metric != stores[0]->metric)
            return fail(VS_E_INVALID_ARG, "stores %u and %u differ in element type or metric", stores[0]->index, s->index);
    }
    std::unique_ptr<Workspace> ws;
    if (int32_t rc = ws_acquire(ctx, &ws)) return rc;
    std::vector<Snapshot> snaps(n_stores);
    for (uint32_t j = 0; j < n_stores; ++j) snapshot_begin(stores[j], &snaps[j]);
    auto body = [&]() -> int32_t {
        // the filter goes to the device ONCE, as words over the docid window all stores lie in; every
        // store reads it from its own offset
        uint64_t lo = ~0ull, hi = 0;
        for (uint32_t j = 0; j < n_stores; ++j) {
            const vs::SlabView& v = snaps[j].v;
            lo = v.docid_base < lo ? v.docid_base : lo;
            hi = v.docid_base + v.n_slots > hi ? v.docid_base + v.n_slots : hi;
        }
        const uint64_t window_words = (hi - lo) / 64;
        const uint64_t* d_window = nullptr;
        if (mask) {
            // a mask made for one of these stores serves all that start at its docid and are not larger
            bool owner = false;
            for (vs_store* s : stores) owner = owner || vs_mask_fits(s, mask);
            if (!owner) return fail(VS_E_INVALID_ARG, "the mask was created for none of these stores as they are now");
            for (uint32_t j = 0; j < n_stores; ++j)
                if (snaps[j].v.docid_base != mask->docid_base || snaps[j].v.n_slots > mask->n_slots)
                    return fail(VS_E_INVALID_ARG, "store %u does not lie inside the mask's docid range", stores[j]->index);
        } else if (mask_words || roaring_bytes) {
            if (int32_t rc = grow(&ws->d_mask, &ws->mask_words, (size_t)window_words * 8)) return rc;
            if (mask_words) {
                if (int32_t rc = copy_dense_window(ws->stream, ws->d_mask, (uint32_t)lo, window_words, mask_words, n_mask_words))
                    return rc;
            } else if (int32_t rc = expand_roaring_window(ws.get(), roaring_bytes, roaring_len, desc, (uint32_t)lo,
                                                          window_words, ws->d_mask)) {
                return rc;
            }
            d_window = ws->d_mask;
        }
        const vs::ScanTuning t = tuning_of(ctx);
        // ONE scan launch takes all stores, a job each (its own slab, the common query, the filter from
        // its own offset), and one merge launch writes every store's k results straight into pinned host
        // memory (mapped into the device's address space): no copy back, one wait.
        uint32_t big = 0;
        for (uint32_t j = 1; j < n_stores; ++j)
            if (snaps[j].v.n_slots > snaps[big].v.n_slots) big = j;
        const vs::SlabView& vbig = snaps[big].v;
        const uint32_t blocks = vs::scan_grid_blocks(vbig, t, true);
        const size_t q_bytes = stores[0]->dtype == VS_BIN1 ? (size_t)stores[0]->row_vecs * 16 : (size_t)dim * 4;
        const size_t off_jobs = (q_bytes + 255) / 256 * 256, jobs_bytes = n_stores * sizeof(vs::ScanJob);
        const size_t res_bytes = (size_t)n_stores * k * 8 + (size_t)n_stores * 4;
        if (int32_t rc = grow(&ws->d_batch, &ws->batch_bytes, off_jobs + jobs_bytes)) return rc;
        if (int32_t rc = grow(&ws->d_partial, &ws->partial_bytes, (size_t)n_stores * blocks * k * 8)) return rc;
        if (int32_t rc = grow_pinned(ws.get(), off_jobs + jobs_bytes + res_bytes)) return rc;
        pack_query(stores[0], query, ws->h_pinned, q_bytes);
        vs::ScanJob* jobs = reinterpret_cast<vs::ScanJob*>(ws->h_pinned + off_jobs);
        for (uint32_t j = 0; j < n_stores; ++j) {
            const vs::SlabView& v = snaps[j].v;
            const uint64_t* d_cand = mask ? mask->d_words : d_window ? d_window + (v.docid_base - lo) / 64 : nullptr;
            jobs[j] = vs::ScanJob{v.rows, v.norms, v.present, d_cand, nullptr, v.n_slots, v.docid_base, 0};
        }
        HIP_TRY(hipMemcpyAsync(ws->d_batch, ws->h_pinned, off_jobs + jobs_bytes, hipMemcpyHostToDevice, ws->stream));
        if (int32_t rc = launch_scan_queries(vbig, t, ws->d_batch, 0, n_stores, nullptr,
                                             reinterpret_cast<const vs::ScanJob*>(ws->d_batch + off_jobs), k,
                                             ws->d_partial, ws->stream))
            return rc;
        unsigned char* hr = ws->h_pinned + off_jobs + jobs_bytes;
        const uint32_t* h_ids = reinterpret_cast<const uint32_t*>(hr);
        const float* h_dist = reinterpret_cast<const float*>(hr + (size_t)n_stores * k * 4);
        const uint32_t* h_n = reinterpret_cast<const uint32_t*>(hr + (size_t)n_stores * k * 8);
        HIP_TRY(vs::launch_merge_batch(ws->d_partial, blocks, 1, blocks, n_stores, (uint64_t)blocks * k, k, vbig.metric,
                                       nullptr, const_cast<uint32_t*>(h_ids), const_cast<float*>(h_dist),
                                       const_cast<uint32_t*>(h_n), ws->stream));
        HIP_TRY(hipStreamSynchronize(ws->stream));
        // every store's list is ascending by (distance, docid); the lists together go out ordered by
        // (distance, docid, store), NaN last, which is the order VectorSort::fill_buffer sorts them into
        struct Hit {
            float d;
            uint32_t id;
            uint8_t store;
        };
        std::vector<Hit> hits;
        hits.reserve((size_t)n_stores * k);
        for (uint32_t j = 0; j < n_stores; ++j) {
            const uint32_t n = h_n[j];
            if (n > k) return fail(VS_E_DEVICE, "device returned %u results for k = %u", n, k);
            for (uint32_t i = 0; i < n; ++i) hits.push_back(Hit{h_dist[(size_t)j * k + i], h_ids[(size_t)j * k + i], (uint8_t)j});
        }
        std::stable_sort(hits.begin(), hits.end(), [](const Hit& a, const Hit& b) {
            const bool an = a.d != a.d, bn = b.d != b.d;
            if (an != bn) return bn;
            if (!an && a.d != b.d) return a.d < b.d;
            return a.id < b.id;  // (equal distance and docid: the stable sort keeps the store order)
        });
        for (size_t i = 0; i < hits.size(); ++i) {
            out_ids[i] = hits[i].id;
            out_dist[i] = hits[i].d;
            out_store[i] = hits[i].store;
        }
        *out_n = (uint32_t)hits.size();
        return VS_OK;
    };
    const int32_t rc = body();
    if (rc != VS_OK) (void)hipStreamSynchronize(ws->stream);  // nothing of a failed call stays in flight
    for (uint32_t j = 0; j < n_stores; ++j) snapshot_end(stores[j], &snaps[j]);
    ws_release(ctx, std::move(ws));
    return rc;
}

// ---- search, device buffers ------------------------------------------------------------------

extern "C" int32_t vs_search_workspace_bytes(vs_store* s, uint32_t k, uint64_t* bytes) {
    if (!s || !bytes) return fail(VS_E_INVALID_ARG, "null argument");
    if (k == 0 || k > VS_MAX_K) return fail(VS_E_INVALID_ARG, "k = %u out of range", k);
    *bytes = vs::scan_workspace_bytes(s->view(0), k);
    return VS_OK;
}

extern "C" int32_t vs_search_device(vs_store* s, const void* d_query, uint32_t dim, uint32_t k,
                                    const uint64_t* d_mask_words, void* d_workspace, uint64_t workspace_bytes,
                                    uint64_t* d_out_keys, void* hip_stream) {
    if (int32_t rc = check_store(s)) return rc;
    if (!d_query || !d_workspace || !d_out_keys) return fail(VS_E_INVALID_ARG, "null device buffer");
    if (int32_t rc = check_device_search(s, dim, k)) return rc;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    Snapshot snap;
    snapshot_begin(s, &snap);
    // against the store as registered: a workspace sized before a vs_store_reserve is too small now
    const uint64_t need = vs::scan_workspace_bytes(snap.v, k);
    const int32_t rc = workspace_bytes < need
                           ? fail(VS_E_CAPACITY, "workspace of %llu bytes, the store needs %llu now (vs_search_workspace_bytes)",
                                  (unsigned long long)workspace_bytes, (unsigned long long)need)
                           : launch_scan(snap.v, tuning_of(s->ctx), d_query, d_mask_words, k,
                                         static_cast<uint64_t*>(d_workspace), d_out_keys, nullptr, nullptr, nullptr,
                                         stream);
    const int32_t rc2 = snapshot_end_async(s, &snap, stream, true);
    return rc ? rc : rc2;
}

// vs_search_device without a workspace argument and without any allocation: the scan's partial
// lists go to this GPU's static scratch array. The scan overwrites every entry of the partial
// buffer it later merges (each workgroup writes its k keys, padded with VS_KEY_NONE), so nothing
// needs clearing between calls; calls that overlap on one GPU would share the array and race.
// No event is recorded for the search (the stream may be under graph capture, and creating an
// event allocates): the caller orders these searches against vs_store_seal itself.
extern "C" int32_t vs_search_device_noalloc(vs_store* s, const void* d_query, uint32_t dim, uint32_t k,
                                            const uint64_t* d_mask_words, uint64_t* d_out_keys,
                                            void* hip_stream) {
    if (int32_t rc = check_store(s)) return rc;
    if (!d_query || !d_out_keys) return fail(VS_E_INVALID_ARG, "null device buffer");
    if (int32_t rc = check_device_search(s, dim, k)) return rc;
    const vs::ScanTuning t = tuning_of(s->ctx);  // one snapshot: the size check and the launch agree
    Snapshot snap;
    snapshot_begin(s, &snap);
    const uint64_t need = vs::scan_partial_bytes(snap.v, k, t);
    int32_t rc;
    if (need > NOALLOC_SCRATCH_BYTES)
        rc = fail(VS_E_CAPACITY,
                  "vs_search_device_noalloc: this store and k = %u need %llu bytes of partial lists, the "
                  "static scratch holds %llu (use vs_search_device with a workspace)",
                  k, (unsigned long long)need, (unsigned long long)NOALLOC_SCRATCH_BYTES);
    else
        rc = launch_scan(snap.v, t, d_query, d_mask_words, k, reinterpret_cast<uint64_t*>(s->ctx->d_scratch),
                         d_out_keys, nullptr, nullptr, nullptr, static_cast<hipStream_t>(hip_stream));
    (void)snapshot_end_async(s, &snap, nullptr, false);
    return rc;
}

namespace vsrt {

uint64_t batch_workspace_bytes_for(vs_store* s, const vs::SlabView& v, uint32_t k, bool* served) {
    *served = true;
    if (on_matrix_cores(s, k)) return vs::batch_workspace_bytes(v, k);
    if (vs::scan_hamming_multi_serves(v, k)) return vs::scan_workspace_bytes(v, k) * 4;
    if (shared_pass_serves(v, k)) return shared_pass_partial_bytes(v, k);
    *served = false;
    return 0;
}

int32_t launch_batch(vs_store* s, const vs::SlabView& v, const vs::ScanTuning& t, const void* d_queries,
                     uint32_t n_q, uint32_t k, const uint64_t* d_mask_words, void* d_workspace,
                     uint64_t* d_out_keys, hipStream_t stream) {
    if (on_matrix_cores(s, k)) {
        // whole tiles if the previous call under this mask counted so many candidates that skipping
        // the rest would save less than the extra launches of the gathered form cost (~80 us of HBM time)
        bool whole_tiles = false;
        const bool counted = d_mask_words && vs::batch_filters_in_place(v);
        if (counted) {
            std::lock_guard<std::mutex> lk(s->hint_mu);
            if (s->hint_event && hipEventQuery(s->hint_event) == hipSuccess) {
                s->known_mask = s->hint_mask;
                s->known_count = *s->hint_count;
            }
            if (s->known_mask == d_mask_words) {
                const uint64_t n = s->known_count < v.n_slots ? s->known_count : v.n_slots;
                whole_tiles = (v.n_slots - n) * s->row_bytes() < 350000000ull;
            }
        }
        const uint32_t* d_count = nullptr;
        HIP_TRY(vs::launch_scan_batch_bf16(v, static_cast<const float*>(d_queries), n_q, d_mask_words, k,
                                           d_workspace, d_out_keys, nullptr, nullptr, nullptr, stream, whole_tiles,
                                           &d_count));
        if (counted && d_count) {
            std::lock_guard<std::mutex> lk(s->hint_mu);
            if (!s->hint_count) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->hint_count), 64, hipHostMallocDefault));
            if (!s->hint_event) HIP_TRY(hipEventCreateWithFlags(&s->hint_event, hipEventDisableTiming));
            HIP_TRY(hipMemcpyAsync(s->hint_count, d_count, 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipEventRecord(s->hint_event, stream));
            s->hint_mask = d_mask_words;
        }
        return VS_OK;
    }
    if (vs::scan_hamming_multi_serves(v, k)) {
        // binary store: d_queries holds n_q packed queries, vs_store_row_bytes bytes each
        const unsigned char* bits = static_cast<const unsigned char*>(d_queries);
        for (uint32_t i = 0; i < n_q; i += 4)
            HIP_TRY(vs::launch_scan_hamming_multi(v, bits + (size_t)i * s->row_bytes(), s->row_vecs,
                                                  n_q - i < 4 ? n_q - i : 4, d_mask_words, k,
                                                  static_cast<uint64_t*>(d_workspace),
                                                  d_out_keys + (size_t)i * k, nullptr, nullptr, nullptr, t, stream));
        return VS_OK;
    }
    if (!shared_pass_serves(v, k)) return fail(VS_E_INVALID_ARG, "no batched scan for this store and k = %u", k);
    return queue_shared_passes(v, t, static_cast<const float*>(d_queries), s->dim, n_q, d_mask_words, k,
                               static_cast<uint64_t*>(d_workspace), d_out_keys, nullptr, nullptr, nullptr, stream);
}

}  // namespace vsrt

extern "C" int32_t vs_search_batch_workspace_bytes(vs_store* s, uint32_t k, uint64_t* bytes) {
    if (!s || !bytes) return fail(VS_E_INVALID_ARG, "null argument");
    bool served;
    *bytes = batch_workspace_bytes_for(s, s->view(0), k, &served);
    if (!served)
        return fail(VS_E_INVALID_ARG,
                    "no batched scan for this store and k = %u (angular bf16 rows with k <= %u; f32 rows "
                    "of up to 1536 or bf16 rows of up to 1536 dimensions with k <= 128)", k, VS_BATCH_MAX_K);
    return VS_OK;
}

extern "C" int32_t vs_search_batch_device(vs_store* s, const float* d_queries, uint32_t n_q,
                                          uint32_t dim, uint32_t k, const uint64_t* d_mask_words,
                                          void* d_workspace, uint64_t workspace_bytes, uint64_t* d_out_keys,
                                          void* hip_stream) {
    if (int32_t rc = check_store(s)) return rc;
    if (!d_queries || !d_workspace || !d_out_keys) return fail(VS_E_INVALID_ARG, "null device buffer");
    if (int32_t rc = check_device_search(s, dim, k)) return rc;
    if (n_q == 0 || n_q > VS_BATCH_MAX_QUERIES)
        return fail(VS_E_INVALID_ARG, "n_q = %u out of range", n_q);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    Snapshot snap;
    snapshot_begin(s, &snap);
    bool served = false;
    const uint64_t need = batch_workspace_bytes_for(s, snap.v, k, &served);
    const int32_t rc = served && workspace_bytes < need
                           ? fail(VS_E_CAPACITY,
                                  "workspace of %llu bytes, the store needs %llu now (vs_search_batch_workspace_bytes)",
                                  (unsigned long long)workspace_bytes, (unsigned long long)need)
                           : launch_batch(s, snap.v, tuning_of(s->ctx), d_queries, n_q, k, d_mask_words, d_workspace,
                                          d_out_keys, stream);
    const int32_t rc2 = snapshot_end_async(s, &snap, stream, true);
    return rc ? rc : rc2;
}

extern "C" int32_t vs_merge_topk_device(vs_ctx* ctx, const uint64_t* d_keys, uint32_t n_lists,
                                        uint32_t k, int32_t metric, uint32_t* d_out_ids,
                                        float* d_out_dist, uint32_t* d_out_n, void* hip_stream) {
    if (!ctx || !d_keys || !d_out_ids || !d_out_dist || !d_out_n)
        return fail(VS_E_INVALID_ARG, "null argument");
    if (k == 0 || k > VS_MAX_K || n_lists == 0)
        return fail(VS_E_INVALID_ARG, "k = %u or n_lists = %u out of range", k, n_lists);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(vs::launch_merge(d_keys, n_lists, k, 1, k, metric, nullptr, d_out_ids, d_out_dist,
                             d_out_n, static_cast<hipStream_t>(hip_stream)));
    return VS_OK;
}

extern "C" int32_t vs_merge_topk_batch_device(vs_ctx* ctx, const uint64_t* d_keys, uint32_t n_lists,
                                              uint64_t list_stride, uint32_t n_queries,
                                              uint64_t query_stride, uint32_t k, int32_t metric,
                                              uint32_t* d_out_ids, float* d_out_dist,
                                              uint32_t* d_out_n, void* hip_stream) {
    if (!ctx || !d_keys || !d_out_ids || !d_out_dist || !d_out_n)
        return fail(VS_E_INVALID_ARG, "null argument");
    if (k == 0 || k > VS_MAX_K || n_lists == 0 || n_queries == 0 || n_queries > 65535)
        return fail(VS_E_INVALID_ARG, "k = %u, n_lists = %u or n_queries = %u out of range", k,
                    n_lists, n_queries);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(vs::launch_merge_batch(d_keys, n_lists, list_stride, 1, n_queries, query_stride, k,
                                   metric, nullptr, d_out_ids, d_out_dist, d_out_n,
                                   static_cast<hipStream_t>(hip_stream)));
    return VS_OK;
}
