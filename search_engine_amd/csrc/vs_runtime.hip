// Host side of libvs_amd.so: contexts, stores (row slabs in HBM), the workspace
// pool that makes searches re-entrant, and the C-ABI entry points declared in
// include/vs_abi.h. No computation happens here; every distance is produced by
// the kernels in vs_kernels.hip, and there is no CPU fallback.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vs_abi.h"
#include "vs_kernels.h"
#include "vs_roaring.h"

namespace {

thread_local std::string g_err;
thread_local uint32_t g_dim_expected = 0, g_dim_received = 0;

int32_t fail(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        const hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess)                                                              \
            return fail(e_ == hipErrorOutOfMemory ? VS_E_OOM : VS_E_DEVICE, "%s failed: %s", \
                        #expr, hipGetErrorString(e_));                                     \
    } while (0)

constexpr uint32_t MAX_DIM_VECS = 16 * 64;  // row vectors the angular scan kernels are built for
constexpr uint32_t MAX_BIN_VECS = 32;       // and the Hamming scan (4096 bits)
constexpr size_t STAGE_BYTES = 32u << 20;

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

    ~Workspace() {
        if (d_query) (void)hipFree(d_query);
        if (d_mask) (void)hipFree(d_mask);
        if (d_partial) (void)hipFree(d_partial);
        if (d_keys) (void)hipFree(d_keys);
        if (d_blob) (void)hipFree(d_blob);
        if (d_batch) (void)hipFree(d_batch);
        if (h_pinned) (void)hipHostFree(h_pinned);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// Scratch of the entry point that allocates nothing (vs_search_device_noalloc): the scan's
// per-workgroup partial lists. One copy per GPU, held until the process ends. Host code never
// uses this name as an address: vs_init resolves the device address of the current GPU's copy
// with hipGetSymbolAddress and keeps it in the context.
constexpr size_t NOALLOC_SCRATCH_BYTES = VS_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

constexpr size_t QUERY_BYTES = (size_t)MAX_DIM_VECS * 16 * 2;  // bf16 rows: 8 f32 per vector
constexpr size_t RESULT_BYTES = (size_t)VS_MAX_K * 8 + 64;

}  // namespace

struct vs_ctx {
    int device = 0;
    std::mutex mu;  // stores map, pool, staging
    std::map<uint16_t, vs_store*> stores;
    std::vector<std::unique_ptr<Workspace>> pool;
    vs::ScanTuning tuning{0, 1};
    unsigned char* d_scratch = nullptr;  // this GPU's copy of `scratch` (vs_search_device_noalloc)
    // upload staging (guarded by upload_mu)
    std::mutex upload_mu;
    unsigned char* h_stage = nullptr;
    unsigned char* d_stage = nullptr;
    hipStream_t upload_stream = nullptr;
};

struct vs_mask {
    vs_store* store;
    uint64_t* d_words;  // store->n_slots / 64 words, aligned with the store
    uint64_t n_slots;   // of the store when the mask was made (vs_store_reserve outdates it)
};

struct vs_store {
    vs_ctx* ctx = nullptr;
    uint16_t index = 0;
    uint32_t dim = 0;
    int32_t dtype = VS_F32, metric = VS_ANGULAR;
    uint32_t docid_base = 0;
    uint64_t capacity = 0, n_slots = 0;
    uint32_t row_vecs = 0;
    void* d_rows = nullptr;
    float* d_norms = nullptr;
    uint64_t* d_present[2] = {nullptr, nullptr};  // searches read [active]; seal fills the other
    std::atomic<int> active{0};
    std::vector<uint64_t> h_present;  // writer's view, published by seal
    uint64_t n_items = 0;             // of h_present
    std::atomic<uint64_t> sealed_items{0};
    uint64_t dirty_lo = 0, dirty_hi = 0;  // slot range uploaded since the last seal
    bool dirty = false;
    std::atomic<bool> sealed{false};

    vs::SlabView view() const {
        vs::SlabView v;
        v.rows = d_rows;
        v.norms = d_norms;
        v.present = d_present[active.load(std::memory_order_acquire)];
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

namespace {

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
    std::lock_guard<std::mutex> g(ctx->mu);
    ctx->pool.push_back(std::move(ws));
}

template <typename T>
int32_t grow(T** p, size_t* have, size_t want) {
    if (*have >= want) return VS_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(p, want));
    *have = want;
    return VS_OK;
}

uint32_t vecs_for(uint32_t dim, int32_t dtype) {
    if (dtype == VS_F32) return (dim + 3) / 4;
    if (dtype == VS_BF16) return (dim + 7) / 8;
    return (dim + 127) / 128;
}

int32_t check_store(vs_store* s) {
    if (!s) return fail(VS_E_INVALID_ARG, "null store handle");
    HIP_TRY(hipSetDevice(s->ctx->device));
    return VS_OK;
}

// Common front half of the host-buffer searches: argument checks and the query upload.
int32_t begin_search(vs_store* s, const float* query, uint32_t dim, uint32_t k, uint32_t* out_ids,
                     float* out_dist, uint32_t* out_n) {
    if (int32_t rc = check_store(s)) return rc;
    if (!query || !out_n || (k && (!out_ids || !out_dist)))
        return fail(VS_E_INVALID_ARG, "null query or output buffer");
    if (!s->sealed.load()) return fail(VS_E_NOT_SEALED, "store %u was never sealed", s->index);
    if (dim != s->dim) {
        g_dim_expected = s->dim;
        g_dim_received = dim;
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
uint64_t shared_pass_partial_bytes(const vs::SlabView& v, uint32_t k, const vs::ScanTuning& t) {
    const uint64_t a = vs::scan_partial_bytes(v, k, t) * vs::scan_multi_queries(v);
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

// One scan of a sealed store under its metric.
int32_t launch_scan(vs_store* s, const vs::ScanTuning& t, const void* d_query, const uint64_t* d_cand,
                    uint32_t k, uint64_t* d_partial, uint64_t* d_keys, uint32_t* d_ids, float* d_dist,
                    uint32_t* d_n, hipStream_t stream) {
    const vs::SlabView v = s->view();
    if (s->metric == VS_HAMMING)
        HIP_TRY(vs::launch_scan_hamming(v, d_query, d_cand, k, d_partial, d_keys, d_ids, d_dist, d_n,
                                        t, stream));
    else
        HIP_TRY(vs::launch_scan_float(v, static_cast<const float*>(d_query), d_cand, k, d_partial,
                                        d_keys, d_ids, d_dist, d_n, t, stream));
    return VS_OK;
}

// Runs one query whose candidate mask (aligned to the store, or null) is already
// on the device, and copies the decoded result back.
int32_t run_and_fetch(vs_store* s, Workspace* ws, const float* h_query, uint32_t k,
                      const uint64_t* d_cand, uint32_t* out_ids, float* out_dist, uint32_t* out_n) {
    const vs::SlabView v = s->view();
    const vs::ScanTuning& t = s->ctx->tuning;
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
    if (int32_t rc = launch_scan(s, t, ws->d_query, d_cand, k, ws->d_partial, nullptr,
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
int32_t run_batch_and_fetch(vs_store* s, Workspace* ws, const float* h_queries, uint32_t n_q,
                            uint32_t k, const uint64_t* d_cand, uint32_t* out_ids, float* out_dist,
                            uint32_t* out_n) {
    const vs::SlabView v = s->view();
    const size_t chunk_q = VS_BATCH_MAX_QUERIES;
    const size_t q_bytes = chunk_q * s->dim * 4;
    const size_t res_bytes = chunk_q * k * 8 + chunk_q * 4;  // ids, distances, counts
    const size_t ws_bytes = (size_t)vs::batch_workspace_bytes(v, k);
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

// Several queries under one candidate mask outside the matrix-core path (f32 and binary stores,
// other metrics, larger k): 32, eight or four queries per pass over the rows where a multi-query
// scan serves the store, else one scan each; the launches of a chunk are queued back to back on the
// workspace's stream, with ONE upload of the queries, ONE download of the results and ONE wait
// per chunk.
int32_t run_many_and_fetch(vs_store* s, Workspace* ws, const float* h_queries, uint32_t n_q, uint32_t k,
                           const uint64_t* d_cand, uint32_t* out_ids, float* out_dist, uint32_t* out_n) {
    const vs::SlabView v = s->view();
    const vs::ScanTuning& t = s->ctx->tuning;
    // several queries share every pass over the rows where one of the multi-query scans serves the store
    const bool shared = shared_pass_serves(v, k);
    const bool shared_bits = vs::scan_hamming_multi_serves(v, k);  // binary rows: four queries per pass
    if (int32_t rc = grow(&ws->d_partial, &ws->partial_bytes,
                          shared        ? (size_t)shared_pass_partial_bytes(v, k, t)
                          : shared_bits ? (size_t)vs::scan_partial_bytes(v, k, t) * 4
                                        : (size_t)vs::scan_partial_bytes(v, k, t)))
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
            const float* q = h_queries + (size_t)(at + i) * s->dim;
            unsigned char* dst = hq.data() + i * q_stride;
            if (bin) {  // quantised like the rows: bit j = (q[j] > 0), zero padded to whole vectors
                uint64_t* w = reinterpret_cast<uint64_t*>(dst);
                memset(w, 0, q_bytes);
                for (uint32_t j = 0; j < s->dim; ++j)
                    if (q[j] > 0.0f) w[j >> 6] |= 1ull << (j & 63);
            } else {
                memcpy(dst, q, q_bytes);
            }
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
        } else if (shared_bits) {
            for (uint32_t i = 0; i < m; i += 4)
                HIP_TRY(vs::launch_scan_hamming_multi(v, ws->d_batch + i * q_stride, (uint32_t)(q_stride / 16),
                                                      m - i < 4 ? m - i : 4, d_cand, k, ws->d_partial, nullptr,
                                                      d_ids + (size_t)i * k, d_dist + (size_t)i * k, d_n + i, t,
                                                      ws->stream));
        } else {
            for (uint32_t i = 0; i < m; ++i)
                if (int32_t rc = launch_scan(s, t, ws->d_batch + i * q_stride, d_cand, k, ws->d_partial, nullptr,
                                             d_ids + (size_t)i * k, d_dist + (size_t)i * k, d_n + i,
                                             ws->stream))
                    return rc;
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

// Copies the caller's docid-indexed mask into ws->d_mask, aligned to the store.
int32_t stage_dense_mask(vs_store* s, Workspace* ws, const uint64_t* mask_words,
                         uint64_t n_mask_words) {
    const uint64_t n_words = s->n_slots / 64;
    if (int32_t rc = grow(&ws->d_mask, &ws->mask_words, (size_t)n_words * 8)) return rc;
    const uint64_t w0 = s->docid_base / 64;
    const uint64_t avail = n_mask_words > w0 ? n_mask_words - w0 : 0;
    const uint64_t ncopy = avail < n_words ? avail : n_words;
    if (ncopy)
        HIP_TRY(hipMemcpyAsync(ws->d_mask, mask_words + w0, (size_t)ncopy * 8, hipMemcpyHostToDevice,
                               ws->stream));
    if (ncopy < n_words)
        HIP_TRY(hipMemsetAsync(ws->d_mask + ncopy, 0, (size_t)(n_words - ncopy) * 8, ws->stream));
    return VS_OK;
}

// Expands a parsed serialised RoaringBitmap into `d_words` (n_slots / 64 words aligned with the store).
int32_t expand_roaring(vs_store* s, Workspace* ws, const uint8_t* roaring_bytes, size_t len,
                       const std::vector<vs::RoaringContainerDesc>& desc, uint64_t* d_words) {
    const uint64_t n_words = s->n_slots / 64;
    HIP_TRY(hipMemsetAsync(d_words, 0, (size_t)n_words * 8, ws->stream));
    if (desc.empty()) return VS_OK;
    const size_t blob = (len + 15) / 16 * 16;
    const size_t desc_b = desc.size() * sizeof(vs::RoaringContainerDesc);
    if (int32_t rc = grow(&ws->d_blob, &ws->blob_bytes, blob + desc_b)) return rc;
    HIP_TRY(hipMemcpyAsync(ws->d_blob, roaring_bytes, len, hipMemcpyHostToDevice, ws->stream));
    HIP_TRY(hipMemcpyAsync(ws->d_blob + blob, desc.data(), desc_b, hipMemcpyHostToDevice, ws->stream));
    HIP_TRY(vs::launch_roaring_expand(ws->d_blob,
                                      reinterpret_cast<const vs::RoaringContainerDesc*>(ws->d_blob + blob),
                                      (uint32_t)desc.size(), s->docid_base, d_words, n_words, ws->stream));
    return VS_OK;
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
    std::unique_ptr<vs_ctx> ctx(new vs_ctx());
    ctx->device = device_ordinal;
    if (const char* v = getenv("VS_SCAN_BLOCKS")) ctx->tuning.blocks = atoi(v);
    if (const char* v = getenv("VS_SCAN_NT")) ctx->tuning.nt = atoi(v);
    if (const char* v = getenv("VS_SCAN_ROWS")) ctx->tuning.rows_in_flight = atoi(v);
    HIP_TRY(hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
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
    if (blocks < 0 || blocks > 65535) return fail(VS_E_INVALID_ARG, "bad tuning values");
    ctx->tuning.blocks = blocks;
    ctx->tuning.nt = nt ? 1 : 0;
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
    s->h_present.assign(n_words, 0);
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        ctx->stores[store_index] = s.get();
    }
    *out = s.release();
    return VS_OK;
}

extern "C" int32_t vs_store_reserve(vs_store* s, uint64_t capacity) {
    if (int32_t rc = check_store(s)) return rc;
    if (capacity <= s->capacity) return VS_OK;
    if ((uint64_t)s->docid_base + capacity > (1ull << 32))
        return fail(VS_E_INVALID_ARG, "docid range [%u, %u + %llu) does not fit u32 docids", s->docid_base,
                    s->docid_base, (unsigned long long)capacity);
    HIP_TRY(hipSetDevice(s->ctx->device));
    HIP_TRY(hipDeviceSynchronize());  // nothing of this store is in flight (the caller's side of the contract)
    const uint64_t old_slots = s->n_slots, new_slots = (capacity + 63) / 64 * 64;
    const size_t rb = s->row_bytes(), old_words = old_slots / 64, new_words = new_slots / 64;
    void* rows = nullptr;
    float* norms = nullptr;
    uint64_t* present[2] = {nullptr, nullptr};
    auto body = [&]() -> int32_t {
        HIP_TRY(hipMalloc(&rows, (size_t)(new_slots + vs::SLAB_PAD_ROWS) * rb));
        HIP_TRY(hipMemcpy(rows, s->d_rows, (size_t)old_slots * rb, hipMemcpyDeviceToDevice));
        HIP_TRY(hipMemset(static_cast<unsigned char*>(rows) + (size_t)old_slots * rb, 0,
                          (size_t)(new_slots - old_slots + vs::SLAB_PAD_ROWS) * rb));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipMalloc(&present[i], new_words * 8));
            HIP_TRY(hipMemcpy(present[i], s->d_present[i], old_words * 8, hipMemcpyDeviceToDevice));
            HIP_TRY(hipMemset(present[i] + old_words, 0, (new_words - old_words) * 8));
        }
        if (s->d_norms) {
            HIP_TRY(hipMalloc(&norms, (size_t)new_slots * 4));
            HIP_TRY(hipMemcpy(norms, s->d_norms, (size_t)old_slots * 4, hipMemcpyDeviceToDevice));
            HIP_TRY(hipMemset(norms + old_slots, 0, (size_t)(new_slots - old_slots) * 4));
        }
        HIP_TRY(hipDeviceSynchronize());
        return VS_OK;
    };
    if (int32_t rc = body()) {
        if (rows) (void)hipFree(rows);
        if (norms) (void)hipFree(norms);
        for (int i = 0; i < 2; ++i)
            if (present[i]) (void)hipFree(present[i]);
        return rc;
    }
    (void)hipFree(s->d_rows);
    s->d_rows = rows;
    if (s->d_norms) {
        (void)hipFree(s->d_norms);
        s->d_norms = norms;
    }
    for (int i = 0; i < 2; ++i) {
        (void)hipFree(s->d_present[i]);
        s->d_present[i] = present[i];
    }
    s->h_present.resize(new_words, 0);
    s->capacity = capacity;
    s->n_slots = new_slots;
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
    (void)hipDeviceSynchronize();
    {
        std::lock_guard<std::mutex> g(s->ctx->mu);
        s->ctx->stores.erase(s->index);
    }
    if (s->d_rows) (void)hipFree(s->d_rows);
    if (s->d_norms) (void)hipFree(s->d_norms);
    if (s->d_present[0]) (void)hipFree(s->d_present[0]);
    if (s->d_present[1]) (void)hipFree(s->d_present[1]);
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
    if (!s->dirty) {
        s->dirty = true;
        s->dirty_lo = s->dirty_hi = slot;
    } else {
        if (slot < s->dirty_lo) s->dirty_lo = slot;
        if (slot > s->dirty_hi) s->dirty_hi = slot;
    }
}
}  // namespace

extern "C" int32_t vs_store_upload(vs_store* s, const uint32_t* docids, const float* rows,
                                   uint64_t n) {
    if (int32_t rc = check_store(s)) return rc;
    if (n == 0) return VS_OK;
    if (!docids || !rows) return fail(VS_E_INVALID_ARG, "null docids or rows");
    for (uint64_t i = 0; i < n; ++i)
        if (docids[i] < s->docid_base || (uint64_t)docids[i] - s->docid_base >= s->capacity)
            return fail(VS_E_CAPACITY, "docid %u outside the store's range [%u, %llu)", docids[i],
                        s->docid_base, (unsigned long long)(s->docid_base + s->capacity));
    vs_ctx* ctx = s->ctx;
    std::lock_guard<std::mutex> g(ctx->upload_mu);
    if (!ctx->h_stage) HIP_TRY(hipHostMalloc(&ctx->h_stage, STAGE_BYTES, hipHostMallocDefault));
    if (!ctx->d_stage) HIP_TRY(hipMalloc(&ctx->d_stage, STAGE_BYTES));
    const size_t per_row = (size_t)s->dim * 4 + 4;
    const uint64_t chunk = STAGE_BYTES / per_row;
    const vs::SlabView v = s->view();
    for (uint64_t at = 0; at < n; at += chunk) {
        const uint64_t m = n - at < chunk ? n - at : chunk;
        const size_t row_b = (size_t)m * s->dim * 4;
        memcpy(ctx->h_stage, rows + at * s->dim, row_b);
        memcpy(ctx->h_stage + row_b, docids + at, (size_t)m * 4);
        HIP_TRY(hipMemcpyAsync(ctx->d_stage, ctx->h_stage, row_b + (size_t)m * 4,
                               hipMemcpyHostToDevice, ctx->upload_stream));
        HIP_TRY(vs::launch_ingest_f32(v, s->d_rows, reinterpret_cast<const float*>(ctx->d_stage),
                                      reinterpret_cast<const uint32_t*>(ctx->d_stage + row_b), m,
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
    const uint64_t slot0 = first_docid - s->docid_base;
    HIP_TRY(hipMemcpyAsync(static_cast<unsigned char*>(s->d_rows) + slot0 * s->row_bytes(),
                           device_rows, (size_t)n * s->row_bytes(), hipMemcpyDeviceToDevice,
                           static_cast<hipStream_t>(hip_stream)));
    std::lock_guard<std::mutex> g(s->ctx->upload_mu);
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
    // uploads may have been enqueued on caller streams
    HIP_TRY(hipDeviceSynchronize());
    if (s->dirty && s->metric == VS_ANGULAR) {
        const vs::SlabView v = s->view();
        HIP_TRY(vs::launch_norms(v, s->d_norms, s->dirty_lo, s->dirty_hi - s->dirty_lo + 1,
                                 ctx->upload_stream));
    }
    const int next = 1 - s->active.load();
    HIP_TRY(hipMemcpyAsync(s->d_present[next], s->h_present.data(), s->h_present.size() * 8,
                           hipMemcpyHostToDevice, ctx->upload_stream));
    HIP_TRY(hipStreamSynchronize(ctx->upload_stream));
    s->active.store(next, std::memory_order_release);
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
    int32_t has = 0;
    vs_store_contains(s, docid, &has);
    if (!has) return fail(VS_E_INVALID_ARG, "item %u is not in store %u", docid, s->index);
    std::vector<unsigned char> row(s->row_bytes());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(row.data(),
                      static_cast<unsigned char*>(s->d_rows) +
                          (uint64_t)(docid - s->docid_base) * s->row_bytes(),
                      row.size(), hipMemcpyDeviceToHost));
    for (uint32_t e = 0; e < s->dim; ++e) {
        if (s->dtype == VS_F32) {
            memcpy(&out[e], row.data() + 4 * e, 4);
        } else if (s->dtype == VS_BF16) {
            uint16_t h;
            memcpy(&h, row.data() + 2 * e, 2);
            const uint32_t u = (uint32_t)h << 16;
            memcpy(&out[e], &u, 4);
        } else {
            uint64_t w;
            memcpy(&w, row.data() + 8 * (e / 64), 8);
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
    std::unique_ptr<Workspace> ws;
    if (int32_t rc = ws_acquire(s->ctx, &ws)) return rc;
    int32_t rc = VS_OK;
    const uint64_t* d_cand = nullptr;
    if (mask_words) {
        rc = stage_dense_mask(s, ws.get(), mask_words, n_mask_words);
        d_cand = ws->d_mask;
    }
    if (rc == VS_OK) rc = run_and_fetch(s, ws.get(), query, k, d_cand, out_ids, out_dist, out_n);
    ws_release(s->ctx, std::move(ws));
    return rc;
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
    std::unique_ptr<Workspace> ws;
    if (int32_t rc = ws_acquire(s->ctx, &ws)) return rc;
    auto body = [&]() -> int32_t {
        const uint64_t n_words = s->n_slots / 64;
        if (int32_t rc = grow(&ws->d_mask, &ws->mask_words, (size_t)n_words * 8)) return rc;
        if (int32_t rc = expand_roaring(s, ws.get(), roaring_bytes, len, desc, ws->d_mask)) return rc;
        return run_and_fetch(s, ws.get(), query, k, ws->d_mask, out_ids, out_dist, out_n);
    };
    const int32_t rc = body();
    ws_release(s->ctx, std::move(ws));
    return rc;
}

// ---- candidate masks kept on the device ------------------------------------------------------

namespace {
int32_t mask_new(vs_store* s, vs_mask** out, std::unique_ptr<vs_mask>* m) {
    if (!out) return fail(VS_E_INVALID_ARG, "null out pointer");
    *out = nullptr;
    if (int32_t rc = check_store(s)) return rc;
    m->reset(new vs_mask{s, nullptr, s->n_slots});
    HIP_TRY(hipSetDevice(s->ctx->device));
    if (hipMalloc(&(*m)->d_words, (size_t)(s->n_slots / 64) * 8) != hipSuccess)
        return fail(VS_E_OOM, "cannot allocate a %llu-byte mask", (unsigned long long)(s->n_slots / 8));
    return VS_OK;
}
}  // namespace

extern "C" int32_t vs_mask_create(vs_store* s, const uint64_t* mask_words, uint64_t n_mask_words,
                                  vs_mask** out) {
    std::unique_ptr<vs_mask> m;
    if (int32_t rc = mask_new(s, out, &m)) return rc;
    if (!mask_words && n_mask_words) {
        (void)hipFree(m->d_words);
        return fail(VS_E_INVALID_ARG, "null mask words");
    }
    std::unique_ptr<Workspace> ws;
    int32_t rc = ws_acquire(s->ctx, &ws);
    if (rc == VS_OK) {
        // stage through the workspace's routine, then keep the words
        rc = stage_dense_mask(s, ws.get(), mask_words, n_mask_words);
        if (rc == VS_OK) {
            rc = [&]() -> int32_t {
                HIP_TRY(hipMemcpyAsync(m->d_words, ws->d_mask, (size_t)(s->n_slots / 64) * 8,
                                       hipMemcpyDeviceToDevice, ws->stream));
                HIP_TRY(hipStreamSynchronize(ws->stream));
                return VS_OK;
            }();
        }
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

extern "C" int32_t vs_search_mask(vs_store* s, const float* query, uint32_t dim, uint32_t k,
                                  const vs_mask* mask, uint32_t* out_ids, float* out_dist,
                                  uint32_t* out_n) {
    if (int32_t rc = begin_search(s, query, dim, k, out_ids, out_dist, out_n)) return rc;
    *out_n = 0;
    if (mask && mask->store != s) return fail(VS_E_INVALID_ARG, "the mask was created for another store");
    if (mask && mask->n_slots != s->n_slots)
        return fail(VS_E_INVALID_ARG, "the mask was created before the store grew: create it again");
    if (k == 0) return VS_OK;
    std::unique_ptr<Workspace> ws;
    if (int32_t rc = ws_acquire(s->ctx, &ws)) return rc;
    const int32_t rc = run_and_fetch(s, ws.get(), query, k, mask ? mask->d_words : nullptr, out_ids, out_dist, out_n);
    ws_release(s->ctx, std::move(ws));
    return rc;
}

extern "C" int32_t vs_search_batch(vs_store* s, const float* queries, uint32_t n_q, uint32_t dim,
                                   uint32_t k, const uint64_t* mask_words, uint64_t n_mask_words,
                                   uint32_t* out_ids, float* out_dist, uint32_t* out_n) {
    if (n_q == 0) return VS_OK;
    if (int32_t rc = begin_search(s, queries, dim, k, out_ids, out_dist, out_n)) return rc;
    for (uint32_t i = 0; i < n_q; ++i) out_n[i] = 0;
    if (k == 0) return VS_OK;
    std::unique_ptr<Workspace> ws;
    if (int32_t rc = ws_acquire(s->ctx, &ws)) return rc;
    int32_t rc = VS_OK;
    const uint64_t* d_cand = nullptr;
    if (mask_words) {
        rc = stage_dense_mask(s, ws.get(), mask_words, n_mask_words);
        d_cand = ws->d_mask;
    }
    if (rc == VS_OK && on_matrix_cores(s, k) && n_q > 1) {
        rc = run_batch_and_fetch(s, ws.get(), queries, n_q, k, d_cand, out_ids, out_dist, out_n);
    } else if (rc == VS_OK) {
        rc = run_many_and_fetch(s, ws.get(), queries, n_q, k, d_cand, out_ids, out_dist, out_n);
    }
    ws_release(s->ctx, std::move(ws));
    return rc;
}

// ---- search, device buffers ------------------------------------------------------------------

extern "C" int32_t vs_search_workspace_bytes(vs_store* s, uint32_t k, uint64_t* bytes) {
    if (!s || !bytes) return fail(VS_E_INVALID_ARG, "null argument");
    if (k == 0 || k > VS_MAX_K) return fail(VS_E_INVALID_ARG, "k = %u out of range", k);
    *bytes = vs::scan_partial_bytes(s->view(), k, s->ctx->tuning);
    return VS_OK;
}

extern "C" int32_t vs_search_device(vs_store* s, const void* d_query, uint32_t dim, uint32_t k,
                                    const uint64_t* d_mask_words, void* d_workspace,
                                    uint64_t* d_out_keys, void* hip_stream) {
    if (int32_t rc = check_store(s)) return rc;
    if (!d_query || !d_workspace || !d_out_keys) return fail(VS_E_INVALID_ARG, "null device buffer");
    if (!s->sealed.load()) return fail(VS_E_NOT_SEALED, "store %u was never sealed", s->index);
    if (dim != s->dim) {
        g_dim_expected = s->dim;
        g_dim_received = dim;
        return fail(VS_E_DIM_MISMATCH, "invalid vector dimensions: expected %u, received %u", s->dim,
                    dim);
    }
    if (k == 0 || k > VS_MAX_K) return fail(VS_E_INVALID_ARG, "k = %u out of range", k);
    return launch_scan(s, s->ctx->tuning, d_query, d_mask_words, k, static_cast<uint64_t*>(d_workspace),
                       d_out_keys, nullptr, nullptr, nullptr, static_cast<hipStream_t>(hip_stream));
}

// vs_search_device without a workspace argument and without any allocation: the scan's partial
// lists go to this GPU's static scratch array. The scan overwrites every entry of the partial
// buffer it later merges (each workgroup writes its k keys, padded with VS_KEY_NONE), so nothing
// needs clearing between calls; calls that overlap on one GPU would share the array and race.
extern "C" int32_t vs_search_device_noalloc(vs_store* s, const void* d_query, uint32_t dim, uint32_t k,
                                            const uint64_t* d_mask_words, uint64_t* d_out_keys,
                                            void* hip_stream) {
    if (int32_t rc = check_store(s)) return rc;
    if (!d_query || !d_out_keys) return fail(VS_E_INVALID_ARG, "null device buffer");
    if (!s->sealed.load()) return fail(VS_E_NOT_SEALED, "store %u was never sealed", s->index);
    if (dim != s->dim) {
        g_dim_expected = s->dim;
        g_dim_received = dim;
        return fail(VS_E_DIM_MISMATCH, "invalid vector dimensions: expected %u, received %u", s->dim,
                    dim);
    }
    if (k == 0 || k > VS_MAX_K) return fail(VS_E_INVALID_ARG, "k = %u out of range", k);
    const vs::ScanTuning t = s->ctx->tuning;  // one snapshot: the size check and the launch agree
    const uint64_t need = vs::scan_partial_bytes(s->view(), k, t);
    if (need > NOALLOC_SCRATCH_BYTES)
        return fail(VS_E_CAPACITY,
                    "vs_search_device_noalloc: this store and k = %u need %llu bytes of partial lists, the "
                    "static scratch holds %llu (use vs_search_device with a workspace)",
                    k, (unsigned long long)need, (unsigned long long)NOALLOC_SCRATCH_BYTES);
    return launch_scan(s, t, d_query, d_mask_words, k, reinterpret_cast<uint64_t*>(s->ctx->d_scratch),
                       d_out_keys, nullptr, nullptr, nullptr, static_cast<hipStream_t>(hip_stream));
}

extern "C" int32_t vs_search_batch_workspace_bytes(vs_store* s, uint32_t k, uint64_t* bytes) {
    if (!s || !bytes) return fail(VS_E_INVALID_ARG, "null argument");
    const vs::SlabView v = s->view();
    if (on_matrix_cores(s, k)) {
        *bytes = vs::batch_workspace_bytes(v, k);
        return VS_OK;
    }
    if (vs::scan_hamming_multi_serves(v, k)) {
        *bytes = vs::scan_partial_bytes(v, k, s->ctx->tuning) * 4;
        return VS_OK;
    }
    if (!shared_pass_serves(v, k))
        return fail(VS_E_INVALID_ARG,
                    "no batched scan for this store and k = %u (angular bf16 rows with k <= %u; f32 rows "
                    "of up to 1536 or bf16 rows of up to 1536 dimensions with k <= 128)", k, VS_BATCH_MAX_K);
    *bytes = shared_pass_partial_bytes(v, k, s->ctx->tuning);
    return VS_OK;
}

extern "C" int32_t vs_search_batch_device(vs_store* s, const float* d_queries, uint32_t n_q,
                                          uint32_t dim, uint32_t k, const uint64_t* d_mask_words,
                                          void* d_workspace, uint64_t* d_out_keys, void* hip_stream) {
    if (int32_t rc = check_store(s)) return rc;
    if (!d_queries || !d_workspace || !d_out_keys) return fail(VS_E_INVALID_ARG, "null device buffer");
    if (!s->sealed.load()) return fail(VS_E_NOT_SEALED, "store %u was never sealed", s->index);
    if (dim != s->dim) {
        g_dim_expected = s->dim;
        g_dim_received = dim;
        return fail(VS_E_DIM_MISMATCH, "invalid vector dimensions: expected %u, received %u", s->dim,
                    dim);
    }
    if (n_q == 0 || n_q > VS_BATCH_MAX_QUERIES)
        return fail(VS_E_INVALID_ARG, "n_q = %u out of range", n_q);
    const vs::SlabView v = s->view();
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (on_matrix_cores(s, k)) {
        HIP_TRY(vs::launch_scan_batch_bf16(v, d_queries, n_q, d_mask_words, k, d_workspace, d_out_keys,
                                           nullptr, nullptr, nullptr, stream));
        return VS_OK;
    }
    if (vs::scan_hamming_multi_serves(v, k)) {
        // binary store: d_queries holds n_q packed queries, vs_store_row_bytes bytes each
        const unsigned char* bits = reinterpret_cast<const unsigned char*>(d_queries);
        for (uint32_t i = 0; i < n_q; i += 4)
            HIP_TRY(vs::launch_scan_hamming_multi(v, bits + (size_t)i * s->row_bytes(), s->row_vecs,
                                                  n_q - i < 4 ? n_q - i : 4, d_mask_words, k,
                                                  static_cast<uint64_t*>(d_workspace),
                                                  d_out_keys + (size_t)i * k, nullptr, nullptr, nullptr,
                                                  s->ctx->tuning, stream));
        return VS_OK;
    }
    if (!shared_pass_serves(v, k))
        return fail(VS_E_INVALID_ARG, "no batched scan for this store and k = %u", k);
    return queue_shared_passes(v, s->ctx->tuning, d_queries, dim, n_q, d_mask_words, k,
                               static_cast<uint64_t*>(d_workspace), d_out_keys, nullptr, nullptr, nullptr,
                               stream);
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
