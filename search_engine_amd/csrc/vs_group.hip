// Multi-GPU searches behind the C ABI: one process per GPU, documents sharded by docid range,
// every rank scans its shard, ONE RCCL all-gather moves the shard-local top-k keys over xGMI
// and one launch merges them. The library owns the communicator, so a host without any
// collective library of its own (the reference is a plain Rust crate) gets the multi-GPU path
// from include/vs_abi.h alone.
//
// RCCL is resolved at run time (dlopen of librccl.so.1) the first time a group is made: a
// single-GPU host never loads it, and a process that already carries a copy (PyTorch bundles
// one under the same SONAME) keeps exactly one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

#include "vs_internal.h"

using vsrt::fail;
using vsrt::grow;

namespace {

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

const Rccl* rccl() {
    static Rccl api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
    });
    return api.ok ? &api : nullptr;
}

#define RCCL_TRY(api, expr)                                                                       \
    do {                                                                                          \
        const ncclResult_t r_ = (expr);                                                           \
        if (r_ != ncclSuccess)                                                                    \
            return fail(VS_E_DEVICE, "%s failed: %s", #expr, (api)->GetErrorString(r_));          \
    } while (0)

static_assert(VS_GROUP_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id crosses the ABI as plain bytes");

}  // namespace

struct vs_group {
    vs_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    std::mutex mu;  // one sharded search at a time: a rank's collectives are issued in one order
    uint64_t* d_partial = nullptr;
    size_t partial_bytes = 0;
    uint64_t* d_local = nullptr;  // this shard's keys [n_q][k]
    size_t local_bytes = 0;
    uint64_t* d_gather = nullptr;  // every shard's keys [world][n_q][k]
    size_t gather_bytes = 0;
    // the buffers above serve one search at a time ON THE DEVICE too: a search on another stream than
    // the last one first waits for that one's end
    hipEvent_t last_done = nullptr;
    hipStream_t last_stream = nullptr;
    bool has_last = false;
};

extern "C" int32_t vs_group_unique_id(uint8_t* out) {
    if (!out) return fail(VS_E_INVALID_ARG, "null output");
    const Rccl* api = rccl();
    if (!api) return fail(VS_E_DEVICE, "librccl.so.1 cannot be loaded: %s", dlerror());
    ncclUniqueId id;
    RCCL_TRY(api, api->GetUniqueId(&id));
    memcpy(out, id.internal, VS_GROUP_ID_BYTES);
    return VS_OK;
}

extern "C" int32_t vs_group_create(vs_ctx* ctx, int32_t rank, int32_t world, const uint8_t* unique_id,
                                   vs_group** out) {
    if (!ctx || !out || !unique_id) return fail(VS_E_INVALID_ARG, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world)
        return fail(VS_E_INVALID_ARG, "rank %d of %d out of range", rank, world);
    const Rccl* api = rccl();
    if (!api) return fail(VS_E_DEVICE, "librccl.so.1 cannot be loaded: %s", dlerror());
    HIP_TRY(hipSetDevice(ctx->device));
    std::unique_ptr<vs_group> g(new vs_group());
    g->ctx = ctx;
    g->rank = rank;
    g->world = world;
    ncclUniqueId id;
    memcpy(id.internal, unique_id, VS_GROUP_ID_BYTES);
    HIP_TRY(hipEventCreateWithFlags(&g->last_done, hipEventDisableTiming));
    {
        const ncclResult_t r = api->CommInitRank(&g->comm, world, id, rank);
        if (r != ncclSuccess) {
            (void)hipEventDestroy(g->last_done);
            return fail(VS_E_DEVICE, "ncclCommInitRank failed: %s", api->GetErrorString(r));
        }
    }
    *out = g.release();
    return VS_OK;
}

extern "C" int32_t vs_group_destroy(vs_group* g) {
    if (!g) return VS_OK;
    (void)hipSetDevice(g->ctx->device);
    if (g->comm)
        if (const Rccl* api = rccl()) (void)api->CommDestroy(g->comm);
    if (g->d_partial) (void)hipFree(g->d_partial);
    if (g->d_local) (void)hipFree(g->d_local);
    if (g->d_gather) (void)hipFree(g->d_gather);
    if (g->last_done) (void)hipEventDestroy(g->last_done);
    delete g;
    return VS_OK;
}

extern "C" int32_t vs_search_sharded_device(vs_group* g, vs_store* s, const void* d_queries, uint32_t n_q,
                                            uint32_t dim, uint32_t k, const uint64_t* d_mask_words,
                                            uint32_t flags, uint32_t* d_out_ids, float* d_out_dist,
                                            uint32_t* d_out_n, void* hip_stream) {
    if (!g) return fail(VS_E_INVALID_ARG, "null group");
    if (int32_t rc = vsrt::check_store(s)) return rc;
    if (s->ctx != g->ctx) return fail(VS_E_INVALID_ARG, "the store belongs to another context than the group");
    if (!d_queries || !d_out_ids || !d_out_dist || !d_out_n) return fail(VS_E_INVALID_ARG, "null device buffer");
    if (int32_t rc = vsrt::check_device_search(s, dim, k)) return rc;
    if (n_q == 0 || n_q > VS_BATCH_MAX_QUERIES) return fail(VS_E_INVALID_ARG, "n_q = %u out of range", n_q);
    const Rccl* api = rccl();
    if (!api) return fail(VS_E_DEVICE, "librccl.so.1 cannot be loaded");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const bool gather = g->world > 1 || (flags & VS_SHARDED_ALWAYS_GATHER);
    const vs::ScanTuning t = vsrt::tuning_of(s->ctx);
    std::lock_guard<std::mutex> lock(g->mu);
    if (g->has_last && g->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, g->last_done, 0));
    vsrt::Snapshot snap;
    vsrt::snapshot_begin(s, &snap);
    const vs::SlabView& v = snap.v;
    auto body = [&]() -> int32_t {
        const size_t list_keys = (size_t)n_q * k;
        if (int32_t rc = grow(&g->d_local, &g->local_bytes, list_keys * 8)) return rc;
        if (gather)
            if (int32_t rc = grow(&g->d_gather, &g->gather_bytes, list_keys * 8 * g->world)) return rc;
        bool served = false;
        const uint64_t batch_ws = (flags & VS_SHARDED_SHARE_PASSES) ? vsrt::batch_workspace_bytes_for(s, v, k, &served) : 0;
        // where the shard-local result goes: straight to the outputs when nothing is gathered
        uint64_t* local_keys = gather ? g->d_local : nullptr;
        if (served) {
            // queries share passes over the shard (the batched scans): keys first, decoded below
            if (int32_t rc = grow(&g->d_partial, &g->partial_bytes, (size_t)batch_ws)) return rc;
            if (int32_t rc = vsrt::launch_batch(s, v, t, d_queries, n_q, k, d_mask_words, g->d_partial, g->d_local,
                                                stream))
                return rc;
            if (!gather)
                HIP_TRY(vs::launch_merge_batch(g->d_local, 1, k, 1, n_q, k, k, v.metric, nullptr, d_out_ids,
                                               d_out_dist, d_out_n, stream));
        } else {
            // ONE launch scans for every query in turn (no row read is shared: each list is bit-identical
            // to vs_search's), ONE launch merges the per-workgroup lists of all queries
            const uint32_t blocks = vs::scan_grid_blocks(v, t, d_mask_words != nullptr);
            const size_t per_query = (size_t)blocks * k;
            if (int32_t rc = grow(&g->d_partial, &g->partial_bytes, per_query * 8 * n_q)) return rc;
            const size_t q_bytes = v.dtype == VS_BIN1 ? (size_t)v.row_vecs * 16 : (size_t)v.dim * 4;
            if (int32_t rc = vsrt::launch_scan_queries(v, t, d_queries, q_bytes, n_q, d_mask_words, nullptr, k,
                                                       g->d_partial, stream))
                return rc;
            HIP_TRY(vs::launch_merge_batch(g->d_partial, blocks, 1, blocks, n_q, per_query, k, v.metric, local_keys,
                                           gather ? nullptr : d_out_ids, gather ? nullptr : d_out_dist,
                                           gather ? nullptr : d_out_n, stream));
        }
        if (gather) {
            // [n_q][k] keys per rank -> [world][n_q][k]; 8 * k * n_q bytes per rank: latency-bound
            RCCL_TRY(api, api->AllGather(g->d_local, g->d_gather, list_keys, ncclUint64, g->comm, stream));
            HIP_TRY(vs::launch_merge_batch(g->d_gather, (uint32_t)g->world, list_keys, 1, n_q, k, k, v.metric, nullptr,
                                           d_out_ids, d_out_dist, d_out_n, stream));
        }
        return VS_OK;
    };
    const int32_t rc = body();
    const int32_t rc2 = vsrt::snapshot_end_async(s, &snap, stream, true);
    if (hipEventRecord(g->last_done, stream) == hipSuccess) {
        g->last_stream = stream;
        g->has_last = true;
    }
    return rc ? rc : rc2;
}
