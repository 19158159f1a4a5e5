"""The pieces of the multi-GPU path that one GPU can run for real: the merge launch on a gathered
[rank][query][k] buffer, eight docid-range shards searched and merged on one device, and
vs_search_sharded_device itself in a group of one (with and without the collective). What is left
for a multi-GPU node is RCCL moving bytes between processes."""
import numpy as np
import pytest
import torch

from conftest import facet_mask, synthetic_vectors

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _merge_on_gpu(ctx, keys, n_lists, list_stride, n_queries, query_stride, k, metric=0):
    d_keys = torch.from_numpy(np.ascontiguousarray(keys).view(np.int64)).to(DEV)
    ids = torch.zeros(n_queries * k, dtype=torch.int32, device=DEV)
    dist = torch.zeros(n_queries * k, dtype=torch.float32, device=DEV)
    cnt = torch.zeros(n_queries, dtype=torch.int32, device=DEV)
    ctx.merge_topk_batch_device(d_keys.data_ptr(), n_lists, list_stride, n_queries, query_stride, k, metric,
                                ids.data_ptr(), dist.data_ptr(), cnt.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return (ids.cpu().numpy().view(np.uint32).reshape(n_queries, k), dist.cpu().numpy().reshape(n_queries, k),
            cnt.cpu().numpy())


@pytest.mark.parametrize("world,n_queries,k", [(8, 16, 10), (8, 3, 100), (2, 256, 10), (5, 7, 1), (8, 4, 300)])
def test_merge_of_a_gathered_buffer_follows_the_layout_contract(gpu_ctx, world, n_queries, k):
    """Synthetic [world][query][k] keys: short lists padded with VS_KEY_NONE, empty shards, distances
    that tie exactly across shards (the docid decides), one query with fewer than k hits overall."""
    from search_engine_amd import sharded
    rng = np.random.default_rng(world * 1000 + k)
    gathered = np.full((world, n_queries, k), sharded.KEY_NONE, np.uint64)
    for q in range(n_queries):
        few = q == n_queries - 1  # the last query: fewer than k hits in all shards together
        levels = rng.random(6).astype(np.float32)  # a handful of distance values: many exact ties
        for r in range(world):
            n = int(rng.integers(0, k + 1))
            if few:
                n = min(n, max(0, (k - 1) // world))
            if r == 1 % world and q % 2 == 0:
                n = 0  # an empty shard
            ids = rng.choice(np.arange(r * 100_000, (r + 1) * 100_000, dtype=np.uint32), n, replace=False)
            d = rng.choice(levels, n) if q % 3 else rng.random(n).astype(np.float32)
            keys = np.sort(sharded.pack_keys(d, ids))
            gathered[r, q, :n] = keys
    want = sharded.merge_keys_cpu(gathered, world, n_queries * k, n_queries, k, k)
    ids, dist, cnt = _merge_on_gpu(gpu_ctx, gathered, world, n_queries * k, n_queries, k, k)
    for q in range(n_queries):
        w_ids, w_d = sharded.unpack_keys(want[q])
        assert cnt[q] == w_ids.size
        assert ids[q, : cnt[q]].tolist() == w_ids.tolist()
        assert dist[q, : cnt[q]].tobytes() == w_d.tobytes()
    assert cnt[-1] < k or k == 1


@pytest.mark.parametrize("dtype_name", ["f32", "bin1"])
def test_eight_shards_on_one_gpu_equal_the_single_store(gpu_ctx, dtype_name):
    """Stores with docid_base = g * per play the eight ranks: vs_search_device on each, the keys laid
    out as the all-gather would, one merge — against one store that holds everything."""
    from search_engine_amd import abi, sharded
    world, n, dim, k, n_q = 8, 1_250_000 if dtype_name == "bin1" else 200_000, 256 if dtype_name == "bin1" else 64, 10, 6
    dtype, metric = (abi.VS_BIN1, abi.VS_HAMMING) if dtype_name == "bin1" else (abi.VS_F32, abi.VS_ANGULAR)
    stream = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=DEV)
    gen.manual_seed(7)
    if dtype_name == "bin1":
        rows = torch.randint(-(1 << 31), 1 << 31, (n, dim // 32), generator=gen, device=DEV, dtype=torch.int64).to(torch.int32)
        queries = torch.randint(-(1 << 31), 1 << 31, (n_q, dim // 32), generator=gen, device=DEV, dtype=torch.int64).to(torch.int32)
    else:
        rows = torch.randn(n, dim, generator=gen, device=DEV)
        queries = torch.randn(n_q, dim, generator=gen, device=DEV)
    mask_np = facet_mask(-(-n // 64) * 64, 10)
    d_mask = torch.from_numpy(mask_np.view(np.int64)).to(DEV)
    whole = gpu_ctx.create_store(800, dim, n, dtype=dtype, metric=metric)
    shards = []
    try:
        whole.upload_device(0, rows.data_ptr(), n, stream)
        whole.seal()
        ws = torch.empty(whole.workspace_bytes(k), dtype=torch.uint8, device=DEV)
        want = torch.zeros(n_q * k, dtype=torch.int64, device=DEV)
        for j in range(n_q):
            whole.search_device(queries[j].data_ptr(), dim, k, d_mask.data_ptr(), ws.data_ptr(),
                                want[j * k:].data_ptr(), stream)
        gathered = torch.zeros(world * n_q * k, dtype=torch.int64, device=DEV)
        for g in range(world):
            first, count = sharded.shard_range(g, world, n)
            st = gpu_ctx.create_store(801 + g, dim, max(count, 64), dtype=dtype, metric=metric, docid_base=first)
            shards.append(st)
            st.upload_device(first, rows[first:first + count].data_ptr(), count, stream)
            st.seal()
            for j in range(n_q):
                st.search_device(queries[j].data_ptr(), dim, k, d_mask[first // 64:].data_ptr(), ws.data_ptr(),
                                 gathered[(g * n_q + j) * k:].data_ptr(), stream)
        torch.cuda.synchronize()
        ids, dist, cnt = _merge_on_gpu(gpu_ctx, gathered.cpu().numpy(), world, n_q * k, n_q, k, k, metric)
        want = want.cpu().numpy().view(np.uint64).reshape(n_q, k)
        for j in range(n_q):
            assert cnt[j] == k
            assert ids[j].tolist() == (want[j] & np.uint64(0xFFFFFFFF)).astype(np.uint32).tolist()
        # and the host restatement agrees with the kernel on real keys
        cpu = sharded.merge_keys_cpu(gathered.cpu().numpy(), world, n_q * k, n_q, k, k)
        assert cpu.tolist() == want.tolist()
    finally:
        whole.destroy()
        for st in shards:
            st.destroy()


@pytest.mark.parametrize("flags_name", ["plain", "always_gather", "share_passes", "share_passes_gather"])
@pytest.mark.parametrize("dtype_name", ["f32", "bf16", "bin1"])
def test_sharded_search_in_a_group_of_one(gpu_ctx, dtype_name, flags_name):
    """vs_search_sharded_device, world 1: without the collective and (VS_SHARDED_ALWAYS_GATHER) through
    a real one-rank RCCL all-gather; one scan per query and queries sharing passes."""
    from search_engine_amd import abi
    flags = {"plain": 0, "always_gather": abi.VS_SHARDED_ALWAYS_GATHER, "share_passes": abi.VS_SHARDED_SHARE_PASSES,
             "share_passes_gather": abi.VS_SHARDED_SHARE_PASSES | abi.VS_SHARDED_ALWAYS_GATHER}[flags_name]
    n, dim, k, n_q = 30_000, 128, 10, 9
    dtype, metric = {"f32": (abi.VS_F32, abi.VS_ANGULAR), "bf16": (abi.VS_BF16, abi.VS_ANGULAR),
                     "bin1": (abi.VS_BIN1, abi.VS_HAMMING)}[dtype_name]
    rows = synthetic_vectors(n, dim, 71)
    queries = synthetic_vectors(n_q, dim, 72)
    mask = facet_mask(-(-n // 64) * 64, 50)
    store = gpu_ctx.create_store(820, dim, n, dtype=dtype, metric=metric)
    group = abi.Group(gpu_ctx, 0, 1, abi.group_unique_id())
    try:
        store.upload(np.arange(n, dtype=np.uint32), rows)
        store.seal()
        if dtype_name == "bin1":
            packed = np.zeros((n_q, store.row_bytes // 8), np.uint64)
            bits = np.packbits(queries > 0, axis=1, bitorder="little")
            packed.view(np.uint8)[:, : bits.shape[1]] = bits
            d_q = torch.from_numpy(packed.view(np.int64)).to(DEV)
        else:
            d_q = torch.from_numpy(queries).to(DEV)
        d_mask = torch.from_numpy(mask.view(np.int64)).to(DEV)
        ids = torch.zeros(n_q * k, dtype=torch.int32, device=DEV)
        dist = torch.zeros(n_q * k, dtype=torch.float32, device=DEV)
        cnt = torch.zeros(n_q, dtype=torch.int32, device=DEV)
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(2):  # the second call reuses the group's buffers
            group.search_sharded_device(store, d_q.data_ptr(), n_q, dim, k, d_mask.data_ptr(), flags,
                                        ids.data_ptr(), dist.data_ptr(), cnt.data_ptr(), stream)
        torch.cuda.synchronize()
        ids = ids.cpu().numpy().view(np.uint32).reshape(n_q, k)
        dist = dist.cpu().numpy().reshape(n_q, k)
        shares = flags & abi.VS_SHARDED_SHARE_PASSES
        want = store.search_batch(queries, k, mask) if shares else [store.search(q, k, mask) for q in queries]
        for j in range(n_q):
            assert cnt[j].item() == k
            assert ids[j].tolist() == want[j][0].tolist(), (j, flags_name)
            assert dist[j].tobytes() == want[j][1].tobytes()
    finally:
        group.destroy()
        store.destroy()


@pytest.mark.parametrize("k", [10, 200])
@pytest.mark.parametrize("pass_percent", [100, 50, 3])
@pytest.mark.parametrize("shape", ["f32_768", "f32_96", "bf16_768", "bin1_1536", "bin1_256"])
def test_a_step_of_queries_in_one_launch_equals_their_own_searches(gpu_ctx, shape, pass_percent, k):
    """vs_search_sharded_device without VS_SHARDED_SHARE_PASSES runs ONE scan launch for all queries of
    the step (a workgroup takes its share of the mask words for one query after the other, several
    queries in flight under a selective filter, one where the filter is dense). No row read is shared,
    so every list must be bit for bit the list of the query's own vs_search_device: long rows (paired
    workgroups), short rows, binary rows with and without the whole-word path, lists in registers
    (k = 10) and in LDS (k = 200), query counts that do and do not fill the lanes."""
    from search_engine_amd import abi
    dtype_name, dim = shape.split("_")
    dim = int(dim)
    dtype, metric = {"f32": (abi.VS_F32, abi.VS_ANGULAR), "bf16": (abi.VS_BF16, abi.VS_ANGULAR),
                     "bin1": (abi.VS_BIN1, abi.VS_HAMMING)}[dtype_name]
    n = 150_000 if dtype_name != "bin1" else 600_000
    stream = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=DEV)
    gen.manual_seed(11)
    if dtype_name == "bin1":
        rows = torch.randint(-(1 << 31), 1 << 31, (n, dim // 32), generator=gen, device=DEV, dtype=torch.int64).to(torch.int32)
        queries = torch.randint(-(1 << 31), 1 << 31, (16, dim // 32), generator=gen, device=DEV, dtype=torch.int64).to(torch.int32)
    else:
        tdt = torch.float32 if dtype_name == "f32" else torch.bfloat16
        rows = torch.randn(n, dim, generator=gen, device=DEV).to(tdt)
        queries = torch.randn(16, dim, generator=gen, device=DEV)
    n_slots = -(-n // 64) * 64
    d_mask = torch.from_numpy(facet_mask(n_slots, pass_percent).view(np.int64)).to(DEV)
    store = gpu_ctx.create_store(830, dim, n, dtype=dtype, metric=metric)
    group = abi.Group(gpu_ctx, 0, 1, abi.group_unique_id())
    try:
        store.upload_device(0, rows.data_ptr(), n, stream)
        torch.cuda.synchronize()
        store.seal()
        ws = torch.empty(store.workspace_bytes(k), dtype=torch.uint8, device=DEV)
        for n_q, masked in ((16, True), (5, True), (7, False), (1, True)):
            mask_ptr = d_mask.data_ptr() if masked else 0
            want = torch.zeros(n_q * k, dtype=torch.int64, device=DEV)
            for j in range(n_q):
                store.search_device(queries[j].data_ptr(), dim, k, mask_ptr, ws.data_ptr(), want[j * k:].data_ptr(), stream)
            ids = torch.zeros(n_q * k, dtype=torch.int32, device=DEV)
            dist = torch.zeros(n_q * k, dtype=torch.float32, device=DEV)
            cnt = torch.zeros(n_q, dtype=torch.int32, device=DEV)
            group.search_sharded_device(store, queries.data_ptr(), n_q, dim, k, mask_ptr, 0,
                                        ids.data_ptr(), dist.data_ptr(), cnt.data_ptr(), stream)
            torch.cuda.synchronize()
            want = want.cpu().numpy().view(np.uint64).reshape(n_q, k)
            got = ids.cpu().numpy().view(np.uint32).reshape(n_q, k)
            assert (cnt.cpu().numpy() == k).all()
            assert got.tolist() == (want & np.uint64(0xFFFFFFFF)).astype(np.uint32).tolist(), (n_q, masked)
            if metric == abi.VS_ANGULAR:  # the key's upper half is the orderable distance: decode and compare bits
                hi = (want >> np.uint64(32)).astype(np.uint32)
                bits = np.where(hi & np.uint32(0x80000000), hi & np.uint32(0x7FFFFFFF), ~hi)
                assert dist.cpu().numpy().view(np.uint32).reshape(n_q, k).tolist() == bits.tolist()
            else:
                assert dist.cpu().numpy().reshape(n_q, k).tolist() == (want >> np.uint64(32)).astype(np.float32).tolist()
    finally:
        group.destroy()
        store.destroy()


def test_eight_bf16_shards_sharing_passes_equal_the_single_store(gpu_ctx):
    """BASELINE config 3's data flow on one GPU: eight docid-range shard stores (bf16), on each the
    batched matrix-core scan of all queries (vs_search_batch_device, what vs_search_sharded_device runs
    under VS_SHARDED_SHARE_PASSES), the keys laid out [rank][query][k] as the all-gather leaves them, ONE
    vs_merge_topk_batch_device -- against vs_search_batch_device on one store holding everything. The
    merge is associative and every shard's list holds its k best, so the ids are the same; distances
    come from the same products in the same order (a row's score does not depend on its shard)."""
    from search_engine_amd import abi, sharded
    world, n, dim, k, n_q = 8, 160_000, 256, 10, 40
    stream = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=DEV)
    gen.manual_seed(23)
    rows = torch.randn(n, dim, generator=gen, device=DEV).to(torch.bfloat16)
    queries = torch.randn(n_q, dim, generator=gen, device=DEV)
    d_mask = torch.from_numpy(facet_mask(-(-n // 64) * 64, 30).view(np.int64)).to(DEV)
    whole = gpu_ctx.create_store(840, dim, n, dtype=abi.VS_BF16)
    shards = []
    try:
        whole.upload_device(0, rows.data_ptr(), n, stream)
        torch.cuda.synchronize()
        whole.seal()
        for masked in (True, False):
            ws = torch.empty(whole.batch_workspace_bytes(k), dtype=torch.uint8, device=DEV)
            want = torch.zeros(n_q * k, dtype=torch.int64, device=DEV)
            whole.search_batch_device(queries.data_ptr(), n_q, dim, k, d_mask.data_ptr() if masked else 0,
                                      ws.data_ptr(), want.data_ptr(), stream)
            gathered = torch.zeros(world * n_q * k, dtype=torch.int64, device=DEV)
            for g in range(world):
                first, count = sharded.shard_range(g, world, n)
                if len(shards) <= g:
                    st = gpu_ctx.create_store(841 + g, dim, max(count, 64), dtype=abi.VS_BF16, docid_base=first)
                    shards.append(st)
                    st.upload_device(first, rows[first:first + count].data_ptr(), count, stream)
                    torch.cuda.synchronize()
                    st.seal()
                st = shards[g]
                sws = torch.empty(st.batch_workspace_bytes(k), dtype=torch.uint8, device=DEV)
                st.search_batch_device(queries.data_ptr(), n_q, dim, k, d_mask[first // 64:].data_ptr() if masked else 0,
                                       sws.data_ptr(), gathered[g * n_q * k:].data_ptr(), stream)
                torch.cuda.synchronize()
            ids, dist, cnt = _merge_on_gpu(gpu_ctx, gathered.cpu().numpy(), world, n_q * k, n_q, k, k, abi.VS_ANGULAR)
            want_keys = want.cpu().numpy().view(np.uint64).reshape(n_q, k)
            assert (cnt == k).all()
            assert ids.tolist() == (want_keys & np.uint64(0xFFFFFFFF)).astype(np.uint32).tolist(), masked
            hi = (want_keys >> np.uint64(32)).astype(np.uint32)
            bits = np.where(hi & np.uint32(0x80000000), hi & np.uint32(0x7FFFFFFF), ~hi)
            assert dist.view(np.uint32).tolist() == bits.tolist()
    finally:
        whole.destroy()
        for st in shards:
            st.destroy()
