"""vs_search_device_noalloc: the device search that takes no workspace and allocates nothing.

It runs the same kernels as vs_search_device with the partial lists in the library's static
device array, so its keys must equal vs_search_device's bit for bit — on two calls in a row
that share the array, the second on a smaller input (the array then still holds the first
call's lists), and under every store type the scan serves. The oracle checks the first result.
"""
import numpy as np
import pytest
import torch

from conftest import facet_mask, synthetic_vectors

pytestmark = pytest.mark.gpu

NOALLOC_SCRATCH_BYTES = 1 << 20


def _device_search(store, q, k, mask, noalloc):
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    d_q = torch.from_numpy(np.ascontiguousarray(q)).to(dev)
    d_mask = None if mask is None else torch.from_numpy(mask.view(np.int64)).to(dev)
    keys = torch.full((k,), 0x55, dtype=torch.int64, device=dev)
    mp = 0 if d_mask is None else d_mask.data_ptr()
    if noalloc:
        store.search_device_noalloc(d_q.data_ptr(), store.dim, k, mp, keys.data_ptr(), stream)
    else:
        ws = torch.empty(store.workspace_bytes(k), dtype=torch.uint8, device=dev)
        store.search_device(d_q.data_ptr(), store.dim, k, mp, ws.data_ptr(), keys.data_ptr(), stream)
    torch.cuda.synchronize()
    return keys.cpu().numpy().view(np.uint64)


def test_two_calls_in_a_row_equal_the_workspace_entry_point(gpu_ctx):
    from oracle import scan as oracle
    from search_engine_amd import sharded

    big_n, dim = 40_000, 96
    rows = synthetic_vectors(big_n, dim, 11)
    big = gpu_ctx.create_store(900, dim, big_n)
    big.upload(np.arange(big_n, dtype=np.uint32), rows)
    big.seal()
    small_n = 1_000
    small = gpu_ctx.create_store(901, dim, small_n)
    small.upload(np.arange(small_n, dtype=np.uint32), rows[:small_n])
    small.seal()
    try:
        q = synthetic_vectors(2, dim, 12)
        mask_big = facet_mask(big_n, 50)
        mask_small = facet_mask(small_n, 10)
        # first call: large store, large k (fills most of what the array can hold per workgroup)
        a = _device_search(big, q[0], 100, mask_big, noalloc=True)
        # second call straight after it, smaller store, smaller k, fewer workgroups
        b = _device_search(small, q[1], 7, mask_small, noalloc=True)
        a_ref = _device_search(big, q[0], 100, mask_big, noalloc=False)
        b_ref = _device_search(small, q[1], 7, mask_small, noalloc=False)
        assert a.tolist() == a_ref.tolist()
        assert b.tolist() == b_ref.tolist()
        # and the first against the oracle
        ids, dist = sharded.unpack_keys(a)
        o_ids, _ = oracle.nns_by_vector(rows, q[0], 100, mask=mask_big)
        assert ids.tolist() == o_ids.tolist()
        d64 = oracle.all_distances_f64(rows, q[0])
        assert np.all(np.abs(dist - d64[ids]) <= 1e-5 * np.abs(d64[ids]) + 1e-6)
        # a third call without a mask, on the small store again (fewer results than k)
        c = _device_search(small, q[0], 10, None, noalloc=True)
        assert c.tolist() == _device_search(small, q[0], 10, None, noalloc=False).tolist()
    finally:
        big.destroy()
        small.destroy()


def test_binary_store_and_the_size_limit(gpu_ctx):
    from search_engine_amd import abi

    n, bits = 160_000, 128
    rows = synthetic_vectors(n, bits, 13)
    store = gpu_ctx.create_store(902, bits, n, dtype=abi.VS_BIN1, metric=abi.VS_HAMMING)
    store.upload(np.arange(n, dtype=np.uint32), rows)
    store.seal()
    try:
        q = synthetic_vectors(1, bits, 14)[0]
        packed = np.packbits(q > 0, bitorder="little").view(np.uint32)
        row_words = store.row_bytes // 4
        packed = np.concatenate([packed, np.zeros(row_words - packed.size, np.uint32)])

        def run(noalloc, k):
            dev = torch.device("cuda", 0)
            d_q = torch.from_numpy(packed.view(np.int32)).to(dev)
            keys = torch.zeros(k, dtype=torch.int64, device=dev)
            stream = torch.cuda.current_stream().cuda_stream
            if noalloc:
                store.search_device_noalloc(d_q.data_ptr(), bits, k, 0, keys.data_ptr(), stream)
            else:
                ws = torch.empty(store.workspace_bytes(k), dtype=torch.uint8, device=dev)
                store.search_device(d_q.data_ptr(), bits, k, 0, ws.data_ptr(), keys.data_ptr(), stream)
            torch.cuda.synchronize()
            return keys.cpu().numpy().view(np.uint64).tolist()

        assert run(True, 100) == run(False, 100)
        assert run(True, 3) == run(False, 3)
        # beyond the static array: refused with a clear error, nothing launched
        k_big = 1024
        assert store.workspace_bytes(k_big) > NOALLOC_SCRATCH_BYTES
        with pytest.raises(abi.VsError) as e:
            run(True, k_big)
        assert e.value.code == abi.VS_E_CAPACITY and "scratch" in e.value.message
    finally:
        store.destroy()


def test_noalloc_search_under_graph_capture(gpu_ctx):
    """What the entry point is for: a caller under HIP graph capture. The call is captured once and
    replayed; a replay reads the query and the mask where they lay at capture and the store as it was
    registered then, so it stays valid until the next vs_store_seal or vs_store_reserve (documented in
    include/vs_abi.h) -- here the store does not change, the query's contents do."""
    dev = torch.device("cuda", 0)
    n, dim, k = 50_000, 96, 10
    rows = synthetic_vectors(n, dim, 21)
    store = gpu_ctx.create_store(910, dim, n)
    try:
        store.upload(np.arange(n, dtype=np.uint32), rows)
        store.seal()
        queries = torch.from_numpy(synthetic_vectors(3, dim, 22)).to(dev)
        d_q = queries[0].clone()
        d_mask = torch.from_numpy(facet_mask(n, 30).view(np.int64)).to(dev)
        keys = torch.zeros(k, dtype=torch.int64, device=dev)
        ws = torch.empty(store.workspace_bytes(k), dtype=torch.uint8, device=dev)
        side = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="relaxed"):
            store.search_device_noalloc(d_q.data_ptr(), dim, k, d_mask.data_ptr(), keys.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream)
        for j in range(3):
            d_q.copy_(queries[j])
            keys.zero_()
            graph.replay()
            torch.cuda.synchronize()
            want = torch.zeros(k, dtype=torch.int64, device=dev)
            store.search_device(queries[j].data_ptr(), dim, k, d_mask.data_ptr(), ws.data_ptr(), want.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert keys.cpu().tolist() == want.cpu().tolist(), j
    finally:
        store.destroy()
