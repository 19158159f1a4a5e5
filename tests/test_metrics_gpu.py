"""GPU parity of the Euclidean and dot-product scans (f32 and bf16 rows), through the C ABI,
against the CPU oracle. Neither metric exists in the reference (Angular stores only,
src/index.rs:155); the definitions under test are arroy's: Euclidean results are ordered by
the squared distance and report its square root, dot-product results are ordered by
descending product and report the product; ties break by ascending docid.

Tolerance, with d64 the float64 value of the reported quantity:
  Euclidean  |d - d64| <= 1e-5 |d64| + 1e-6                      (tests/helpers.py)
  dot        |d - d64| <= 1e-5 |d64| + 1e-6 * ||row|| * ||query||  (the same bound as the
             angular one, whose absolute floor is 1e-6 of a cosine: a product of two long
             vectors that nearly cancels has no meaningful relative error in f32)
Ids equal the oracle's position by position except where two candidates are closer than
those bounds in float64."""
import numpy as np
import pytest

from conftest import facet_mask, synthetic_vectors
from oracle import scan as oracle

pytestmark = pytest.mark.gpu

_next_index = [0x3000]


def check(metric, vectors, query, k, ids, dist, mask, docids):
    from search_engine_amd import abi
    o_metric = oracle.METRIC_DOT if metric == abi.VS_DOT else oracle.METRIC_EUCLIDEAN
    o_ids, _ = oracle.nns_by_vector_metric(vectors, query, k, o_metric, docids=docids, mask=mask)
    f64 = oracle.np_metric_f64(vectors, query, o_metric)
    d64 = dict(zip(docids.tolist(), f64.tolist()))
    if metric == abi.VS_DOT:
        qn = float(np.linalg.norm(query.astype(np.float64)))
        floor = dict(zip(docids.tolist(),
                         (1e-6 * np.linalg.norm(vectors.astype(np.float64), axis=1) * qn).tolist()))
    else:
        floor = {int(i): 1e-6 for i in docids}
    tol = lambda i: 1e-5 * abs(d64[int(i)]) + floor[int(i)]  # noqa: E731
    ok = oracle.mask_to_bool(mask, int(docids.max()) + 1)[docids] if mask is not None else np.ones(len(docids), bool)
    assert len(ids) == len(dist) == min(k, int(ok.sum())) == len(o_ids)
    assert len(set(ids.tolist())) == len(ids)
    for i, d in zip(ids, dist):
        assert ok[np.searchsorted(docids, i)]
        assert abs(float(d) - d64[int(i)]) <= tol(i), (int(i), float(d), d64[int(i)])
    sign = -1.0 if metric == abi.VS_DOT else 1.0
    for a in range(1, len(ids)):  # order of the returned values, ties by docid
        x, y = sign * float(dist[a - 1]), sign * float(dist[a])
        assert x < y or (x == y and ids[a - 1] < ids[a]), f"order broken at {a}"
    swaps = 0
    for a in range(len(ids)):
        if ids[a] != o_ids[a]:
            assert abs(d64[int(ids[a])] - d64[int(o_ids[a])]) <= tol(ids[a]) + tol(o_ids[a]), a
            swaps += 1
    return swaps


def run(ctx, metric, vectors, query, k, mask=None, docids=None, dtype=None, docid_base=0):
    from search_engine_amd import abi
    n, dim = vectors.shape
    if docids is None:
        docids = np.arange(n, dtype=np.uint32) + docid_base
    _next_index[0] += 1
    dtype = abi.VS_F32 if dtype is None else dtype
    store = ctx.create_store(_next_index[0], dim, int(docids.max()) - docid_base + 1, dtype=dtype,
                             metric=metric, docid_base=docid_base)
    try:
        store.upload(docids, vectors)
        store.seal()
        ids, dist = store.search(query, k, mask)
        r_ids, r_dist = store.search_roaring(query, k, abi.mask_to_roaring(mask)) if mask is not None else (ids, dist)
        assert r_ids.tolist() == ids.tolist() and r_dist.tolist() == dist.tolist()
    finally:
        store.destroy()
    stored = oracle.bf16_round(vectors) if dtype == abi.VS_BF16 else vectors
    return check(metric, stored, query, k, ids, dist, mask, docids), ids, dist


def metrics():
    from search_engine_amd import abi
    return [abi.VS_EUCLIDEAN, abi.VS_DOT]


@pytest.mark.parametrize("m", [0, 1])
def test_known_answers(gpu_ctx, m):
    metric = metrics()[m]
    v = np.array([[0, 1, 2], [6, 7, 8], [9, 10, 11], [0, -1, -2]], np.float32)
    q = np.array([0, 1, 2], np.float32)
    _, ids, dist = run(gpu_ctx, metric, v, q, 10)
    if m == 0:  # sqrt(0), sqrt(108), sqrt(243), sqrt(20)
        assert ids.tolist() == [0, 3, 1, 2]
        np.testing.assert_allclose(dist, [0.0, 20 ** 0.5, 108 ** 0.5, 243 ** 0.5], rtol=1e-6)
    else:       # 5, 23, 32, -5: descending
        assert ids.tolist() == [2, 1, 0, 3]
        assert dist.tolist() == [32.0, 23.0, 5.0, -5.0]


@pytest.mark.parametrize("m", [0, 1])
@pytest.mark.parametrize("n,dim,k,pass_percent", [
    (5000, 768, 10, 100), (5000, 768, 10, 10), (20000, 384, 100, 50), (3000, 100, 10, 1),
    (777, 3, 5, 100), (4096, 1536, 10, 30), (2500, 2048, 200, 100), (300, 768, 1024, 100),
])
def test_parity_with_oracle(gpu_ctx, m, n, dim, k, pass_percent):
    metric = metrics()[m]
    vectors = synthetic_vectors(n, dim, seed=n + dim + m)
    for s in range(2):
        q = synthetic_vectors(1, dim, seed=900 + s)[0]
        mask = facet_mask(n, pass_percent) if pass_percent < 100 else None
        swaps, _, _ = run(gpu_ctx, metric, vectors, q, k, mask)
        assert swaps <= 2


@pytest.mark.parametrize("m", [0, 1])
def test_bf16_rows(gpu_ctx, m):
    from search_engine_amd import abi
    metric = metrics()[m]
    vectors = synthetic_vectors(6000, 768, seed=77)
    q = synthetic_vectors(1, 768, seed=78)[0]
    swaps, _, _ = run(gpu_ctx, metric, vectors, q, 10, facet_mask(6000, 40), dtype=abi.VS_BF16)
    assert swaps <= 1


@pytest.mark.parametrize("m", [0, 1])
def test_ties_holes_and_docid_base(gpu_ctx, m):
    """Duplicated rows tie exactly and come out by ascending docid; sparse docids on a base."""
    metric = metrics()[m]
    rng = np.random.default_rng(5)
    base_rows = rng.standard_normal((40, 64)).astype(np.float32)
    vectors = np.concatenate([base_rows, base_rows, base_rows])  # every row three times
    docids = (np.sort(rng.choice(5000, 120, replace=False)) + 6400).astype(np.uint32)
    q = rng.standard_normal(64).astype(np.float32)
    swaps, ids, dist = run(gpu_ctx, metric, vectors, q, 30, docids=docids, docid_base=6400)
    assert swaps == 0
    for a in range(0, 30, 3):
        assert dist[a] == dist[a + 1] == dist[a + 2] and ids[a] < ids[a + 1] < ids[a + 2]


@pytest.mark.parametrize("m", [0, 1])
def test_batch_and_device_entry_points(gpu_ctx, m):
    """vs_search_batch (no matrix-core path for these metrics) and the device-buffer
    search + shard merge give what single searches give."""
    import torch
    from search_engine_amd import abi
    metric = metrics()[m]
    n, dim, k = 4096, 256, 10
    vectors = synthetic_vectors(n, dim, seed=31)
    queries = synthetic_vectors(5, dim, seed=32)
    mask = facet_mask(n, 50)
    store = gpu_ctx.create_store(0x3F00 + m, dim, n, dtype=abi.VS_BF16, metric=metric)
    halves = [gpu_ctx.create_store(0x3F10 + 2 * m + h, dim, n // 2, dtype=abi.VS_BF16, metric=metric,
                                   docid_base=h * (n // 2)) for h in range(2)]
    try:
        store.upload(np.arange(n, dtype=np.uint32), vectors)
        store.seal()
        singles = [store.search(q, k, mask) for q in queries]
        batch = store.search_batch(queries, k, mask)
        for (a, b), (c, d) in zip(singles, batch):
            assert a.tolist() == c.tolist() and b.tolist() == d.tolist()
        # the device-buffer batch entry point: groups of four queries per pass, keys as vs_search_device's
        dev = torch.device("cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        d_all = torch.from_numpy(queries).to(dev)
        d_full_mask = torch.from_numpy(mask.view(np.int64).copy()).to(dev)
        bws = torch.empty(store.batch_workspace_bytes(k), dtype=torch.uint8, device=dev)
        bkeys = torch.empty(len(queries) * k, dtype=torch.int64, device=dev)
        store.search_batch_device(d_all.data_ptr(), len(queries), dim, k, d_full_mask.data_ptr(),
                                  bws.data_ptr(), bkeys.data_ptr(), stream)
        sws = torch.empty(store.workspace_bytes(k), dtype=torch.uint8, device=dev)
        skeys = torch.empty(k, dtype=torch.int64, device=dev)
        for j in range(len(queries)):
            store.search_device(d_all[j].data_ptr(), dim, k, d_full_mask.data_ptr(), sws.data_ptr(),
                                skeys.data_ptr(), stream)
            torch.cuda.synchronize()
            assert bkeys[j * k:(j + 1) * k].tolist() == skeys.tolist()
            assert (skeys.cpu().numpy().view(np.uint64) & 0xFFFFFFFF).tolist() == singles[j][0].tolist()
        with pytest.raises(abi.VsError):
            store.batch_workspace_bytes(129)  # beyond the register lists of the shared-pass scan
        # two shards, device buffers, one merge
        keys = torch.empty(2 * k, dtype=torch.int64, device=dev)
        dq = torch.from_numpy(queries[0]).to(dev)
        keep = []
        for h, st in enumerate(halves):
            lo = h * (n // 2)
            st.upload(np.arange(lo, lo + n // 2, dtype=np.uint32), vectors[lo:lo + n // 2])
            st.seal()
            dm = torch.from_numpy(mask[lo // 64:(lo + n // 2) // 64].view(np.int64).copy()).to(dev)
            ws = torch.empty(st.workspace_bytes(k), dtype=torch.uint8, device=dev)
            keep += [dm, ws]
            st.search_device(dq.data_ptr(), dim, k, dm.data_ptr(), ws.data_ptr(),
                             keys[h * k:].data_ptr(), stream)
        ids = torch.empty(k, dtype=torch.int32, device=dev)
        dist = torch.empty(k, dtype=torch.float32, device=dev)
        cnt = torch.empty(1, dtype=torch.int32, device=dev)
        gpu_ctx.merge_topk_device(keys.data_ptr(), 2, k, metric, ids.data_ptr(), dist.data_ptr(),
                                  cnt.data_ptr(), stream)
        torch.cuda.synchronize()
        assert int(cnt.item()) == k
        assert ids.cpu().numpy().view(np.uint32).tolist() == singles[0][0].tolist()
        assert dist.cpu().numpy().tolist() == singles[0][1].tolist()
    finally:
        store.destroy()
        for st in halves:
            st.destroy()


def test_metric_and_element_type_must_match(gpu_ctx):
    from search_engine_amd import abi
    for dtype, metric in [(abi.VS_BIN1, abi.VS_EUCLIDEAN), (abi.VS_BIN1, abi.VS_DOT),
                          (abi.VS_F32, abi.VS_HAMMING), (abi.VS_F32, 7)]:
        with pytest.raises(abi.VsError) as e:
            gpu_ctx.create_store(0x3FF0, 128, 64, dtype=dtype, metric=metric)
        assert e.value.code == abi.VS_E_INVALID_ARG
