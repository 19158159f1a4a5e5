"""GPU parity tests of the f32 angular scan, through the C ABI (ctypes), against
the CPU oracle on the same seeded inputs. Tolerances: tests/helpers.py."""
import ctypes

import numpy as np
import pytest

from conftest import facet_mask, synthetic_vectors
from helpers import check_hits
from oracle import scan as oracle

pytestmark = pytest.mark.gpu

_next_index = [0]


def make_store(ctx, vectors, docids=None, dtype=None, capacity=None, docid_base=0):
    from search_engine_amd import abi
    n, dim = vectors.shape
    if docids is None:
        docids = np.arange(n, dtype=np.uint32) + docid_base
    _next_index[0] += 1
    cap = capacity or int(docids.max()) - docid_base + 1
    store = ctx.create_store(_next_index[0], dim, cap, dtype=abi.VS_F32 if dtype is None else dtype,
                             docid_base=docid_base)
    store.upload(docids, vectors)
    store.seal()
    return store


def run_case(ctx, vectors, query, k, mask=None, docids=None, docid_base=0, store=None):
    n = vectors.shape[0]
    if docids is None:
        docids = np.arange(n, dtype=np.uint32) + docid_base
    own = store is None
    if own:
        store = make_store(ctx, vectors, docids, docid_base=docid_base)
    try:
        ids, dist = store.search(query, k, mask)
    finally:
        if own:
            store.destroy()
    o_ids, _ = oracle.nns_by_vector(vectors, query, k, docids=docids, mask=mask)
    d64 = dict(zip(docids.tolist(), oracle.all_distances_f64(vectors, query).tolist()))
    ok = oracle.mask_to_bool(mask, int(docids.max()) + 1)[docids] if mask is not None else np.ones(n, bool)
    return check_hits(ids, dist, o_ids, d64, k, int(ok.sum())), ids, dist


def test_appendix_b_known_answers(gpu_ctx):
    """Hand-computed answers for the reference's commented-out vector scenario
    (src/update/index_documents/mod.rs:2599-2619): query [0,1,2]."""
    q = np.array([0, 1, 2], np.float32)
    store0 = np.array([[0, 1, 2], [6, 7, 8], [9, 10, 11]], np.float32)  # docs 0,1,2, vector #0
    _, ids, dist = run_case(gpu_ctx, store0, q, 10)
    assert ids.tolist() == [0, 1, 2]
    np.testing.assert_allclose(dist, [0.0, 0.078673, 0.088252], atol=1e-6)
    store1 = np.array([[3, 4, 5], [12, 13, 14]], np.float32)  # docs 0 and 2, vector #1
    _, ids, dist = run_case(gpu_ctx, store1, q, 10, docids=np.array([0, 2], np.uint32))
    assert ids.tolist() == [0, 2]
    np.testing.assert_allclose(dist, [0.057281, 0.093641], atol=1e-6)
    # filter {1, 2}
    _, ids, _ = run_case(gpu_ctx, store0, q, 10, mask=np.array([0b110], np.uint64))
    assert ids.tolist() == [1, 2]


def test_scan_small_golden_fixture(gpu_ctx):
    """tests/golden/scan_small.json: ids exactly as committed, distances against its float64 values."""
    import json
    import os
    from helpers import tol
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scan_small.json")) as f:
        g = json.load(f)
    vectors = np.array(g["vectors"], np.float32)
    q = np.array(g["query"], np.float32)
    d64 = np.array(g["distances_f64"])
    mask = oracle.bool_to_mask(np.isin(np.arange(vectors.shape[0]), g["candidates"]))
    store = make_store(gpu_ctx, vectors)
    try:
        ids, dist = store.search(q, 10, mask)
        assert ids.tolist() == g["top10_filtered"]
        assert np.all(np.abs(dist - d64[ids]) <= tol(d64[ids]))
        ids, dist = store.search(q, 10, None)
        assert ids.tolist() == g["top10_unfiltered"] and dist[0] == 0.0  # the zero-norm row
    finally:
        store.destroy()


@pytest.mark.parametrize("pass_percent", [None, 100, 50, 10, 1])
def test_parity_768_filtered(gpu_ctx, pass_percent):
    n, dim, k = 40000, 768, 10
    vectors = synthetic_vectors(n, dim, 0xC0FFEE)
    queries = synthetic_vectors(3, dim, 42)
    mask = None if pass_percent is None else facet_mask(n, pass_percent)
    store = make_store(gpu_ctx, vectors)
    try:
        swaps = 0
        for q in queries:
            s, _, _ = run_case(gpu_ctx, vectors, q, k, mask=mask, store=store)
            swaps += s
        assert swaps <= 1  # near ties among the top 10 of 40k random rows are rare
    finally:
        store.destroy()


@pytest.mark.parametrize("dim", [1, 2, 3, 5, 100, 255, 256, 257, 384, 1024, 1536, 2048, 3072, 4096])
def test_parity_dimensions(gpu_ctx, dim):
    n = 3000
    vectors = synthetic_vectors(n, dim, 7 + dim)
    q = synthetic_vectors(1, dim, 1000 + dim)[0]
    run_case(gpu_ctx, vectors, q, 10, mask=facet_mask(n, 50))


@pytest.mark.parametrize("dim", [1, 7, 8, 9, 384, 768, 1536, 4096])
def test_parity_bf16_rows(gpu_ctx, dim):
    """bf16 stores have no counterpart in the reference: rows are rounded to bf16 (nearest
    even) at upload, the query stays f32, norms come from the rounded rows. The oracle gets
    the same rounded rows."""
    from search_engine_amd import abi
    n = 4000
    vectors = synthetic_vectors(n, dim, 300 + dim)
    rounded = oracle.bf16_round(vectors)
    q = synthetic_vectors(1, dim, 400 + dim)[0]
    mask = facet_mask(n, 50)
    store = make_store(gpu_ctx, vectors, dtype=abi.VS_BF16)
    try:
        assert store.row_bytes == -(-dim // 8) * 16
        np.testing.assert_array_equal(store.item_vector(17), rounded[17])
        run_case(gpu_ctx, rounded, q, 10, mask=mask, store=store)
    finally:
        store.destroy()


@pytest.mark.parametrize("k", [1, 2, 10, 63, 64, 65, 100, 128, 129, 1000, 1024])
def test_parity_k(gpu_ctx, k):
    n, dim = 6000, 96
    vectors = synthetic_vectors(n, dim, 11)
    q = synthetic_vectors(1, dim, 12)[0]
    run_case(gpu_ctx, vectors, q, k, mask=facet_mask(n, 50))


@pytest.mark.parametrize("k", [129, 500, 1024])
@pytest.mark.parametrize("order", ["improving", "worsening", "random"])
def test_lists_in_lds_filled_in_batches(gpu_ctx, k, order):
    """Beyond k = 128 (deep pagination: limit = offset + limit, src/search/search.rs:115) a wave's list
    lives in LDS and takes its keys a batch at a time: up to 128 parked candidates are sorted and merged
    into the list in place. Adversarial arrival orders: every row closer to the query than all before
    it (every one passes the bound, merge after merge, each landing in front), every row farther (once
    the list is full nothing passes), and random; no filter, a dense one, a selective one, and one
    that leaves fewer candidates than k. Exact ties are planted as well (the docid decides)."""
    n, dim = 40_000, 64
    rng = np.random.default_rng(k + len(order))
    q = rng.standard_normal(dim).astype(np.float32)
    noise = rng.standard_normal((n, dim)).astype(np.float32)
    if order == "improving":
        scale = np.linspace(3.0, 0.01, n)
    elif order == "worsening":
        scale = np.linspace(0.01, 3.0, n)
    else:
        scale = rng.uniform(0.01, 3.0, n)
    vectors = (q[None, :] + noise * scale[:, None].astype(np.float32)).astype(np.float32)
    vectors[30_000:30_040] = vectors[100:140]  # exact ties
    store = make_store(gpu_ctx, vectors)
    try:
        few = np.zeros(n, bool)
        few[rng.choice(n, k - 7, replace=False)] = True
        for mask in (None, facet_mask(n, 60), facet_mask(n, 4), oracle.bool_to_mask(few)):
            run_case(gpu_ctx, vectors, q, k, mask=mask, store=store)
    finally:
        store.destroy()


def test_k_larger_than_candidates(gpu_ctx):
    n, dim = 500, 64
    vectors = synthetic_vectors(n, dim, 3)
    q = synthetic_vectors(1, dim, 4)[0]
    mask = facet_mask(n, 10)
    _, ids, _ = run_case(gpu_ctx, vectors, q, 200, mask=mask)
    assert ids.size == int(oracle.mask_to_bool(mask, n).sum()) < 200


def test_empty_candidates_and_k0(gpu_ctx):
    n, dim = 300, 32
    vectors = synthetic_vectors(n, dim, 5)
    q = synthetic_vectors(1, dim, 6)[0]
    store = make_store(gpu_ctx, vectors)
    try:
        ids, dist = store.search(q, 10, np.zeros(5, np.uint64))
        assert ids.size == 0 and dist.size == 0
        ids, _ = store.search(q, 0, None)  # arroy: count == 0 -> empty
        assert ids.size == 0
        # a mask shorter than the docid range: docids beyond it are not candidates
        ids, _ = store.search(q, 300, np.array([~np.uint64(0)], np.uint64))
        assert sorted(ids.tolist()) == list(range(64))
    finally:
        store.destroy()


def test_exact_ties_break_by_docid(gpu_ctx):
    """Bit-identical rows give bit-identical distances; order must be by docid."""
    dim = 768
    base = synthetic_vectors(4, dim, 21)
    vectors = np.concatenate([base[[0]]] * 40 + [base[[1]]] * 40 + [base[2:]], 0)
    rng = np.random.default_rng(0)
    perm = rng.permutation(vectors.shape[0])
    vectors = vectors[perm]
    q = base[0] + 0.1 * base[1]
    store = make_store(gpu_ctx, vectors)
    try:
        ids, dist = store.search(q, 82, None)
    finally:
        store.destroy()
    o_ids, o_dist = oracle.nns_by_vector(vectors, q, 82)
    assert ids.tolist() == o_ids.tolist()  # two runs of 40 exact ties, each ascending by docid
    assert len(set(dist[:40].tolist())) == 1 and len(set(dist[40:80].tolist())) == 1
    assert ids[:40].tolist() == sorted(ids[:40].tolist())
    assert ids[40:80].tolist() == sorted(ids[40:80].tolist())


def test_zero_norm_rows_and_query(gpu_ctx):
    """arroy: pn * qn == 0 -> distance 0."""
    n, dim = 200, 48
    vectors = synthetic_vectors(n, dim, 31)
    vectors[[5, 77, 150]] = 0.0
    q = synthetic_vectors(1, dim, 32)[0]
    _, ids, dist = run_case(gpu_ctx, vectors, q, 5)
    assert ids[:3].tolist() == [5, 77, 150] and np.all(dist[:3] == 0.0)
    _, ids, dist = run_case(gpu_ctx, vectors, np.zeros(dim, np.float32), 7)
    assert ids.tolist() == list(range(7)) and np.all(dist == 0.0)


def test_identical_vector_distance_near_zero(gpu_ctx):
    n, dim = 1000, 768
    vectors = synthetic_vectors(n, dim, 41)
    _, ids, dist = run_case(gpu_ctx, vectors, vectors[123] * 3.5, 3)
    assert ids[0] == 123 and abs(dist[0]) <= 1e-6


def test_sparse_docids_base_and_removal(gpu_ctx):
    from search_engine_amd import abi
    dim, base = 128, 64 * 1000
    rng = np.random.default_rng(9)
    docids = np.sort(rng.choice(50000, 8000, replace=False)).astype(np.uint32) + base  # holes
    vectors = synthetic_vectors(docids.size, dim, 51)
    q = synthetic_vectors(1, dim, 52)[0]
    mask = facet_mask(base + 50000, 50)
    store = make_store(gpu_ctx, vectors, docids, capacity=50000, docid_base=base)
    try:
        assert len(store) == 8000
        run_case(gpu_ctx, vectors, q, 20, mask=mask, docids=docids, docid_base=base, store=store)
        # delete a third of the items, among them the current best hits
        ids, _ = store.search(q, 20, mask)
        gone = np.concatenate([ids[:10], docids[::3]])
        store.remove(gone)
        # not visible before seal (a search sees the last sealed state)
        ids2, _ = store.search(q, 20, mask)
        assert ids2.tolist() == ids.tolist()
        store.seal()
        keep = ~np.isin(docids, gone)
        assert len(store) == int(keep.sum())
        run_case(gpu_ctx, vectors[keep], q, 20, mask=mask, docids=docids[keep], docid_base=base,
                 store=store)
        assert not store.contains(int(gone[0])) and store.contains(int(docids[keep][0]))
        np.testing.assert_array_equal(store.item_vector(int(docids[keep][5])), vectors[keep][5])
        with pytest.raises(abi.VsError) as e:
            store.upload(np.array([base - 1], np.uint32), vectors[:1])
        assert e.value.code == abi.VS_E_CAPACITY
    finally:
        store.destroy()


def test_overwrite_item(gpu_ctx):
    n, dim = 400, 64
    vectors = synthetic_vectors(n, dim, 61)
    q = synthetic_vectors(1, dim, 62)[0]
    store = make_store(gpu_ctx, vectors)
    try:
        vectors2 = vectors.copy()
        vectors2[17] = q * 2  # becomes the best hit
        store.upload(np.array([17], np.uint32), vectors2[[17]])
        store.seal()
        _, ids, _ = run_case(gpu_ctx, vectors2, q, 5, store=store)
        assert ids[0] == 17
    finally:
        store.destroy()


def test_error_codes(gpu_ctx):
    from search_engine_amd import abi
    vectors = synthetic_vectors(100, 16, 71)
    with pytest.raises(abi.VsError) as e:
        gpu_ctx.open_store(0xFFFF)  # MissingMetadata: the reader loop's terminator
    assert e.value.code == abi.VS_E_NO_STORE
    _next_index[0] += 1
    store = gpu_ctx.create_store(_next_index[0], 16, 100)
    try:
        with pytest.raises(abi.VsError) as e:
            gpu_ctx.open_store(_next_index[0])  # never sealed = never built
        assert e.value.code == abi.VS_E_NO_STORE
        store.upload(np.arange(100, dtype=np.uint32), vectors)
        with pytest.raises(abi.VsError) as e:
            store.search(vectors[0], 5)
        assert e.value.code == abi.VS_E_NOT_SEALED
        store.seal()
        assert gpu_ctx.open_store(_next_index[0]).handle.value == store.handle.value
        with pytest.raises(abi.VsError) as e:
            store.search(np.zeros(17, np.float32), 5)
        assert e.value.code == abi.VS_E_DIM_MISMATCH
        assert (e.value.expected, e.value.received) == (16, 17)
        with pytest.raises(abi.VsError) as e:
            store.search(vectors[0], abi.VS_MAX_K + 1)
        assert e.value.code == abi.VS_E_INVALID_ARG
        with pytest.raises(abi.VsError) as e:
            gpu_ctx.create_store(_next_index[0], 16, 100)
        assert e.value.code == abi.VS_E_INVALID_ARG
    finally:
        store.destroy()


def test_schedules_agree(gpu_ctx):
    """Every scan schedule returns the same list bit for bit."""
    n, dim = 30000, 384
    vectors = synthetic_vectors(n, dim, 81)
    q = synthetic_vectors(1, dim, 82)[0]
    mask = facet_mask(n, 10)
    store = make_store(gpu_ctx, vectors)
    try:
        results = []
        for blocks, nt in [(0, 1), (0, 0), (7, 1), (300, 0), (2048, 1)]:
            gpu_ctx.set_scan_tuning(blocks, nt)
            ids, dist = store.search(q, 50, mask)
            results.append((ids.tolist(), dist.tolist()))
        assert all(r == results[0] for r in results[1:])
    finally:
        gpu_ctx.set_scan_tuning(0, 1)
        store.destroy()


def test_search_batch_and_roaring(gpu_ctx):
    from search_engine_amd import abi
    n, dim = 70000 + 13, 64
    vectors = synthetic_vectors(n, dim, 91)
    queries = synthetic_vectors(5, dim, 92)
    mask = facet_mask(n, 10)
    mask[0:1024] = ~np.uint64(0)  # one full container -> bitmap container in roaring form
    store = make_store(gpu_ctx, vectors)
    try:
        single = [store.search(q, 10, mask) for q in queries]
        batch = store.search_batch(queries, 10, mask)
        blob = abi.mask_to_roaring(mask)
        for (i1, d1), (i2, d2), q in zip(single, batch, queries):
            assert i1.tolist() == i2.tolist() and d1.tolist() == d2.tolist()
            i3, d3 = store.search_roaring(q, 10, blob)
            assert i1.tolist() == i3.tolist() and d1.tolist() == d3.tolist()
        with pytest.raises(abi.VsError) as e:
            store.search_roaring(queries[0], 10, b"\x01\x02\x03\x04\x05\x06\x07\x08\x09")
        assert e.value.code == abi.VS_E_FORMAT
    finally:
        store.destroy()


@pytest.mark.parametrize("kind", ["f32", "bin1", "bf16 k=40"])
def test_search_batch_without_the_matrix_core_path(gpu_ctx, kind):
    """vs_search_batch where it is a queue of single-query scans (f32 and binary stores, bf16 with
    k beyond 32): the same lists as vs_search, for more queries than one chunk holds."""
    from search_engine_amd import abi
    n, dim = 9000, 192
    vectors = synthetic_vectors(n, dim, 201)
    queries = synthetic_vectors(150, dim, 202)
    mask = facet_mask(n, 50)
    if kind == "bin1":
        _next_index[0] += 1
        store = gpu_ctx.create_store(_next_index[0], dim, n, dtype=abi.VS_BIN1, metric=abi.VS_HAMMING)
        store.upload(np.arange(n, dtype=np.uint32), vectors)
        store.seal()
    else:
        store = make_store(gpu_ctx, vectors, dtype=abi.VS_F32 if kind == "f32" else abi.VS_BF16)
    k = 40 if kind == "bf16 k=40" else 10
    try:
        batch = store.search_batch(queries, k, mask)
        assert len(batch) == len(queries)
        for q, (ids, dist) in zip(queries, batch):
            s_ids, s_dist = store.search(q, k, mask)
            if kind == "f32":
                # 150 f32 queries go 32 at a time through the f32 matrix cores: another summation
                # order, so equal up to rounding (parity with the oracle: the test below)
                assert len(set(ids.tolist()) ^ set(s_ids.tolist())) <= 2
                both = [i for i in range(k) if ids[i] == s_ids[i]]
                np.testing.assert_allclose(dist[both], s_dist[both], rtol=1e-5, atol=1e-6)
            else:
                assert ids.tolist() == s_ids.tolist() and dist.tolist() == s_dist.tolist()
        assert store.search_batch(queries[:3], 0, mask)[0][0].size == 0
    finally:
        store.destroy()


def test_masks_kept_on_the_device(gpu_ctx):
    """vs_mask_*: a filter uploaded once (dense words or serialised roaring) gives the results of
    vs_search with the same filter, for every query, before and after the store changes."""
    from search_engine_amd import abi
    dim, base = 96, 64 * 50
    rng = np.random.default_rng(3)
    docids = np.sort(rng.choice(40000, 9000, replace=False)).astype(np.uint32) + base
    vectors = synthetic_vectors(docids.size, dim, 93)
    queries = synthetic_vectors(6, dim, 94)
    mask = facet_mask(base + 40000 + 500, 10)  # longer than the store: the tail is ignored
    store = make_store(gpu_ctx, vectors, docids, capacity=40000, docid_base=base)
    other = make_store(gpu_ctx, vectors[:100])
    handles = []
    try:
        handles = [store.create_mask(mask), store.create_mask(roaring_bytes=abi.mask_to_roaring(mask)),
                   store.create_mask(mask[: base // 64 + 100]),  # shorter than the store: the rest is no candidate
                   store.create_mask(np.zeros(0, np.uint64))]
        short = mask.copy()
        short[base // 64 + 100:] = 0
        for q in queries:
            want = store.search(q, 15, mask)
            for h in handles[:2]:
                got = store.search_mask(q, 15, h)
                assert got[0].tolist() == want[0].tolist() and got[1].tolist() == want[1].tolist()
            got, want_short = store.search_mask(q, 15, handles[2]), store.search(q, 15, short)
            assert got[0].tolist() == want_short[0].tolist()
            assert store.search_mask(q, 15, handles[3])[0].size == 0
            none = store.search_mask(q, 15, None)  # NULL = every item
            assert none[0].tolist() == store.search(q, 15)[0].tolist()
        # the mask is a set of docids: after a removal + seal it still applies
        gone = store.search(queries[0], 5, mask)[0]
        store.remove(gone)
        store.seal()
        got, want = store.search_mask(queries[0], 15, handles[0]), store.search(queries[0], 15, mask)
        assert got[0].tolist() == want[0].tolist() and not set(got[0].tolist()) & set(gone.tolist())
        with pytest.raises(abi.VsError) as e:
            other.search_mask(synthetic_vectors(1, dim, 1)[0], 5, handles[0])
        assert e.value.code == abi.VS_E_INVALID_ARG
        with pytest.raises(abi.VsError) as e:
            store.create_mask(roaring_bytes=b"\x01\x02\x03")
        assert e.value.code == abi.VS_E_FORMAT
    finally:
        for h in handles:
            store.destroy_mask(h)
        store.destroy()
        other.destroy()


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
def test_store_grows(gpu_ctx, dtype_name):
    """vs_store_reserve: contents and the sealed state survive, new docids fit afterwards, masks
    made for the smaller store are refused."""
    from search_engine_amd import abi
    dtype = abi.VS_F32 if dtype_name == "f32" else abi.VS_BF16
    dim, base = 64, 128
    vectors = synthetic_vectors(5000, dim, 401)
    if dtype_name == "bf16":
        vectors = oracle.bf16_round(vectors)
    q = synthetic_vectors(1, dim, 402)[0]
    docids = np.arange(5000, dtype=np.uint32) + base
    store = make_store(gpu_ctx, vectors[:1000], docids[:1000], dtype=dtype, capacity=1000, docid_base=base)
    old_mask = store.create_mask(facet_mask(base + 1000, 50))
    try:
        before = store.search(q, 10)
        with pytest.raises(abi.VsError) as e:
            store.upload(docids[3000:3001], vectors[3000:3001])
        assert e.value.code == abi.VS_E_CAPACITY
        store.reserve(500)   # not larger: nothing happens
        store.reserve(5000)
        after = store.search(q, 10)
        assert after[0].tolist() == before[0].tolist() and after[1].tolist() == before[1].tolist()
        with pytest.raises(abi.VsError) as e:
            store.search_mask(q, 10, old_mask)
        assert e.value.code == abi.VS_E_INVALID_ARG
        store.upload(docids[3000:5000], vectors[3000:5000])
        store.seal()
        keep = np.r_[0:1000, 3000:5000]
        mask = facet_mask(base + 5000, 50)
        run_case(gpu_ctx, vectors[keep], q, 20, mask=mask, docids=docids[keep], docid_base=base, store=store)
        if dtype_name == "bf16":  # the batched scan reads whole 256-row tiles of the new slab
            got = store.search_batch(np.stack([q, -q]), 10, mask)
            for qq, (ids, dist) in zip((q, -q), got):
                s_ids, _ = store.search(oracle.bf16_round(qq[None])[0], 10, mask)
                assert ids.tolist() == s_ids.tolist()
        with pytest.raises(abi.VsError) as e:
            store.upload(np.array([base + 5000], np.uint32), vectors[:1])
        assert e.value.code == abi.VS_E_CAPACITY
    finally:
        store.destroy_mask(old_mask)
        store.destroy()


def test_concurrent_searches_from_many_threads(gpu_ctx):
    """The reference shares one Index between request threads (search is synchronous per thread);
    vs_search* must be re-entrant on a sealed store: every thread gets the single-threaded answer."""
    import threading
    n, dim, k = 40000, 128, 10
    vectors = synthetic_vectors(n, dim, 301)
    queries = synthetic_vectors(48, dim, 302)
    mask = facet_mask(n, 10)
    store = make_store(gpu_ctx, vectors)
    kept = store.create_mask(mask)
    try:
        want = [store.search(q, k, mask) for q in queries]
        errors = []

        def work(t):
            try:
                for rep in range(3):
                    for j in range(t, len(queries), 8):
                        for got in (store.search(queries[j], k, mask), store.search_mask(queries[j], k, kept)):
                            if got[0].tolist() != want[j][0].tolist() or got[1].tolist() != want[j][1].tolist():
                                errors.append((t, j))
            except Exception as e:  # noqa: BLE001
                errors.append((t, repr(e)))

        threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, errors[:5]
    finally:
        store.destroy_mask(kept)
        store.destroy()


def test_device_api_and_shard_merge(gpu_ctx):
    """Two docid-range shards searched through the device entry point, keys
    concatenated as an all-gather would, merged: identical to one store."""
    import torch
    from search_engine_amd import abi
    n, dim, k = 64 * 500, 128, 10
    vectors = synthetic_vectors(n, dim, 101)
    q = synthetic_vectors(1, dim, 102)[0]
    mask = facet_mask(n, 50)
    half = n // 2
    whole = make_store(gpu_ctx, vectors)
    a = make_store(gpu_ctx, vectors[:half])
    b = make_store(gpu_ctx, vectors[half:], docid_base=half)
    try:
        ref_ids, ref_dist = whole.search(q, k, mask)
        dev = torch.device("cuda:0")
        d_q = torch.from_numpy(q).to(dev)
        d_mask = torch.from_numpy(mask.view(np.int64)).to(dev)
        keys = torch.empty(2 * k, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        for j, st in enumerate((a, b)):
            ws = torch.empty(st.workspace_bytes(k), dtype=torch.uint8, device=dev)
            words0 = (j * half) // 64
            st.search_device(d_q.data_ptr(), dim, k, d_mask.data_ptr() + 8 * words0, ws.data_ptr(),
                             keys.data_ptr() + 8 * k * j, stream)
        ids = torch.empty(k, dtype=torch.int32, device=dev)
        dist = torch.empty(k, dtype=torch.float32, device=dev)
        cnt = torch.empty(1, dtype=torch.int32, device=dev)
        gpu_ctx.merge_topk_device(keys.data_ptr(), 2, k, abi.VS_ANGULAR, ids.data_ptr(),
                                  dist.data_ptr(), cnt.data_ptr(), stream)
        torch.cuda.synchronize()
        assert cnt.item() == k
        assert ids.cpu().numpy().astype(np.uint32).tolist() == ref_ids.tolist()
        assert dist.cpu().numpy().tolist() == ref_dist.tolist()
    finally:
        for st in (whole, a, b):
            st.destroy()


@pytest.mark.parametrize("metric_name", ["angular", "euclidean", "dot"])
@pytest.mark.parametrize("dtype_name,dim,k", [
    ("f32", 3, 5), ("f32", 100, 10), ("f32", 768, 10), ("f32", 768, 128), ("f32", 1000, 17),
    ("f32", 1536, 10), ("f32", 2048, 10),     # 2048-d f32: beyond the multi-query scan, one scan per query
    ("f32", 768, 129),                        # k beyond its register lists: likewise
    ("bf16", 768, 64), ("bf16", 1536, 40), ("bf16", 2000, 33),
])
def test_queries_sharing_a_pass_equal_single_searches(gpu_ctx, metric_name, dtype_name, dim, k):
    """vs_search_batch outside the matrix-core path shares each pass over the rows between four
    or eight queries; every query's list must be bit-identical to its own vs_search (same ids,
    same float bits), for batch sizes that do and do not fill the last group, with and without
    a filter, holes in the store included."""
    from search_engine_amd import abi
    metric = {"angular": abi.VS_ANGULAR, "euclidean": abi.VS_EUCLIDEAN, "dot": abi.VS_DOT}[metric_name]
    dtype = abi.VS_F32 if dtype_name == "f32" else abi.VS_BF16
    n = 3000
    rng = np.random.default_rng(dim * 7 + k)
    docids = np.sort(rng.choice(n + 700, n, replace=False)).astype(np.uint32)
    vectors = synthetic_vectors(n, dim, 300 + dim)
    vectors[5] = 0.0          # a zero row: angular distance 0 by definition
    vectors[11] = vectors[10]  # an exact tie
    queries = synthetic_vectors(19, dim, 301 + dim)
    queries[2] = vectors[10]
    _next_index[0] += 1
    store = gpu_ctx.create_store(_next_index[0], dim, n + 700, dtype=dtype, metric=metric)
    try:
        store.upload(docids, vectors)
        store.seal()
        for mask in (None, facet_mask(n + 700, 30)):
            singles = [store.search(q, k, mask) for q in queries]
            for n_q in (1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 16):  # from 17 on f32 batches may take the matrix cores
                batch = store.search_batch(queries[:n_q], k, mask)
                assert len(batch) == n_q
                for (ids, dist), (s_ids, s_dist) in zip(batch, singles):
                    assert ids.tolist() == s_ids.tolist()
                    assert dist.view(np.uint32).tolist() == s_dist.view(np.uint32).tolist()
    finally:
        store.destroy()


@pytest.mark.parametrize("metric_name", ["angular", "dot"])
@pytest.mark.parametrize("dim,k,n_q", [
    (3, 1, 17), (36, 5, 40), (100, 10, 33), (760, 10, 50), (768, 32, 32), (768, 10, 70), (384, 7, 64),
])
def test_batches_on_the_f32_matrix_cores(gpu_ctx, metric_name, dim, k, n_q):
    """f32 stores, 17 or more queries: 32 queries per pass on v_mfma_f32_32x32x2_f32. Parity with
    the CPU oracle under the usual rules (ids position by position up to near ties in float64,
    distances within 1e-5 relative + the absolute floor; exact ties by docid), with and without
    filters, with holes, a zero row, duplicated rows and a query equal to a row."""
    from search_engine_amd import abi
    from test_metrics_gpu import check as check_metric
    metric = abi.VS_ANGULAR if metric_name == "angular" else abi.VS_DOT
    n = 5000
    rng = np.random.default_rng(dim * 13 + k)
    docids = np.sort(rng.choice(n + 900, n, replace=False)).astype(np.uint32)
    vectors = synthetic_vectors(n, dim, 400 + dim)
    vectors[5] = 0.0
    vectors[11] = vectors[10]
    vectors[12] = vectors[10]
    queries = synthetic_vectors(n_q, dim, 401 + dim)
    queries[2] = vectors[10]
    queries[n_q - 1] = vectors[77] * 2.0
    _next_index[0] += 1
    store = gpu_ctx.create_store(_next_index[0], dim, n + 900, metric=metric)
    try:
        store.upload(docids, vectors)
        store.seal()
        for pass_percent in (100, 30, 1):
            mask = facet_mask(n + 900, pass_percent) if pass_percent < 100 else None
            ok = oracle.mask_to_bool(mask, n + 900)[docids] if mask is not None else np.ones(n, bool)
            batch = store.search_batch(queries, k, mask)
            assert len(batch) == n_q
            for q, (ids, dist) in zip(queries, batch):
                if metric == abi.VS_ANGULAR:
                    o_ids, _ = oracle.nns_by_vector(vectors, q, k, docids=docids, mask=mask)
                    d64 = dict(zip(docids.tolist(), oracle.all_distances_f64(vectors, q).tolist()))
                    # (check_hits allows a swap only between candidates closer than the tolerance in
                    # float64; the planted rows and the zero row are four such candidates at distance ~0)
                    assert check_hits(ids, dist, o_ids, d64, k, int(ok.sum())) <= 4
                else:
                    assert check_metric(metric, vectors, q, k, ids, dist, mask, docids) <= 4
            # the duplicated rows tie exactly for the query that equals them: docid order
            ids, dist = batch[2]
            if pass_percent == 100 and k >= 4 and metric == abi.VS_ANGULAR:
                # (the zero row is at distance 0 by definition and may come first)
                at = [ids.tolist().index(int(docids[r])) for r in (10, 11, 12)]
                assert at[1] == at[0] + 1 and at[2] == at[0] + 2 and dist[at[0]] == dist[at[1]] == dist[at[2]]
    finally:
        store.destroy()


def test_f32_matrix_core_batches_with_a_sample_pass(gpu_ctx):
    """Stores of a million slots or more: a first pass over 1/64 of the mask words gives every
    query a bound, the main pass covers the rest. Same parity rules; the sample's own
    neighbours (a planted row inside a sample word) and rows outside it both have to show up."""
    from search_engine_amd import abi
    n, dim, k, n_q = 1_100_000, 24, 10, 40
    rng = np.random.default_rng(77)
    vectors = rng.standard_normal((n, dim), dtype=np.float32)
    queries = rng.standard_normal((n_q, dim), dtype=np.float32)
    vectors[0] = queries[0] * 3.0          # word 0 of wave 0 is a sample word
    vectors[999_999] = queries[1] * 0.5
    docids = np.arange(n, dtype=np.uint32)
    _next_index[0] += 1
    store = gpu_ctx.create_store(_next_index[0], dim, n)
    try:
        for lo in range(0, n, 1 << 18):
            store.upload(docids[lo:lo + (1 << 18)], vectors[lo:lo + (1 << 18)])
        store.seal()
        norms = oracle.norms(vectors)
        for pass_percent in (100, 5):
            mask = facet_mask(n, pass_percent) if pass_percent < 100 else None
            n_cand = n if mask is None else int(oracle.mask_to_bool(mask, n).sum())
            batch = store.search_batch(queries, k, mask)
            o_ids, _, o_n, _ = oracle.nns_batch_mt(vectors, queries, k, mask=mask, norms=norms)
            for j, (ids, dist) in enumerate(batch):
                cand = sorted(set(ids.tolist()) | set(o_ids[j].tolist()))
                d64 = dict(zip(cand, oracle.all_distances_f64(vectors[cand], queries[j]).tolist()))
                assert check_hits(ids, dist, o_ids[j], d64, k, n_cand) <= 2
            if pass_percent == 100:
                assert batch[0][0][0] == 0 and batch[1][0][0] == 999_999
    finally:
        store.destroy()
