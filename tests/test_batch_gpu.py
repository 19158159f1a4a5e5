"""GPU parity tests of the batched bf16 scan on the matrix cores (vs_search_batch on a
VS_BF16 store). No counterpart in the reference; the definition under test: rows AND
queries rounded to bf16 (nearest even), arroy's Angular formula on the rounded values,
f32 accumulation, results per query ascending by (distance, docid). The oracle gets the
same rounded inputs; tolerances as in tests/helpers.py."""
import numpy as np
import pytest

from conftest import facet_mask, synthetic_vectors
from helpers import check_hits
from oracle import scan as oracle

pytestmark = pytest.mark.gpu

_next_index = [0x5000]


def make_bf16_store(ctx, vectors, docids=None, docid_base=0):
    from search_engine_amd import abi
    n, dim = vectors.shape
    if docids is None:
        docids = np.arange(n, dtype=np.uint32) + docid_base
    _next_index[0] += 1
    store = ctx.create_store(_next_index[0], dim, int(docids.max()) - docid_base + 1, dtype=abi.VS_BF16,
                             docid_base=docid_base)
    store.upload(docids, vectors)
    store.seal()
    return store


def check_batch(store, vectors, queries, k, mask=None, docids=None):
    n = vectors.shape[0]
    if docids is None:
        docids = np.arange(n, dtype=np.uint32)
    rv, rq = oracle.bf16_round(vectors), oracle.bf16_round(queries)
    results = store.search_batch(queries, k, mask)
    assert len(results) == queries.shape[0]
    ok = oracle.mask_to_bool(mask, int(docids.max()) + 1)[docids] if mask is not None else np.ones(n, bool)
    swaps = 0
    for (ids, dist), q in zip(results, rq):
        o_ids, _ = oracle.nns_by_vector(rv, q, k, docids=docids, mask=mask)
        d64 = dict(zip(docids.tolist(), oracle.all_distances_f64(rv, q).tolist()))
        swaps += check_hits(ids, dist, o_ids, d64, k, int(ok.sum()))
    return swaps


@pytest.mark.parametrize("dim", [8, 24, 64, 100, 384, 768, 1536])
def test_batch_parity_dimensions(gpu_ctx, dim):
    n = 3000  # not a multiple of the 128-row tile
    vectors = synthetic_vectors(n, dim, 700 + dim)
    queries = synthetic_vectors(9, dim, 800 + dim)
    store = make_bf16_store(gpu_ctx, vectors)
    try:
        check_batch(store, vectors, queries, 10, facet_mask(n, 50))
    finally:
        store.destroy()


@pytest.mark.parametrize("n_q,k", [(2, 1), (7, 10), (64, 32), (256, 10), (300, 5)])
def test_batch_parity_query_counts(gpu_ctx, n_q, k):
    n, dim = 20000, 768
    vectors = synthetic_vectors(n, dim, 0xBA7C)
    queries = synthetic_vectors(n_q, dim, 44)
    store = make_bf16_store(gpu_ctx, vectors)
    try:
        swaps = check_batch(store, vectors, queries, k, facet_mask(n, 10))
        assert swaps <= max(2, n_q // 20)
    finally:
        store.destroy()


@pytest.mark.parametrize("pass_percent", [None, 100, 1])
def test_batch_parity_filters_and_holes(gpu_ctx, pass_percent):
    dim, base = 128, 64 * 7
    rng = np.random.default_rng(12)
    docids = np.sort(rng.choice(20000, 6000, replace=False)).astype(np.uint32) + base
    vectors = synthetic_vectors(docids.size, dim, 91)
    vectors[5] = 0.0                                   # zero-norm row: distance 0 to everything
    vectors[100:140] = vectors[99]                     # 41 identical rows: exact ties by docid
    queries = synthetic_vectors(12, dim, 92)
    queries[3] = 0.0                                   # zero query: every distance 0, order by docid
    queries[4] = vectors[99] * 2
    mask = None if pass_percent is None else facet_mask(base + 20000, pass_percent)
    store = make_bf16_store(gpu_ctx, vectors, docids, docid_base=base)
    try:
        check_batch(store, vectors, queries, 32, mask, docids)
    finally:
        store.destroy()


@pytest.mark.parametrize("n_q", [19, 100, 150])
@pytest.mark.parametrize("dim", [100, 160])
def test_streaming_pass_with_two_and_three_step_tiles(gpu_ctx, dim, n_q):
    """The streaming second pass requests a step's rows two steps ahead, across tile boundaries: tiles
    of two steps (100 dimensions) and of three (160), next to the one-step tiles of the test below.
    Batches of up to 128 queries take the narrow build of that pass (64 query columns per wave; 100
    queries reach into its second column group), larger ones the full-width build; three-step tiles
    under the 40 % filter are fetched in place, so both builds of the gathered form run as well."""
    n = 256 * 256 * 2 + 64 * 5
    vectors = synthetic_vectors(n, dim, 0x51DE + dim)
    queries = synthetic_vectors(n_q, dim, 0x51DF + dim)
    rng = np.random.default_rng(dim)
    for j in range(queries.shape[0]):
        for pos in rng.choice(n, 4, replace=False):
            vectors[pos] = queries[j] + 0.05 * rng.standard_normal(dim).astype(np.float32)
    store = make_bf16_store(gpu_ctx, vectors)
    try:
        check_batch(store, vectors, queries, 10, facet_mask(n, 40))
        check_batch(store, vectors, queries, 10, None)
    finally:
        store.destroy()


@pytest.mark.parametrize("awkward", [True, False])
@pytest.mark.parametrize("pass_percent,k", [(None, 10), (3, 10), (50, 32), ("top third", 10)])
def test_batch_parity_beyond_the_sample_pass(gpu_ctx, pass_percent, k, awkward):
    """More than 3 x 256 tiles of 256 rows: 256 tiles spread over the store (every third) are
    scanned without a bound, the rest against the k-th best of that sample. The best rows are
    planted all over the store so that every path has to deliver; one filter selects the top
    third of the docids only. awkward = False: the streaming second pass delivers (survivors
    through the queue, the regions, the buckets, the exact judgement). awkward = True: a zero
    query and non-finite rows make that pass give up, and the gated rerun of the general kernel
    (register test, owner threads, chunked walk) delivers instead."""
    n, dim = 256 * 256 * 3 + 64 * 3, 64
    vectors = synthetic_vectors(n, dim, 0x7A11)
    queries = synthetic_vectors(37, dim, 0x7A12)
    rng = np.random.default_rng(5)
    for j in range(queries.shape[0]):  # near-duplicates of each query scattered over the store
        for pos in rng.choice(n, 6, replace=False):
            vectors[pos] = queries[j] * (1.0 + 0.01 * rng.standard_normal()) + 0.05 * rng.standard_normal(dim).astype(np.float32)
    vectors[n - 1] = queries[0]
    vectors[70000] = 0.0          # zero-norm row in the bounded pass
    if awkward:
        vectors[100000, 0] = np.inf   # non-finite rows: their tile takes the chunked walk; NaN distances sort last
        vectors[100300, 3] = np.nan
        queries[5] = 0.0              # zero query: the first k candidate docids, some of them in second-pass tiles
    if pass_percent is None:
        mask = None
    elif pass_percent == "top third":
        mask = oracle.bool_to_mask(np.arange(n) >= 2 * n // 3)
    else:
        mask = facet_mask(n, pass_percent)
    store = make_bf16_store(gpu_ctx, vectors)
    try:
        check_batch(store, vectors, queries, k, mask)
    finally:
        store.destroy()


def test_batch_under_a_selective_filter_gathers_the_candidates(gpu_ctx):
    """At most a quarter of the rows pass: vs_search_batch copies the candidates into a dense slab
    (slot order kept) and scans that. 20 % of 393 k rows are more than the 256 sample tiles, so the
    gathered slab goes through both passes; the planted near-duplicates outside the filter must not
    come back, the identical rows inside it tie by docid."""
    n, dim, base = 256 * 256 * 6 + 64, 64, 64 * 3
    vectors = synthetic_vectors(n, dim, 0x6A7)
    queries = synthetic_vectors(23, dim, 0x6A8)
    rng = np.random.default_rng(77)
    keep = rng.random(n) < 0.2
    for j in range(queries.shape[0]):
        for pos in rng.choice(n, 8, replace=False):
            vectors[pos] = queries[j] + 0.05 * rng.standard_normal(dim).astype(np.float32)
    vectors[5000:5030] = vectors[4999]
    keep[4999:5030] = True
    queries[2] = vectors[4999] * 3
    vectors[9000] = 0.0
    keep[9000] = True
    docids = np.arange(n, dtype=np.uint32) + base
    mask = oracle.bool_to_mask(np.concatenate([np.zeros(base, bool), keep]))
    store = make_bf16_store(gpu_ctx, vectors, docids, docid_base=base)
    try:
        assert keep.sum() * 4 <= n and keep.sum() > 256 * 256
        check_batch(store, vectors, queries, 10, mask, docids)
    finally:
        store.destroy()


@pytest.mark.parametrize("dim", [32, 192])
def test_batch_under_a_filter_of_a_few_rows(gpu_ctx, dim):
    """Three candidates (fewer than k, fewer than a tile), then candidates that are not items; short
    rows (the copied form) and long ones (fetched in place)."""
    n = 5000
    rng = np.random.default_rng(3)
    docids = np.sort(rng.choice(9000, n, replace=False)).astype(np.uint32)
    vectors = synthetic_vectors(n, dim, 31)
    queries = synthetic_vectors(5, dim, 32)
    store = make_bf16_store(gpu_ctx, vectors, docids)
    try:
        few = np.zeros(9000, bool)
        few[docids[[17, 2500, 4999]]] = True
        results = store.search_batch(queries, 10, oracle.bool_to_mask(few))
        assert all(sorted(ids.tolist()) == sorted(docids[[17, 2500, 4999]].tolist()) for ids, _ in results)
        check_batch(store, vectors, queries, 10, oracle.bool_to_mask(few), docids)
        absent = np.ones(9000, bool)
        absent[docids] = False
        assert all(ids.size == 0 for ids, _ in store.search_batch(queries, 10, oracle.bool_to_mask(absent)))
    finally:
        store.destroy()


def test_batch_matches_single_query_ids(gpu_ctx):
    """Same store, same (already bf16-exact) queries: the batched and the single-query path
    agree on ids wherever distances are not within tolerance of each other."""
    n, dim = 10000, 256
    vectors = synthetic_vectors(n, dim, 5)
    queries = oracle.bf16_round(synthetic_vectors(20, dim, 6))
    store = make_bf16_store(gpu_ctx, vectors)
    try:
        batch = store.search_batch(queries, 10, None)
        for (b_ids, b_dist), q in zip(batch, queries):
            s_ids, s_dist = store.search(q, 10, None)
            np.testing.assert_allclose(b_dist, s_dist, rtol=1e-5, atol=1e-6)
            for a, b, da, db in zip(b_ids, s_ids, b_dist, s_dist):
                assert a == b or abs(float(da) - float(db)) <= 2e-5 * abs(float(da)) + 2e-6
    finally:
        store.destroy()


def test_batch_device_api_sharded(gpu_ctx):
    """Two shards through vs_search_batch_device, keys gathered [shard][query][k], one merge launch."""
    import torch
    from search_engine_amd import abi
    n, dim, k, n_q = 64 * 300, 64, 10, 33
    vectors = synthetic_vectors(n, dim, 21)
    queries = synthetic_vectors(n_q, dim, 22)
    half = n // 2
    whole = make_bf16_store(gpu_ctx, vectors)
    a = make_bf16_store(gpu_ctx, vectors[:half])
    b = make_bf16_store(gpu_ctx, vectors[half:], docid_base=half)
    try:
        ref = whole.search_batch(queries, k, None)
        dev = torch.device("cuda:0")
        d_q = torch.from_numpy(queries).to(dev)
        keys = torch.empty(2 * n_q * k, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        keep = []
        for j, st in enumerate((a, b)):
            ws = torch.empty(st.batch_workspace_bytes(k), dtype=torch.uint8, device=dev)
            keep.append(ws)
            st.search_batch_device(d_q.data_ptr(), n_q, dim, k, None, ws.data_ptr(),
                                   keys.data_ptr() + 8 * n_q * k * j, stream)
        ids = torch.empty(n_q * k, dtype=torch.int32, device=dev)
        dist = torch.empty(n_q * k, dtype=torch.float32, device=dev)
        cnt = torch.empty(n_q, dtype=torch.int32, device=dev)
        gpu_ctx.merge_topk_batch_device(keys.data_ptr(), 2, n_q * k, n_q, k, k, abi.VS_ANGULAR,
                                        ids.data_ptr(), dist.data_ptr(), cnt.data_ptr(), stream)
        torch.cuda.synchronize()
        ids = ids.cpu().numpy().astype(np.uint32).reshape(n_q, k)
        dist = dist.cpu().numpy().reshape(n_q, k)
        assert (cnt.cpu().numpy() == k).all()
        for j, (r_ids, r_dist) in enumerate(ref):
            assert ids[j].tolist() == r_ids.tolist() and dist[j].tolist() == r_dist.tolist()
    finally:
        for st in (whole, a, b):
            st.destroy()


@pytest.mark.parametrize("dim,pass_percent,k,awkward", [(192, 25, 10, False), (768, 3, 10, False), (200, 60, 10, False),
                                                        (192, 25, 32, False), (192, 25, 10, True)])
def test_batch_fetches_the_candidates_of_a_filter_in_place(gpu_ctx, dim, pass_percent, k, awkward):
    """Rows of more than 128 elements: under a filter the streaming pass takes a tile's 256 rows from
    wherever the candidates lie (three steps per tile or more, the offsets swapped two steps before a
    tile ends). 25 % of 400 k rows fill more tiles than the sample, 3 % of 100 k fewer; dim 200 has a
    K tail; k = 32 takes the staged build of the sample pass. Planted near-duplicates in and outside
    the filter, identical rows, a zero row. awkward: a zero query and non-finite candidate rows make
    the streaming pass give up, and the gated unbounded scan of the store itself delivers."""
    n = 256 * 256 * 6 + 64 * 5 if pass_percent != 3 else 100_000
    base = 64 * 2
    vectors = synthetic_vectors(n, dim, 0x1A + dim)
    queries = synthetic_vectors(21, dim, 0x1B + dim)
    rng = np.random.default_rng(dim)
    keep = rng.random(n) < pass_percent / 100
    for j in range(queries.shape[0]):
        for pos in rng.choice(n, 8, replace=False):
            vectors[pos] = queries[j] + 0.05 * rng.standard_normal(dim).astype(np.float32)
    vectors[7000:7020] = vectors[6999]
    keep[6999:7020] = True
    queries[1] = vectors[6999] * 0.5
    vectors[n - 3] = 0.0
    keep[n - 3] = True
    if awkward:
        vectors[150000, 0] = np.inf
        vectors[150300, 3] = np.nan
        keep[[150000, 150300]] = True
        queries[5] = 0.0
    docids = np.arange(n, dtype=np.uint32) + base
    mask = oracle.bool_to_mask(np.concatenate([np.zeros(base, bool), keep]))
    store = make_bf16_store(gpu_ctx, vectors, docids, docid_base=base)
    try:
        check_batch(store, vectors, queries, k, mask, docids)
    finally:
        store.destroy()
