"""End-to-end plumbing on the GPU: the C++ host mirror (VectorSort -> bucket_sort ->
hybrid merge) driving libvs_amd.so through GpuContext, against the pure-Python
restatement of the reference's host logic fed by the CPU oracle. This is the
10k x 384 "plumbing" configuration: user-provided vectors, cosine top-10 with a
facet filter, several vectors per document spread over stores (embedder << 8) | k."""
import numpy as np
import pytest

from conftest import facet_mask, synthetic_vectors
from oracle import host as ref
from oracle import scan as oracle

pytestmark = pytest.mark.gpu

CATEGORY = 2  # embedder_category_id of "default" in this test


def build_world(ctx, n_docs, dim, n_stores, seed, index_base):
    """Docs own 1..n_stores vectors; the j-th vector of a doc lives in store j (packed stores,
    src/update/index_documents/typed_chunk.rs:597-607). Returns GPU stores and CPU readers."""
    rng = np.random.default_rng(seed)
    counts = rng.integers(1, n_stores + 1, n_docs)
    stores, readers = [], []
    for k in range(n_stores):
        docids = np.nonzero(counts > k)[0].astype(np.uint32)
        vectors = synthetic_vectors(docids.size, dim, seed * 10 + k)
        s = ctx.create_store(index_base | k, dim, n_docs)
        s.upload(docids, vectors)
        s.seal()
        stores.append(s)

        def nns(target, count, cand, docids=docids, vectors=vectors):
            mask = oracle.bool_to_mask(np.isin(np.arange(n_docs), list(cand)))
            ids, dist = oracle.nns_by_vector(vectors, np.asarray(target, np.float32), count,
                                             docids=docids, mask=mask)
            return list(zip(ids.tolist(), dist.tolist()))
        readers.append(nns)
    return stores, readers


def test_vector_search_plumbing_10k_384(gpu_ctx):
    from search_engine_amd import _host
    n_docs, dim, n_stores = 10_000, 384, 3
    stores, readers = build_world(gpu_ctx, n_docs, dim, n_stores, 5, CATEGORY << 8)
    try:
        all_docids = list(range(n_docs))
        ctx = _host.GpuContext(gpu_ctx.handle.value, {"default": CATEGORY}, oracle.bool_to_mask(np.ones(n_docs, bool)))
        universe_mask = facet_mask(n_docs, 10)
        universe = np.nonzero(oracle.mask_to_bool(universe_mask, n_docs))[0].tolist()
        for qi in range(3):
            q = synthetic_vectors(1, dim, 900 + qi)[0].tolist()
            for skip, limit, dist in ((0, 10, None), (5, 10, (0.5, 0.2)), (0, 20, None)):
                got = _host.execute_vector_search(ctx, q, universe_mask, skip, limit, "default", dist)
                want = ref.execute_vector_search(readers, all_docids, q, universe, skip, limit, dist)
                assert got["documents_ids"] == want["documents_ids"]
                assert len(got["documents_ids"]) == limit
                assert set(got["documents_ids"]) <= set(universe)
                for gs, ws in zip(got["document_scores"], want["document_scores"]):
                    # similarity = shift(1 - d): |error| <= |factor| * (1e-5 d + 1e-6); factor <= 2 here
                    assert gs[0]["similarity"] == pytest.approx(ws[0]["similarity"], abs=1.2e-5)
        # unknown embedder -> InvalidEmbedder; wrong dimension -> InvalidVectorDimensions{expected, found}
        with pytest.raises(_host.HostError) as e:
            _host.execute_vector_search(ctx, q, universe_mask, 0, 10, "nope", None)
        assert e.value.args[1] == "InvalidEmbedder"
        with pytest.raises(_host.HostError) as e:
            _host.execute_vector_search(ctx, q[:100], universe_mask, 0, 10, "default", None)
        assert e.value.args[1:] == ("InvalidVectorDimensions", dim, 100)
    finally:
        for s in stores:
            s.destroy()


def test_vector_search_batch_plumbing(gpu_ctx):
    """execute_vector_search_batch through GpuContext: every store is asked once for all vectors
    (vs_search_batch). f32 stores: identical to the single searches. bf16 stores (batched scan on
    the matrix cores, queries rounded to bf16): the oracle's result for the rounded inputs."""
    from search_engine_amd import _host, abi
    n_docs, dim, n_stores = 6000, 128, 2
    stores, readers = build_world(gpu_ctx, n_docs, dim, n_stores, 8, (CATEGORY + 2) << 8)
    bf = []
    try:
        ctx = _host.GpuContext(gpu_ctx.handle.value, {"default": CATEGORY + 2, "half": CATEGORY + 3},
                               oracle.bool_to_mask(np.ones(n_docs, bool)))
        universe_mask = facet_mask(n_docs, 50)
        queries = synthetic_vectors(9, dim, 333)
        for skip, limit, dist in ((0, 10, None), (3, 7, (0.5, 0.2))):
            got = _host.execute_vector_search_batch(ctx, queries.tolist(), universe_mask, skip, limit, "default", dist)
            for q, g in zip(queries, got):
                assert g == _host.execute_vector_search(ctx, q.tolist(), universe_mask, skip, limit, "default", dist)
        with pytest.raises(_host.HostError) as e:
            _host.execute_vector_search_batch(ctx, queries[:, :100].tolist(), universe_mask, 0, 10, "default", None)
        assert e.value.args[1:] == ("InvalidVectorDimensions", dim, 100)
        # one bf16 store under a second embedder
        vectors = synthetic_vectors(n_docs, dim, 444)
        s = gpu_ctx.create_store((CATEGORY + 3) << 8, dim, n_docs, dtype=abi.VS_BF16)
        s.upload(np.arange(n_docs, dtype=np.uint32), vectors)
        s.seal()
        bf.append(s)
        got = _host.execute_vector_search_batch(ctx, queries.tolist(), universe_mask, 0, 10, "half", None)
        rv, rq = oracle.bf16_round(vectors), oracle.bf16_round(queries)
        for q, g in zip(rq, got):
            want, _ = oracle.nns_by_vector(rv, q, 10, mask=universe_mask)
            d64 = oracle.all_distances_f64(rv, q)
            assert len(g["documents_ids"]) == 10
            for a, b in zip(g["documents_ids"], want.tolist()):  # ids equal unless the two are closer than the tolerance
                assert a == b or abs(d64[a] - d64[b]) <= 2e-5 * abs(d64[b]) + 2e-6
    finally:
        for s in stores + bf:
            s.destroy()


def test_hybrid_plumbing(gpu_ctx):
    """semanticRatio 0.5 with a stubbed keyword list: the vector half comes from the GPU."""
    from search_engine_amd import _host
    n_docs, dim = 4000, 64
    stores, readers = build_world(gpu_ctx, n_docs, dim, 1, 8, (CATEGORY + 1) << 8)
    try:
        all_docids = list(range(n_docs))
        ctx = _host.GpuContext(gpu_ctx.handle.value, {"default": CATEGORY + 1},
                               oracle.bool_to_mask(np.ones(n_docs, bool)))
        for pass_percent in (1, 10, 50):
            universe_mask = facet_mask(n_docs, pass_percent)
            universe = np.nonzero(oracle.mask_to_bool(universe_mask, n_docs))[0].tolist()
            q = synthetic_vectors(1, dim, 77)[0].tolist()
            limit, offset = 10, 2
            # keyword side: 8 documents of the universe with middling scores
            kw_ids = universe[:8]
            keyword = {"documents_ids": kw_ids, "candidates": kw_ids,
                       "document_scores": [[{"kind": "words", "value": 2, "max": 3},
                                            {"kind": "typo", "value": i % 3, "max": 2}] for i in range(8)]}
            dist = (0.5, 0.15)

            def gpu_vector_search(v):
                return _host.execute_vector_search(ctx, list(v), universe_mask, 0, limit + offset, "default", dist)

            def cpu_vector_search(v):
                return ref.execute_vector_search(readers, all_docids, list(v), universe, 0, limit + offset, dist)

            got, got_n = _host.execute_hybrid("query", True, q, offset, limit, lambda: keyword,
                                              gpu_vector_search, None, 0.5)
            want, want_n = ref.execute_hybrid("query", True, q, offset, limit, lambda: keyword,
                                              cpu_vector_search, None, 0.5)
            assert got["documents_ids"] == want["documents_ids"] and got_n == want_n
            assert got["candidates"] == want["candidates"]
            assert len(got["documents_ids"]) <= limit
    finally:
        for s in stores:
            s.destroy()


def test_universe_kept_on_the_device_and_refills(gpu_ctx):
    """A universe object that comes back: from its second search on a reader names the device copy
    of its words (vs_search_mask) instead of sending them again; results are those of a search that
    sends them. And VectorSort driven past its cache: every refill searches what is left (the
    candidates without the documents handed out so far), until the universe is used up."""
    from search_engine_amd import _host
    n_docs, dim, n_stores = 6_000, 64, 2
    stores, readers = build_world(gpu_ctx, n_docs, dim, n_stores, 11, (CATEGORY + 5) << 8)
    try:
        ctx = _host.GpuContext(gpu_ctx.handle.value, {"default": CATEGORY + 5},
                               oracle.bool_to_mask(np.ones(n_docs, bool)))
        universe_mask = facet_mask(n_docs, 30)
        queries = synthetic_vectors(6, dim, 1234).tolist()
        for skip, limit, dist in ((0, 10, None), (3, 7, (0.5, 0.2))):
            kept = _host.execute_vector_search_same_universe(ctx, queries, universe_mask, skip, limit, "default", dist)
            for q, got in zip(queries, kept):
                assert got == _host.execute_vector_search(ctx, q, universe_mask, skip, limit, "default", dist)
        # the same through a small (sorted array) universe and through the whole index (a range)
        few = [5, 77, 4000, 5999]
        for universe in (few, oracle.bool_to_mask(np.ones(n_docs, bool))):
            kept = _host.execute_vector_search_same_universe(ctx, queries, universe, 0, 10, "default", None)
            for q, got in zip(queries, kept):
                assert got == _host.execute_vector_search(ctx, q, universe, 0, 10, "default", None)

        universe = sorted(np.random.default_rng(3).choice(n_docs, 41, replace=False).tolist())
        q = queries[0]
        rule = _host.VectorSort(ctx, q, list(range(n_docs)), 4, "default", None)
        rule.start_iteration(universe)
        order = []
        while True:
            b = rule.next_bucket()
            if b is None or not b["candidates"]:
                break
            assert len(b["candidates"]) == 1
            order.append(b["candidates"][0])
        # what the rule does, replayed on the oracle's readers: a fill asks every store for its nearest
        # 4 of what is left, hands the documents out in (distance, docid) order -- a document that two
        # stores returned only once -- and the next fill starts from what is left then
        left, expect = set(universe), []
        while left:
            fill = []
            for nns in readers:
                fill += nns(q, 4, left)
            if not fill:
                break
            fill.sort(key=lambda t: (t[1], t[0]))
            for d, _ in fill:
                if d in left:
                    expect.append(d)
                    left.remove(d)
        assert order == expect and len(order) > 3 * 4 * n_stores  # several refills happened
    finally:
        for s in stores:
            s.destroy()


def test_vector_search_batch_with_a_universe_per_request(gpu_ctx):
    """B requests with B filters in one call (vs_search_batch_masks under the host mirror): result i
    equals the single search of vector i under universe i; requests that hand over the same universe
    object share passes, and on a second call their universe is already on the device."""
    from search_engine_amd import _host
    n_docs, dim, n_stores = 5_000, 96, 2
    stores, _ = build_world(gpu_ctx, n_docs, dim, n_stores, 21, (CATEGORY + 6) << 8)
    try:
        ctx = _host.GpuContext(gpu_ctx.handle.value, {"default": CATEGORY + 6},
                               oracle.bool_to_mask(np.ones(n_docs, bool)))
        queries = synthetic_vectors(9, dim, 4321).tolist()
        shared = facet_mask(n_docs, 20)
        universes = [shared, facet_mask(n_docs, 50), [3, 4, 4999], shared, [], oracle.bool_to_mask(np.ones(n_docs, bool)),
                     shared, facet_mask(n_docs, 1), list(range(100, 200))]
        for skip, limit, dist in ((0, 10, None), (4, 6, (0.5, 0.2))):
            for _ in range(2):
                got = _host.execute_vector_search_batch_universes(ctx, queries, universes, skip, limit, "default", dist)
                for q, u, g in zip(queries, universes, got):
                    assert g == _host.execute_vector_search(ctx, q, u, skip, limit, "default", dist)
    finally:
        for s in stores:
            s.destroy()
