"""The CPU oracle against the committed golden vectors and against an independent
numpy restatement of the same definitions (no GPU)."""
import json
import os

import numpy as np
import pytest

from conftest import facet_mask, synthetic_vectors
from helpers import tol
from oracle import scan as oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.mark.parametrize("mode", [oracle.ACC_F32_SEQ, oracle.ACC_F32_SIMD8, oracle.ACC_F64])
def test_appendix_b_known_answers(mode):
    g = load("appendix_b.json")
    q = np.array(g["query"], np.float32)
    for store in g["stores"]:
        vectors = np.array([e["vector"] for e in store], np.float32)
        docids = np.array([e["docid"] for e in store], np.uint32)
        ids, dist = oracle.nns_by_vector(vectors, q, 10, docids=docids, mode=mode)
        expect = sorted(store, key=lambda e: (e["distance"], e["docid"]))
        assert ids.tolist() == [e["docid"] for e in expect]
        np.testing.assert_allclose(dist, [e["distance"] for e in expect], atol=1e-6)


def test_scan_small_golden():
    g = load("scan_small.json")
    vectors = np.array(g["vectors"], np.float32)
    q = np.array(g["query"], np.float32)
    d64 = np.array(g["distances_f64"])
    mask = oracle.bool_to_mask(np.isin(np.arange(vectors.shape[0]), g["candidates"]))
    for mode in (oracle.ACC_F32_SEQ, oracle.ACC_F32_SIMD8, oracle.ACC_F64):
        ids, dist = oracle.nns_by_vector(vectors, q, 10, mask=mask, mode=mode)
        assert ids.tolist() == g["top10_filtered"]
        assert np.all(np.abs(dist - d64[ids]) <= tol(d64[ids]))
        ids, _ = oracle.nns_by_vector(vectors, q, 10, mode=mode)
        assert ids.tolist() == g["top10_unfiltered"]
    # zero-norm row -> distance exactly 0; exact duplicates -> tie broken by docid
    assert d64[40] == 0.0 and g["top10_unfiltered"][0] == 40
    np.testing.assert_allclose(oracle.all_distances_f64(vectors, q), d64, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("dim,pass_percent", [(3, 100), (96, 50), (384, 10), (768, 1)])
def test_c_oracle_matches_numpy_restatement(dim, pass_percent):
    n = 5000
    vectors = synthetic_vectors(n, dim, 100 + dim)
    q = synthetic_vectors(1, dim, 200 + dim)[0]
    mask = facet_mask(n, pass_percent)
    ids, dist = oracle.nns_by_vector(vectors, q, 10, mask=mask)
    np_ids, np_dist = oracle.np_nns_by_vector(vectors, q, 10, mask=mask)
    d64 = oracle.np_angular_f64(vectors, q)
    assert np.all(np.abs(dist - d64[ids]) <= tol(d64[ids]))
    for a, b in zip(ids, np_ids):  # identical up to near ties
        assert a == b or abs(d64[a] - d64[b]) <= tol(d64[a]) + tol(d64[b])


def test_ordering_rules():
    # all distances equal (every row identical): order is by docid, count respected
    v = np.tile(synthetic_vectors(1, 16, 1), (50, 1))
    docids = np.arange(50, dtype=np.uint32)[::-1].copy()
    ids, dist = oracle.nns_by_vector(v, v[0] * 2, 7, docids=docids)
    assert ids.tolist() == list(range(7)) and len(set(dist.tolist())) == 1
    # NaN distances sort after every number (OrderedFloat)
    v = synthetic_vectors(6, 8, 2)
    v[2, 0] = np.nan
    ids, dist = oracle.nns_by_vector(v, synthetic_vectors(1, 8, 3)[0], 6)
    assert ids[-1] == 2 and np.isnan(dist[-1]) and not np.isnan(dist[:-1]).any()
    # count == 0 -> empty; count > candidates -> all candidates
    assert oracle.nns_by_vector(v, v[0], 0)[0].size == 0
    assert oracle.nns_by_vector(v, v[0], 100)[0].size == 6
    # pn * qn == 0 -> 0
    assert oracle.lib().oracle_angular_distance(1.0, 0.0, 3.0) == 0.0
    assert oracle.lib().oracle_angular_distance(-6.0, 2.0, 3.0) == 1.0  # cos = -1
    assert oracle.lib().oracle_angular_distance(0.0, 2.0, 3.0) == 0.5   # cos = 0


def test_batch_mt_equals_single():
    vectors = synthetic_vectors(3000, 64, 5)
    queries = synthetic_vectors(6, 64, 6)
    mask = facet_mask(3000, 50)
    ids, dist, cnt, used = oracle.nns_batch_mt(vectors, queries, 10, mask=mask, n_threads=3)
    assert used >= 1
    for i, q in enumerate(queries):
        a, b = oracle.nns_by_vector(vectors, q, 10, mask=mask)
        assert cnt[i] == 10 and ids[i].tolist() == a.tolist() and dist[i].tolist() == b.tolist()


def test_hamming_oracle_matches_numpy():
    rng = np.random.default_rng(7)
    x = rng.standard_normal((2000, 1536)).astype(np.float32)
    codes = oracle.binarize(x)
    assert codes.shape == (2000, 24)
    q = oracle.binarize(rng.standard_normal((1, 1536)).astype(np.float32))[0]
    mask = facet_mask(2000, 50)
    ids, dist = oracle.hamming_topk(codes, q, 100, mask=mask)
    bits = np.unpackbits((codes ^ q).view(np.uint8), axis=1).sum(1)
    ok = oracle.mask_to_bool(mask, 2000)
    cand = np.nonzero(ok)[0]
    order = cand[np.lexsort((cand, bits[cand]))][:100]
    assert ids.tolist() == order.tolist() and dist.tolist() == bits[order].tolist()


def test_bf16_round_helper():
    x = np.array([1.0, 1.00390625, 1.01171875, -2.5, 3.14159265, 1e-40, np.inf], np.float32)
    r = oracle.bf16_round(x)
    assert r.dtype == np.float32 and np.all((r.view(np.uint32) & 0xFFFF) == 0)
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] == np.float32(1.015625)  # ties to even


@pytest.mark.parametrize("metric", [oracle.METRIC_EUCLIDEAN, oracle.METRIC_DOT])
def test_euclidean_and_dot_oracle_matches_numpy(metric):
    """The C restatement of arroy's Euclidean / DotProduct order against the numpy one
    (float64), known answers first."""
    v = np.array([[0, 1, 2], [6, 7, 8], [9, 10, 11], [0, -1, -2]], np.float32)
    q = np.array([0, 1, 2], np.float32)
    ids, dist = oracle.nns_by_vector_metric(v, q, 10, metric)
    if metric == oracle.METRIC_EUCLIDEAN:
        assert ids.tolist() == [0, 3, 1, 2]
        np.testing.assert_allclose(dist, np.sqrt([0.0, 20.0, 108.0, 243.0]), rtol=1e-7)
    else:
        assert ids.tolist() == [2, 1, 0, 3] and dist.tolist() == [32.0, 23.0, 5.0, -5.0]
    vectors = synthetic_vectors(3000, 96, seed=12)
    query = synthetic_vectors(1, 96, seed=13)[0]
    mask = facet_mask(3000, 30)
    ok = oracle.mask_to_bool(mask, 3000)
    f64 = oracle.np_metric_f64(vectors, query, metric)
    order_key = -f64 if metric == oracle.METRIC_DOT else f64
    expect = [i for i in np.argsort(order_key, kind="stable") if ok[i]][:20]
    for mode in (oracle.ACC_F32_SEQ, oracle.ACC_F64):
        ids, dist = oracle.nns_by_vector_metric(vectors, query, 20, metric, mask=mask, mode=mode)
        assert ids.tolist() == expect
        np.testing.assert_allclose(dist, f64[expect], rtol=1e-5, atol=1e-5)
    # exact ties come out by ascending id; an empty filter gives nothing
    twice = np.concatenate([vectors[:50], vectors[:50]])
    ids, dist = oracle.nns_by_vector_metric(twice, query, 6, metric)
    assert ids[0] + 50 == ids[1] and ids[2] + 50 == ids[3] and dist[0] == dist[1]
    assert len(oracle.nns_by_vector_metric(vectors, query, 5, metric, mask=np.zeros(47, np.uint64))[0]) == 0
