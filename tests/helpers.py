"""Shared checks of the parity tests.

Float tolerance (stated once, used everywhere): a GPU distance d must satisfy
    |d - d64| <= 1e-5 * |d64| + 1e-6
against the float64 value d64 of the same formula. 1e-5 relative is the bar the
task sets for float distances; the absolute floor covers d -> 0, where
(1 - cos) / 2 cancels and no f32 evaluation (the reference's included) has a
meaningful relative error.

Id parity: result lists must be IDENTICAL to the oracle's, position by
position, except where two candidates are closer than the tolerance above in
float64 (a near tie: the reference's own SIMD summation order would not pin
their order either). Exact ties (bit-identical distances) must break by docid.
"""
import numpy as np

RTOL = 1e-5
ATOL = 1e-6


def tol(d64):
    return RTOL * np.abs(d64) + ATOL


def check_hits(ids, dist, oracle_ids, d64_by_docid, k, n_candidates):
    """ids/dist: GPU result. oracle_ids: oracle result for the same inputs.
    d64_by_docid: mapping docid -> float64 distance (dict or array)."""
    ids = np.asarray(ids)
    dist = np.asarray(dist)
    assert ids.size == dist.size == min(k, n_candidates), (ids.size, k, n_candidates)
    assert ids.size == np.unique(ids).size, "duplicate docids in result"
    if ids.size == 0:
        return 0
    d64 = np.array([d64_by_docid[int(i)] for i in ids])
    # 1. distances within tolerance of float64
    err = np.abs(dist.astype(np.float64) - d64)
    assert np.all(err <= tol(d64)), f"distance error {err.max()} beyond tolerance"
    # 2. ascending by (distance, docid), bit-exact on the returned f32 values
    for i in range(1, ids.size):
        assert (dist[i - 1], ids[i - 1]) < (dist[i], ids[i]) or (
            dist[i - 1] == dist[i] and ids[i - 1] < ids[i]), f"order broken at {i}"
    # 3. same ids as the oracle, position by position, up to near ties
    oracle_ids = np.asarray(oracle_ids)
    assert oracle_ids.size == ids.size
    swaps = 0
    for i in range(ids.size):
        if ids[i] != oracle_ids[i]:
            a, b = d64_by_docid[int(ids[i])], d64_by_docid[int(oracle_ids[i])]
            assert abs(a - b) <= tol(a) + tol(b), (
                f"position {i}: gpu id {ids[i]} (d64 {a}) vs oracle id {oracle_ids[i]} (d64 {b})")
            swaps += 1
    return swaps
