"""ctypes front end of oracle/scan_oracle.c plus an independent numpy version.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). The C functions restate
arroy 0.2.0 as the reference calls it from
src/search/ranking/vector_sort.rs:73-77; the numpy functions below restate the
same definitions a second time, sharing no code with the C file, so that each
checks the other.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

ACC_F32_SEQ, ACC_F32_SIMD8, ACC_F64 = 0, 1, 2

_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)


def build():
    """Compile liboracle.so (gcc, a second or two)."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_nns_by_vector.restype = ctypes.c_uint32
        L.oracle_nns_by_vector.argtypes = [
            _f32p, _f32p, _u32p, ctypes.c_uint64, ctypes.c_uint32, _f32p, ctypes.c_uint32,
            _u64p, ctypes.c_uint64, ctypes.c_int, _u32p, _f32p]
        L.oracle_nns_batch_mt.restype = ctypes.c_int
        L.oracle_nns_batch_mt.argtypes = [
            _f32p, _f32p, _u32p, ctypes.c_uint64, ctypes.c_uint32, _f32p, ctypes.c_uint32,
            ctypes.c_uint32, _u64p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, _u32p, _f32p, _u32p]
        L.oracle_all_distances_f64.restype = None
        L.oracle_all_distances_f64.argtypes = [_f32p, ctypes.c_uint64, ctypes.c_uint32, _f32p, _f64p]
        L.oracle_norms.restype = None
        L.oracle_norms.argtypes = [_f32p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, _f32p]
        L.oracle_hamming_topk.restype = ctypes.c_uint32
        L.oracle_hamming_topk.argtypes = [
            _u64p, _u32p, ctypes.c_uint64, ctypes.c_uint32, _u64p, ctypes.c_uint32, _u64p,
            ctypes.c_uint64, _u32p, _u32p]
        L.oracle_nns_by_vector_metric.restype = ctypes.c_uint32
        L.oracle_nns_by_vector_metric.argtypes = [
            _f32p, _u32p, ctypes.c_uint64, ctypes.c_uint32, _f32p, ctypes.c_uint32, _u64p,
            ctypes.c_uint64, ctypes.c_int, ctypes.c_int, _u32p, _f32p]
        L.oracle_angular_distance.restype = ctypes.c_float
        L.oracle_angular_distance.argtypes = [ctypes.c_float] * 3
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def nns_by_vector(vectors, query, count, docids=None, mask=None, mode=ACC_F32_SIMD8, norms=None):
    """Exhaustive arroy nns_by_vector: (ids, distances) ascending by (distance, id)."""
    vectors = _c(vectors, np.float32)
    query = _c(query, np.float32)
    docids = _c(docids, np.uint32)
    mask = _c(mask, np.uint64)
    norms = _c(norms, np.float32)
    n, dim = vectors.shape
    assert query.shape == (dim,)
    ids = np.zeros(max(count, 1), np.uint32)
    dist = np.zeros(max(count, 1), np.float32)
    m = lib().oracle_nns_by_vector(
        _p(vectors, _f32p), _p(norms, _f32p), _p(docids, _u32p), n, dim, _p(query, _f32p), count,
        _p(mask, _u64p), 0 if mask is None else mask.size, mode, _p(ids, _u32p), _p(dist, _f32p))
    return ids[:m].copy(), dist[:m].copy()


def nns_batch_mt(vectors, queries, count, mask=None, mode=ACC_F32_SIMD8, n_threads=0, norms=None):
    """One query per thread. Returns (ids[n_q, count], dist, n[n_q], threads_used)."""
    vectors = _c(vectors, np.float32)
    queries = _c(queries, np.float32)
    mask = _c(mask, np.uint64)
    norms = _c(norms, np.float32)
    n, dim = vectors.shape
    n_q = queries.shape[0]
    ids = np.zeros((n_q, count), np.uint32)
    dist = np.zeros((n_q, count), np.float32)
    cnt = np.zeros(n_q, np.uint32)
    used = lib().oracle_nns_batch_mt(
        _p(vectors, _f32p), _p(norms, _f32p), None, n, dim, _p(queries, _f32p), n_q, count,
        _p(mask, _u64p), 0 if mask is None else mask.size, mode, n_threads, _p(ids, _u32p),
        _p(dist, _f32p), _p(cnt, _u32p))
    return ids, dist, cnt, used


def all_distances_f64(vectors, query):
    vectors = _c(vectors, np.float32)
    query = _c(query, np.float32)
    out = np.zeros(vectors.shape[0], np.float64)
    lib().oracle_all_distances_f64(_p(vectors, _f32p), vectors.shape[0], vectors.shape[1],
                                   _p(query, _f32p), _p(out, _f64p))
    return out


def norms(vectors, mode=ACC_F32_SIMD8):
    vectors = _c(vectors, np.float32)
    out = np.zeros(vectors.shape[0], np.float32)
    lib().oracle_norms(_p(vectors, _f32p), vectors.shape[0], vectors.shape[1], mode, _p(out, _f32p))
    return out


def hamming_topk(codes, query, count, docids=None, mask=None):
    codes = _c(codes, np.uint64)
    query = _c(query, np.uint64)
    docids = _c(docids, np.uint32)
    mask = _c(mask, np.uint64)
    n, words = codes.shape
    ids = np.zeros(max(count, 1), np.uint32)
    dist = np.zeros(max(count, 1), np.uint32)
    m = lib().oracle_hamming_topk(_p(codes, _u64p), _p(docids, _u32p), n, words, _p(query, _u64p),
                                  count, _p(mask, _u64p), 0 if mask is None else mask.size,
                                  _p(ids, _u32p), _p(dist, _u32p))
    return ids[:m].copy(), dist[:m].copy()


METRIC_EUCLIDEAN, METRIC_DOT = 2, 3


def nns_by_vector_metric(vectors, query, count, metric, docids=None, mask=None, mode=ACC_F32_SEQ):
    """Exhaustive top-`count` under arroy's Euclidean or DotProduct (no counterpart in the
    reference, which has Angular stores only: src/index.rs:155). Returns (ids, reported
    distances): Euclidean = sqrt of the squared distance the results are ordered by; dot =
    the product itself, results in descending order of it."""
    vectors = _c(vectors, np.float32)
    query = _c(query, np.float32)
    docids = _c(docids, np.uint32)
    mask = _c(mask, np.uint64)
    n, dim = vectors.shape
    ids = np.zeros(max(count, 1), np.uint32)
    dist = np.zeros(max(count, 1), np.float32)
    m = lib().oracle_nns_by_vector_metric(
        _p(vectors, _f32p), _p(docids, _u32p), n, dim, _p(query, _f32p), count, _p(mask, _u64p),
        0 if mask is None else mask.size, metric, mode, _p(ids, _u32p), _p(dist, _f32p))
    return ids[:m].copy(), dist[:m].copy()


def np_metric_f64(vectors, query, metric):
    """Reported distance of every row in float64 (numpy restatement of the function above)."""
    v = np.asarray(vectors, np.float64)
    q = np.asarray(query, np.float64)
    if metric == METRIC_DOT:
        return v @ q
    diff = v - q
    return np.sqrt((diff * diff).sum(1))


# ---- independent numpy restatement ------------------------------------------------------


def mask_to_bool(mask, n):
    """Dense u64 mask -> bool[n] (docid d = bit d & 63 of word d >> 6)."""
    if mask is None:
        return np.ones(n, bool)
    mask = np.ascontiguousarray(mask, np.uint64)
    bits = np.unpackbits(mask.view(np.uint8), bitorder="little")
    out = np.zeros(n, bool)
    m = min(n, bits.size)
    out[:m] = bits[:m].astype(bool)
    return out


def bool_to_mask(b):
    b = np.asarray(b, bool)
    n_words = (b.size + 63) // 64
    padded = np.zeros(n_words * 64, np.uint8)
    padded[: b.size] = b
    return np.packbits(padded, bitorder="little").view(np.uint64).copy()


def np_angular_f64(vectors, query):
    """Angular distance of every row in float64: (1 - cos) / 2, 0 when a norm is 0."""
    v = np.asarray(vectors, np.float64)
    q = np.asarray(query, np.float64)
    pn = np.sqrt((v * v).sum(1))
    qn = np.sqrt((q * q).sum())
    pnqn = pn * qn
    with np.errstate(divide="ignore", invalid="ignore"):
        d = (1.0 - (v @ q) / pnqn) / 2.0
    return np.where(pnqn != 0.0, d, 0.0)


def np_nns_by_vector(vectors, query, count, docids=None, mask=None):
    """Top-`count` by (float32(distance), id), distances computed in float64 then rounded."""
    n = vectors.shape[0]
    docids = np.arange(n, dtype=np.uint32) if docids is None else np.asarray(docids, np.uint32)
    d = np_angular_f64(vectors, query).astype(np.float32)
    if mask is not None:
        ok = mask_to_bool(mask, int(docids.max()) + 1 if n else 0)[docids]
    else:
        ok = np.ones(n, bool)
    idx = np.nonzero(ok)[0]
    order = np.lexsort((docids[idx], d[idx]))[:count]
    sel = idx[order]
    return docids[sel], d[sel]


def bf16_round(x):
    """Round f32 to bf16 (nearest even) and widen back to f32."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def binarize(x):
    """bit i = (x[i] > 0), LSB first into u64 words."""
    x = np.asarray(x)
    n, dim = x.shape
    words = (dim + 63) // 64
    bits = np.zeros((n, words * 64), np.uint8)
    bits[:, :dim] = x > 0
    return np.packbits(bits, axis=1, bitorder="little").view(np.uint64).reshape(n, words).copy()
