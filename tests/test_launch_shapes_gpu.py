"""One scan launch for many queries, under filters whose density changes from one stretch of the store to
the next.

A workgroup decides from a sample of its own mask words whether it pairs its waves over the words
or lets half of them idle, and whether its query lanes collapse into one; rows of a sparse batch are
unpacked by rounds across lanes, rows of a dense one word by word. None of that may show in the
result: every query's list must be bit for bit the list vs_search returns for it alone (which
tests/test_scan_gpu.py holds against the oracle), whatever the mix of stretches a workgroup meets.
One query of every case is checked against the oracle here as well. The reference searches one
request at a time (src/search/ranking/vector_sort.rs:73-77); requests are concurrent in a server.
"""
import numpy as np
import pytest

from conftest import synthetic_vectors
from helpers import check_hits

pytestmark = pytest.mark.gpu


def _striped_mask(rng, n, stretch):
    """Candidate bits whose density is drawn per stretch of `stretch` docids: none, 1 %, half, all."""
    from oracle.scan import bool_to_mask
    keep = np.zeros(n, bool)
    for lo in range(0, n, stretch):
        p = rng.choice([0.0, 0.01, 0.5, 1.0], p=[0.2, 0.35, 0.25, 0.2])
        hi = min(n, lo + stretch)
        keep[lo:hi] = rng.random(hi - lo) < p
    return bool_to_mask(keep)


@pytest.mark.parametrize("seed", range(8))
def test_many_queries_under_striped_filters(gpu_ctx, seed):
    from oracle import scan as oracle
    from search_engine_amd import abi
    rng = np.random.default_rng(7000 + seed)
    kind = ["f32", "bf16", "bin1", "f32"][seed % 4]
    dim = int(rng.choice({"f32": [64, 200, 768], "bf16": [64, 768], "bin1": [256, 1536]}[kind]))
    n = int(rng.integers(30_000, 250_000 if dim <= 256 else 90_000))
    k = int(rng.choice([1, 10, 100, 200]))
    n_q = int(rng.integers(2, 17))
    dtype, metric = {"f32": (abi.VS_F32, abi.VS_ANGULAR), "bf16": (abi.VS_BF16, abi.VS_ANGULAR),
                     "bin1": (abi.VS_BIN1, abi.VS_HAMMING)}[kind]
    rows = synthetic_vectors(n, dim, 7100 + seed)
    store = gpu_ctx.create_store(720 + seed, dim, n, dtype=dtype, metric=metric)
    handles = []
    try:
        store.upload(np.arange(n, dtype=np.uint32), rows)
        store.seal()
        queries = synthetic_vectors(n_q, dim, 7200 + seed)
        filters = []
        for i in range(n_q):
            m = _striped_mask(rng, n, int(rng.choice([640, 4096, 30_000])))
            if i % 5 == 4:
                handles.append(store.create_mask(m))
                filters.append(handles[-1])
            elif i == 6:  # (once: queries that share a filter share passes, which on a bf16 store rounds the queries)
                filters.append(None)
            else:
                filters.append(m)
        got = store.search_batch_masks(queries, k, filters)
        for i, f in enumerate(filters):
            want = store.search(queries[i], k, f) if f is None or isinstance(f, np.ndarray) else \
                store.search_mask(queries[i], k, f)
            assert got[i][0].tolist() == want[0].tolist(), (kind, dim, n, k, n_q, i)
            assert got[i][1].tobytes() == want[1].tobytes(), (kind, dim, n, k, n_q, i)
        if kind == "f32":  # and the first filtered query against the oracle
            i = next(j for j, f in enumerate(filters) if isinstance(f, np.ndarray))
            o_ids, _ = oracle.nns_by_vector(rows, queries[i], k, mask=filters[i])
            d64 = oracle.all_distances_f64(rows, queries[i])
            n_cand = int(np.unpackbits(filters[i].view(np.uint8), bitorder="little")[:n].sum())
            check_hits(got[i][0], got[i][1], o_ids, d64, k, n_cand)
    finally:
        for h in handles:
            store.destroy_mask(h)
        store.destroy()
