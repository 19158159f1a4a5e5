"""Parity at the sizes BASELINE.json names, where the CPU oracle would take minutes per query:
the store is filled on the device in chunks and the same chunks feed a plain PyTorch fp32 (or
integer) reference that keeps a running top-k, so no copy of the data is ever held twice. On top
of that the size-independent properties of the result: ascending (distance, docid) order, ids
inside the filter, top-5 = prefix of top-10, the same answer twice, planted neighbours found.

Tolerances are those of tests/helpers.py: ids equal position by position unless the two
candidates are closer than 1e-5 relative (+1e-6) in distance; float distances within the same
bound of the reference; Hamming distances and their order bit-exact."""
import numpy as np
import pytest
import torch

from conftest import hash32
from helpers import check_hits

pytestmark = pytest.mark.gpu

MARGIN = 6  # the torch reference nominates k + MARGIN candidates; float64 decides among them


def device_facet_mask(n_slots, pass_percent, dev):
    """Dense mask words for hash32(docid) % 100 < pass_percent, built on the device."""
    ids = torch.arange(n_slots, device=dev, dtype=torch.int64)
    x = ids
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & 0xFFFFFFFF
    x = ((x ^ (x >> 15)) * 0x846CA68B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    ok = (x % 100) < pass_percent
    w = (ok.view(-1, 64).to(torch.int64) << torch.arange(64, device=dev, dtype=torch.int64)).sum(dim=1)
    return ok, w


def test_device_facet_matches_the_host_one():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ok, _ = device_facet_mask(64 * 100, 10, torch.device("cuda:0"))
    np.testing.assert_array_equal(ok.cpu().numpy(), hash32(np.arange(6400)) % 100 < 10)


def running_topk(best, dist, ids, k):
    """Keeps the k smallest (distance, docid) of `best` and the new candidates."""
    d = torch.cat([best[0], dist])
    i = torch.cat([best[1], ids])
    # order by distance, ties by docid: sort by docid first (stable), then by distance
    o = torch.argsort(i, stable=True)
    d, i = d[o], i[o]
    o = torch.argsort(d, stable=True)[:k]
    return d[o], i[o]


def compare(store, query, ids, dist, ref_ids, k, n_candidates):
    """ids/dist: the library's answer. ref_ids: the k + MARGIN nominees of the fp32 torch reference.
    The rows of both are read back from the store and judged in float64 with the rules of
    helpers.check_hits (the query is the one the path computes with)."""
    cand = sorted(set(ids.tolist()) | set(int(i) for i in ref_ids.cpu().tolist()))
    q = np.asarray(query, np.float64)
    d64 = {}
    for c in cand:
        v = store.item_vector(c).astype(np.float64)
        pnqn = np.sqrt(v @ v) * np.sqrt(q @ q)
        d64[c] = (1.0 - (v @ q) / pnqn) / 2.0 if pnqn != 0 else 0.0
    expected = sorted(cand, key=lambda c: (d64[c], c))[:k]
    return check_hits(ids, dist, expected, d64, k, n_candidates)


@pytest.mark.parametrize("k", [10, 100])
def test_config2_1m_x_768_f32_against_the_cpu_oracle(gpu_ctx, k):
    """BASELINE config 2 is small enough for the C oracle itself (one query per thread):
    1 M x 768 f32, facet filters of 100 / 10 / 1 %."""
    from conftest import facet_mask
    from oracle import scan as oracle
    n, dim = 1_000_000, 768
    rng = np.random.default_rng(2024)
    vectors = rng.standard_normal((n, dim), dtype=np.float32)
    queries = rng.standard_normal((3, dim), dtype=np.float32)
    docids = np.arange(n, dtype=np.uint32)
    store = gpu_ctx.create_store(0x7004 + k, dim, n)
    try:
        for lo in range(0, n, 1 << 17):
            store.upload(docids[lo:lo + (1 << 17)], vectors[lo:lo + (1 << 17)])
        store.seal()
        norms = oracle.norms(vectors)
        for pass_percent in (100, 10, 1):
            mask = facet_mask(n, pass_percent) if pass_percent < 100 else None
            n_cand = n if mask is None else int(oracle.mask_to_bool(mask, n).sum())
            o_ids, _, o_n, _ = oracle.nns_batch_mt(vectors, queries, k, mask=mask, norms=norms)
            for j, q in enumerate(queries):
                ids, dist = store.search(q, k, mask)
                assert o_n[j] == k
                cand = sorted(set(ids.tolist()) | set(o_ids[j].tolist()))
                d64 = dict(zip(cand, oracle.all_distances_f64(vectors[cand], q).tolist()))
                assert check_hits(ids, dist, o_ids[j], d64, k, n_cand) <= 2
    finally:
        store.destroy()


@pytest.mark.parametrize("dtype_name,dim,n", [
    ("f32", 128, 1_100_000),   # 512-byte rows: four workgroups per CU
    ("bf16", 384, 1_100_000),  # 768-byte rows: likewise
    ("f32", 384, 600_000),     # 1.5 KiB rows: two per CU, with and without a mask
    ("bf16", 768, 600_000),    # BASELINE's bf16 row shape: likewise
])
def test_short_rows_on_their_larger_default_grids(gpu_ctx, dtype_name, dim, n):
    """The single-query scan sizes its default grid by the row shape (short and bf16 rows take two
    or four workgroups per CU instead of one); stores this large reach those grids. Against the C
    oracle, unfiltered and under 50 / 1 % facet filters, k = 10 and k = 100."""
    from conftest import facet_mask
    from oracle import scan as oracle
    from search_engine_amd import abi
    rng = np.random.default_rng(77 + dim)
    vectors = rng.standard_normal((n, dim), dtype=np.float32)
    if dtype_name == "bf16":
        vectors = oracle.bf16_round(vectors)
    queries = rng.standard_normal((2, dim), dtype=np.float32)
    docids = np.arange(n, dtype=np.uint32)
    store = gpu_ctx.create_store(0x7100 + dim + (1 if dtype_name == "bf16" else 0), dim, n,
                                 dtype=abi.VS_BF16 if dtype_name == "bf16" else abi.VS_F32)
    try:
        for lo in range(0, n, 1 << 17):
            store.upload(docids[lo:lo + (1 << 17)], vectors[lo:lo + (1 << 17)])
        store.seal()
        norms = oracle.norms(vectors)
        for pass_percent, k in ((100, 10), (50, 100), (1, 10)):
            mask = facet_mask(n, pass_percent) if pass_percent < 100 else None
            n_cand = n if mask is None else int(oracle.mask_to_bool(mask, n).sum())
            o_ids, _, o_n, _ = oracle.nns_batch_mt(vectors, queries, k, mask=mask, norms=norms)
            for j, q in enumerate(queries):
                ids, dist = store.search(q, k, mask)
                assert o_n[j] == k
                cand = sorted(set(ids.tolist()) | set(o_ids[j].tolist()))
                d64 = dict(zip(cand, oracle.all_distances_f64(vectors[cand], q).tolist()))
                assert check_hits(ids, dist, o_ids[j], d64, k, n_cand) <= 2
    finally:
        store.destroy()


@pytest.mark.parametrize("pass_percent", [100, 10])
def test_headline_10m_x_768_f32(gpu_ctx, pass_percent):
    from search_engine_amd import abi
    dev = torch.device("cuda:0")
    n, dim, k, n_q = 10_000_000, 768, 10, 3
    n_slots = -(-n // 64) * 64
    store = gpu_ctx.create_store(0x7001, dim, n)
    try:
        gen = torch.Generator(device=dev)
        gen.manual_seed(0xC0FFEE)
        queries = torch.randn(n_q, dim, generator=gen, device=dev)
        qn = queries.norm(dim=1)
        ok, words = device_facet_mask(n_slots, pass_percent, dev)
        best = [(torch.full((0,), 0.0, device=dev), torch.zeros(0, dtype=torch.int64, device=dev)) for _ in range(n_q)]
        stream = torch.cuda.current_stream().cuda_stream
        planted = {0: 7 if pass_percent == 100 else None, 1: n - 1 if pass_percent == 100 else None}
        chunk = 1 << 18
        for lo in range(0, n, chunk):
            m = min(chunk, n - lo)
            rows = torch.randn(m, dim, generator=gen, device=dev)
            for j, at in planted.items():  # an exact (scaled) copy of the query: distance ~ 0, rank 0
                if at is not None and lo <= at < lo + m:
                    rows[at - lo] = queries[j] * 3.0
            store.upload_device(lo, rows.data_ptr(), m, stream)
            pn = rows.norm(dim=1)
            pq = rows @ queries.T  # [m][n_q], fp32
            for j in range(n_q):
                d = (1.0 - pq[:, j] / (pn * qn[j])) / 2.0
                sel = ok[lo:lo + m]
                ids = torch.arange(lo, lo + m, device=dev)[sel]
                dj = d[sel]
                if dj.numel() > k + MARGIN:
                    top = torch.topk(dj, k + MARGIN, largest=False)
                    # everything within tolerance of the k-th is kept so that ties are ordered by docid below
                    keep = dj <= top.values[-1] * (1 + 1e-6) + 1e-9
                    dj, ids = dj[keep], ids[keep]
                best[j] = running_topk(best[j], dj, ids, k + MARGIN)
            torch.cuda.synchronize()
            del rows, pq
        store.seal()
        mask = words.cpu().numpy().view(np.uint64)
        for j in range(n_q):
            q = queries[j].cpu().numpy()
            ids, dist = store.search(q, k, mask)
            compare(store, q, ids, dist, best[j][1], k, int(ok[:n].sum().item()))
            assert ok[torch.from_numpy(ids.astype(np.int64)).to(dev)].all()           # inside the filter
            assert len(set(ids.tolist())) == k
            ids5, dist5 = store.search(q, 5, mask)                                     # prefix property
            assert ids5.tolist() == ids[:5].tolist() and dist5.tolist() == dist[:5].tolist()
            again = store.search(q, k, mask)                                           # idempotence
            assert again[0].tolist() == ids.tolist() and again[1].tolist() == dist.tolist()
            r_ids, r_dist = store.search_roaring(q, k, abi.mask_to_roaring(mask))      # the roaring form of the filter
            assert r_ids.tolist() == ids.tolist() and r_dist.tolist() == dist.tolist()
            if planted.get(j) is not None:
                assert ids[0] == planted[j] and dist[0] < 1e-6
    finally:
        store.destroy()
        torch.cuda.empty_cache()


def test_batched_1_25m_x_768_bf16(gpu_ctx):
    """One GPU's share of BASELINE config 3: 256 queries per pass on the matrix cores."""
    from search_engine_amd import abi
    dev = torch.device("cuda:0")
    n, dim, k, n_q = 1_250_000, 768, 10, 256
    n_slots = -(-n // 64) * 64
    store = gpu_ctx.create_store(0x7002, dim, n, dtype=abi.VS_BF16)
    try:
        gen = torch.Generator(device=dev)
        gen.manual_seed(99)
        queries = torch.randn(n_q, dim, generator=gen, device=dev)
        rq = queries.to(torch.bfloat16).to(torch.float32)  # the path rounds queries to bf16 (nearest even)
        qn = rq.norm(dim=1)
        ok, words = device_facet_mask(n_slots, 50, dev)
        check = [0, 1, 17, 128, 255]
        best = {j: (torch.full((0,), 0.0, device=dev), torch.zeros(0, dtype=torch.int64, device=dev)) for j in check}
        stream = torch.cuda.current_stream().cuda_stream
        chunk = 1 << 17
        for lo in range(0, n, chunk):
            m = min(chunk, n - lo)
            rows = torch.randn(m, dim, generator=gen, device=dev).to(torch.bfloat16)
            store.upload_device(lo, rows.data_ptr(), m, stream)
            r32 = rows.to(torch.float32)
            pn = r32.norm(dim=1)
            sel = ok[lo:lo + m]
            ids_all = torch.arange(lo, lo + m, device=dev)[sel]
            for j in check:
                d = ((1.0 - (r32 @ rq[j]) / (pn * qn[j])) / 2.0)[sel]
                top = torch.topk(d, k + MARGIN, largest=False)
                keep = d <= top.values[-1] * (1 + 1e-6) + 1e-9
                best[j] = running_topk(best[j], d[keep], ids_all[keep], k + MARGIN)
            torch.cuda.synchronize()
            del rows, r32
        store.seal()
        mask = words.cpu().numpy().view(np.uint64)
        results = store.search_batch(queries.cpu().numpy(), k, mask)
        assert len(results) == n_q
        for j, (ids, dist) in enumerate(results):
            assert len(ids) == k and (np.diff(dist) >= 0).all() and len(set(ids.tolist())) == k
            assert ok[torch.from_numpy(ids.astype(np.int64)).to(dev)].all()
            if j in best:
                compare(store, rq[j].cpu().numpy(), ids, dist, best[j][1], k, int(ok[:n].sum().item()))
        again = store.search_batch(queries.cpu().numpy(), k, mask)
        for (a, b), (c, d) in zip(results, again):
            assert a.tolist() == c.tolist() and b.tolist() == d.tolist()
    finally:
        store.destroy()
        torch.cuda.empty_cache()


def test_hamming_25m_x_1536_bits_top100(gpu_ctx):
    """One of two GPUs' share of BASELINE config 4; distances and order bit-exact."""
    from search_engine_amd import abi
    dev = torch.device("cuda:0")
    n, dim, k = 25_000_000, 1536, 100
    store = gpu_ctx.create_store(0x7003, dim, n, dtype=abi.VS_BIN1, metric=abi.VS_HAMMING)
    try:
        gen = torch.Generator(device=dev)
        gen.manual_seed(4)
        q = torch.randn(dim, generator=gen, device=dev)
        bitw = (1 << torch.arange(8, device=dev, dtype=torch.int32)).to(torch.uint8)
        qbits = ((q > 0).view(-1, 8).to(torch.uint8) * bitw).sum(dim=1).to(torch.uint8)  # LSB-first packing
        popc = torch.tensor([bin(v).count("1") for v in range(256)], dtype=torch.int16, device=dev)
        best = (torch.zeros(0, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int64, device=dev))
        stream = torch.cuda.current_stream().cuda_stream
        chunk = 1 << 20
        for lo in range(0, n, chunk):
            m = min(chunk, n - lo)
            rows = torch.randint(0, 256, (m, dim // 8), generator=gen, device=dev, dtype=torch.uint8)
            if lo == 0:
                rows[123] = qbits           # distance 0
                rows[124] = qbits
                rows[124, 5] ^= 1           # distance 1
            store.upload_device(lo, rows.data_ptr(), m, stream)
            d = popc[(rows ^ qbits).to(torch.int64)].sum(dim=1, dtype=torch.int64)
            ids = torch.arange(lo, lo + m, device=dev)
            key = d * (1 << 32) + ids  # (distance, docid) in one integer
            key = torch.cat([best[0], torch.topk(key, k, largest=False).values])
            best = (torch.sort(key).values[:k], best[1])
            torch.cuda.synchronize()
            del rows, d, key
        store.seal()
        ids, dist = store.search(q.cpu().numpy(), k)
        ref = best[0].cpu().numpy()
        assert ids.tolist() == (ref & 0xFFFFFFFF).tolist()
        assert dist.tolist() == (ref >> 32).astype(np.float32).tolist()
        assert ids[0] == 123 and dist[0] == 0 and ids[1] == 124 and dist[1] == 1
    finally:
        store.destroy()
        torch.cuda.empty_cache()
