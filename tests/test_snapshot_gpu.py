"""The snapshot rule while a writer works, and the per-query filters of vs_search_batch_masks.

A search must see the items and the rows of ONE sealed generation (the reference's readers hold
an LMDB RoTxn while the single writer removes and re-adds the same docids,
src/update/index_documents/typed_chunk.rs:618,650,691,702): never a new row under an old norm,
never a half-published item set.
"""
import threading

import numpy as np
import pytest

from conftest import facet_mask, synthetic_vectors
from helpers import check_hits

pytestmark = pytest.mark.gpu


def _oracle(rows_by_docid, present, q, k, mask=None):
    """Oracle answer for the generation described by (rows, present)."""
    from oracle import scan as oracle
    cand = oracle.bool_to_mask(present)
    if mask is not None:
        cand = cand & mask[: cand.size]
    return oracle.nns_by_vector(rows_by_docid, q, k, mask=cand)


def test_replaced_rows_stay_invisible_until_the_seal(gpu_ctx):
    from oracle import scan as oracle
    n, dim, k = 3000, 64, 10
    rows0 = synthetic_vectors(n, dim, 21)
    store = gpu_ctx.create_store(700, dim, n)
    try:
        store.upload(np.arange(n, dtype=np.uint32), rows0)
        store.seal()
        q = synthetic_vectors(1, dim, 22)[0]
        ids0, dist0 = store.search(q, k)
        # replace the current winners (and one more docid twice in one call: the last row wins)
        rows1 = rows0.copy()
        victims = ids0[:5].astype(np.uint32)
        rows1[victims] = synthetic_vectors(5, dim, 23) * 7.0
        dup = np.uint32((int(ids0[6]) + 1) % n)
        first_try = synthetic_vectors(1, dim, 24)[0]
        rows1[dup] = q * 3.0  # parallel to the query: distance 0 once visible
        store.remove(victims)  # the indexer deletes, then adds again (typed_chunk.rs:618,650)
        store.upload(np.concatenate([victims, [dup, dup]]).astype(np.uint32),
                     np.concatenate([rows1[victims], first_try[None], rows1[dup][None]]))
        # the writer's own view already shows the new rows ...
        assert np.array_equal(store.item_vector(int(victims[0])), rows1[victims[0]])
        assert np.array_equal(store.item_vector(int(dup)), rows1[dup])
        # ... a search still sees the sealed generation, bit for bit
        ids_mid, dist_mid = store.search(q, k)
        assert ids_mid.tolist() == ids0.tolist() and dist_mid.tobytes() == dist0.tobytes()
        store.seal()
        ids1, dist1 = store.search(q, k)
        o_ids, _ = oracle.nns_by_vector(rows1, q, k)
        d64 = oracle.all_distances_f64(rows1, q)
        check_hits(ids1, dist1, o_ids, d64, k, n)
        assert ids1[0] == dup and dist1[0] <= 1e-6
    finally:
        store.destroy()


@pytest.mark.parametrize("dtype_name", ["bf16", "bin1"])
def test_replaced_rows_other_element_types(gpu_ctx, dtype_name):
    from search_engine_amd import abi
    n, dim, k = 2000, 128, 5
    dtype, metric = (abi.VS_BF16, abi.VS_ANGULAR) if dtype_name == "bf16" else (abi.VS_BIN1, abi.VS_HAMMING)
    rows0 = synthetic_vectors(n, dim, 31)
    store = gpu_ctx.create_store(701, dim, n, dtype=dtype, metric=metric)
    fresh = gpu_ctx.create_store(702, dim, n, dtype=dtype, metric=metric)
    try:
        store.upload(np.arange(n, dtype=np.uint32), rows0)
        store.seal()
        q = synthetic_vectors(1, dim, 32)[0]
        before = store.search(q, k)
        rows1 = rows0.copy()
        changed = np.arange(0, n, 3, dtype=np.uint32)
        rows1[changed] = synthetic_vectors(changed.size, dim, 33)
        store.upload(changed, rows1[changed])
        mid = store.search(q, k)
        assert mid[0].tolist() == before[0].tolist() and mid[1].tobytes() == before[1].tobytes()
        store.seal()
        # the same rows loaded into a fresh store give the reference answer, bit for bit
        fresh.upload(np.arange(n, dtype=np.uint32), rows1)
        fresh.seal()
        a, b = store.search(q, k), fresh.search(q, k)
        assert a[0].tolist() == b[0].tolist() and a[1].tobytes() == b[1].tobytes()
        assert np.array_equal(store.item_vector(3), fresh.item_vector(3))
    finally:
        store.destroy()
        fresh.destroy()


def test_searches_race_a_writer_and_always_see_one_generation(gpu_ctx):
    """4 searching threads against 1 writer that removes, replaces, adds and seals in a loop."""
    n, dim, k, generations = 4096, 48, 8, 12
    rng = np.random.default_rng(41)
    base = synthetic_vectors(n, dim, 42)
    queries = synthetic_vectors(4, dim, 43)
    mask = facet_mask(n, 50)
    # generation g: rows and item set, precomputed so that the oracle can answer for each
    gens = []
    rows = base.copy()
    present = np.zeros(n, bool)
    present[: n // 2] = True
    gens.append((rows.copy(), present.copy(), None))
    for g in range(1, generations):
        live = np.flatnonzero(present)
        removed = rng.choice(live, 40, replace=False)
        replaced = rng.choice(np.setdiff1d(live, removed), 60, replace=False)
        added = rng.choice(np.flatnonzero(~present), 50, replace=False)
        rows = rows.copy()
        rows[replaced] = rng.standard_normal((replaced.size, dim), dtype=np.float32) * (1 + g)
        rows[added] = rng.standard_normal((added.size, dim), dtype=np.float32)
        present = present.copy()
        present[removed] = False
        present[added] = True
        gens.append((rows, present, (removed, replaced, added)))
    # what a quiet store holding generation g answers (bit-exact reference for the racing searches),
    # itself checked against the oracle
    from oracle import scan as oracle
    answers = [[] for _ in queries]
    for (r, p, _) in gens:
        quiet = gpu_ctx.create_store(706, dim, n)
        try:
            ids_g = np.flatnonzero(p).astype(np.uint32)
            quiet.upload(ids_g, r[ids_g])
            quiet.seal()
            for t, q in enumerate(queries):
                ids, dist = quiet.search(q, k, mask)
                o_ids, _ = _oracle(r, p, q, k, mask)
                check_hits(ids, dist, o_ids, oracle.all_distances_f64(r, q), k, int(1e9))
                answers[t].append(ids.tolist())
        finally:
            quiet.destroy()

    store = gpu_ctx.create_store(703, dim, n)
    try:
        r0, p0, _ = gens[0]
        ids0 = np.flatnonzero(p0).astype(np.uint32)
        store.upload(ids0, r0[ids0])
        store.seal()
        stop = threading.Event()
        errors, seen = [], [set() for _ in queries]

        def searcher(t):
            try:
                while not stop.is_set():
                    ids, _ = store.search(queries[t], k, mask)
                    got = ids.tolist()
                    hit = [g for g, a in enumerate(answers[t]) if a == got]
                    if not hit:
                        errors.append((t, got))
                        return
                    seen[t].update(hit)
            except Exception as e:  # noqa: BLE001
                errors.append((t, repr(e)))

        threads = [threading.Thread(target=searcher, args=(t,)) for t in range(len(queries))]
        for th in threads:
            th.start()
        try:
            for g in range(1, generations):
                r, _, (removed, replaced, added) = gens[g]
                store.remove(removed.astype(np.uint32))
                store.remove(replaced.astype(np.uint32))  # del_item then add_item, like the indexer
                store.upload(replaced.astype(np.uint32), r[replaced])
                store.upload(added.astype(np.uint32), r[added])
                store.seal()
        finally:
            stop.set()
            for th in threads:
                th.join()
        assert not errors, errors[:3]
        # after the last seal every thread's next search is the last generation
        for t, q in enumerate(queries):
            assert store.search(q, k, mask)[0].tolist() == answers[t][-1]
    finally:
        store.destroy()


def test_device_api_searches_are_waited_for_by_the_seal(gpu_ctx):
    """Searches enqueued on a caller stream, then remove + seal twice: the keys come from the
    generation that was sealed when each search was enqueued."""
    import torch
    from search_engine_amd import sharded
    n, dim, k = 200_000, 64, 10
    dev = torch.device("cuda", 0)
    rows = synthetic_vectors(n, dim, 51)
    store = gpu_ctx.create_store(704, dim, n)
    try:
        store.upload(np.arange(n, dtype=np.uint32), rows)
        store.seal()
        q = synthetic_vectors(1, dim, 52)[0]
        d_q = torch.from_numpy(q).to(dev)
        ws = torch.empty(store.workspace_bytes(k), dtype=torch.uint8, device=dev)
        stream = torch.cuda.Stream()
        outs = []
        expect = []
        for gen in range(3):
            want, _ = store.search(q, k)
            expect.append(want.tolist())
            keys = torch.zeros(k, dtype=torch.int64, device=dev)
            with torch.cuda.stream(stream):
                for _ in range(20):  # keep the stream busy so that the seal really has to wait
                    store.search_device(d_q.data_ptr(), dim, k, 0, ws.data_ptr(), keys.data_ptr(), stream.cuda_stream)
            outs.append(keys)
            # drop this generation's winners and publish, twice in a row (the second seal reuses
            # the present buffer the searches above were launched on)
            store.remove(want[:3])
            store.seal()
            store.remove(want[3:4])
            store.seal()
        torch.cuda.synchronize()
        for keys, want in zip(outs, expect):
            ids, _ = sharded.unpack_keys(keys.cpu().numpy())
            assert ids.tolist() == want
    finally:
        store.destroy()


def test_one_filter_per_query(gpu_ctx):
    n, dim, k = 20_000, 96, 10
    rows = synthetic_vectors(n, dim, 61)
    store = gpu_ctx.create_store(705, dim, n)
    try:
        store.upload(np.arange(n, dtype=np.uint32), rows)
        store.seal()
        queries = synthetic_vectors(11, dim, 62)
        m10, m50 = facet_mask(n, 10), facet_mask(n, 50)
        h1 = store.create_mask(facet_mask(n, 1))
        empty = np.zeros(n // 64 + 1, np.uint64)
        filters = [None, m10, m50, h1, m10, None, h1, m50, m10, empty, None]
        got = store.search_batch_masks(queries, k, filters)
        for i, f in enumerate(filters):
            if f is None or isinstance(f, np.ndarray):
                want = store.search(queries[i], k, f)
            else:
                want = store.search_mask(queries[i], k, f)
            assert got[i][0].tolist() == want[0].tolist(), i
            assert got[i][1].tobytes() == want[1].tobytes(), i
        assert got[9][0].size == 0
        store.destroy_mask(h1)
    finally:
        store.destroy()


@pytest.mark.parametrize("dtype_name,dim,k", [("f32", 96, 10), ("f32", 768, 300), ("bin1", 1536, 100)])
def test_a_filter_of_its_own_for_every_query(gpu_ctx, dtype_name, dim, k):
    """A batch of requests with a universe each: no two queries share a filter, so every one is a group
    of its own, and the call scans for all of them in ONE launch, each query under its own mask (host
    words staged side by side, device masks named in place, one query without a filter). Result i is
    bit for bit vs_search of query i under filter i."""
    from search_engine_amd import abi
    n = 40_000
    dtype, metric = (abi.VS_BIN1, abi.VS_HAMMING) if dtype_name == "bin1" else (abi.VS_F32, abi.VS_ANGULAR)
    rows = synthetic_vectors(n, dim, 63)
    store = gpu_ctx.create_store(706, dim, n, dtype=dtype, metric=metric)
    handles = []
    try:
        store.upload(np.arange(n, dtype=np.uint32), rows)
        store.seal()
        queries = synthetic_vectors(9, dim, 64)
        handles = [store.create_mask(facet_mask(n, p)) for p in (2, 30)]
        short = facet_mask(n, 60)[: n // 128]  # fewer words than the store has: the rest are no candidates
        filters = [facet_mask(n, 10), handles[0], None, facet_mask(n, 50), handles[1], short,
                   facet_mask(n, 100), facet_mask(n, 1), np.zeros(n // 64 + 1, np.uint64)]
        got = store.search_batch_masks(queries, k, filters)
        for i, f in enumerate(filters):
            if f is None or isinstance(f, np.ndarray):
                want = store.search(queries[i], k, f)
            else:
                want = store.search_mask(queries[i], k, f)
            assert got[i][0].tolist() == want[0].tolist(), i
            assert got[i][1].tobytes() == want[1].tobytes(), i
        assert got[8][0].size == 0 and got[2][0].size == k
    finally:
        for h in handles:
            store.destroy_mask(h)
        store.destroy()


def test_parked_rows_survive_the_growth_of_their_buffer(gpu_ctx):
    """Rows that replace visible items are parked beside the slab; the buffer starts at 1024 rows and is
    regrown (copy, then free) when an upload needs more. Copies of an earlier vs_store_upload_device may
    still be queued on the caller's stream at that moment: the growth has to wait for them, or the rows
    they park are lost and the seal scatters garbage."""
    import torch
    dev = torch.device("cuda", 0)
    n, dim, k = 4096, 64, 5
    rows_a = synthetic_vectors(n, dim, 81)
    rows_b = synthetic_vectors(n, dim, 82)
    store = gpu_ctx.create_store(707, dim, n)
    try:
        store.upload(np.arange(n, dtype=np.uint32), rows_a)
        store.seal()
        side = torch.cuda.Stream(device=dev)
        d_b = torch.from_numpy(rows_b).to(dev)
        busy = torch.randn(4096, 4096, device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(40):  # the copies below queue up behind this
                busy = busy @ busy * 1e-4
            store.upload_device(0, d_b[:700].data_ptr(), 700, side.cuda_stream)            # parks 700 rows
            store.upload_device(700, d_b[700:2500].data_ptr(), 1800, side.cuda_stream)      # 2500 > 1024: regrown
            store.upload_device(2500, d_b[2500:3000].data_ptr(), 500, side.cuda_stream)     # 3000 > 2048: again
        # until the seal, searches still see generation A
        got, _ = store.search(rows_a[10], k)
        assert got[0] == 10
        store.seal()
        torch.cuda.synchronize()
        for d in (0, 5, 699, 700, 1023, 1024, 2047, 2048, 2499, 2500, 2999):
            assert store.item_vector(d).tobytes() == rows_b[d].tobytes(), d
            got, dist = store.search(rows_b[d], k)
            assert got[0] == d and dist[0] < 1e-6
        for d in (3000, 4095):
            assert store.item_vector(d).tobytes() == rows_a[d].tobytes(), d
    finally:
        store.destroy()


def test_a_workspace_sized_before_the_store_grew_is_refused(gpu_ctx):
    """Device-API searches take the size of their workspace and check it against the store as it is
    now: the batched scan's workspace grows with the store, so one sized before vs_store_reserve would
    be overrun."""
    import torch
    from search_engine_amd import abi
    dev = torch.device("cuda", 0)
    dim, k, n_q = 256, 10, 32
    store = gpu_ctx.create_store(708, dim, 100_000, dtype=abi.VS_BF16)
    try:
        rows = synthetic_vectors(100_000, dim, 83)
        store.upload(np.arange(100_000, dtype=np.uint32), rows)
        store.seal()
        before = store.batch_workspace_bytes(k)
        ws = torch.empty(before, dtype=torch.uint8, device=dev)
        d_q = torch.from_numpy(synthetic_vectors(n_q, dim, 84)).to(dev)
        keys = torch.zeros(n_q * k, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        store.search_batch_device(d_q.data_ptr(), n_q, dim, k, 0, ws.data_ptr(), keys.data_ptr(), stream,
                                  workspace_bytes=before)
        torch.cuda.synchronize()
        want = keys.clone()
        store.reserve(3_000_000)
        after = store.batch_workspace_bytes(k)
        assert after > before
        with pytest.raises(abi.VsError) as err:
            store.search_batch_device(d_q.data_ptr(), n_q, dim, k, 0, ws.data_ptr(), keys.data_ptr(), stream,
                                      workspace_bytes=before)
        assert err.value.code == abi.VS_E_CAPACITY
        ws = torch.empty(after, dtype=torch.uint8, device=dev)
        keys.zero_()
        store.search_batch_device(d_q.data_ptr(), n_q, dim, k, 0, ws.data_ptr(), keys.data_ptr(), stream,
                                  workspace_bytes=after)
        torch.cuda.synchronize()
        assert keys.cpu().tolist() == want.cpu().tolist()
        # the single-query entry checks its size too
        one = torch.zeros(k, dtype=torch.int64, device=dev)
        with pytest.raises(abi.VsError) as err:
            store.search_device(d_q.data_ptr(), dim, k, 0, ws.data_ptr(), one.data_ptr(), stream, workspace_bytes=4096)
        assert err.value.code == abi.VS_E_CAPACITY
    finally:
        store.destroy()
