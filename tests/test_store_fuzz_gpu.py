"""Random sequences of the writer's operations against a model of the store.

The reference's index is one LMDB environment with a single writer: readers see the last committed
generation, the writer sees its own changes at once (src/update/index_documents/typed_chunk.rs:610-707
deletes and re-adds items inside one RwTxn; src/search/ranking/vector_sort.rs:55-78 reads under a
RoTxn). The store keeps that contract with two views -- the sealed generation every search scans and
the writer's pending changes (parked replacement rows, item bits, removals) -- and this test walks
both through uploads from host and device memory, replacements, duplicates inside one call,
removals, capacity growth and seals in random order, checking after every operation that

* a search (plain, under a filter, through a retained mask, batched) answers for the SEALED model,
  position by position against the oracle;
* contains / item_vector / writer_len answer for the WRITER's model.
"""
import numpy as np
import pytest

from conftest import synthetic_vectors
from helpers import check_hits

pytestmark = pytest.mark.gpu


class Model:
    def __init__(self, cap, dim):
        self.dim = dim
        self.rows = np.zeros((cap, dim), np.float32)
        self.present = np.zeros(cap, bool)

    def grow(self, cap):
        rows = np.zeros((cap, self.dim), np.float32)
        rows[: self.rows.shape[0]] = self.rows
        present = np.zeros(cap, bool)
        present[: self.present.size] = self.present
        self.rows, self.present = rows, present

    def copy(self):
        m = Model(self.rows.shape[0], self.dim)
        m.rows, m.present = self.rows.copy(), self.present.copy()
        return m


def _check_search(store, sealed, rng, k, how):
    from oracle import scan as oracle
    q = rng.standard_normal(sealed.dim).astype(np.float32)
    cap = sealed.present.size
    cand = oracle.bool_to_mask(sealed.present)
    mask = None
    if how != "plain":
        keep = rng.random(cap) < rng.choice([0.02, 0.3, 0.9])
        mask = oracle.bool_to_mask(keep)
        cand = cand & mask[: cand.size]
    n_cand = int(sum(bin(int(w)).count("1") for w in cand))
    o_ids, _ = oracle.nns_by_vector(sealed.rows, q, k, mask=cand)
    d64 = oracle.all_distances_f64(sealed.rows, q)
    if how == "plain":
        ids, dist = store.search(q, k)
    elif how == "words":
        ids, dist = store.search(q, k, mask=mask)
    elif how == "handle":
        h = store.create_mask(mask=mask)
        try:
            assert store.mask_fits(h)
            ids, dist = store.search_mask(q, k, h)
        finally:
            store.destroy_mask(h)
    else:  # three queries sharing the filter; the first is the one checked
        qs = np.stack([q, -q, rng.standard_normal(sealed.dim).astype(np.float32)])
        res = store.search_batch(qs, k, mask=mask)
        ids, dist = res[0]
    check_hits(ids, dist, o_ids, d64, k, n_cand)


@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_writer_histories(gpu_ctx, seed, dtype_name):
    """bf16 stores round rows at upload (nearest even); the model keeps the rounded rows, device
    uploads hand over bf16 rows, and the batched form is left out there (it rounds the queries too)."""
    import torch
    from oracle import scan as oracle
    from search_engine_amd import abi
    bf16 = dtype_name == "bf16"
    rng = np.random.default_rng(1000 + seed + (50 if bf16 else 0))
    dim = int(rng.choice([24, 48, 100, 7]))
    cap = 512
    store = gpu_ctx.create_store(900 + seed, dim, cap, dtype=abi.VS_BF16 if bf16 else abi.VS_F32)
    hows = ["plain", "words", "handle"] + ([] if bf16 else ["batch"])
    writer, sealed = Model(cap, dim), Model(cap, dim)
    side = torch.cuda.Stream()
    keep_alive = []  # device rows of upload_device calls, until the next seal
    try:
        store.seal()
        for step in range(80):
            op = rng.choice(["upload", "upload", "upload_device", "remove", "reserve", "seal", "seal"])
            if op == "upload":
                m = int(rng.integers(1, 400))
                docids = rng.integers(0, cap, m).astype(np.uint32)  # duplicates on purpose: the last row wins
                rows = synthetic_vectors(m, dim, int(rng.integers(1 << 30)))
                rows[rng.random(m) < 0.01] = 0.0  # zero vectors are legal items
                store.upload(docids, rows)
                if bf16:
                    rows = oracle.bf16_round(rows)
                for d, r in zip(docids, rows):
                    writer.rows[d] = r
                    writer.present[d] = True
            elif op == "upload_device":
                m = int(rng.integers(1, min(cap, 1500)))
                first = int(rng.integers(0, cap - m + 1))
                rows = synthetic_vectors(m, dim, int(rng.integers(1 << 30)))
                with torch.cuda.stream(side):
                    # device rows come in the store's own layout: its element type, vs_store_row_bytes apart
                    elems = store.row_bytes // (2 if bf16 else 4)
                    t = torch.zeros(m, elems, device="cuda", dtype=torch.bfloat16 if bf16 else torch.float32)
                    t[:, :dim] = torch.from_numpy(rows).to("cuda", non_blocking=False)
                    if bf16:
                        rows = oracle.bf16_round(rows)
                    store.upload_device(first, t.data_ptr(), m, side.cuda_stream)
                keep_alive.append(t)
                writer.rows[first: first + m] = rows
                writer.present[first: first + m] = True
            elif op == "remove":
                m = int(rng.integers(1, 200))
                docids = rng.integers(0, cap, m).astype(np.uint32)
                store.remove(docids)
                writer.present[docids] = False
            elif op == "reserve":
                cap = cap + int(rng.integers(1, 1500))
                store.reserve(cap)
                writer.grow(cap)
                sealed.grow(cap)
            else:
                side.synchronize()
                store.seal()
                sealed = writer.copy()
                keep_alive.clear()
            # the writer's view
            assert store.writer_len() == int(writer.present.sum()), (step, op)
            assert len(store) == int(sealed.present.sum()), (step, op)
            for d in rng.integers(0, cap, 6):
                assert store.contains(int(d)) == bool(writer.present[d]), (step, op, int(d))
                if writer.present[d]:
                    assert np.array_equal(store.item_vector(int(d)), writer.rows[d]), (step, op, int(d))
            # the readers' view
            k = int(rng.choice([1, 7, 33, 150]))
            _check_search(store, sealed, rng, k, str(rng.choice(hows)))
    finally:
        side.synchronize()
        store.destroy()
