"""An index that already exists on disk, and the write path, on the GPU.

The reference's commented scenario (src/update/index_documents/mod.rs:2599-2619: documents with
2 / 1 / 3 user-provided vectors, query [0, 1, 2]) is replayed twice: from synthesized arroy item
leaves through vs_ingest_arroy_items (the bytes a host reads from `vector-arroy` with a cursor,
tree and metadata nodes mixed in), and through the write-path mirror over GpuWriters, followed by
deletions and replacements. Expected answers: tests/golden/appendix_b.json (hand-derived)."""
import json
import os
import struct

import numpy as np
import pytest

from conftest import synthetic_vectors

pytestmark = pytest.mark.gpu

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "appendix_b.json")))
DOCS = {0: [[0, 1, 2], [3, 4, 5]], 1: [[6, 7, 8]], 2: [[9, 10, 11], [12, 13, 14], [15, 16, 17]]}


def _search(gpu_ctx, category, name, universe, limit=10):
    from search_engine_amd import _host
    ctx = _host.GpuContext(gpu_ctx.handle.value, {name: category}, [0, 1, 2])
    return _host.execute_vector_search(ctx, GOLDEN["query"], universe, 0, limit, name, None)


def _check_emission(got, want):
    assert got["documents_ids"] == [e["docid"] for e in want]
    for s, e in zip(got["document_scores"], want):
        assert s[0]["similarity"] == pytest.approx(e["similarity"], abs=2e-6)


def _destroy_stores(gpu_ctx, category, n=4):
    for k in range(n):
        try:
            gpu_ctx.open_store((category << 8) | k).destroy()
        except Exception:  # noqa: BLE001  (never sealed or never made)
            pass


def arroy_key(index, mode, item):
    """arroy 0.2.0's key as published: index u16 big-endian | mode u8 | item u32 big-endian."""
    return struct.pack(">HBI", index, mode, item)


def arroy_leaf(vector):
    """An item leaf: tag u8 = 0 | Angular header { norm f32 } | dim x f32 little endian."""
    norm = float(np.sqrt(np.sum(np.asarray(vector, np.float64) ** 2)))
    return struct.pack("<Bf", 0, norm) + struct.pack(f"<{len(vector)}f", *vector)


def test_existing_index_from_arroy_leaves(gpu_ctx):
    """The parser is fed bytes it did not write: the pairs are packed here with struct.pack from the
    published layout and must equal the committed fixture (tests/golden/arroy_leaves.json, written the
    same way by make_golden.py), tree nodes, metadata and an unknown node mode mixed in. The layout
    itself is restated, not verified against the crate (arroy is not in the reference tree)."""
    from search_engine_amd import abi
    leaves = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "arroy_leaves.json")))
    category = leaves["category"]
    pairs = [(bytes.fromhex(k), bytes.fromhex(v)) for k, v in leaves["pairs"]]
    packed = [(arroy_key((category << 8) | j, 0, docid), arroy_leaf(v))
              for docid, vectors in DOCS.items() for j, v in enumerate(vectors)]
    assert sorted(packed) == [p for p in pairs if p[0][0] == category and p[0][2] == 0]
    assert pairs == sorted(pairs)  # LMDB order
    want = leaves["expect"]
    stores = [gpu_ctx.create_store((category << 8) | k, 3, 64) for k in range(3)]
    try:
        assert gpu_ctx.ingest_arroy_items(pairs) == (want["n_items"], want["n_other_nodes"], want["n_unrouted"])
        for s in stores:
            s.seal()
        assert [len(s) for s in stores] == want["store_lens"]
        for k, want_k in enumerate(GOLDEN["stores"]):
            ids, dist = stores[k].search(np.array(GOLDEN["query"], np.float32), 10)
            assert ids.tolist() == [e["docid"] for e in want_k]
            assert dist == pytest.approx([e["distance"] for e in want_k], abs=1e-6)
            for e in want_k:
                assert stores[k].item_vector(e["docid"]).tolist() == e["vector"]
        _check_emission(_search(gpu_ctx, category, "manual", [0, 1, 2]), GOLDEN["emission"])
        _check_emission(_search(gpu_ctx, category, "manual", [1, 2]), GOLDEN["emission_filtered_1_2"])
        # a leaf of the wrong dimension, bytes that are no leaf, a short key: every pair is checked
        # before the first upload, so the good pair in front of the bad one changes nothing either
        good = (arroy_key(category << 8, 0, 40), arroy_leaf([7, 7, 7]))
        key = arroy_key(category << 8, 0, 5)
        with pytest.raises(abi.VsError) as e:
            gpu_ctx.ingest_arroy_items([good, (key, arroy_leaf([1, 2, 3, 4]))])
        assert e.value.code == abi.VS_E_DIM_MISMATCH and (e.value.expected, e.value.received) == (3, 4)
        with pytest.raises(abi.VsError) as e:
            gpu_ctx.ingest_arroy_items([good, (key, bytes([1]) + arroy_leaf([1, 2, 3])[1:])])
        assert e.value.code == abi.VS_E_FORMAT
        with pytest.raises(abi.VsError) as e:
            gpu_ctx.ingest_arroy_items([good, (key[:6], arroy_leaf([1, 2, 3]))])
        assert e.value.code == abi.VS_E_FORMAT
        assert stores[0].writer_len() == 3 and not stores[0].contains(40)
    finally:
        for s in stores:
            s.destroy()


def test_bulk_ingest_equals_upload(gpu_ctx):
    from search_engine_amd import abi
    n, dim, index = 5000, 96, (11 << 8)
    rows = synthetic_vectors(n, dim, 81)
    docids = np.random.default_rng(82).permutation(20_000)[:n].astype(np.uint32)
    a = gpu_ctx.create_store(index, dim, 20_000)
    b = gpu_ctx.create_store(index | 1, dim, 20_000)
    try:
        assert gpu_ctx.ingest_arroy_items(abi.arroy_encode_item(index, int(d), r) for d, r in zip(docids, rows)) == (n, 0, 0)
        a.seal()
        b.upload(docids, rows)
        b.seal()
        q = synthetic_vectors(1, dim, 83)[0]
        ra, rb = a.search(q, 50), b.search(q, 50)
        assert ra[0].tolist() == rb[0].tolist() and ra[1].tobytes() == rb[1].tobytes()
    finally:
        a.destroy()
        b.destroy()


def test_write_path_over_gpu_stores(gpu_ctx):
    """write_vector_points / build_vector_stores (typed_chunk.rs:562-710, index_documents/mod.rs:554-563)
    driving vs_store_upload / remove / item_vector / seal."""
    from search_engine_amd import _host
    category = 12
    writers = _host.GpuWriters(gpu_ctx.handle.value, category, 3, 64)
    try:
        for docid, vectors in DOCS.items():  # three add_documents calls, each ending in the build loop
            _host.write_vector_points(writers, {"manual_vectors": [(docid, None, [float(x) for x in v]) for v in vectors]})
            _host.build_vector_stores(writers)
        _check_emission(_search(gpu_ctx, category, "manual", [0, 1, 2]), GOLDEN["emission"])
        _check_emission(_search(gpu_ctx, category, "manual", [1, 2]), GOLDEN["emission_filtered_1_2"])
        # document 0 loses [0,1,2]: its best vector is now [3,4,5] in store 1 (similarity 1 - 0.057281)
        _host.write_vector_points(writers, {"manual_vectors": [(0, [0.0, 1.0, 2.0], None)]})
        got = _search(gpu_ctx, category, "manual", [0, 1, 2])
        assert got["documents_ids"] == [0, 1, 2]  # not sealed yet: searches see the last build
        assert _host.build_vector_stores(writers) == 3
        got = _search(gpu_ctx, category, "manual", [0, 1, 2])
        assert got["documents_ids"] == [0, 1, 2]
        assert got["document_scores"][0][0]["similarity"] == pytest.approx(1 - 0.05728112757642695, abs=2e-6)
        # document 1 is replaced by a vector parallel to the query (distance 0): deletion + addition
        _host.write_vector_points(writers, {"manual_vectors": [(1, [6.0, 7.0, 8.0], [0.0, 2.0, 4.0])]})
        _host.build_vector_stores(writers)
        got = _search(gpu_ctx, category, "manual", [0, 1, 2])
        assert got["documents_ids"] == [1, 0, 2]
        assert got["document_scores"][0][0]["similarity"] == pytest.approx(1.0, abs=2e-6)
        # document 2 is deleted: remove_vectors walks the packed writers
        _host.write_vector_points(writers, {"remove_vectors": [2]})
        _host.build_vector_stores(writers)
        got = _search(gpu_ctx, category, "manual", [0, 1, 2])
        # (store 2 is now empty: the build loop stops there, and the emptied store stops answering)
        assert got["documents_ids"][:2] == [1, 0]
        assert len(got["documents_ids"]) == 2 or got["document_scores"][2][0]["similarity"] is None
        # a vector of the wrong dimension is the reference's user error
        with pytest.raises(_host.HostError) as e:
            _host.write_vector_points(writers, {"manual_vectors": [(3, None, [1.0, 2.0])]})
        assert e.value.args[1] == "InvalidVectorDimensions" and e.value.args[2:] == (3, 2)
        # docids beyond the first capacity make the store grow
        _host.write_vector_points(writers, {"manual_vectors": [(1000, None, [0.0, 1.0, 2.0])]})
        _host.build_vector_stores(writers)
        ctx = _host.GpuContext(gpu_ctx.handle.value, {"manual": category}, [0, 1, 1000])
        got = _host.execute_vector_search(ctx, GOLDEN["query"], [0, 1, 1000], 0, 10, "manual", None)
        assert got["documents_ids"] == [1, 1000, 0]
    finally:
        del writers
        _destroy_stores(gpu_ctx, category)


class _ListWriter:
    """A plain in-memory writer (the same fake tests/test_write_path.py drives the mirror with)."""

    def __init__(self):
        self.items, self.built = {}, {}

    def del_item(self, d):
        return self.items.pop(d, None) is not None

    def add_item(self, d, v):
        self.items[d] = [float(x) for x in v]

    def item_vector(self, d):
        return self.items.get(d)

    def contains_item(self, d):
        return d in self.items

    def is_empty(self):
        return not self.items

    def build(self):
        self.built = dict(self.items)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_random_indexing_runs_leave_the_stores_a_plain_writer_would(gpu_ctx, seed):
    """The mirror's writers over GPU stores buffer a chunk's changes and hand them over in one upload
    (and must flush whenever the chunk reads its own writes back: typed_chunk.rs:618-707 deletes an
    item, looks vectors up by value and re-adds them inside one transaction). Random chunks of
    removals, generated embeddings and manual diffs go through write_vector_points twice -- over plain
    in-memory writers and over the GPU writers -- and after every build the sealed stores must hold
    exactly what the plain writers built, vector for vector."""
    from search_engine_amd import _host
    rng = np.random.default_rng(4000 + seed)
    category, dim, n_docs = 20 + seed, 4, 150
    pool = [[float(x) for x in rng.integers(-3, 4, dim)] for _ in range(40)]  # few distinct vectors: deletions by value hit
    pool = [v for v in pool if any(v)]
    plain = {}

    def plain_writers(k):
        return plain.setdefault(k, _ListWriter())

    gpu = _host.GpuWriters(gpu_ctx.handle.value, category, dim, 64)  # 64 slots: growth on the way
    try:
        for chunk in range(30):
            points = {"remove_vectors": [], "embeddings": [], "manual_vectors": []}
            kind = rng.choice(["manual", "manual", "embeddings", "remove", "mixed"])
            if kind in ("remove", "mixed"):
                points["remove_vectors"] = sorted({int(d) for d in rng.integers(0, n_docs, int(rng.integers(1, 12)))})
            if kind in ("embeddings", "mixed"):
                for d in sorted({int(d) for d in rng.integers(0, n_docs, int(rng.integers(1, 25)))}):
                    vecs = [pool[int(i)] for i in rng.integers(0, len(pool), int(rng.integers(1, 4)))]
                    points["embeddings"].append((d, vecs))
            if kind in ("manual", "mixed"):
                for _ in range(int(rng.integers(1, 30))):
                    d = int(rng.integers(0, n_docs))
                    have = [w.items[d] for _, w in sorted(plain.items()) if d in w.items]
                    deletion = addition = None
                    r = rng.random()
                    if have and r < 0.5:  # delete one of its vectors (sometimes one it does not have)
                        deletion = have[int(rng.integers(0, len(have)))] if rng.random() < 0.8 else pool[0]
                    if r > 0.3 and len(have) < 6:
                        addition = pool[int(rng.integers(0, len(pool)))]
                    if deletion is None and addition is None:
                        addition = pool[int(rng.integers(0, len(pool)))]
                    points["manual_vectors"].append((d, deletion, addition))
            # generated embeddings replace a document's vectors: the reference deletes them first in the same chunk
            if points["embeddings"]:
                points["remove_vectors"] = sorted(set(points["remove_vectors"]) | {d for d, _ in points["embeddings"]})
            _host.write_vector_points(plain_writers, points)
            _host.write_vector_points(gpu, points)
            n_plain = _host.build_vector_stores(plain_writers)
            n_gpu = _host.build_vector_stores(gpu)
            assert n_plain == n_gpu, (chunk, kind)
            # the one deviation from the reference's build loop (index_documents/mod.rs:559-561 leaves an
            # emptied writer unbuilt): a store must not go on answering with removed items, so the
            # emptied stores behind the last built one are sealed empty
            for k, w in plain.items():
                if k >= n_plain and not w.items:
                    w.built = {}
            for k, w in sorted(plain.items()):
                if not w.built and k >= n_gpu:
                    continue  # never built: no store opens
                try:
                    store = gpu_ctx.open_store((category << 8) | k)
                except Exception:  # noqa: BLE001
                    assert not w.built, (chunk, k)
                    continue
                assert len(store) == len(w.built), (chunk, kind, k)
                ids, _ = store.search(np.ones(dim, np.float32), 1024)
                assert sorted(ids.tolist()) == sorted(w.built), (chunk, kind, k)
                if k < n_gpu:  # built in this run: the writer's view is the sealed one
                    for d, v in w.built.items():
                        assert store.item_vector(d).tolist() == v, (chunk, kind, k, d)
    finally:
        del gpu
        _destroy_stores(gpu_ctx, category, n=12)
