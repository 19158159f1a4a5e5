"""The vector write path of the reference restated above the C ABI, on the CPU with an in-memory
fake of arroy::Writer (as the reference's TestContext fakes the search seam):
  write_vector_points   src/update/index_documents/typed_chunk.rs:562-710
  build_vector_stores   src/update/index_documents/mod.rs:554-563
  embedder category ids src/update/settings.rs:775-785
and arroy's key / leaf layout as vs_ingest_arroy_items reads it (host-only codec calls)."""
import struct

import numpy as np
import pytest

from search_engine_amd import _host, abi


class FakeWriter:
    def __init__(self, log, k):
        self.items, self.built, self.log, self.k = {}, None, log, k

    def del_item(self, d):
        self.log.append(("del", self.k, d))
        return self.items.pop(d, None) is not None

    def add_item(self, d, v):
        self.log.append(("add", self.k, d))
        self.items[d] = list(v)

    def item_vector(self, d):
        return self.items.get(d)

    def contains_item(self, d):
        return d in self.items

    def is_empty(self):
        return not self.items

    def build(self):
        self.built = dict(self.items)


class FakeWriters:
    def __init__(self):
        self.log, self.w = [], {}

    def __call__(self, k):
        return self.w.setdefault(k, FakeWriter(self.log, k))

    def layout(self):
        return {k: dict(w.items) for k, w in sorted(self.w.items()) if w.items}


def _scenario():
    """src/update/index_documents/mod.rs:2599-2609: documents with 2, 1 and 3 user-provided vectors."""
    w = FakeWriters()
    docs = {0: [[0, 1, 2], [3, 4, 5]], 1: [[6, 7, 8]], 2: [[9, 10, 11], [12, 13, 14], [15, 16, 17]]}
    for docid, vectors in docs.items():  # one add_documents call each; vectors arrive as manual additions
        _host.write_vector_points(w, {"manual_vectors": [(docid, None, [float(x) for x in v]) for v in vectors]})
        _host.build_vector_stores(w)
    return w


def test_vectors_are_packed_in_the_first_writers():
    w = _scenario()
    assert w.layout() == {0: {0: [0, 1, 2], 1: [6, 7, 8], 2: [9, 10, 11]},
                          1: {0: [3, 4, 5], 2: [12, 13, 14]},
                          2: {2: [15, 16, 17]}}
    assert w.w[2].built == {2: [15, 16, 17]}  # the build loop reached the third store and stopped there
    assert 3 not in w.w or w.w[3].built is None


def test_deleting_a_vector_follows_the_reference_including_its_hole():
    """typed_chunk.rs:679-693 means to move the document's last vector into the writer that lost one,
    but its search for "the last writer with a vector" starts AT the writer it has just deleted from
    (`.skip(deleted_index)`, :681) and stops at the first writer without the item (:682-684) — that
    very writer. Nothing is moved; the hole stays. Restated as it is, not as it was meant."""
    w = _scenario()
    _host.write_vector_points(w, {"manual_vectors": [(2, [9.0, 10.0, 11.0], None)]})
    assert w.layout() == {0: {0: [0, 1, 2], 1: [6, 7, 8]},
                          1: {0: [3, 4, 5], 2: [12, 13, 14]},
                          2: {2: [15, 16, 17]}}
    # the hole is filled by the document's next addition (first writer without the item, :700-705)
    _host.write_vector_points(w, {"manual_vectors": [(2, None, [7.0, 7.0, 7.0])]})
    assert w.layout()[0][2] == [7, 7, 7]
    # replace: deletion and addition in one diff entry — the new vector takes the writer just freed
    _host.write_vector_points(w, {"manual_vectors": [(0, [3.0, 4.0, 5.0], [1.0, 1.0, 1.0])]})
    assert w.layout()[1] == {0: [1, 1, 1], 2: [12, 13, 14]}
    # a deletion that matches nothing changes nothing
    before = w.layout()
    _host.write_vector_points(w, {"manual_vectors": [(1, [0.0, 0.0, 0.0], None)]})
    assert w.layout() == before
    # a vector stored twice is deleted from every writer that holds it; the last match is the one
    # whose writer would have been refilled
    _host.write_vector_points(w, {"manual_vectors": [(5, None, [2.0, 2.0, 2.0]), (5, None, [2.0, 2.0, 2.0]),
                                                     (5, None, [3.0, 3.0, 3.0])]})
    _host.write_vector_points(w, {"manual_vectors": [(5, [2.0, 2.0, 2.0], None)]})
    assert [k for k, items in w.layout().items() if 5 in items] == [2]


def test_remove_vectors_stops_at_the_first_writer_without_the_item():
    w = _scenario()
    w.log.clear()
    _host.write_vector_points(w, {"remove_vectors": [1, 2]})
    assert w.layout() == {0: {0: [0, 1, 2]}, 1: {0: [3, 4, 5]}}
    # document 1 had one vector: writer 0 deleted, writer 1 asked and missed, no further writer touched
    assert [e for e in w.log if e[2] == 1] == [("del", 0, 1), ("del", 1, 1)]
    assert [e for e in w.log if e[2] == 2] == [("del", 0, 2), ("del", 1, 2), ("del", 2, 2), ("del", 3, 2)]
    # the build loop stops at the first empty writer: store 2 is not rebuilt
    assert _host.build_vector_stores(w) == 2


def test_generated_embeddings_go_to_writers_in_order_and_overflow_is_an_error():
    w = FakeWriters()
    _host.write_vector_points(w, {"embeddings": [(7, [[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]])]})
    assert w.layout() == {0: {7: [1, 0]}, 1: {7: [0, 1]}, 2: {7: [1, 1]}}
    with pytest.raises(_host.HostError) as e:
        _host.write_vector_points(w, {"embeddings": [(8, [[0.0, 0.0]] * 256)]})
    assert e.value.args[1] == "TooManyVectors"
    _host.write_vector_points(w, {"embeddings": [(9, [[0.5, 0.5]] * 255)]})  # 255 is the limit
    assert len(w.layout()) == 255


def test_embedder_category_ids_follow_name_order():
    assert _host.assign_embedder_category_ids(["manual", "default", "bge", "Zeta"]) == {
        "Zeta": 0, "bge": 1, "default": 2, "manual": 3}
    # removing an embedder renumbers the rest (the database is cleared and rewritten)
    assert _host.assign_embedder_category_ids(["manual", "default"]) == {"default": 0, "manual": 1}
    assert len(_host.assign_embedder_category_ids([f"e{i:03d}" for i in range(256)])) == 256
    with pytest.raises(_host.HostError) as e:
        _host.assign_embedder_category_ids([f"e{i:03d}" for i in range(257)])
    assert e.value.args[1] == "TooManyEmbedders"


def test_arroy_key_and_leaf_layout():
    key, leaf = abi.arroy_encode_item(0x0102, 0x0A0B0C0D, [3.0, 4.0])
    assert key == bytes([0x01, 0x02, 0x00, 0x0A, 0x0B, 0x0C, 0x0D])   # index BE | mode item | item BE
    assert leaf == bytes([0]) + struct.pack("<f", 5.0) + struct.pack("<2f", 3.0, 4.0)  # tag | norm | vector
    assert abi.arroy_parse_key(key) == (0x0102, 0, 0x0A0B0C0D)
    assert abi.arroy_parse_key(bytes([0, 5, 2, 0, 0, 0, 0])) == (5, 2, 0)  # a metadata key
    assert abi.arroy_parse_key(bytes([0, 0, 3, 0, 0, 0, 0])) == (0, 3, 0)  # a mode of another arroy version: handed back
    for bad in (b"", bytes(6), bytes(8)):
        with pytest.raises(abi.VsError) as e:
            abi.arroy_parse_key(bad)
        assert e.value.code == abi.VS_E_FORMAT
