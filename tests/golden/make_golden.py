#!/usr/bin/env python3
"""Writes the fixtures under tests/golden/.

The reference cannot be built or run here (Rust, nightly features, out-of-tree
path dependencies) and its tests hold no expected output for the vector path,
so these vectors do not come from the reference. They are
  appendix_b.json   hand-derived known answers for the scenario of the reference's
                    commented-out test_multiple_vectors
                    (src/update/index_documents/mod.rs:2567-2623): exact rational
                    arithmetic evaluated in float64 below, no oracle code involved;
  shift.json        DistributionShift::shift (src/vector/mod.rs:405-432) evaluated
                    with numpy.float32 one operation at a time;
  scan_small.json   a small seeded scan case with float64 distances from plain numpy;
  arroy_leaves.json the (key, value) pairs a cursor over the reference's `vector-arroy`
                    database (src/index.rs:152-155, written by
                    src/update/index_documents/typed_chunk.rs:601-606) would yield for that
                    scenario, packed HERE with struct.pack from the layout arroy 0.2.0
                    publishes (the crate is not in the reference tree: restated, not
                    verified against it) -- bytes the library's own encoder never touched.
Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import math
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def angular(p, q):
    pq = sum(a * b for a, b in zip(p, q))
    pn = math.sqrt(sum(a * a for a in p))
    qn = math.sqrt(sum(a * a for a in q))
    return (1.0 - pq / (pn * qn)) / 2.0 if pn * qn != 0 else 0.0


def appendix_b():
    q = [0.0, 1.0, 2.0]
    docs = {0: [[0, 1, 2], [3, 4, 5]], 1: [[6, 7, 8]], 2: [[9, 10, 11], [12, 13, 14], [15, 16, 17]]}
    stores = {}
    for doc, vecs in docs.items():
        for k, v in enumerate(vecs):
            stores.setdefault(k, []).append({"docid": doc, "vector": v, "distance": angular(v, q)})
    # emission: first occurrence of each doc in ascending distance
    flat = sorted(((e["distance"], e["docid"], k) for k, es in stores.items() for e in es))
    seen, emission = set(), []
    for d, doc, _ in flat:
        if doc not in seen:
            seen.add(doc)
            emission.append({"docid": doc, "similarity": 1.0 - d})
    return {"query": q, "stores": [stores[k] for k in sorted(stores)], "emission": emission,
            "emission_filtered_1_2": [e for e in emission if e["docid"] in (1, 2)]}


def shift_cases():
    f = np.float32

    def shift(mean, sigma, score):
        factor = f(f(0.4) / f(sigma))
        offset = f(f(0.5) - f(factor * f(mean)))
        s = f(f(factor * f(score)) + offset)
        if s <= f(0):
            s = f(np.finfo(np.float32).eps)
        if s > f(1):
            s = f(1)
        return float(s)

    cases = []
    for mean, sigma in [(0.85, 0.1), (0.9, 0.08), (0.7, 0.1), (0.75, 0.1), (0.5, 0.4), (0.0, 1.0), (1.0, 0.01)]:
        for score in [1.0, 0.921327, 0.911748, 0.85, 0.70, 0.5, 0.0, -0.25, 1.5]:
            cases.append({"mean": mean, "sigma": sigma, "score": score, "shifted": shift(mean, sigma, score)})
    return cases


def scan_small():
    rng = np.random.default_rng(20240117)
    n, dim = 257, 24
    vectors = rng.standard_normal((n, dim)).astype(np.float32)
    vectors[40] = 0.0            # zero-norm row: distance 0
    vectors[100] = vectors[7]    # exact tie: broken by docid
    query = rng.standard_normal(dim).astype(np.float32)
    mask_ids = sorted(rng.choice(n, 120, replace=False).tolist() + [7, 40, 100])
    v = vectors.astype(np.float64)
    qv = query.astype(np.float64)
    pn = np.sqrt((v * v).sum(1))
    qn = math.sqrt((qv * qv).sum())
    with np.errstate(invalid="ignore", divide="ignore"):
        d = np.where(pn * qn != 0, (1.0 - (v @ qv) / (pn * qn)) / 2.0, 0.0)
    cand = sorted(set(mask_ids))
    order = sorted(cand, key=lambda i: (np.float32(d[i]), i))
    return {"dim": dim, "vectors": vectors.tolist(), "query": query.tolist(), "candidates": cand,
            "distances_f64": d.tolist(), "top10_filtered": order[:10],
            "top10_unfiltered": sorted(range(n), key=lambda i: (np.float32(d[i]), i))[:10]}


def arroy_key(index, mode, item):
    """index u16 big-endian | mode u8 (0 item, 1 tree, 2 metadata) | item u32 big-endian"""
    return struct.pack(">HBI", index, mode, item)


def arroy_leaf(vector):
    """tag u8 = 0 | Angular header { norm f32 } | dim x f32, little endian"""
    norm = math.sqrt(sum(float(np.float32(x)) ** 2 for x in vector))
    return struct.pack("<Bf", 0, norm) + struct.pack(f"<{len(vector)}f", *vector)


def arroy_leaves(category=7):
    docs = {0: [[0, 1, 2], [3, 4, 5]], 1: [[6, 7, 8]], 2: [[9, 10, 11], [12, 13, 14], [15, 16, 17]]}
    pairs = []
    for doc, vecs in docs.items():
        for j, v in enumerate(vecs):  # the j-th vector of a document lives in index (category << 8) | j
            pairs.append((arroy_key((category << 8) | j, 0, doc), arroy_leaf(v)))
    # what else a cursor yields: a split-plane tree node, a descendants node, metadata (payloads arbitrary:
    # an exhaustive scan reads none of them), another embedder's item, a node mode of a later arroy
    pairs.append((arroy_key(category << 8, 1, 9), struct.pack("<B3fII", 2, 0.5, -0.25, 0.125, 3, 4)))
    pairs.append((arroy_key(category << 8, 1, 10), bytes([1]) + bytes.fromhex("3a300000010000000000020000000100")))
    pairs.append((arroy_key(category << 8, 2, 0), struct.pack("<I", 3) + b"angular" + bytes(9)))
    pairs.append((arroy_key(9 << 8, 0, 1), arroy_leaf([1, 1, 1])))
    pairs.append((arroy_key(category << 8, 3, 2), b"\x00"))
    pairs.sort()  # LMDB hands keys over in byte order
    return {"category": category, "dim": 3, "pairs": [[k.hex(), v.hex()] for k, v in pairs],
            "expect": {"n_items": 6, "n_other_nodes": 4, "n_unrouted": 1, "store_lens": [3, 2, 1]}}


if __name__ == "__main__":
    for name, obj in (("appendix_b", appendix_b()), ("shift", shift_cases()), ("scan_small", scan_small()),
                      ("arroy_leaves", arroy_leaves())):
        with open(os.path.join(HERE, name + ".json"), "w") as fh:
            json.dump(obj, fh, indent=1)
        print("wrote", name)
