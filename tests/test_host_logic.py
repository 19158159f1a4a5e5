"""The C++ host mirror (search_engine_amd/csrc/host_engine.cpp, reached through the
_host binding) against the pure-Python restatement of the reference's host logic
(oracle/host.py) and the committed known answers. No GPU: the scan is injected
as a fake Reader, the way the reference's TestContext mocks `trait Context`
(src/search/query_graph.rs:536-733)."""
import json
import os
import random

import numpy as np
import pytest

from oracle import host as ref
from oracle import scan as oracle_scan

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def host():
    from search_engine_amd import _host
    return _host


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


# ---- DistributionShift ----------------------------------------------------------------------


def test_distribution_shift_golden():
    for c in load("shift.json"):
        assert host().distribution_shift(c["mean"], c["sigma"], c["score"]) == c["shifted"], c
        assert ref.distribution_shift(c["mean"], c["sigma"], c["score"]) == c["shifted"], c


def test_distribution_shift_properties():
    h = host()
    assert h.distribution_shift(0.5, 0.0, 0.7) is None and h.distribution_shift(0.5, -1.0, 0.7) is None
    # bge defaults (hf.rs:199-202): 1.0 clamps to 1.0, 0.70 clamps to f32::EPSILON
    assert h.distribution_shift(0.85, 0.1, 1.0) == 1.0
    assert h.distribution_shift(0.85, 0.1, 0.70) == float(np.finfo(np.float32).eps)
    assert abs(h.distribution_shift(0.85, 0.1, 0.921327) - 0.78531) < 1e-4
    rng = np.random.default_rng(0)
    for _ in range(2000):
        m, s, x = float(rng.random()), float(rng.random()) + 1e-3, float(rng.normal(0.7, 0.4))
        got = h.distribution_shift(m, s, x)
        assert got == ref.distribution_shift(m, s, x) and 0.0 < got <= 1.0


# ---- score details ------------------------------------------------------------------------------


def random_details(rng, with_sort=True):
    out = []
    for _ in range(rng.randint(0, 6)):
        kind = rng.choice(["words", "typo", "proximity", "exactness", "attribute", "vector"]
                          + (["sort"] if with_sort else []))
        if kind == "vector":
            out.append({"kind": "vector", "similarity": rng.choice([None, rng.random()])})
        elif kind == "sort":
            out.append({"kind": "sort", "field": rng.choice(["price", "name"]), "ascending": rng.random() < 0.5,
                        "value": rng.choice([None, rng.randint(0, 5) * 1.5, rng.choice(["a", "b", "é"])])})
        else:
            mx = rng.randint(0, 9)
            out.append({"kind": kind, "value": rng.randint(0, mx), "max": mx})
    return out


def test_rank_and_score_values_match_oracle():
    h, rng = host(), random.Random(1)
    assert h.rank_merge((3, 4), (2, 5)) == ref.rank_merge((3, 4), (2, 5)) == (12, 20)
    assert h.rank_merge((0, 4), (2, 5)) == (2, 20)  # saturating_sub
    for _ in range(500):
        ranks = [(rng.randint(0, 6), rng.randint(1, 6)) for _ in range(rng.randint(0, 5))]
        assert h.rank_global_score(ranks) == ref.rank_global_score(ranks)
        d = random_details(rng)
        got = [(k, v if k == "score" else dict(v)) for k, v in h.score_values(d)]
        want = ref.score_values(d)
        assert len(got) == len(want)
        for (gk, gv), (wk, wv) in zip(got, want):
            assert gk == wk and (gv == wv if gk == "score" else gv["field"] == wv["field"])
        assert h.global_score(d) == ref.global_score(d)


def test_to_json_map_matches_oracle():
    h, rng = host(), random.Random(2)
    d = [{"kind": "vector", "similarity": 0.92132738}, {"kind": "words", "value": 2, "max": 3},
         {"kind": "typo", "value": 1, "max": 2}, {"kind": "sort", "field": "price", "ascending": True, "value": 12.5},
         {"kind": "vector", "similarity": None}]
    parsed = json.loads(h.to_json_map(d))
    assert list(parsed) == ["vectorSort", "words", "typo", "price:asc"]  # second vectorSort overwrites in place
    assert parsed["vectorSort"] == {"order": 4, "similarity": None}
    assert parsed["words"] == {"order": 1, "matchingWords": 2, "maxMatchingWords": 3, "score": 0.75}
    for _ in range(300):
        d = random_details(rng)
        got, want = json.loads(h.to_json_map(d)), ref.to_json_map(d)
        assert list(got) == list(want) and got == want


def test_sort_partial_cmp_matches_oracle():
    h = host()
    vals = [None, 1.0, 2.5, "a", "b", "é"]
    for fa in ("x", "y"):
        for asc_a in (True, False):
            for asc_b in (True, False):
                for va in vals:
                    for vb in vals:
                        a = {"kind": "sort", "field": fa, "ascending": asc_a, "value": va}
                        b = {"kind": "sort", "field": "x", "ascending": asc_b, "value": vb}
                        assert h.sort_partial_cmp(a, b) == ref.sort_partial_cmp(a, b)


# ---- hybrid -------------------------------------------------------------------------------------


def comparable_details(rng):
    """Details that never hit the reference's unwrap() on incomparable sorts."""
    out = [{"kind": "words", "value": rng.randint(0, 3), "max": 3}]
    if rng.random() < 0.5:
        out.append({"kind": "typo", "value": rng.randint(0, 2), "max": 2})
    if rng.random() < 0.3:
        out.append({"kind": "sort", "field": "price", "ascending": True, "value": float(rng.randint(0, 3))})
    if rng.random() < 0.5:
        out.append({"kind": "attribute", "value": rng.randint(0, 4), "max": 4})
    return out


def vector_details(rng):
    # similarities are f32 in the reference: keep the test values f32-representable
    return [{"kind": "vector", "similarity": rng.choice([None, float(np.float32(rng.random()))])}] + (
        [{"kind": "words", "value": 0, "max": 0}] if rng.random() < 0.5 else [])


def test_compare_scores_matches_oracle():
    h, rng = host(), random.Random(3)
    for _ in range(3000):
        l = rng.choice([comparable_details, vector_details])(rng)
        r = rng.choice([comparable_details, vector_details])(rng)
        lr, rr = rng.choice([0.0, 0.3, 0.5, 1.0]), rng.choice([0.0, 0.5, 0.7, 1.0])
        assert h.compare_scores(l, lr, r, rr) == ref.compare_scores(l, lr, r, rr)
    # equality within f64::EPSILON continues to the next value; shorter list is Less
    a = [{"kind": "vector", "similarity": 0.5}]
    assert h.compare_scores(a, 0.5, a, 0.5) == 0
    assert h.compare_scores(a, 0.5, a + [{"kind": "words", "value": 1, "max": 1}], 0.5) == -1
    # Score vs Sort: a zero score loses, a non-zero score wins (hybrid.rs:48-53)
    s = [{"kind": "sort", "field": "p", "ascending": True, "value": 1.0}]
    assert h.compare_scores([{"kind": "vector", "similarity": None}], 1.0, s, 1.0) == -1
    assert h.compare_scores(a, 1.0, s, 1.0) == 1 and h.compare_scores(s, 1.0, a, 1.0) == -1


def random_result(rng, n, make, id_pool):
    ids = rng.sample(id_pool, n)
    return {"documents_ids": ids, "document_scores": [make(rng) for _ in ids], "candidates": sorted(ids)}


def test_merge_matches_oracle():
    h, rng = host(), random.Random(4)
    for _ in range(400):
        pool = list(range(30))
        v = random_result(rng, rng.randint(0, 12), vector_details, pool)
        k = random_result(rng, rng.randint(0, 12), comparable_details, pool)
        ratio = rng.choice([0.0, 0.2, 0.5, 0.9, 1.0])
        frm, length = rng.randint(0, 5), rng.randint(0, 12)
        got, got_n = h.merge(v, ratio, k, 1.0 - ratio, frm, length)
        want, want_n = ref.merge(v, ratio, k, float(np.float32(1.0) - np.float32(ratio)), frm, length)
        assert got == want and got_n == want_n
        assert len(got["documents_ids"]) == len(set(got["documents_ids"])) <= length


def test_results_good_enough():
    h = host()
    full = {"documents_ids": [1, 2], "candidates": [1, 2],
            "document_scores": [[{"kind": "words", "value": 9, "max": 9}], [{"kind": "words", "value": 9, "max": 9}]]}
    assert h.results_good_enough(full, 0.5, 2, 0) and ref.results_good_enough(full, 0.5, 2, 0)
    assert not h.results_good_enough(full, 0.5, 2, 1)  # page not full
    weak = json.loads(json.dumps(full))
    weak["document_scores"][1] = [{"kind": "words", "value": 7, "max": 9}]  # 0.8 * 0.5 < 0.45
    assert not h.results_good_enough(weak, 0.5, 2, 0) and not ref.results_good_enough(weak, 0.5, 2, 0)
    assert h.results_good_enough(weak, 0.2, 2, 0)      # 0.8 * 0.8 >= 0.45


def test_execute_hybrid_branches():
    h = host()
    good = {"documents_ids": [1], "candidates": [1], "document_scores": [[{"kind": "words", "value": 3, "max": 3}]]}
    weak = {"documents_ids": [1], "candidates": [1], "document_scores": [[{"kind": "words", "value": 1, "max": 3}]]}
    vec = {"documents_ids": [2, 1], "candidates": [1, 2],
           "document_scores": [[{"kind": "vector", "similarity": 0.875}], [{"kind": "vector", "similarity": 0.75}]]}
    calls = []

    def run(query, has_semantic, vector, keyword, embed=None, limit=1, offset=0, ratio=0.5):
        args = (query, has_semantic, vector, offset, limit, lambda: keyword,
                lambda v: calls.append(list(v)) or vec, embed, ratio)
        got = h.execute_hybrid(*args)
        want = ref.execute_hybrid(*args)
        assert got == tuple(want) or list(got) == list(want)
        return got

    assert run("q", True, [1.0], good) == (good, 0) and not calls            # good enough: no vector search
    assert run(None, True, [1.0], weak) == (weak, 0)                         # placeholder query
    assert run("q", False, None, weak) == (weak, 0)                          # no embedder
    assert run("q", True, None, weak, embed=lambda q: None) == (weak, 0)     # embedding failed
    res, n = run("q", True, None, weak, embed=lambda q: [0.5, 0.5], limit=2)
    assert calls[-1] == [0.5, 0.5] and n == 2 and res["documents_ids"] == [2, 1]
    res, n = run("q", True, [1.0, 0.0], weak, limit=1, offset=1)
    assert res["documents_ids"] == [1] and n == 1


# ---- VectorSort / bucket_sort with a fake Context ---------------------------------------------------


class FakeReader:
    """arroy Reader stand-in backed by the CPU oracle."""

    def __init__(self, docids, vectors):
        self.docids = np.asarray(docids, np.uint32)
        self.vectors = np.asarray(vectors, np.float32)
        self.calls = 0

    def nns_by_vector(self, target, count, candidates):
        self.calls += 1
        mask = oracle_scan.bool_to_mask(np.isin(np.arange(int(self.docids.max()) + 1), list(candidates)))
        ids, dist = oracle_scan.nns_by_vector(self.vectors, np.asarray(target, np.float32), count,
                                              docids=self.docids, mask=mask)
        return list(zip(ids.tolist(), dist.tolist()))


class FakeContext:
    def __init__(self, stores, all_docids, embedders=None):
        self.stores = stores  # {index: FakeReader}
        self._all = list(all_docids)
        self.embedders = embedders or {"manual": 0}

    def all_docids(self):
        return self._all

    def embedder_category_id(self, name):
        return self.embedders.get(name)

    def vector_reader(self, index):
        return self.stores.get(index)


def appendix_b_context(category=0):
    g = load("appendix_b.json")
    stores = {}
    for k, entries in enumerate(g["stores"]):
        stores[(category << 8) | k] = FakeReader([e["docid"] for e in entries], [e["vector"] for e in entries])
    return g, FakeContext(stores, [0, 1, 2], {"manual": category})


def test_vector_sort_appendix_b_emission():
    g, ctx = appendix_b_context(category=3)
    vs = host().VectorSort(ctx, g["query"], [0, 1, 2], 10, "manual", None)
    vs.start_iteration([0, 1, 2])
    out = []
    while True:
        b = vs.next_bucket()
        if b is None:
            break
        out.append(b)
    assert [b["candidates"] for b in out] == [[e["docid"]] for e in g["emission"]]
    for b, e in zip(out, g["emission"]):
        assert abs(b["score"]["similarity"] - e["similarity"]) < 1e-6
    # with the bge distribution: 1.0 (clamped), ~0.78531, ~0.74699
    vs = host().VectorSort(ctx, g["query"], [0, 1, 2], 10, "manual", (0.85, 0.1))
    vs.start_iteration([1, 2])  # filter {1, 2}
    sims = []
    while (b := vs.next_bucket()) is not None:
        sims.append((b["candidates"], b["score"]["similarity"]))
    assert [c for c, _ in sims] == [[1], [2]]
    assert abs(sims[0][1] - 0.78531) < 1e-4 and abs(sims[1][1] - 0.74699) < 1e-4


def test_vector_sort_errors_and_missing_vectors():
    h = host()
    g, ctx = appendix_b_context()
    with pytest.raises(h.HostError) as e:
        h.VectorSort(ctx, g["query"], [0, 1, 2], 10, "nope", None)
    assert e.value.args[1] == "InvalidEmbedder"
    # documents 5 and 6 have no vector: after the vector hits, one empty bucket with
    # similarity None ends the iteration and they are never returned
    ctx = FakeContext(ctx.stores, [0, 1, 2, 5, 6])
    vs = h.VectorSort(ctx, g["query"], [0, 1, 2, 5, 6], 10, "manual", None)
    vs.start_iteration([0, 1, 2, 5, 6])
    buckets = []
    while (b := vs.next_bucket()) is not None:
        buckets.append(b)
    assert [b["candidates"] for b in buckets] == [[0], [1], [2], []]
    assert buckets[-1]["score"]["similarity"] is None


def make_random_world(rng, n_docs, dim, n_stores):
    """Docs own 1..n_stores vectors; vector j of a doc lives in store j (packed stores)."""
    counts = rng.integers(1, n_stores + 1, n_docs)
    stores, py_readers = {}, []
    for k in range(n_stores):
        docids = np.nonzero(counts > k)[0].astype(np.uint32)
        vectors = rng.standard_normal((docids.size, dim)).astype(np.float32)
        r = FakeReader(docids, vectors)
        stores[k] = r
        py_readers.append(lambda t, c, cand, r=r: r.nns_by_vector(t, c, cand))
    return stores, py_readers


class CountRule:
    """A downstream rule: splits its candidates into buckets by docid % modulus (best first)."""

    def __init__(self, modulus):
        self.m = modulus

    def start_iteration(self, candidates):
        self.cand = set(candidates)
        self.cur = 0

    def next_bucket(self):
        if self.cur >= self.m:
            return None
        b = sorted(d for d in self.cand if d % self.m == self.cur)
        self.cur += 1
        return {"score": {"kind": "words", "value": self.m - self.cur, "max": self.m - 1}, "candidates": b}


@pytest.mark.parametrize("seed", range(6))
def test_execute_vector_search_matches_oracle(seed):
    rng = np.random.default_rng(seed)
    n_docs, dim, n_stores = 60, 8, 3
    stores, py_readers = make_random_world(rng, n_docs, dim, n_stores)
    q = rng.standard_normal(dim).astype(np.float32).tolist()
    all_docids = list(range(n_docs))
    ctx = FakeContext(stores, all_docids)
    for universe in (all_docids, [d for d in all_docids if d % 3], [4, 9], []):
        for skip, limit in ((0, 10), (3, 5), (0, 100), (7, 1), (0, 0), (2, 58)):
            for dist in (None, (0.7, 0.1)):
                got = host().execute_vector_search(ctx, q, universe, skip, limit, "manual", dist)
                want = ref.execute_vector_search(py_readers, all_docids, q, universe, skip, limit, dist)
                assert got["documents_ids"] == want["documents_ids"], (universe[:5], skip, limit)
                assert got["candidates"] == want["candidates"]
                for gs, ws in zip(got["document_scores"], want["document_scores"]):
                    assert len(gs) == len(ws) == 1
                    assert gs[0]["similarity"] == pytest.approx(ws[0]["similarity"], abs=1e-7)
    # pagination: page (skip, limit) == slice of the first skip+limit results
    full = host().execute_vector_search(ctx, q, all_docids, 0, 20, "manual", None)["documents_ids"]
    page = host().execute_vector_search(ctx, q, all_docids, 5, 15, "manual", None)["documents_ids"]
    assert page == full[5:20]
    # with a downstream rule every one-doc bucket recurses into it
    got = host().execute_vector_search(ctx, q, all_docids, 2, 6, "manual", None, lambda: [CountRule(2)])
    want = ref.execute_vector_search(py_readers, all_docids, q, all_docids, 2, 6, None, [CountRule(2)])
    assert got["documents_ids"] == want["documents_ids"] == full[2:8]
    assert [len(s) for s in got["document_scores"]] == [2] * 6
    assert got["document_scores"][0][1]["kind"] == "words"


def test_bucket_sort_matches_oracle_on_fake_rules():
    rng = random.Random(9)
    for _ in range(100):
        universe = rng.sample(range(40), rng.randint(0, 40))
        mods = [rng.randint(1, 4) for _ in range(rng.randint(0, 3))]
        limit, skip = rng.randint(0, 15), rng.randint(0, 10)
        got = host().bucket_sort([CountRule(m) for m in mods], limit, skip, universe)
        want = ref.bucket_sort([CountRule(m) for m in mods], limit, skip, universe)
        assert got == want, (mods, limit, skip)


def test_docsets_beyond_the_small_array_limit():
    """More than 4096 documents: the host mirror's docid sets switch from a sorted array to a
    dense bitmap (as a roaring container does); results stay those of the oracle on either
    side of the switch and across it."""
    rng = np.random.default_rng(77)
    n_docs, dim = 6000, 4
    stores, py_readers = make_random_world(rng, n_docs, dim, 1)
    q = rng.standard_normal(dim).astype(np.float32).tolist()
    all_docids = list(range(n_docs))
    ctx = FakeContext(stores, all_docids)
    for universe in (all_docids, all_docids[::2], all_docids[:4096], all_docids[:4097], [5, 4500, 5999]):
        for skip, limit in ((0, 10), (4, 3), (0, 0)):
            got = host().execute_vector_search(ctx, q, universe, skip, limit, "manual", None)
            want = ref.execute_vector_search(py_readers, all_docids, q, universe, skip, limit, None)
            assert got["documents_ids"] == want["documents_ids"]
            assert got["candidates"] == want["candidates"]
    # `visited` grows through the limit while buckets of every size are subtracted from it
    universe = all_docids[::-1]
    for limit, skip in ((5000, 0), (4500, 1000), (10, 5990)):
        got = host().bucket_sort([CountRule(3), CountRule(2)], limit, skip, universe)
        want = ref.bucket_sort([CountRule(3), CountRule(2)], limit, skip, universe)
        assert got == want


def test_execute_vector_search_batch_equals_single_searches():
    """One universe, several vectors: result i is execute_vector_search of vector i (stores are
    asked once for all vectors; the fake readers take the default path, a loop)."""
    rng = np.random.default_rng(11)
    n_docs, dim, n_stores = 80, 6, 3
    stores, _ = make_random_world(rng, n_docs, dim, n_stores)
    all_docids = list(range(n_docs))
    ctx = FakeContext(stores, all_docids)
    vectors = rng.standard_normal((7, dim)).astype(np.float32).tolist()
    for universe in (all_docids, [d for d in all_docids if d % 4 != 1], [3], []):
        for skip, limit, dist in ((0, 10, None), (4, 6, (0.6, 0.2)), (0, 0, None), (0, 200, None)):
            got = host().execute_vector_search_batch(ctx, vectors, universe, skip, limit, "manual", dist)
            assert len(got) == len(vectors)
            for v, g in zip(vectors, got):
                assert g == host().execute_vector_search(ctx, v, universe, skip, limit, "manual", dist)
    assert host().execute_vector_search_batch(ctx, [], all_docids, 0, 5, "manual", None) == []
    with pytest.raises(host().HostError) as e:
        host().execute_vector_search_batch(ctx, vectors, all_docids, 0, 5, "nope", None)
    assert e.value.args[1] == "InvalidEmbedder"


def test_docset_algebra_matches_python_sets():
    """The three shapes a host-side docid set takes (sorted array up to 4096 documents, the range
    [0, n) of an index without holes, dense bitmap) against Python sets: every operation the path
    uses, on every pair of shapes, with the cached cardinality kept honest through modifications."""
    rng = random.Random(31)
    n = 9000

    def make(shape):
        if shape == "array":
            ids = rng.sample(range(n), rng.randint(0, 300))
            return ids, set(ids)
        if shape == "range":
            m = rng.randint(1, n)
            return m, set(range(m))
        if shape == "range-from-ids":  # every docid below m, handed over as a list: recognised as a range
            m = rng.randint(4097, n)
            return list(range(m)), set(range(m))
        if shape == "words":
            ids = rng.sample(range(n), rng.randint(4100, n))
            words = np.zeros(-(-n // 64), np.uint64)
            for d in ids:
                words[d >> 6] |= np.uint64(1) << np.uint64(d & 63)
            return words, set(ids)
        ids = rng.sample(range(n), rng.randint(4100, n - 1))
        if len(set(ids)) == max(ids) + 1:
            ids.remove(0)
        return ids, set(ids)

    expected_repr = {"array": "array", "range": "range", "range-from-ids": "range", "words": "bitmap",
                     "bitmap": "bitmap"}
    shapes = list(expected_repr)
    for sa in shapes:
        for sb in shapes:
            for _ in range(3):
                a, set_a = make(sa)
                b, set_b = make(sb)
                script, c, want_effects = [], set(set_a), []
                for _ in range(12):
                    op = rng.choice(["insert", "remove", "insert", "remove", "or", "sub"])
                    if op in ("insert", "remove"):
                        d = rng.choice([rng.randrange(n + 200), len(c), max(c) if c else 0])
                        script.append((op, d))
                        if op == "insert":
                            want_effects.append(d not in c)
                            c.add(d)
                        else:
                            want_effects.append(d in c)
                            c.discard(d)
                    else:
                        script.append((op,))
                        c = c | set_b if op == "or" else c - set_b
                    want_effects.append(len(c))
                probes = [0, 1, 63, 64, 4095, 4096, n - 1, n, n + 500] + [rng.randrange(n) for _ in range(20)]
                got = host().docset_probe(a, b, script, probes)
                assert got["a_repr"] == expected_repr[sa] and got["b_repr"] == expected_repr[sb]
                assert got["and"] == sorted(set_a & set_b)
                assert got["and_len"] == got["and_len_again"] == len(set_a & set_b)
                assert got["or"] == sorted(set_a | set_b)
                assert got["sub"] == sorted(set_a - set_b) and got["sub_len"] == len(set_a - set_b)
                assert got["a_after_ops"] == sorted(set_a)
                assert got["effects"] == want_effects, (sa, sb, script)
                assert got["c"] == sorted(c) and got["c_empty"] == (not c)
                assert got["contains"] == [(d in set_a, d in c) for d in probes]
                words = got["c_words"]
                assert sorted(64 * i + j for i, w in enumerate(words.tolist()) for j in range(64) if (w >> j) & 1) == sorted(c)
    # a range over a bitmap that it covers hands the bitmap back as it is
    words = np.zeros(100, np.uint64)
    words[3] = 5
    got = host().docset_probe(6400, words, [], [])
    assert got["and_repr"] == "bitmap" and got["and"] == [192, 194] and got["and_len"] == 2
    got = host().docset_probe(193, words, [], [])  # ... and cuts one that reaches beyond it
    assert got["and"] == [192]


def test_execute_vector_search_batch_with_a_universe_per_request():
    """Several requests in one call, each under the universe of its own filter: result i is
    execute_vector_search of vector i under universe i (fake readers: the default loop)."""
    rng = np.random.default_rng(12)
    n_docs, dim, n_stores = 90, 5, 2
    stores, _ = make_random_world(rng, n_docs, dim, n_stores)
    all_docids = list(range(n_docs))
    ctx = FakeContext(stores, all_docids)
    vectors = rng.standard_normal((6, dim)).astype(np.float32).tolist()
    shared = [d for d in all_docids if d % 3]
    universes = [all_docids, shared, [7], [], shared, [d for d in all_docids if d > 40]]
    for skip, limit, dist in ((0, 10, None), (2, 5, (0.6, 0.2)), (0, 0, None)):
        got = host().execute_vector_search_batch_universes(ctx, vectors, universes, skip, limit, "manual", dist)
        assert len(got) == len(vectors)
        for v, u, g in zip(vectors, universes, got):
            assert g == host().execute_vector_search(ctx, v, u, skip, limit, "manual", dist)
