"""Pure-Python restatement of the reference's host logic around the vector scan.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py). Each function cites the
reference lines it follows. Floats that the reference computes in f32 are
computed with numpy.float32, one operation at a time, in the reference's order.

Score details are dicts:
  {"kind": "words"|"typo"|"proximity"|"exactness"|"attribute", "value": a, "max": b}
  {"kind": "sort", "field": str, "ascending": bool, "value": None | float | str}
  {"kind": "vector", "similarity": None | float}
Search results are dicts {"documents_ids": [...], "document_scores": [[...]], "candidates": [...]}.
"""
import math

import numpy as np

f32 = np.float32
F32_EPSILON = float(np.finfo(np.float32).eps)
F64_EPSILON = float(np.finfo(np.float64).eps)


# ---- DistributionShift: src/vector/mod.rs:395-433 ----------------------------------------


def distribution_shift(mean, sigma, score):
    """DistributionShift::new(mean, sigma).map(|d| d.shift(score)); None if sigma <= 0."""
    if f32(sigma) <= f32(0.0):
        return None
    current_mean, current_sigma, score = f32(mean), f32(sigma), f32(score)
    target_mean, target_sigma = f32(0.5), f32(0.4)
    factor = f32(target_sigma / current_sigma)
    offset = f32(target_mean - f32(factor * current_mean))
    score = f32(f32(factor * score) + offset)
    if score <= f32(0.0):
        score = f32(F32_EPSILON)
    if score > f32(1.0):
        score = f32(1.0)
    return float(score)


# ---- Rank / ScoreDetails: src/score_details.rs --------------------------------------------

_U32 = 0xFFFFFFFF


def rank_of(d):
    """ScoreDetails::rank (score_details.rs:36-46) with the per-kind formulas (:183-271)."""
    k = d["kind"]
    if k in ("words", "exactness", "attribute"):
        return (d["value"] + 1, d["max"] + 1)
    if k in ("typo", "proximity"):
        top = d["max"] + 1
        return (max(top - d["value"], 0), top)
    return None


def rank_merge(outer, inner):
    """Rank::merge (score_details.rs:308-317); u32 arithmetic."""
    r, m = outer
    r = max(r - 1, 0)
    r = (r * inner[1]) & _U32
    m = (m * inner[1]) & _U32
    r = (r + inner[0]) & _U32
    return (r, m)


def rank_local_score(rank):
    return rank[0] / rank[1]


def rank_global_score(ranks):
    """Rank::global_score (score_details.rs:300-306)."""
    rank = (1, 1)
    for inner in ranks:
        rank = rank_merge(rank, inner)
    return rank_local_score(rank)


def score_values(details):
    """ScoreDetails::score_values (score_details.rs:59-89): adjacent ranks coalesce."""
    out, pending = [], None
    for d in details:
        r = rank_of(d)
        if r is not None:
            pending = r if pending is None else rank_merge(pending, r)
            continue
        if pending is not None:
            out.append(("score", rank_local_score(pending)))
            pending = None
        if d["kind"] == "sort":
            out.append(("sort", d))
        else:
            s = d["similarity"]
            out.append(("score", float(f32(s)) if s is not None else 0.0))
    if pending is not None:
        out.append(("score", rank_local_score(pending)))
    return out


def global_score(details):
    """ScoreDetails::global_score (score_details.rs:48-57)."""
    for kind, v in score_values(details):
        if kind == "score":
            return v
    return 1.0


def sort_partial_cmp(a, b):
    """Sort::partial_cmp (score_details.rs:328-360): -1/0/1 or None."""
    if a["field"] != b["field"] or a["ascending"] != b["ascending"]:
        return None
    l, r = a["value"], b["value"]
    if l is None and r is None:
        return 0
    if l is None:
        return -1
    if r is None:
        return 1
    ln, rn = isinstance(l, (int, float)), isinstance(r, (int, float))
    if ln and isinstance(r, str):
        return 1
    if isinstance(l, str) and rn:
        return -1
    if ln and rn:
        if math.isnan(l) or math.isnan(r):
            return None
        order = (l > r) - (l < r)
        return -order if a["ascending"] else order
    if isinstance(l, str) and isinstance(r, str):
        lb, rb = l.encode(), r.encode()  # Rust compares strings bytewise
        order = (lb > rb) - (lb < rb)
        return -order if a["ascending"] else order
    return None


def to_json_map(details):
    """ScoreDetails::to_json_map (score_details.rs:91-171) as a Python dict in insertion order."""
    out, order = {}, 0
    names = {"words": ("words", "matchingWords", "maxMatchingWords"),
             "typo": ("typo", "typoCount", "maxTypoCount"),
             "proximity": ("proximity", "proximity", "maxProximity"),
             "attribute": ("attribute", "attribute", "maxAttribute"),
             "exactness": ("exactness", "exactWords", "maxExactWords")}
    for d in details:
        k = d["kind"]
        if k in names:
            key, a, b = names[k]
            out[key] = {"order": order, a: d["value"], b: d["max"],
                        "score": rank_local_score(rank_of(d))}
        elif k == "sort":
            out[f"{d['field']}:{'asc' if d['ascending'] else 'desc'}"] = {"order": order,
                                                                         "value": d["value"]}
        else:
            s = d["similarity"]
            out["vectorSort"] = {"order": order, "similarity": None if s is None else float(f32(s))}
        order += 1
    return out


# ---- VectorSort: src/search/ranking/vector_sort.rs -------------------------------------------


def _ordered_float_key(d):
    return (1, 0.0) if math.isnan(d) else (0, float(d))


class VectorSort:
    """readers: list of callables nns(target, count, candidate_set) -> [(docid, distance)],
    store k at position k (the loop over ctx.vector_reader stops at the first missing store,
    vector_sort.rs:55-66). Sets are Python sets of docids."""

    def __init__(self, readers, target, vector_candidates, limit, distribution=None):
        self.readers = readers
        self.target = target
        self.candidates = set()
        self.vector_candidates = set(vector_candidates)
        self.cached = iter(())
        self.cached_len = 0
        self.limit = limit
        self.distribution = distribution

    def _fill_buffer(self, vector_candidates):  # vector_sort.rs:50-82
        results = []
        for store, nns in enumerate(self.readers):
            results.extend((docid, dist, store)
                           for docid, dist in nns(self.target, self.limit, vector_candidates))
        # the reference sorts by distance only (unstable); equal distances are put in the
        # canonical (distance, docid, store) order
        results.sort(key=lambda r: (_ordered_float_key(r[1]), r[0], r[2]))
        self.cached = iter([(r[0], r[1]) for r in results])
        self.cached_len = len(results)

    def start_iteration(self, candidates):  # vector_sort.rs:87-93
        self.candidates = set(candidates)
        self._fill_buffer(self.vector_candidates & self.candidates)

    def next_bucket(self):  # vector_sort.rs:95-144
        while True:
            if not self.candidates:
                return None
            vector_candidates = self.vector_candidates & self.candidates
            if not vector_candidates:
                self.candidates = set()
                return {"score": {"kind": "vector", "similarity": None}, "candidates": []}
            for docid, distance in self.cached:
                if docid in vector_candidates:
                    score = float(f32(1.0) - f32(distance))
                    if self.distribution is not None:
                        shifted = distribution_shift(self.distribution[0], self.distribution[1], score)
                        score = shifted if shifted is not None else score
                    self.candidates.discard(docid)
                    return {"score": {"kind": "vector", "similarity": score}, "candidates": [docid]}
            self._fill_buffer(vector_candidates)
            if self.cached_len == 0:
                self.candidates = set()
                return {"score": {"kind": "vector", "similarity": None}, "candidates": []}


# ---- bucket_sort: src/search/ranking/bucket_sort.rs:17-84 ---------------------------------------


def bucket_sort(ranking_rules, limit, skip, candidates):
    candidates = set(candidates)
    if not ranking_rules:
        ids = sorted(candidates)
        return {"documents_ids": ids, "document_scores": [[] for _ in ids], "candidates": ids}
    state = {"limit": limit, "skip": skip}
    visited, score, output = set(), [], []

    def recursive_sort(current_rule):
        while True:
            bucket = ranking_rules[current_rule].next_bucket()
            if bucket is None:
                break
            if state["limit"] == 0:
                return
            docids = set(bucket["candidates"]) - visited
            n = len(docids)
            if n == 0:
                continue
            if state["skip"] > 0 and n < state["skip"]:
                visited.update(docids)
                state["skip"] -= n
                continue
            score.append(bucket["score"])
            if current_rule == len(ranking_rules) - 1:
                visited.update(docids)
                for d in sorted(docids)[state["skip"]: state["skip"] + state["limit"]]:
                    output.append((d, list(score)))
                state["limit"] = max(state["limit"] - (0 if n < state["skip"] else n - state["skip"]), 0)
                state["skip"] = state["skip"] - n if n < state["skip"] else 0
                score.pop()
            else:
                ranking_rules[current_rule + 1].start_iteration(sorted(docids))
                recursive_sort(current_rule + 1)
        if score:
            score.pop()

    ranking_rules[0].start_iteration(sorted(candidates))
    recursive_sort(0)
    ids = [d for d, _ in output]
    return {"documents_ids": ids, "document_scores": [s for _, s in output], "candidates": sorted(set(ids))}


def execute_vector_search(readers, all_docids, vector, universe, skip, limit, distribution=None,
                          extra_rules=()):
    """execute_vector_search (search.rs:98-135) + get_ranking_rules_for_vector (ranking_rule.rs:74-132)."""
    rules = [VectorSort(readers, vector, all_docids, skip + limit, distribution)] + list(extra_rules)
    return bucket_sort(rules, limit, skip, universe)


# ---- hybrid: src/search/hybrid.rs -----------------------------------------------------------------


def compare_scores(left, left_ratio, right, right_ratio):
    """compare_scores (hybrid.rs:19-56): -1 Less, 0 Equal, 1 Greater."""
    l, r = score_values(left), score_values(right)
    lr, rr = float(f32(left_ratio)), float(f32(right_ratio))
    i = 0
    while True:
        lh, rh = i < len(l), i < len(r)
        if not lh and not rh:
            return 0
        if not lh:
            return -1
        if not rh:
            return 1
        (lk, lv), (rk, rv) = l[i], r[i]
        i += 1
        if lk == "score" and rk == "score":
            a, b = lv * lr, rv * rr
            if abs(a - b) <= F64_EPSILON:
                continue
            return -1 if a < b else 1
        if lk == "sort" and rk == "sort":
            c = sort_partial_cmp(lv, rv)
            if c is None:
                raise ValueError("incomparable sort values (the reference unwraps)")
            if c == 0:
                continue
            return c
        if lk == "score":
            return -1 if lv == 0.0 else 1
        return 1 if rv == 0.0 else -1


def results_good_enough(keyword_results, semantic_ratio, limit, offset):
    """Search::results_good_enough (hybrid.rs:201-219)."""
    good_enough_score = 0.45
    if len(keyword_results["document_scores"]) < limit + offset:
        return False
    keep = float(f32(1.0) - f32(semantic_ratio))
    for score in keyword_results["document_scores"]:
        if global_score(score) * keep < good_enough_score:
            return False
    return True


def merge(vector_results, vector_ratio, keyword_results, keyword_ratio, from_, length):
    """ScoreWithRatioResult::merge (hybrid.rs:74-135)."""
    v = list(zip(vector_results["documents_ids"], vector_results["document_scores"]))
    k = list(zip(keyword_results["documents_ids"], keyword_results["document_scores"]))
    merged, vi, ki = [], 0, 0
    while vi < len(v) or ki < len(k):
        if vi < len(v) and ki < len(k):
            take_v = compare_scores(v[vi][1], vector_ratio, k[ki][1], keyword_ratio) >= 0
        else:
            take_v = vi < len(v)
        if take_v:
            merged.append((v[vi], "semantic"))
            vi += 1
        else:
            merged.append((k[ki], "keyword"))
            ki += 1
    seen, filtered = set(), []
    for (docid, scores), source in merged:
        if docid not in seen:
            seen.add(docid)
            filtered.append((docid, scores, source))
    page = filtered[from_: from_ + length]
    return ({"documents_ids": [d for d, _, _ in page],
             "document_scores": [s for _, s, _ in page],
             "candidates": sorted(set(vector_results["candidates"]) | set(keyword_results["candidates"]))},
            sum(1 for _, _, s in page if s == "semantic"))


def execute_hybrid(query, has_semantic, vector, offset, limit, keyword_search, vector_search,
                   embed_one, semantic_ratio):
    """Search::execute_hybrid (hybrid.rs:138-199). keyword_search() / vector_search(v) run the
    search with offset 0 and limit = limit + offset; embed_one(query) returns None on failure."""
    keyword_results = keyword_search()
    if results_good_enough(keyword_results, semantic_ratio, limit, offset):
        return keyword_results, 0
    if query is None:
        return keyword_results, 0
    if not has_semantic:
        return keyword_results, 0
    if vector is None:
        vector = embed_one(query) if embed_one is not None else None
        if vector is None:
            return keyword_results, 0
    vector_results = vector_search(vector)
    merged, count = merge(vector_results, semantic_ratio, keyword_results,
                          float(f32(1.0) - f32(semantic_ratio)), offset, limit)
    assert len(merged["documents_ids"]) <= limit
    return merged, count
