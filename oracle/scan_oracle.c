/*
 * scan_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the exhaustive regime of the reference's vector search,
 * used only as the checker in tests/, in __graft_entry__.smoke() and as the
 * `cpu_baseline` leg of bench.py. Nothing under search_engine_amd/ links,
 * loads or calls it.
 *
 * What it restates
 *   VectorSort::fill_buffer      src/search/ranking/vector_sort.rs:50-82
 *     -> reader.nns_by_vector(txn, target, limit, None, Some(candidates))   :73-77
 * The arithmetic of that call lives in a third-party crate that is NOT in the
 * reference tree: `arroy = "0.2.0"` (reference Cargo.toml:78; no lockfile, no
 * vendored copy). Its published algorithm, restated here:
 *   - Angular leaf header: norm = sqrt(dot(v, v))        (distance/angular.rs)
 *   - built_distance(p, q): pn*qn != 0 ? (1 - dot(p,q)/(pn*qn)) / 2 : 0
 *   - nns_by_vector: when the candidate set is smaller than search_k the tree
 *     descent collects every candidate item, computes its distance and returns
 *     the `count` smallest ordered by the tuple (OrderedFloat(distance), item)
 *     (reader.rs). OrderedFloat orders NaN above +inf and all NaN equal.
 *
 * PARITY UNPINNED: the reference holds no test, fixture or golden vector for
 * this path (its only vector test is commented out, src/update/
 * index_documents/mod.rs:2567-2623, and asserted nothing but a length), and the
 * crate cannot be built here (no Rust toolchain). The restatement is pinned
 * only by hand-computed known answers for that commented-out scenario
 * (tests/golden/appendix_b.json) and by an independent numpy implementation.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* accumulation modes of the dot product */
enum { ACC_F32_SEQ = 0, ACC_F32_SIMD8 = 1, ACC_F64 = 2 };

/* scalar left-to-right f32 sum (built with -ffp-contract=off: no fused multiply-add) */
float oracle_dot_f32_seq(const float* a, const float* b, uint32_t n) {
    float s = 0.0f;
    for (uint32_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* eight running f32 sums as an 8-lane SIMD unit keeps them, combined pairwise
 * at the end (how arroy's dot product runs on an AVX machine) */
float oracle_dot_f32_simd8(const float* a, const float* b, uint32_t n) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int l = 0; l < 8; ++l) acc[l] += a[i + l] * b[i + l];
    float s = ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
    for (; i < n; ++i) s += a[i] * b[i];
    return s;
}

double oracle_dot_f64(const float* a, const float* b, uint32_t n) {
    double s = 0.0;
    for (uint32_t i = 0; i < n; ++i) s += (double)a[i] * (double)b[i];
    return s;
}

static float dot_mode(const float* a, const float* b, uint32_t n, int mode) {
    if (mode == ACC_F32_SIMD8) return oracle_dot_f32_simd8(a, b, n);
    if (mode == ACC_F64) return (float)oracle_dot_f64(a, b, n);
    return oracle_dot_f32_seq(a, b, n);
}

/* arroy Angular::new_header */
float oracle_norm(const float* v, uint32_t n, int mode) { return sqrtf(dot_mode(v, v, n, mode)); }

/* arroy Angular::built_distance */
float oracle_angular_distance(float pq, float pn, float qn) {
    const float pnqn = pn * qn;
    if (pnqn != 0.0f) {
        const float c = pq / pnqn;
        return (1.0f - c) / 2.0f;
    }
    return 0.0f;
}

/* the same formula carried in f64 end to end: the reference value tolerances are measured against */
double oracle_angular_distance_f64(const float* p, const float* q, uint32_t n) {
    const double pn = sqrt(oracle_dot_f64(p, p, n)), qn = sqrt(oracle_dot_f64(q, q, n));
    const double pnqn = pn * qn;
    if (pnqn != 0.0) return (1.0 - oracle_dot_f64(p, q, n) / pnqn) / 2.0;
    return 0.0;
}

typedef struct {
    float d;
    uint32_t id;
} hit_t;

/* (OrderedFloat(d), id): NaN above everything, all NaN equal, -0 == +0 */
static int hit_cmp(const void* x, const void* y) {
    const hit_t* a = (const hit_t*)x;
    const hit_t* b = (const hit_t*)y;
    const int an = isnan(a->d), bn = isnan(b->d);
    if (an != bn) return an ? 1 : -1;
    if (!an) {
        if (a->d < b->d) return -1;
        if (a->d > b->d) return 1;
    }
    return a->id < b->id ? -1 : a->id > b->id ? 1 : 0;
}

static int is_candidate(uint32_t docid, const uint64_t* mask, uint64_t n_words) {
    if (!mask) return 1;
    const uint64_t w = docid >> 6;
    return w < n_words && ((mask[w] >> (docid & 63)) & 1);
}

/* bounded max-heap of the `count` best hits under hit_cmp */
static void heap_sift_down(hit_t* h, uint32_t n, uint32_t i) {
    for (;;) {
        uint32_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && hit_cmp(&h[l], &h[m]) > 0) m = l;
        if (r < n && hit_cmp(&h[r], &h[m]) > 0) m = r;
        if (m == i) return;
        hit_t t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
}
static void heap_sift_up(hit_t* h, uint32_t i) {
    while (i) {
        const uint32_t p = (i - 1) / 2;
        if (hit_cmp(&h[i], &h[p]) <= 0) return;
        hit_t t = h[i];
        h[i] = h[p];
        h[p] = t;
        i = p;
    }
}

/*
 * Exhaustive nns_by_vector over one store.
 *   vectors  [n_items][dim] f32 row-major, docids[n_items] the item ids
 *   mask     dense candidate bitmask over docids (NULL = no filter)
 * Writes at most `count` (id, distance) pairs ascending by (distance, id) and
 * returns how many. `norms` may be NULL (computed with the same mode).
 */
uint32_t oracle_nns_by_vector(const float* vectors, const float* norms, const uint32_t* docids,
                              uint64_t n_items, uint32_t dim, const float* query, uint32_t count,
                              const uint64_t* mask, uint64_t n_mask_words, int mode,
                              uint32_t* out_ids, float* out_dist) {
    if (count == 0) return 0; /* arroy: count == 0 -> empty */
    hit_t* heap = (hit_t*)malloc((size_t)count * sizeof(hit_t));
    uint32_t n = 0;
    const float qn = oracle_norm(query, dim, mode);
    for (uint64_t i = 0; i < n_items; ++i) {
        const uint32_t id = docids ? docids[i] : (uint32_t)i;
        if (!is_candidate(id, mask, n_mask_words)) continue;
        const float* p = vectors + i * dim;
        const float pn = norms ? norms[i] : oracle_norm(p, dim, mode);
        hit_t h;
        h.d = oracle_angular_distance(dot_mode(p, query, dim, mode), pn, qn);
        h.id = id;
        if (n < count) {
            heap[n] = h;
            heap_sift_up(heap, n);
            ++n;
        } else if (hit_cmp(&h, &heap[0]) < 0) {
            heap[0] = h;
            heap_sift_down(heap, n, 0);
        }
    }
    qsort(heap, n, sizeof(hit_t), hit_cmp);
    for (uint32_t i = 0; i < n; ++i) {
        out_ids[i] = heap[i].id;
        out_dist[i] = heap[i].d;
    }
    free(heap);
    return n;
}

/* f64 distances of every candidate (for tolerance and near-tie analysis in tests) */
void oracle_all_distances_f64(const float* vectors, uint64_t n_items, uint32_t dim,
                              const float* query, double* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n_items; ++i)
        out[i] = oracle_angular_distance_f64(vectors + (uint64_t)i * dim, query, dim);
}

/* `n_q` independent queries, one per thread: the CPU throughput baseline
 * ("one query per core"). Returns the threads used. */
int oracle_nns_batch_mt(const float* vectors, const float* norms, const uint32_t* docids,
                        uint64_t n_items, uint32_t dim, const float* queries, uint32_t n_q,
                        uint32_t count, const uint64_t* mask, uint64_t n_mask_words, int mode,
                        int n_threads, uint32_t* out_ids, float* out_dist, uint32_t* out_n) {
    int used = 1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
    }
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int32_t q = 0; q < (int32_t)n_q; ++q)
        out_n[q] = oracle_nns_by_vector(vectors, norms, docids, n_items, dim,
                                        queries + (size_t)q * dim, count, mask, n_mask_words, mode,
                                        out_ids + (size_t)q * count, out_dist + (size_t)q * count);
    return used;
}

void oracle_norms(const float* vectors, uint64_t n_items, uint32_t dim, int mode, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n_items; ++i)
        out[i] = oracle_norm(vectors + (uint64_t)i * dim, dim, mode);
}

/*
 * Binary-quantised stores have no counterpart in the reference (it has one
 * metric, Angular: src/index.rs:155); this is the build's own definition:
 * distance = popcount(code XOR query) over `words` u64 per row, results
 * ascending by (distance, id).
 */
typedef struct {
    uint32_t d;
    uint32_t id;
} hhit_t;
static int hhit_cmp(const void* x, const void* y) {
    const hhit_t* a = (const hhit_t*)x;
    const hhit_t* b = (const hhit_t*)y;
    if (a->d != b->d) return a->d < b->d ? -1 : 1;
    return a->id < b->id ? -1 : a->id > b->id ? 1 : 0;
}
uint32_t oracle_hamming_topk(const uint64_t* codes, const uint32_t* docids, uint64_t n_items,
                             uint32_t words, const uint64_t* query, uint32_t count,
                             const uint64_t* mask, uint64_t n_mask_words, uint32_t* out_ids,
                             uint32_t* out_dist) {
    if (count == 0) return 0;
    hhit_t* all = (hhit_t*)malloc((size_t)(n_items ? n_items : 1) * sizeof(hhit_t));
    uint64_t n = 0;
    for (uint64_t i = 0; i < n_items; ++i) {
        const uint32_t id = docids ? docids[i] : (uint32_t)i;
        if (!is_candidate(id, mask, n_mask_words)) continue;
        uint32_t d = 0;
        for (uint32_t w = 0; w < words; ++w)
            d += (uint32_t)__builtin_popcountll(codes[i * words + w] ^ query[w]);
        all[n].d = d;
        all[n].id = id;
        ++n;
    }
    qsort(all, n, sizeof(hhit_t), hhit_cmp);
    const uint32_t m = n < count ? (uint32_t)n : count;
    for (uint32_t i = 0; i < m; ++i) {
        out_ids[i] = all[i].id;
        out_dist[i] = all[i].d;
    }
    free(all);
    return m;
}

/*
 * Euclidean and dot-product stores have no counterpart in the reference either
 * (src/index.rs:155 again); they follow the arroy 0.2.0 distances of the same
 * name as published (distance/euclidean.rs, distance/dot_product.rs):
 *   Euclidean   built_distance = sum (p_i - q_i)^2, results ordered by it,
 *               reported distance = normalized_distance = its square root;
 *   DotProduct  built_distance on a built index = -dot(p, q), results ordered by
 *               it, reported distance = normalized_distance = dot(p, q).
 * Ties break by ascending id as everywhere else. f32 accumulation left to right,
 * or f64 (mode ACC_F64) rounded once. PARITY UNPINNED like the rest of this file.
 */
enum { METRIC_EUCLIDEAN = 2, METRIC_DOT = 3 };

static float order_value(const float* p, const float* q, uint32_t n, int metric, int mode) {
    if (metric == METRIC_DOT) return -dot_mode(p, q, n, mode);
    if (mode == ACC_F64) {
        double s = 0.0;
        for (uint32_t i = 0; i < n; ++i) {
            const double d = (double)p[i] - (double)q[i];
            s += d * d;
        }
        return (float)s;
    }
    float s = 0.0f;
    for (uint32_t i = 0; i < n; ++i) {
        const float d = p[i] - q[i];
        s += d * d;
    }
    return s;
}

uint32_t oracle_nns_by_vector_metric(const float* vectors, const uint32_t* docids, uint64_t n_items,
                                     uint32_t dim, const float* query, uint32_t count,
                                     const uint64_t* mask, uint64_t n_mask_words, int metric,
                                     int mode, uint32_t* out_ids, float* out_dist) {
    if (count == 0 || (metric != METRIC_EUCLIDEAN && metric != METRIC_DOT)) return 0;
    hit_t* all = (hit_t*)malloc((size_t)(n_items ? n_items : 1) * sizeof(hit_t));
    uint64_t n = 0;
    for (uint64_t i = 0; i < n_items; ++i) {
        const uint32_t id = docids ? docids[i] : (uint32_t)i;
        if (!is_candidate(id, mask, n_mask_words)) continue;
        all[n].d = order_value(vectors + i * dim, query, dim, metric, mode);
        all[n].id = id;
        ++n;
    }
    qsort(all, n, sizeof(hit_t), hit_cmp);
    const uint32_t m = n < count ? (uint32_t)n : count;
    for (uint32_t i = 0; i < m; ++i) {
        out_ids[i] = all[i].id;
        out_dist[i] = metric == METRIC_DOT ? -all[i].d : sqrtf(all[i].d);
    }
    free(all);
    return m;
}
