// HIP kernels of the filtered vector scan (gfx950 / CDNA4, wave64).
//
// The scan replaces the exhaustive regime of arroy 0.2.0
// `Reader<Angular>::nns_by_vector` as the reference calls it from
// VectorSort::fill_buffer (src/search/ranking/vector_sort.rs:73-77): every item
// of the store that is in the candidate bitmap gets its Angular distance to the
// query, and the `k` smallest by (distance, docid) come back.
//
// Shape of the work: one query against N rows is a matrix-vector product that
// reads every passing row exactly once, so it is bound by HBM bandwidth. The
// kernel therefore
//   * keeps the query in registers (each lane owns the slice of the query that
//     matches the 16-byte vectors it loads from every row),
//   * gives one wave one row at a time: a row is read by full-wave 1 KiB
//     loads straight into registers (no LDS staging: nothing is reused),
//   * tests the candidate bit BEFORE a row is loaded, so rows that the filter
//     rejects cost no HBM traffic at all,
//   * reduces the 64 per-lane partial sums on DPP (no LDS), and
//   * keeps a sorted top-k list per wave in LDS that is only touched when a
//     row beats the wave's current k-th best (rare after the first few rows).
#include "vs_kernels.h"

#include "../../include/vs_abi.h"
#include "vs_device.h"

namespace vs {

constexpr int WAVES = 4;               // waves per workgroup
constexpr int QCAP = 128 + 64;         // row queue entries per wave: the largest scoring group + one mask word
// Compute units of the device the grids are sized for: 256 on a whole MI355X; vs_init reports what
// the runtime sees (a partitioned or CU-masked device has fewer).
static uint32_t g_num_cus = 256;
#define NUM_CUS g_num_cus

struct ScanArgs {
    const u32x4* rows;
    const float* norms;
    const uint64_t* present;
    const uint64_t* cand;
    const float* query;
    uint64_t* partial;  // [k][gridDim.x], rank-major
    uint64_t n_slots;
    uint32_t row_vecs;
    uint32_t dim;
    uint32_t docid_base;
    uint32_t k;
    int32_t metric;  // host-side dispatch only
    uint32_t pair = 0;  // non-zero: workgroups b and b + gridDim.x / 2 share their mask words (scan_float_rows)
    // Several independent queries in one launch: workgroup (b, y) takes the b-th share of the mask
    // words for the queries y, y + gridDim.y, ... one after the other, so that a workgroup which has
    // finished its share of one query goes straight on to the next (no ramp and tail per query, one
    // launch per step). No row read is shared: every list is what a launch of its own would write.
    uint32_t n_q = 1;
    uint32_t query_stride = 0;            // floats (Hamming: 16-byte vectors) between consecutive queries
    uint64_t partial_stride = 0;          // keys between consecutive queries' partial lists
    const ScanJob* jobs = nullptr;        // nullable: query y is searched as jobs[y] says instead (n_q entries)
    uint32_t list_bufs = 1;               // 2: the waves' LDS lists alternate between two buffers from query to query
};

__device__ __forceinline__ uint32_t lanes_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

// Elements of one 16-byte row vector: 4 f32 or 8 bf16.
template <int ELEM>
struct Epv {
    static constexpr int value = ELEM == VS_F32 ? 4 : 8;
};

// acc += <one 16-byte row vector, the matching query slice>, fixed order.
template <int ELEM>
__device__ __forceinline__ float dot_vec(u32x4 x, const float* q, float acc) {
    if (ELEM == VS_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_fmaf(__uint_as_float(x[j]), q[j], acc);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc = __builtin_fmaf(__uint_as_float(x[j] << 16), q[2 * j], acc);
            acc = __builtin_fmaf(__uint_as_float(x[j] & 0xFFFF0000u), q[2 * j + 1], acc);
        }
    }
    return acc;
}
// acc += ‖vector − query slice‖² (differences rounded to f32, then fused into the sum).
template <int ELEM>
__device__ __forceinline__ float sqdiff_vec(u32x4 x, const float* q, float acc) {
    if (ELEM == VS_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = __uint_as_float(x[j]) - q[j];
            acc = __builtin_fmaf(d, d, acc);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(x[j] << 16) - q[2 * j];
            const float hi = __uint_as_float(x[j] & 0xFFFF0000u) - q[2 * j + 1];
            acc = __builtin_fmaf(lo, lo, acc);
            acc = __builtin_fmaf(hi, hi, acc);
        }
    }
    return acc;
}
// acc += <vector, vector> for the norm.
template <int ELEM>
__device__ __forceinline__ float self_dot_vec(u32x4 x, float acc) {
    if (ELEM == VS_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = __uint_as_float(x[j]);
            acc = __builtin_fmaf(v, v, acc);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(x[j] << 16), hi = __uint_as_float(x[j] & 0xFFFF0000u);
            acc = __builtin_fmaf(lo, lo, acc);
            acc = __builtin_fmaf(hi, hi, acc);
        }
    }
    return acc;
}

template <bool NT>
__device__ __forceinline__ u32x4 load_vec(const u32x4* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// arroy 0.2.0 Angular::built_distance: cos = <p,q> / (‖p‖‖q‖); d = (1 - cos) / 2,
// and 0 when ‖p‖‖q‖ == 0.
__device__ __forceinline__ float angular_distance(float pq, float pn, float qn) {
    const float pnqn = pn * qn;
    if (pnqn != 0.0f) {
        const float c = pq / pnqn;
        return (1.0f - c) / 2.0f;
    }
    return 0.0f;
}

// The skeleton both scan kernels share. The wave `w` of `W` owns the 64-slot mask
// words w, w + W, w + 2W, ... Each lane fetches one of the next 64 of them (the
// following batch is requested before this one is scored); words with a set bit
// are then taken one at a time, their passing slots appended to the wave's queue
// in LDS, and score(head, n) is called for GROUP queued slots at a time (the
// last call may get fewer: n < GROUP). A kernel may also offer dense(slot0, word):
// it is then given the (nearly) full words whole, bypassing the queue.
struct NoDenseFn {
    __device__ __forceinline__ bool wants(uint64_t) const { return false; }
    __device__ __forceinline__ void operator()(uint32_t, uint64_t) const {}
};
// keep(t): whether the wave takes the t-th of its words at all (a pass over a sample of the
// words, or over the rest)
struct EveryWord {
    __device__ __forceinline__ bool keep(uint64_t) const { return true; }
};
// ROUNDS > 0: batches with fewer than ROUNDS candidates per non-empty word are unpacked by rounds (below).
// The code costs a few registers, which the scan over long f32 rows does not have to spare (its four
// waves per SIMD under a selective filter end at 128): 0 there.
template <int GROUP, int ROUNDS, typename ScoreFn, typename DenseFn = NoDenseFn, typename WordFn = EveryWord>
__device__ __forceinline__ void scan_masked_slots(const uint64_t* __restrict__ present,
                                                  const uint64_t* __restrict__ cand,
                                                  uint64_t n_slots, uint32_t w, uint32_t W, int lane,
                                                  uint32_t* queue, ScoreFn&& score, DenseFn&& dense = DenseFn(),
                                                  WordFn&& words = WordFn(), bool have_first = false,
                                                  uint64_t first_batch = 0) {
    static_assert(GROUP <= QCAP - 64, "queue too small");
    const uint64_t n_words = n_slots >> 6;
    const uint64_t my_words = w < n_words ? (n_words - w + W - 1) / W : 0;
    auto fetch_words = [&](uint64_t t0) -> uint64_t {
        const uint64_t t = t0 + (uint64_t)lane;
        uint64_t wd = 0;
        if (t < my_words && words.keep(t)) {
            const uint64_t wi = (uint64_t)w + t * W;
            wd = present[wi];
            if (cand) wd &= cand[wi];
        }
        return wd;
    };
    uint32_t qcount = 0;
    // scores whole groups off the front of the queue and moves the (< GROUP) leftovers there
    auto drain = [&]() {
        uint32_t head = 0;
        do {
            score(head, (uint32_t)GROUP);
            head += GROUP;
        } while (qcount - head >= (uint32_t)GROUP);
        const uint32_t rem = qcount - head;
        uint32_t v = 0;
        if ((uint32_t)lane < rem) v = queue[head + lane];
        wave_lds_fence();
        if ((uint32_t)lane < rem) queue[lane] = v;
        wave_lds_fence();
        qcount = rem;
    };
    // (a caller that had to look at the wave's first 64 words anyway hands them over)
    uint64_t wd_next = have_first ? first_batch : fetch_words(0);
    for (uint64_t t0 = 0; t0 < my_words; t0 += 64) {
        uint64_t wd = wd_next;
        if (t0 + 64 < my_words) wd_next = fetch_words(t0 + 64);
        // words the kernel takes whole, one at a time
        uint64_t whole = __ballot(wd != 0 && dense.wants(wd));
        while (whole) {
            const int b = __builtin_ctzll(whole);
            whole &= whole - 1;
            dense((uint32_t)(((uint64_t)w + (t0 + (uint64_t)b) * W) << 6), readlane_u64(wd, b));
        }
        if (dense.wants(wd)) wd = 0;
        uint64_t live = __ballot(wd != 0);
        if (!live) continue;
        // Under a selective filter (fewer than ROUNDS candidates per word with any) the words are unpacked by
        // all lanes together: in every round each lane that still has a candidate in ITS word appends the
        // lowest one (ballot + prefix count give it its place), so a batch of 64 words costs as many rounds
        // as its fullest word holds candidates -- four or so under a 1 % filter, where a walk word by word
        // took thirty steps. The queue's order is then round-major; which rows share a scoring group
        // changes nothing in any row's distance or in the set the lists keep. Denser batches keep the walk
        // word by word: their candidates are neighbours in memory, and reading them in slot order is what
        // lets a dense scan stream.
        if (ROUNDS > 0 && wave_sum_u32((uint32_t)__popcll(wd)) < (uint32_t)ROUNDS * (uint32_t)__popcll(live)) {
            const uint32_t my_slot0 = (uint32_t)(((uint64_t)w + (t0 + (uint64_t)lane) * W) << 6);
            while (live) {
                if (wd) {
                    queue[qcount + lanes_below(live)] = my_slot0 + (uint32_t)__builtin_ctzll(wd);
                    wd &= wd - 1;
                }
                qcount += (uint32_t)__popcll(live);
                wave_lds_fence();
                if (qcount >= (uint32_t)GROUP) drain();
                live = __ballot(wd != 0);
            }
            continue;
        }
        while (live) {
            const int b = __builtin_ctzll(live);
            live &= live - 1;
            const uint64_t word = readlane_u64(wd, b);
            const uint32_t slot0 = (uint32_t)(((uint64_t)w + (t0 + (uint64_t)b) * W) << 6);
            if ((word >> lane) & 1) queue[qcount + lanes_below(word)] = slot0 + (uint32_t)lane;
            qcount += (uint32_t)__popcll(word);
            wave_lds_fence();
            if (qcount >= (uint32_t)GROUP) drain();
        }
    }
    if (qcount) score(0, qcount);
}

// The first 64 mask words of wave `w` of `W` (present & candidates), one per lane: what
// scan_masked_slots fetches first, for callers that want to look at them before.
__device__ __forceinline__ uint64_t first_mask_words(const uint64_t* __restrict__ present,
                                                     const uint64_t* __restrict__ cand, uint64_t n_slots,
                                                     uint32_t w, uint32_t W, int lane) {
    const uint64_t wi = (uint64_t)w + (uint64_t)lane * W;
    if (wi >= (n_slots >> 6)) return 0;
    uint64_t x = present[wi];
    if (cand) x &= cand[wi];
    return x;
}

// Two waves per SIMD (twice as many workgroups as CUs) win under a selective filter, where one
// wave's mask decoding overlaps the other's row loads; one wave per SIMD wins when most rows
// pass (fewer, longer streams: 7.19 against 7.00 TB/s on 10 M x 768 f32 at 100 %, 6.9 against 6.8
// at 50 %; 3.7 against 4.2 at 1 %). A launch does not know its filter's density, so in a paired
// launch wave i of workgroup b and wave i of workgroup b + gridDim.x / 2 look at the same sample --
// the first 64 mask words the two own together -- and take the same decision: dense (at least
// `num` candidates in `den` slots), and the first wave takes all their words while the second has
// nothing to do; sparse, and they split them as two independent waves would. Returns whether
// this wave is the idle one; otherwise *w / *W are the wave's stride over the mask words and
// *first_batch its first 64 words (one per lane, present & candidates).
__device__ __forceinline__ bool pair_up_waves(const uint64_t* __restrict__ present,
                                              const uint64_t* __restrict__ cand, uint64_t n_slots, int lane,
                                              uint32_t wib, uint32_t num, uint32_t den, uint32_t* w, uint32_t* W,
                                              uint64_t* first_batch, bool* dense = nullptr) {
    const uint32_t half = gridDim.x >> 1, hi = blockIdx.x >= half ? 1u : 0u;
    const uint32_t Wp = half * WAVES, wp = (blockIdx.x - hi * half) * WAVES + wib;
    const uint64_t n_words = n_slots >> 6;
    auto word_at = [&](uint64_t wi) -> uint64_t {
        if (wi >= n_words) return 0;
        uint64_t x = present[wi];
        if (cand) x &= cand[wi];
        return x;
    };
    const uint64_t shared_words = word_at((uint64_t)wp + (uint64_t)lane * Wp);  // the pair's first 64
    const uint64_t own_words = word_at((uint64_t)*w + (uint64_t)lane * *W);     // this wave's first 64 if they split
    const uint64_t sampled = wp < n_words ? (n_words - wp + Wp - 1) / Wp : 0;
    const float bits = wave_sum_f32((float)__popcll(shared_words));             // <= 4096: exact
    const float slots = 64.0f * (float)(sampled < 64 ? sampled : 64);
    const bool is_dense = bits * (float)den >= slots * (float)num;
    if (dense) *dense = is_dense;
    if (is_dense) {
        if (hi) return true;
        *w = wp;
        *W = Wp;
        *first_batch = shared_words;
    } else {
        *first_batch = own_words;
    }
    return false;
}

// Query lanes of a many-query launch. gridDim.y workgroups share each share of the mask words and
// split the launch's queries between them (lane y takes y, y + gridDim.y, ...): under a selective
// filter a wave spends most of a query waiting for a handful of scattered rows, and several queries in
// flight per SIMD hide that (1 M x 768 f32 at 1 %: 11.1 us per query with one lane, 7.1 with four).
// A dense scan wants the opposite, one streaming wave per SIMD (10 M x 768 f32 at 100 %: 0.899 of the
// HBM peak with one lane, 0.876 with four). Every workgroup of a share looks at the same sample of
// the mask, so they agree without talking: where all four waves see a dense sample, lane 0 takes
// every query and the other lanes leave. Sets the first query and the stride of this workgroup's
// loop; returns false if it has nothing to do.
__device__ __forceinline__ bool split_queries(bool dense_here, uint32_t* y0, uint32_t* y_step) {
    *y0 = blockIdx.y;
    *y_step = gridDim.y;
    if (gridDim.y == 1) return true;
    if (__syncthreads_and(dense_here ? 1 : 0)) {
        if (blockIdx.y != 0) return false;
        *y_step = 1;
    }
    return true;
}
// Whether at least `num` in `den` of the slots of a wave's first 64 mask words are candidates.
__device__ __forceinline__ bool sample_is_dense(uint64_t first_batch, uint64_t n_slots, uint32_t w, uint32_t W,
                                                uint32_t num, uint32_t den) {
    const uint64_t n_words = n_slots >> 6;
    const uint64_t mine = w < n_words ? (n_words - w + W - 1) / W : 0;
    const float bits = wave_sum_f32((float)__popcll(first_batch));
    const float slots = 64.0f * (float)(mine < 64 ? mine : 64);
    return bits * (float)den >= slots * (float)num;
}

// Workgroup epilogue: wave 0 absorbs the other waves' lists and writes the
// workgroup's k best, rank-major, to partial[rank * gridDim.x + blockIdx.x].
template <typename TopK>
__device__ __forceinline__ void finish_block(TopK& top, unsigned char* smem, int k, uint32_t wib,
                                             int lane, uint64_t* partial) {
    top.spill();
    __syncthreads();
    if (wib == 0) {
        for (int v = 1; v < WAVES; ++v)
            top.absorb(reinterpret_cast<uint64_t*>(smem) + (size_t)v * k);
        for (int base = 0; base < k; base += 64) {
            const int i = base + lane;
            const uint64_t key = top.at_lane(base);
            if (i < k) partial[(uint64_t)i * gridDim.x + blockIdx.x] = key;
        }
    }
}

// One query against every candidate row of an f32 or bf16 slab. METRIC picks what
// a row's sort key is made of:
//   VS_ANGULAR    arroy's Angular distance (above);
//   VS_EUCLIDEAN  the SQUARED distance Σ(p − q)², as arroy's Euclidean orders its
//                 results; the merge reports its square root (normalized_distance);
//   VS_DOT        −⟨p, q⟩, so that larger products come first, as arroy's DotProduct
//                 orders once an index is built; the merge reports ⟨p, q⟩.
template <int METRIC, int ELEM, int NCH, int U, bool NT>
__device__ __forceinline__ void scan_float_rows(const ScanArgs& a, unsigned char* smem) {
    constexpr int EPV = Epv<ELEM>::value;
    const int lane = threadIdx.x & 63;
    const uint32_t wib = uniform_u32(threadIdx.x >> 6);
    const uint32_t W0 = gridDim.x * WAVES, w0 = blockIdx.x * WAVES + wib;
    const int k = (int)a.k;
    const size_t list_bytes = (size_t)WAVES * k * 8;

    uint32_t* queue = reinterpret_cast<uint32_t*>(smem + a.list_bufs * list_bytes) + (size_t)wib * QCAP;

    // What a wave takes of the mask words depends on the mask alone: decided once when every query
    // of the launch is searched under the same one.
    uint32_t W = W0, w = w0;
    bool idle = false, have_first = false;
    uint64_t first_batch = 0;
    bool dense = a.cand == nullptr;
    if (a.pair && !a.jobs) {
        idle = pair_up_waves(a.present, a.cand, a.n_slots, lane, wib, 1, 6, &w, &W, &first_batch, &dense);
        have_first = true;
    } else if (gridDim.y > 1 && a.cand && !a.jobs) {
        first_batch = first_mask_words(a.present, a.cand, a.n_slots, w, W, lane);
        have_first = true;
        dense = sample_is_dense(first_batch, a.n_slots, w, W, 1, 6);
    }
    uint32_t y0, y_step;
    if (!split_queries(dense && !a.jobs, &y0, &y_step)) return;

    // the common arguments, or (per query) what its job says
    const u32x4* rows = a.rows;
    const float* norms = a.norms;
    const uint64_t* present = a.present;
    const uint64_t* cand = a.cand;
    uint64_t n_slots = a.n_slots;
    uint32_t docid_base = a.docid_base;

    uint32_t turn = 0;
    for (uint32_t y = y0; y < a.n_q; y += y_step, ++turn) {
        const float* query = a.query + (size_t)y * a.query_stride;
        if (a.jobs) {
            const ScanJob& job = a.jobs[y];
            rows = reinterpret_cast<const u32x4*>(job.rows);
            norms = job.norms;
            present = job.present;
            cand = job.cand;
            n_slots = job.n_slots;
            docid_base = job.docid_base;
            if (job.query) query = static_cast<const float*>(job.query);
            W = W0;
            w = w0;
            idle = have_first = false;
            if (a.pair && cand) {
                idle = pair_up_waves(present, cand, n_slots, lane, wib, 1, 6, &w, &W, &first_batch);
                have_first = true;
            }
        }
        unsigned char* lists = smem + (a.list_bufs > 1 ? (turn & 1) * list_bytes : 0);
        WaveTopK top;
        top.init(reinterpret_cast<uint64_t*>(lists) + (size_t)wib * k, k, lane);
        if (WaveTopK::needs_park_buffer(a.k))  // (the launcher adds the buffers behind the queues)
            top.park_in(reinterpret_cast<uint64_t*>(smem + a.list_bufs * list_bytes + (size_t)WAVES * QCAP * 4) +
                        (size_t)wib * 128);

        // this lane's slice of the query, and ‖q‖ summed in the row order
        float q[NCH][EPV];
        float qq = 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                const uint32_t e = (uint32_t)(c * 64 + lane) * EPV + j;
                q[c][j] = e < a.dim ? query[e] : 0.0f;
                qq = __builtin_fmaf(q[c][j], q[c][j], qq);
            }
        }
        const float qn = sqrtf(wave_sum_f32(qq));
        wave_lds_fence();

        // Scores rows queue[head .. head+U); the first `nvalid` of them count.
        auto score_rows = [&](uint32_t head, uint32_t nvalid) {
            top.make_room(U);
            uint32_t slot[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t at = head + ((uint32_t)u < nvalid ? (uint32_t)u : nvalid - 1);
                slot[u] = uniform_u32(queue[at]);
            }
            u32x4 x[U][NCH];
            float pn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32x4* rp = rows + (uint64_t)slot[u] * a.row_vecs;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const uint32_t idx = (uint32_t)(c * 64 + lane);
                    if (c < NCH - 1 || idx < a.row_vecs) {
                        x[u][c] = load_vec<NT>(rp + idx);
                    } else {
                        x[u][c] = u32x4{0u, 0u, 0u, 0u};
                    }
                }
                pn[u] = METRIC == VS_ANGULAR ? norms[slot[u]] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float acc = 0.0f;
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    acc = METRIC == VS_EUCLIDEAN ? sqdiff_vec<ELEM>(x[u][c], q[c], acc)
                                                 : dot_vec<ELEM>(x[u][c], q[c], acc);
                const float pq = wave_sum_f32(acc);
                const float d = METRIC == VS_ANGULAR     ? angular_distance(pq, pn[u], qn)
                                : METRIC == VS_EUCLIDEAN ? pq
                                                         : -pq;
                const uint64_t key = make_key(uniform_u32(f32_orderable(d)), docid_base + slot[u]);
                if ((uint32_t)u < nvalid && key < top.tau) top.insert(key);
            }
        };

        if (!idle)
            scan_masked_slots<U, (ELEM == VS_BF16 || NCH <= 2) ? 4 : 0>(present, cand, n_slots, w, W, lane, queue, score_rows,
                                                                NoDenseFn(), EveryWord(), have_first, first_batch);
        finish_block(top, lists, k, wib, lane, a.partial + (uint64_t)y * a.partial_stride);
        // one list buffer only: wave 0 must be through with it before the next query's lists go there
        if (a.list_bufs < 2 && y + y_step < a.n_q) __syncthreads();
    }
}

template <int ELEM, int NCH, int U, bool NT>
__global__ __launch_bounds__(WAVES * 64) void scan_angular_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    scan_float_rows<VS_ANGULAR, ELEM, NCH, U, NT>(a, smem);
}
// The two metrics below have no counterpart in the reference (its stores are all
// Angular, src/index.rs:155); they follow the arroy distances of the same name.
template <int ELEM, int NCH, int U>
__global__ __launch_bounds__(WAVES * 64) void scan_euclidean_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    scan_float_rows<VS_EUCLIDEAN, ELEM, NCH, U, true>(a, smem);
}
template <int ELEM, int NCH, int U>
__global__ __launch_bounds__(WAVES * 64) void scan_dot_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    scan_float_rows<VS_DOT, ELEM, NCH, U, true>(a, smem);
}

// ---- several queries per pass ---------------------------------------------------------
//
// The same scan with Q = 4 or 8 queries sharing every row read (vs_search_batch on stores the
// matrix-core path does not serve: f32 rows, other metrics, k beyond its limit). A row's 12
// FMAs per lane become 12 Q, still below what the row's HBM time allows, so a dense pass
// costs what a single-query pass costs. Each lane carries Q partial sums; the reduction
// transposes while it adds: at lane distance 1 the two lanes of a pair split the queries
// between them, at distance 2 (and 4 for Q = 8) they split again, so that after log2 Q steps
// every lane holds ONE query's partial sum, and the remaining steps up to distance 32 finish
// all Q at once. Every level adds the same pairs as the single-query reduction does, so a
// query's distances are bit-identical to vs_search's.
struct MultiArgs {
    ScanArgs s;               // query = first query; partial = first query's lists
    uint32_t query_stride;    // floats between consecutive queries
    uint32_t n_q;             // 1 .. Q (missing queries are not scored)
    uint64_t partial_stride;  // keys between consecutive queries' partial lists
};

template <int XOR>
__device__ __forceinline__ float swizzle_xor_f32(float v) {  // lane ^ XOR inside each half of the wave
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (XOR << 10) | 0x1F));
}

// Two queries' sums advance together in one packed FMA per row element (each half is the
// IEEE operation the single-query scan performs).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int METRIC>
__device__ __forceinline__ f32x2 pair_step(float xe, f32x2 q, f32x2 acc) {
    const f32x2 xx = {xe, xe};
    if (METRIC == VS_EUCLIDEAN) {
        const f32x2 d = xx - q;
        return __builtin_elementwise_fma(d, d, acc);
    }
    return __builtin_elementwise_fma(xx, q, acc);
}
template <int METRIC, int ELEM>
__device__ __forceinline__ f32x2 pair_vec(u32x4 x, const f32x2* q, f32x2 acc) {
    if (ELEM == VS_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = pair_step<METRIC>(__uint_as_float(x[j]), q[j], acc);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc = pair_step<METRIC>(__uint_as_float(x[j] << 16), q[2 * j], acc);
            acc = pair_step<METRIC>(__uint_as_float(x[j] & 0xFFFF0000u), q[2 * j + 1], acc);
        }
    }
    return acc;
}

template <int METRIC, int ELEM, int NCH, int U, int Q>
__global__ __launch_bounds__(WAVES * 64) void scan_multi_kernel(MultiArgs m) {
    static_assert(Q == 4 || Q == 8, "four or eight queries per pass");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int EPV = Epv<ELEM>::value;
    const ScanArgs& a = m.s;
    const int lane = threadIdx.x & 63;
    const uint32_t wib = uniform_u32(threadIdx.x >> 6);
    const uint32_t W = gridDim.x * WAVES;
    const uint32_t w = blockIdx.x * WAVES + wib;
    const int k = (int)a.k;

    uint32_t* queue =
        reinterpret_cast<uint32_t*>(smem + (size_t)Q * WAVES * k * 8) + (size_t)wib * QCAP;
    WaveTopKRegs top[Q];  // (k <= 128 here)
#pragma unroll
    for (int q = 0; q < Q; ++q)
        top[q].init(reinterpret_cast<uint64_t*>(smem) + ((size_t)q * WAVES + wib) * k, k, lane);

    // Lane bits 0, 1 (, 2) choose the half, quarter (, eighth) of the queries a lane keeps in
    // the transposing steps: it ends up speaking for query `mine`, and query q is spoken for
    // by the lane whose low bits are q's bits in reverse order.
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    const int mine = Q == 4 ? (b0 ? 2 : 0) + (b1 ? 1 : 0) : (b0 ? 4 : 0) + (b1 ? 2 : 0) + (b2 ? 1 : 0);

    f32x2 qv[Q / 2][NCH][EPV];  // queries 2i and 2i + 1 side by side
    float my_qn = 0.0f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const float* src = a.query + (size_t)(q < (int)m.n_q ? q : 0) * m.query_stride;
        float qq = 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int j = 0; j < EPV; ++j) {
                const uint32_t e = (uint32_t)(c * 64 + lane) * EPV + j;
                const float val = e < a.dim ? src[e] : 0.0f;
                qv[q / 2][c][j][q & 1] = val;
                qq = __builtin_fmaf(val, val, qq);
            }
        }
        const float n = sqrtf(wave_sum_f32(qq));
        if (mine == q) my_qn = n;
    }
    // k-th best key of this lane's query so far; 0 (nothing is smaller) for a query the batch lacks
    uint64_t my_tau = (uint32_t)mine < m.n_q ? KEY_NONE : 0ull;
    wave_lds_fence();

    auto score_rows = [&](uint32_t head, uint32_t nvalid) {
        uint32_t slot[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t at = head + ((uint32_t)u < nvalid ? (uint32_t)u : nvalid - 1);
            slot[u] = uniform_u32(queue[at]);
        }
        u32x4 x[U][NCH];
        float pn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32x4* rp = a.rows + (uint64_t)slot[u] * a.row_vecs;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t idx = (uint32_t)(c * 64 + lane);
                if (c < NCH - 1 || idx < a.row_vecs) {
                    x[u][c] = load_vec<true>(rp + idx);
                } else {
                    x[u][c] = u32x4{0u, 0u, 0u, 0u};
                }
            }
            pn[u] = METRIC == VS_ANGULAR ? a.norms[slot[u]] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float acc[Q];
            if (ELEM == VS_BF16) {
                // packed FMAs pay where each element also costs an unpack (3.83 -> 3.55 ms per pass
                // of 10 M x 768 bf16); on f32 rows they were slower than plain ones (6.3 against 5.5 ms)
#pragma unroll
                for (int i = 0; i < Q / 2; ++i) {
                    f32x2 s2 = {0.0f, 0.0f};
#pragma unroll
                    for (int c = 0; c < NCH; ++c) s2 = pair_vec<METRIC, ELEM>(x[u][c], qv[i][c], s2);
                    acc[2 * i] = s2[0];
                    acc[2 * i + 1] = s2[1];
                }
            } else {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    acc[q] = 0.0f;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float xe = __uint_as_float(x[u][c][j]), qe = qv[q / 2][c][j][q & 1];
                            if (METRIC == VS_EUCLIDEAN) {
                                const float df = xe - qe;
                                acc[q] = __builtin_fmaf(df, df, acc[q]);
                            } else {
                                acc[q] = __builtin_fmaf(xe, qe, acc[q]);
                            }
                        }
                    }
                }
            }
            // distance 1: the even lane keeps the first half of the queries, the odd one the second
            float h[Q / 2];
#pragma unroll
            for (int i = 0; i < Q / 2; ++i) {
                const float keep = b0 ? acc[i + Q / 2] : acc[i], send = b0 ? acc[i] : acc[i + Q / 2];
                h[i] = keep + dpp_f32<0xB1, 0xF>(send);  // quad_perm [1,0,3,2]
            }
            // distance 2: halves again
            float g[Q / 4];
#pragma unroll
            for (int i = 0; i < Q / 4; ++i) {
                const float keep = b1 ? h[i + Q / 4] : h[i], send = b1 ? h[i] : h[i + Q / 4];
                g[i] = keep + dpp_f32<0x4E, 0xF>(send);  // quad_perm [2,3,0,1]
            }
            float v;
            if (Q == 8) {  // distance 4: the last split
                const float keep = b2 ? g[Q / 4 - 1] : g[0], send = b2 ? g[0] : g[Q / 4 - 1];
                v = keep + swizzle_xor_f32<4>(send);
            } else {
                v = g[0];
                v += swizzle_xor_f32<4>(v);
            }
            v += swizzle_xor_f32<8>(v);
            v += swizzle_xor_f32<16>(v);
            v += __shfl_xor(v, 32, 64);
            const float d = METRIC == VS_ANGULAR     ? angular_distance(v, pn[u], my_qn)
                            : METRIC == VS_EUCLIDEAN ? v
                                                     : -v;
            const uint64_t key = make_key(f32_orderable(d), a.docid_base + slot[u]);
            // the first Q lanes speak for one query each and compare against its k-th best
            // themselves; the common case (no query wants the row) costs one ballot
            const uint64_t hits = __ballot((uint32_t)u < nvalid && lane < Q && key < my_tau);
            if (hits) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int speaker = Q == 4 ? ((q & 1) << 1) | (q >> 1)
                                               : ((q & 1) << 2) | (q & 2) | (q >> 2);
                    if ((hits >> speaker) & 1) {
                        top[q].insert(readlane_u64(key, speaker));
                        if (mine == q) my_tau = top[q].tau;
                    }
                }
            }
        }
    };

    scan_masked_slots<U, 0>(a.present, a.cand, a.n_slots, w, W, lane, queue, score_rows);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if ((uint32_t)q < m.n_q)
            finish_block(top[q], smem + (size_t)q * WAVES * k * 8, k, wib, lane,
                         a.partial + (uint64_t)q * m.partial_stride);
    }
}

// ---- 32 queries per pass on the f32 matrix cores ----------------------------------------
//
// v_mfma_f32_32x32x2_f32 multiplies exact f32 operands and accumulates like a chain of fmaf,
// at 1/16 of the bf16 rate: 2 * 32 * dim flop per row are about what a CU can do in the row's
// HBM time, so THIRTY-TWO queries ride on one pass over an f32 slab (measured on 10 M x 768,
// k = 10: 5.5 to 5.7 ms for the main pass with two waves per SIMD, 6.55 ms with one, against
// 4.4 ms for a single-query pass; 24.6 k of a tile's cycles are matrix instructions).
// One wave = one 32-row x 32-query tile:
//   * the 32 queries sit in LDS for the whole kernel (row-major, 4 floats of padding);
//   * rows come from the candidate queue, so a filter skips rows exactly as in the
//     single-query scan. A chunk = 32 floats of each of the 32 rows: four loads in which
//     eight lanes cover one row's 128-byte line, four or eight chunks requested ahead;
//   * a chunk goes through a per-wave LDS staging tile (stride 36 floats) to turn "lane =
//     piece of a row" into the operand layout "lane = row (mod 32), k = lane / 32"; lanes of
//     the upper half read the floats 4 further, which merely permutes the order in which a
//     row's products are added;
//   * the accumulators give a lane 16 rows of ONE query (lane & 31). It evaluates the
//     distance, compares with that query's k-th best (LDS, per wave) and inserts survivors
//     into the query's sorted list in LDS itself -- the two halves of the wave take turns,
//     they share queries.
// Sums are two interleaved k-ordered fmaf chains, not the lane tree of the scans above: distances agree with
// vs_search within rounding, not bit for bit. Angular and dot product only (a squared
// distance assembled from norms and a product would cancel for near neighbours).
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MQ = 32;                   // queries per pass
constexpr uint32_t MQ_MAX_K = 32;        // per-wave lists of every query live in LDS
constexpr uint32_t MQ_MAX_VECS = 192;    // 768 f32 per row: the query block must fit LDS
constexpr int MQ_STAGE_STRIDE = 36;      // floats per staged row (32 + 4 against bank conflicts)
constexpr int MQ_QCAP = 32 + 64;         // candidate queue entries per wave: one tile + one mask word
// Waves per workgroup. A wave's matrix instructions and its other instructions do not overlap (the
// co-execution counter of a one-wave-per-SIMD run reads zero), so the staging, address and queue
// work of a tile hides only behind ANOTHER wave's matrix instructions: eight waves (two per SIMD)
// whenever their lists fit beside the query block (k <= 10 at 768 dimensions), else four.
// Chunks requested ahead per wave (even: the fragment sets alternate): 8 with four waves, 4 with
// eight, 128 KiB per CU either way.
constexpr size_t MQ_LDS_LIMIT = 160 * 1024;

struct Mq32Args {
    ScanArgs s;               // query = first query; partial = first query's lists
    uint32_t query_stride;    // floats between consecutive queries
    uint32_t n_q;             // 1 .. 32
    uint64_t partial_stride;  // keys between consecutive queries' partial lists
    const uint64_t* bound;    // nullable: [n_q][k] ascending keys of an earlier pass over the SAMPLE words
    uint32_t words;           // 0 every word, 1 the sample words only, 2 all but the sample words
};
constexpr uint32_t MQ_SAMPLE_PERIOD = 128;  // a wave's t-th word is a sample word iff (t + wave) % 128 == 0
struct Mq32Words {
    uint32_t mode, w;
    __device__ __forceinline__ bool keep(uint64_t t) const {
        return mode == 0 || (((t + w) % MQ_SAMPLE_PERIOD == 0) == (mode == 1));
    }
};

__host__ __device__ inline uint32_t mq32_chunks(uint32_t row_vecs) { return (row_vecs + 7) / 8; }
__host__ __device__ inline uint32_t mq32_qstride(uint32_t row_vecs) { return mq32_chunks(row_vecs) * 32 + 4; }
static size_t mq32_lds_bytes(uint32_t row_vecs, uint32_t k, uint32_t waves) {
    return (size_t)MQ * mq32_qstride(row_vecs) * 4          // queries
           + (size_t)waves * 32 * MQ_STAGE_STRIDE * 4       // staging tiles
           + (size_t)waves * MQ * k * 8                     // lists
           + (size_t)waves * MQ * 8                         // k-th best per query
           + (size_t)waves * 32 * 8                         // slot and norm of the tile's rows
           + (size_t)waves * MQ_QCAP * 4                    // candidate queues
           + (size_t)MQ * 4;                                // query norms
}
static uint32_t mq32_waves(uint32_t row_vecs, uint32_t k) {
    return mq32_lds_bytes(row_vecs, k, 8) <= MQ_LDS_LIMIT ? 8 : 4;
}

// WORDS (= Mq32Args::words) is a template argument so that the sample pass and the main pass carry
// different kernel names in a profile; the kernel reads the run-time copy.
template <int METRIC, int WORDS, int NW>
__global__ __launch_bounds__(NW * 64) void scan_mq32_kernel(Mq32Args m) {
    constexpr int WAVES = NW, QCAP = MQ_QCAP, MQ_DEPTH = NW == 8 ? 4 : 8;
    static_assert(METRIC == VS_ANGULAR || METRIC == VS_DOT, "angular or dot product");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ScanArgs& a = m.s;
    const int lane = threadIdx.x & 63;
    const uint32_t wib = uniform_u32(threadIdx.x >> 6);
    const uint32_t W = gridDim.x * WAVES;
    const uint32_t w = blockIdx.x * WAVES + wib;
    const int k = (int)a.k;
    const uint32_t n_chunks = mq32_chunks(a.row_vecs);
    const uint32_t qstride = mq32_qstride(a.row_vecs);

    float* qmat = reinterpret_cast<float*>(smem);
    float* stage = qmat + (size_t)MQ * qstride + (size_t)wib * 32 * MQ_STAGE_STRIDE;
    unsigned char* after_stage =
        reinterpret_cast<unsigned char*>(qmat + (size_t)MQ * qstride + (size_t)WAVES * 32 * MQ_STAGE_STRIDE);
    uint64_t* lists = reinterpret_cast<uint64_t*>(after_stage) + (size_t)wib * MQ * k;
    uint64_t* tau = reinterpret_cast<uint64_t*>(after_stage) + (size_t)WAVES * MQ * k + (size_t)wib * MQ;
    unsigned char* after_tau = after_stage + (size_t)WAVES * MQ * k * 8 + (size_t)WAVES * MQ * 8;
    uint32_t* row_slot = reinterpret_cast<uint32_t*>(after_tau) + (size_t)wib * 64;
    float* row_norm = reinterpret_cast<float*>(row_slot + 32);
    uint32_t* queue = reinterpret_cast<uint32_t*>(after_tau + (size_t)WAVES * 32 * 8) + (size_t)wib * QCAP;
    float* qnorm = reinterpret_cast<float*>(after_tau + (size_t)WAVES * 32 * 8 + (size_t)WAVES * QCAP * 4);

    // the query block, zero padded to whole chunks; missing queries are zero
    for (uint32_t i = threadIdx.x; i < (uint32_t)MQ * qstride; i += blockDim.x) {
        const uint32_t q = i / qstride, e = i - q * qstride;
        qmat[i] = (q < m.n_q && e < a.dim) ? a.query[(size_t)q * m.query_stride + e] : 0.0f;
    }
    __syncthreads();
    // norms of the queries, summed in the order the single-query scan uses
    for (uint32_t q = wib; q < (uint32_t)MQ; q += WAVES) {
        float qq = 0.0f;
        for (uint32_t c = 0; c * 64 < a.row_vecs; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t e = (c * 64 + (uint32_t)lane) * 4 + j;
                const float v = e < a.dim ? qmat[q * qstride + e] : 0.0f;
                qq = __builtin_fmaf(v, v, qq);
            }
        }
        const float n = sqrtf(wave_sum_f32(qq));
        if (lane == 0) qnorm[q] = n;
    }
    for (int i = lane; i < MQ * k; i += 64) lists[i] = KEY_NONE;
    // k-th best so far: the bound of the sample pass if there is one; 0 (nothing is smaller) for a
    // query the batch lacks
    if (lane < MQ)
        tau[lane] = (uint32_t)lane >= m.n_q ? 0ull
                    : m.bound           ? m.bound[(uint64_t)lane * k + (k - 1)]
                                        : KEY_NONE;
    __syncthreads();
    const uint32_t col = (uint32_t)lane & 31, half = (uint32_t)lane >> 5;
    const float my_qn = qnorm[col];
    uint64_t* my_list = lists + (size_t)col * k;

#ifdef VS_MQ_STAMP
    uint64_t cyc_loop = 0, cyc_epi = 0, n_tiles = 0;
#endif
    auto score_rows = [&](uint32_t head, uint32_t nvalid) {
#ifdef VS_MQ_STAMP
        const uint64_t s0 = __builtin_amdgcn_s_memtime();
#endif
        if (lane < 32) {
            const uint32_t slot = queue[head + ((uint32_t)lane < nvalid ? (uint32_t)lane : nvalid - 1)];
            row_slot[lane] = slot;
            row_norm[lane] = METRIC == VS_ANGULAR ? a.norms[slot] : 0.0f;
        }
        wave_lds_fence();
        const uint32_t piece = (uint32_t)lane & 7, r0 = (uint32_t)lane >> 3;
        const u32x4* base[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            base[i] = a.rows + (uint64_t)row_slot[r0 + 8 * i] * a.row_vecs + piece;
        // Nothing in the steady state below is conditional: a load that is issued only on some
        // paths makes the compiler wait for ALL outstanding loads at the next use, which would
        // undo the prefetch. Chunks past the end are clamped to the last one (a repeated read
        // that is never used), pieces past the end of a row read the row's last vector and are
        // zeroed on their way into the staging tile.
        const uint32_t last_chunk = n_chunks - 1;
        auto request = [&](uint32_t c, u32x4 (&dst)[4]) {
            const uint32_t cc = c < last_chunk ? c : last_chunk;
            uint32_t v = cc * 8 + piece;
            v = v < a.row_vecs ? v : a.row_vecs - 1;
            v -= piece;  // base[] already points at this lane's piece
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[i] = load_vec<true>(base[i] + (int32_t)v);
        };
        // two accumulators take alternate k-steps: a link of either chain waits for the link two
        // back, which has left the pipeline by then, instead of for the one just issued
        f32x16 acc, acc_odd;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = acc_odd[e] = 0.0f;
        u32x4 buf[MQ_DEPTH][4];
#pragma unroll
        for (int d = 0; d < MQ_DEPTH; ++d) request((uint32_t)d, buf[d]);
        const float* arow = stage + col * MQ_STAGE_STRIDE + 4 * half;
        const float* brow = qmat + (size_t)col * qstride + 4 * half;
        // rows i0, i0 + 1 of the four a lane stages / requests; fragment pairs mm0, mm0 + 1
        auto to_stage2 = [&](uint32_t c, const u32x4 (&src)[4], const int i0) {
            const bool in_row = c * 8 + piece < a.row_vecs;
#pragma unroll
            for (int i = i0; i < i0 + 2; ++i)
                *reinterpret_cast<u32x4*>(stage + (r0 + 8 * i) * MQ_STAGE_STRIDE + piece * 4) =
                    in_row ? src[i] : u32x4{0u, 0u, 0u, 0u};
        };
        auto request2 = [&](uint32_t c, u32x4 (&dst)[4], const int i0) {
            const uint32_t cc = c < last_chunk ? c : last_chunk;
            uint32_t v = cc * 8 + piece;
            v = v < a.row_vecs ? v : a.row_vecs - 1;
            v -= piece;
#pragma unroll
            for (int i = i0; i < i0 + 2; ++i) dst[i] = load_vec<true>(base[i] + (int32_t)v);
        };
        auto frag_a2 = [&](f32x4 (&fa)[4], const int mm0) {
#pragma unroll
            for (int mm = mm0; mm < mm0 + 2; ++mm) fa[mm] = *reinterpret_cast<const f32x4*>(arow + 8 * mm);
        };
        auto frag_b2 = [&](uint32_t c, f32x4 (&fb)[4], const int mm0) {
#pragma unroll
            for (int mm = mm0; mm < mm0 + 2; ++mm)
                fb[mm] = *reinterpret_cast<const f32x4*>(brow + c * 32 + 8 * mm);
        };
        // Software pipeline, one wave per SIMD. The 16 MFMAs of chunk c form one dependent chain
        // (64 cycles each), and a wave issues in order: whatever follows the chain in program
        // order starts only when its last link has been issued. So the work for the next chunks is
        // cut into eight pieces of about ten instructions and placed BETWEEN the links, where it
        // issues in the shadow of the MFMA before it: chunk c + 1 goes through the staging tile
        // into the other fragment set (LDS executes a wave's operations in order, so the reads of
        // chunk c were done before the overwrite), the rows of chunk c + MQ_DEPTH are requested
        // from HBM. One step past the last chunk stages and reads a repeated chunk that no MFMA
        // consumes (the fragment read of the query block then runs at most 32 floats past a
        // query's row: still inside LDS).
        f32x4 fa[2][4], fb[2][4];
        to_stage2(0, buf[0], 0);
        to_stage2(0, buf[0], 2);
        request((uint32_t)MQ_DEPTH, buf[0]);
        wave_lds_fence();
        frag_a2(fa[0], 0);
        frag_a2(fa[0], 2);
        frag_b2(0, fb[0], 0);
        frag_b2(0, fb[0], 2);
        auto step = [&](uint32_t c, const int d) {
            const int nb = (d + 1) % MQ_DEPTH, ns = (d + 1) & 1, cs = d & 1;
            auto mfma2 = [&](const int mm, const int t0) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cs][mm][t0], fb[cs][mm][t0], acc, 0, 0, 0);
                acc_odd = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cs][mm][t0 + 1], fb[cs][mm][t0 + 1], acc_odd, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            wave_lds_fence();
            mfma2(0, 0);
            to_stage2(c + 1, buf[nb], 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma2(0, 2);
            to_stage2(c + 1, buf[nb], 2);
            __builtin_amdgcn_sched_barrier(0);
            mfma2(1, 0);
            request2(c + 1 + MQ_DEPTH, buf[nb], 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma2(1, 2);
            request2(c + 1 + MQ_DEPTH, buf[nb], 2);
            wave_lds_fence();
            __builtin_amdgcn_sched_barrier(0);
            mfma2(2, 0);
            frag_a2(fa[ns], 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma2(2, 2);
            frag_a2(fa[ns], 2);
            __builtin_amdgcn_sched_barrier(0);
            mfma2(3, 0);
            frag_b2(c + 1, fb[ns], 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma2(3, 2);
            frag_b2(c + 1, fb[ns], 2);
            __builtin_amdgcn_sched_barrier(0);
        };
        uint32_t c0 = 0;
        for (; c0 + MQ_DEPTH <= n_chunks; c0 += MQ_DEPTH) {
#pragma unroll
            for (int d = 0; d < MQ_DEPTH; ++d) step(c0 + (uint32_t)d, d);
        }
#pragma unroll
        for (int d = 0; d < MQ_DEPTH; ++d)  // fewer than MQ_DEPTH chunks are left
            if (c0 + (uint32_t)d < n_chunks) step(c0 + (uint32_t)d, d);
        wave_lds_fence();
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += acc_odd[e];
#ifdef VS_MQ_STAMP
        const uint64_t s1 = __builtin_amdgcn_s_memtime();
#endif
        // lane: query `col`, rows (e & 3) + 8 (e >> 2) + 4 half
        uint64_t key[16];
        if (METRIC == VS_ANGULAR) {
            // Most rows lose against the query's k-th best d_k. d < d_k  <=>  <p,q> > |p| |q| (1 - 2 d_k), so
            // one multiply-add per score on the accumulator decides (with a margin of 1e-6 |q|, so that
            // rounding can only admit too many); the formula itself, with its division, is evaluated
            // for the survivors only. Anything that does not compare (NaN) survives.
            const uint64_t tk = tau[col];
            const float dk = f32_from_orderable((uint32_t)(tk >> 32));
            float c2 = (1.0f - 2.0f * dk) * my_qn - 1e-6f * my_qn;
            if (tk == KEY_NONE || !(c2 == c2)) c2 = -__builtin_inff();
            uint32_t alive = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 pn4 = *reinterpret_cast<const f32x4*>(row_norm + 8 * g + 4 * half);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = __builtin_fmaf(-pn4[i], c2, acc[4 * g + i]);
                    if (!(t < 0.0f) && (uint32_t)(8 * g + i) + 4 * half < nvalid) alive |= 1u << (4 * g + i);
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) key[e] = KEY_NONE;
            if (__ballot(alive != 0)) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t r = (uint32_t)((e & 3) + 8 * (e >> 2)) + 4 * half;
                    if ((alive >> e) & 1)
                        key[e] = make_key(f32_orderable(angular_distance(acc[e], row_norm[r], my_qn)),
                                          a.docid_base + row_slot[r]);
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t r = (uint32_t)((e & 3) + 8 * (e >> 2)) + 4 * half;
                key[e] = r < nvalid ? make_key(f32_orderable(-acc[e]), a.docid_base + row_slot[r]) : KEY_NONE;
            }
        }
#pragma unroll 1
        for (uint32_t turn = 0; turn < 2; ++turn) {
            if (half == turn) {
                uint64_t t = tau[col];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (key[e] < t) {
                        int pos = k - 1;
                        while (pos > 0 && my_list[pos - 1] > key[e]) {
                            my_list[pos] = my_list[pos - 1];
                            --pos;
                        }
                        my_list[pos] = key[e];
                        t = min_u64(t, my_list[k - 1]);
                    }
                }
                tau[col] = t;
            }
            wave_lds_fence();
        }
#ifdef VS_MQ_STAMP
        const uint64_t s2 = __builtin_amdgcn_s_memtime();
        cyc_loop += s1 - s0;
        cyc_epi += s2 - s1;
        ++n_tiles;
#endif
    };

#ifdef VS_MQ_STAMP
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
#endif
    scan_masked_slots<32, 0>(a.present, a.cand, a.n_slots, w, W, lane, queue, score_rows, NoDenseFn(),
                                 Mq32Words{m.words, w});
#ifdef VS_MQ_STAMP
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 7 && threadIdx.x == 0)
        printf("mq32 clk %llu real %llu -> %.1f MHz; tiles %llu loop %llu epi %llu per tile\n", (unsigned long long)(t1 - t0), (unsigned long long)(r1_ - r0_), (double)(t1 - t0) / (double)(r1_ - r0_) * 100.0, (unsigned long long)n_tiles, (unsigned long long)(cyc_loop / n_tiles), (unsigned long long)(cyc_epi / n_tiles));
#endif
    __syncthreads();
    // thread q of the workgroup merges the four waves' lists of query q
    if (threadIdx.x < m.n_q) {
        const uint64_t* all = reinterpret_cast<const uint64_t*>(after_stage);
        uint32_t at[WAVES];
#pragma unroll
        for (int v = 0; v < WAVES; ++v) at[v] = 0;
        // list-major: list `blockIdx.x` of the query; the sample pass's list rides along as list gridDim.x
        uint64_t* out = a.partial + (uint64_t)threadIdx.x * m.partial_stride + (uint64_t)blockIdx.x * k;
        if (m.bound && blockIdx.x == 0)
            for (int r = 0; r < k; ++r)
                a.partial[(uint64_t)threadIdx.x * m.partial_stride + (uint64_t)gridDim.x * k + r] =
                    m.bound[(uint64_t)threadIdx.x * k + r];
        for (int r = 0; r < k; ++r) {
            uint64_t best = KEY_NONE;
            int from = 0;
#pragma unroll
            for (int v = 0; v < WAVES; ++v) {
                const uint64_t cand =
                    at[v] < (uint32_t)k ? all[((size_t)v * MQ + threadIdx.x) * k + at[v]] : KEY_NONE;
                if (cand < best) {
                    best = cand;
                    from = v;
                }
            }
#pragma unroll
            for (int v = 0; v < WAVES; ++v)
                if (v == from && best != KEY_NONE) ++at[v];
            out[r] = best;
        }
    }
}

// ---- Hamming scan of binary-quantised rows ----------------------------------------------
//
// No counterpart in the reference (its only metric is Angular): distance =
// popcount(row XOR query) over the row's bits, results ascending by (distance,
// docid). Rows are short (1536 bits = 192 B), so a wave-per-row schedule would
// waste most lanes; instead FOUR lanes share a row: each loads every fourth
// 16-byte vector of it (a quad reads 64 contiguous bytes per instruction),
// popcounts its share, and two DPP adds give the row's distance to all four.
// One pass scores 16 rows per wave, U passes are in flight together.
struct HammingArgs {
    const u32x4* rows;
    const uint64_t* present;
    const uint64_t* cand;
    const u32x4* query;  // row_vecs vectors, zero padded
    uint64_t* partial;
    uint64_t n_slots;
    uint32_t row_vecs;
    uint32_t docid_base;
    uint32_t k;
    uint32_t park = 0;  // non-zero: candidates are parked and merged in batches (WaveTopK::park)
    uint32_t pair = 0;  // non-zero: workgroups b and b + gridDim.x / 2 share their mask words (pair_up_waves)
    // several independent queries in one launch, as in ScanArgs
    uint32_t n_q = 1;
    uint32_t query_stride = 0;    // 16-byte vectors between consecutive queries
    uint64_t partial_stride = 0;  // keys between consecutive queries' partial lists
    const ScanJob* jobs = nullptr;
    uint32_t list_bufs = 1;
};

// Dense stretches of 192-byte rows. A row is one and a half 128-byte lines, so the quad
// schedule above asks for half lines; here EIGHT lanes share a PAIR of rows (384 bytes = three
// whole lines): lane j of the eight loads vectors j, j + 8, j + 16 of the pair, each
// instruction reads one whole line per pair, and a wave covers the 64 slots of a mask word
// in four passes of eight pairs. Vector v of the pair belongs to row v / 12 and lines up with
// query vector v % 12; which is which is fixed per lane. Three DPP adds per row finish it.
template <bool NT>
struct HammingPairs {
    const u32x4* rows;
    uint32_t docid_base;
    int lane;
    WaveTopK* top;
    u32x4 q[3];  // query vectors j, (j + 8) % 12, j + 4 for this lane's j = lane & 7

    __device__ __forceinline__ void load_query(const u32x4* query) {
        const uint32_t j = (uint32_t)lane & 7;
        q[0] = query[j];
        q[1] = query[(j + 8) % 12];
        q[2] = query[j + 4];
    }
    // enough candidates that reading all 64 rows beats picking them one by one
    __device__ __forceinline__ bool wants(uint64_t word) const { return __popcll(word) >= 56; }
    __device__ __forceinline__ void operator()(uint32_t slot0, uint64_t word) {
        const uint32_t j = (uint32_t)lane & 7, g = (uint32_t)lane >> 3;
        u32x4 x[4][3];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const u32x4* rp = rows + ((uint64_t)slot0 + 2u * ((uint32_t)p * 8 + g)) * 12 + j;
#pragma unroll
            for (int i = 0; i < 3; ++i) x[p][i] = load_vec<NT>(rp + 8 * i);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint32_t c[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                c[i] = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) c[i] += (uint32_t)__popc(x[p][i][e] ^ q[i][e]);
            }
            // vector j: first row; vector j + 16: second row; vector j + 8: first row iff j < 4
            uint32_t ca = c[0] + (j < 4 ? c[1] : 0u), cb = c[2] + (j < 4 ? 0u : c[1]);
            ca += dpp_u32<0xB1, 0xF>(ca);   // quad_perm [1,0,3,2]
            cb += dpp_u32<0xB1, 0xF>(cb);
            ca += dpp_u32<0x4E, 0xF>(ca);   // quad_perm [2,3,0,1]
            cb += dpp_u32<0x4E, 0xF>(cb);
            ca += dpp_u32<0x141, 0xF>(ca);  // row_half_mirror: the other quad of the eight
            cb += dpp_u32<0x141, 0xF>(cb);
            // lanes 0 and 1 of the eight speak for the first and the second row
            const uint32_t r = 2u * ((uint32_t)p * 8 + g) + (j & 1);
            const uint64_t key = make_key(j == 0 ? ca : cb, docid_base + slot0 + r);
            const bool live = j < 2 && ((word >> r) & 1);
            if (top->parks()) {
                top->park(live && key < top->tau, key);
                continue;
            }
            uint64_t m = __ballot(live && key < top->tau);
            while (m) {
                const int src = __builtin_ctzll(m);
                top->insert(readlane_u64(key, src));
                m &= m - 1;
                m &= __ballot(live && key < top->tau);
            }
        }
    }
};

template <int NI, int U, bool NT>
__global__ __launch_bounds__(WAVES * 64) void scan_hamming_kernel(HammingArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int quad = lane >> 2, sub = lane & 3;
    const uint32_t wib = uniform_u32(threadIdx.x >> 6);
    const uint32_t W0 = gridDim.x * WAVES, w0 = blockIdx.x * WAVES + wib;
    const int k = (int)a.k;
    const size_t list_bytes = (size_t)WAVES * k * 8;
    uint32_t* queue = reinterpret_cast<uint32_t*>(smem + a.list_bufs * list_bytes) + (size_t)wib * QCAP;
    // (the launcher adds the buffers of the batched insertion behind the queues)
    uint64_t* park_buffer =
        reinterpret_cast<uint64_t*>(smem + a.list_bufs * list_bytes + (size_t)WAVES * QCAP * 4) + (size_t)wib * 128;
    // paired launch: where nearly every slot is a candidate (the whole-word path below) one wave per
    // SIMD streams faster (50 M x 1536 bit, all-ones mask: 6.94 against 6.78 TB/s). The decision
    // depends on the mask alone: taken once when all queries share it.
    uint32_t W = W0, w = w0;
    bool idle = false, have_first = false;
    uint64_t first_batch = 0;
    bool dense = a.cand == nullptr;
    if (a.pair && !a.jobs) {
        idle = pair_up_waves(a.present, a.cand, a.n_slots, lane, wib, 7, 8, &w, &W, &first_batch, &dense);
        have_first = true;
    } else if (gridDim.y > 1 && a.cand && !a.jobs) {
        first_batch = first_mask_words(a.present, a.cand, a.n_slots, w, W, lane);
        have_first = true;
        dense = sample_is_dense(first_batch, a.n_slots, w, W, 7, 8);
    }
    uint32_t y0, y_step;
    if (!split_queries(dense && !a.jobs, &y0, &y_step)) return;

    // the common arguments, or (per query) what its job says
    const u32x4* rows = a.rows;
    const uint64_t* present = a.present;
    const uint64_t* cand = a.cand;
    uint64_t n_slots = a.n_slots;
    uint32_t docid_base = a.docid_base;

    uint32_t turn = 0;
    for (uint32_t y = y0; y < a.n_q; y += y_step, ++turn) {
        const u32x4* query = a.query + (size_t)y * a.query_stride;
        if (a.jobs) {
            const ScanJob& job = a.jobs[y];
            rows = reinterpret_cast<const u32x4*>(job.rows);
            present = job.present;
            cand = job.cand;
            n_slots = job.n_slots;
            docid_base = job.docid_base;
            if (job.query) query = static_cast<const u32x4*>(job.query);
            W = W0;
            w = w0;
            idle = have_first = false;
            if (a.pair && cand) {
                idle = pair_up_waves(present, cand, n_slots, lane, wib, 7, 8, &w, &W, &first_batch);
                have_first = true;
            }
        }
        unsigned char* lists = smem + (a.list_bufs > 1 ? (turn & 1) * list_bytes : 0);
        WaveTopK top;
        top.init(reinterpret_cast<uint64_t*>(lists) + (size_t)wib * k, k, lane);
        if (a.park) top.park_in(park_buffer);

        // this lane's share of the query: vectors sub, sub + 4, ...
        u32x4 q[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const uint32_t p = (uint32_t)(i * 4 + sub);
            q[i] = p < a.row_vecs ? query[p] : u32x4{0u, 0u, 0u, 0u};
        }

        auto score_rows = [&](uint32_t head, uint32_t nvalid) {
            uint32_t slot[U];
            bool valid[U];
            u32x4 x[U][NI];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t r = (uint32_t)(u * 16 + quad);
                valid[u] = r < nvalid;
                slot[u] = queue[head + (valid[u] ? r : 0u)];
                const u32x4* rp = rows + (uint64_t)slot[u] * a.row_vecs;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const uint32_t p = (uint32_t)(i * 4 + sub);
                    if (i < NI - 1 || p < a.row_vecs) {
                        x[u][i] = load_vec<NT>(rp + p);
                    } else {
                        x[u][i] = u32x4{0u, 0u, 0u, 0u};
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t cnt = 0;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) cnt += (uint32_t)__popc(x[u][i][j] ^ q[i][j]);
                }
                cnt += dpp_u32<0xB1, 0xF>(cnt);  // quad_perm [1,0,3,2]
                cnt += dpp_u32<0x4E, 0xF>(cnt);  // quad_perm [2,3,0,1]
                const uint64_t key = make_key(cnt, docid_base + slot[u]);
                const bool live = valid[u] && sub == 0;
                if (top.parks()) {
                    top.park(live && key < top.tau, key);
                    continue;
                }
                uint64_t m = __ballot(live && key < top.tau);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    top.insert(readlane_u64(key, src));
                    m &= m - 1;
                    m &= __ballot(live && key < top.tau);
                }
            }
        };

        bool scanned = false;
        if constexpr (NI == 3) {
            if (a.row_vecs == 12) {  // 1536-bit rows: whole words of candidates go through the pair path
                HammingPairs<NT> pairs{rows, docid_base, lane, &top};
                pairs.load_query(query);
                if (!idle)
                    scan_masked_slots<16 * U, 16>(present, cand, n_slots, w, W, lane, queue, score_rows, pairs,
                                                    EveryWord(), have_first, first_batch);
                scanned = true;
            }
        }
        if (!scanned && !idle)
            scan_masked_slots<16 * U, 16>(present, cand, n_slots, w, W, lane, queue, score_rows, NoDenseFn(),
                                            EveryWord(), have_first, first_batch);
        top.merge_parked();
        finish_block(top, lists, k, wib, lane, a.partial + (uint64_t)y * a.partial_stride);
        if (a.list_bufs < 2 && y + y_step < a.n_q) __syncthreads();
    }
}

// ---- Hamming scan, four queries per pass ------------------------------------------------
//
// vs_search_batch on binary stores: four queries share every row read. The quad schedule of
// the single-query scan (four lanes per row, 16 rows per pass); each lane counts its share of
// the row against all four queries, and the two reduction steps transpose while they add, so
// that lane `sub` of a quad ends up with the whole distance of ONE query (2 (sub & 1) +
// (sub >> 1)) and compares it with that query's k-th best itself. Distances are integers:
// results equal vs_search's exactly.
constexpr int HQ = 4;
struct HammingMultiArgs {
    HammingArgs s;            // query = first query; partial = first query's lists
    uint32_t query_stride;    // 16-byte vectors between consecutive queries
    uint32_t n_q;             // 1 .. HQ
    uint64_t partial_stride;  // keys between consecutive queries' partial lists
};

template <int NI, int U>
__global__ __launch_bounds__(WAVES * 64) void scan_hamming_multi_kernel(HammingMultiArgs m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const HammingArgs& a = m.s;
    const int lane = threadIdx.x & 63;
    const int quad = lane >> 2, sub = lane & 3;
    const uint32_t wib = uniform_u32(threadIdx.x >> 6);
    const uint32_t W = gridDim.x * WAVES;
    const uint32_t w = blockIdx.x * WAVES + wib;
    const int k = (int)a.k;
    uint32_t* queue =
        reinterpret_cast<uint32_t*>(smem + (size_t)HQ * WAVES * k * 8) + (size_t)wib * QCAP;
    WaveTopKRegs top[HQ];  // (k <= 128 here)
#pragma unroll
    for (int q = 0; q < HQ; ++q)
        top[q].init(reinterpret_cast<uint64_t*>(smem) + ((size_t)q * WAVES + wib) * k, k, lane);

    // this lane's share of every query: vectors sub, sub + 4, ...
    u32x4 qv[HQ][NI];
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
        const u32x4* src = a.query + (size_t)(q < (int)m.n_q ? q : 0) * m.query_stride;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const uint32_t p = (uint32_t)(i * 4 + sub);
            qv[q][i] = p < a.row_vecs ? src[p] : u32x4{0u, 0u, 0u, 0u};
        }
    }
    const bool b0 = sub & 1, b1 = sub & 2;
    const int mine = (b0 ? 2 : 0) + (b1 ? 1 : 0);
    // k-th best key of this lane's query so far; 0 (nothing is smaller) for a query the batch lacks
    uint64_t my_tau = (uint32_t)mine < m.n_q ? KEY_NONE : 0ull;
    wave_lds_fence();

    auto score_rows = [&](uint32_t head, uint32_t nvalid) {
        uint32_t slot[U];
        bool valid[U];
        u32x4 x[U][NI];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t r = (uint32_t)(u * 16 + quad);
            valid[u] = r < nvalid;
            slot[u] = queue[head + (valid[u] ? r : 0u)];
            const u32x4* rp = a.rows + (uint64_t)slot[u] * a.row_vecs;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint32_t p = (uint32_t)(i * 4 + sub);
                if (i < NI - 1 || p < a.row_vecs) {
                    x[u][i] = load_vec<true>(rp + p);
                } else {
                    x[u][i] = u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t cnt[HQ];
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                cnt[q] = 0;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) cnt[q] += (uint32_t)__popc(x[u][i][j] ^ qv[q][i][j]);
                }
            }
            // lane pairs: the even lane keeps queries 0 and 1, the odd one 2 and 3; then the halves
            // of a quad split again
            const uint32_t pa = (b0 ? cnt[2] : cnt[0]) + dpp_u32<0xB1, 0xF>(b0 ? cnt[0] : cnt[2]);
            const uint32_t pb = (b0 ? cnt[3] : cnt[1]) + dpp_u32<0xB1, 0xF>(b0 ? cnt[1] : cnt[3]);
            const uint32_t d = (b1 ? pb : pa) + dpp_u32<0x4E, 0xF>(b1 ? pa : pb);
            const uint64_t key = make_key(d, a.docid_base + slot[u]);
            uint64_t hits = __ballot(valid[u] && key < my_tau);
            if (hits) {
#pragma unroll
                for (int q = 0; q < HQ; ++q) {
                    // query q is spoken for by lane ((q & 1) << 1 | q >> 1) of every quad
                    const uint64_t lanes_q = 0x1111111111111111ull << (((q & 1) << 1) | (q >> 1));
                    uint64_t hq = hits & lanes_q;
                    while (hq) {
                        const int src = __builtin_ctzll(hq);
                        top[q].insert(readlane_u64(key, src));
                        if (mine == q) my_tau = top[q].tau;
                        hq &= hq - 1;
                        hq &= __ballot(valid[u] && key < my_tau);
                    }
                }
            }
        }
    };

    scan_masked_slots<16 * U, 0>(a.present, a.cand, a.n_slots, w, W, lane, queue, score_rows);
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
        if ((uint32_t)q < m.n_q)
            finish_block(top[q], smem + (size_t)q * WAVES * k * 8, k, wib, lane,
                         a.partial + (uint64_t)q * m.partial_stride);
    }
}

// ---- merge of sorted key lists ------------------------------------------------

// One workgroup per query: blockIdx.x selects the query's key lists
// (keys + blockIdx.x * query_stride) and its slice of every output. Entry r of
// list l is keys[l * list_stride + r * rank_stride], r < n_ranks; every list is
// ascending. The four waves each merge a quarter of the lists into a WaveTopK,
// rank by rank with the next rank's keys already requested, then wave 0 absorbs
// the other three.
__global__ __launch_bounds__(256) void merge_kernel(const uint64_t* keys, uint32_t n_lists,
                                                    uint64_t list_stride, uint64_t rank_stride,
                                                    uint32_t n_ranks, uint64_t query_stride,
                                                    uint32_t k_, int32_t metric, uint64_t* out_keys,
                                                    uint32_t* out_ids, float* out_dist,
                                                    uint32_t* out_n, const uint32_t* gate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (gate && *gate == 0) return;  // a launch queued for the rare case only (nullable)
    keys += (uint64_t)blockIdx.x * query_stride;
    if (out_keys) out_keys += (uint64_t)blockIdx.x * k_;
    if (out_ids) {
        out_ids += (uint64_t)blockIdx.x * k_;
        out_dist += (uint64_t)blockIdx.x * k_;
    }
    if (out_n) out_n += blockIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t wv = uniform_u32(threadIdx.x >> 6);
    const int k = (int)k_;
    WaveTopK top;
    top.init(reinterpret_cast<uint64_t*>(smem) + (size_t)wv * k, k, lane);
    if (WaveTopK::needs_park_buffer(k_)) top.park_in(reinterpret_cast<uint64_t*>(smem) + (size_t)4 * k + (size_t)wv * 128);

    for (uint32_t base = 0; base < n_lists; base += 1024) {
        auto load_rank = [&](uint32_t r, uint64_t (&dst)[4]) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t l = base + (uint32_t)b * 256 + wv * 64 + lane;
                dst[b] = l < n_lists ? keys[(uint64_t)l * list_stride + (uint64_t)r * rank_stride]
                                     : KEY_NONE;
            }
        };
        uint64_t nxt[4];
        load_rank(0, nxt);
        // Once no list of this wave has its rank-r entry below tau, no later
        // rank can have one either.
        for (uint32_t r = 0; r < n_ranks; ++r) {
            uint64_t key[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) key[b] = nxt[b];
            if (r + 1 < n_ranks) load_rank(r + 1, nxt);
            bool any = false;
            int b0 = 0;
            // the list is empty: 128 keys at the price of one sort (a handful of keys are cheaper inserted)
            if (r == 0 && base == 0 && top.in_regs &&
                __popcll(__ballot(key[0] != KEY_NONE)) + __popcll(__ballot(key[1] != KEY_NONE)) >= 24) {
                top.fill128(key[0], key[1]);
                any = true;
                b0 = 2;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b < b0) continue;
                if (!top.in_regs) {  // lists in LDS take their keys a batch at a time
                    const bool mine = key[b] < top.tau;
                    if (__ballot(mine)) any = true;
                    top.park(mine, key[b]);
                    continue;
                }
                uint64_t m = __ballot(key[b] < top.tau);
                while (m) {
                    const int src = __builtin_ctzll(m);
                    top.insert(readlane_u64(key[b], src));
                    any = true;
                    m &= m - 1;
                    m &= __ballot(key[b] < top.tau);
                }
            }
            if (!any) break;
        }
    }
    top.spill();
    __syncthreads();
    if (wv == 0) {
        for (int v = 1; v < 4; ++v)
            top.absorb(reinterpret_cast<uint64_t*>(smem) + (size_t)v * k);
        uint32_t n = 0;
        for (int base = 0; base < k; base += 64) {
            const int i = base + lane;
            uint64_t key = top.at_lane(base);
            if (i >= k) key = KEY_NONE;
            if (i < k) {
                if (out_keys) out_keys[i] = key;
                if (out_ids) {
                    const uint32_t hi = (uint32_t)(key >> 32);
                    out_ids[i] = (uint32_t)key;
                    float d = 0.0f;
                    if (key != KEY_NONE) {
                        if (metric == VS_HAMMING) d = (float)hi;
                        else if (metric == VS_EUCLIDEAN) d = sqrtf(f32_from_orderable(hi));
                        else if (metric == VS_DOT) d = -f32_from_orderable(hi);
                        else d = f32_from_orderable(hi);
                    }
                    out_dist[i] = d;
                }
            }
            n += (uint32_t)__popcll(__ballot(key != KEY_NONE));
        }
        if (out_n && lane == 0) *out_n = n;
    }
}

// ---- norms ---------------------------------------------------------------------

template <int ELEM>
__global__ __launch_bounds__(256) void norms_kernel(const u32x4* rows, uint32_t row_vecs,
                                                    float* norms, uint64_t slot0, uint64_t n) {
    const int lane = threadIdx.x & 63;
    const uint64_t W = (uint64_t)gridDim.x * 4;
    for (uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += W) {
        const u32x4* rp = rows + (slot0 + i) * row_vecs;
        float acc = 0.0f;
        // same per-lane order as the scan: chunk by chunk, element by element
        for (uint32_t idx = lane; idx < row_vecs; idx += 64) acc = self_dot_vec<ELEM>(rp[idx], acc);
        const float s = wave_sum_f32(acc);
        if (lane == 0) norms[slot0 + i] = sqrtf(s);
    }
}

// the same for a list of slots (rows rewritten in place when a seal applies replaced items)
template <int ELEM>
__global__ __launch_bounds__(256) void norms_slots_kernel(const u32x4* rows, uint32_t row_vecs,
                                                          float* norms, const uint64_t* slots, uint64_t n) {
    const int lane = threadIdx.x & 63;
    const uint64_t W = (uint64_t)gridDim.x * 4;
    for (uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += W) {
        const uint64_t slot = slots[i];
        const u32x4* rp = rows + slot * row_vecs;
        float acc = 0.0f;
        for (uint32_t idx = lane; idx < row_vecs; idx += 64) acc = self_dot_vec<ELEM>(rp[idx], acc);
        const float s = wave_sum_f32(acc);
        if (lane == 0) norms[slot] = sqrtf(s);
    }
}

// Copies staged rows (native element type, densely packed) to their slots: one wave per row.
__global__ __launch_bounds__(256) void scatter_rows_kernel(u32x4* rows_out, uint32_t row_vecs,
                                                           const u32x4* staged, const uint64_t* slots,
                                                           uint64_t n) {
    const int lane = threadIdx.x & 63;
    const uint64_t W = (uint64_t)gridDim.x * 4;
    for (uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += W) {
        const u32x4* src = staged + i * row_vecs;
        u32x4* dst = rows_out + slots[i] * row_vecs;
        for (uint32_t idx = lane; idx < row_vecs; idx += 64) dst[idx] = src[idx];
    }
}

// ---- ingest ----------------------------------------------------------------------

__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if (f != f) return 0x7FC0u;
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

// one wave per staged row
// dest[i]: INGEST_SKIP = drop the row; INGEST_SIDE | j = row j of `side_out` (a densely packed
// side buffer of rows in the slab's element type); otherwise the slab slot.
__global__ __launch_bounds__(256) void ingest_f32_kernel(unsigned char* rows_out, unsigned char* side_out,
                                                         uint32_t row_vecs, uint32_t dim,
                                                         int32_t dtype, const float* staged,
                                                         const uint64_t* dest, uint64_t n) {
    const int lane = threadIdx.x & 63;
    const uint64_t W = (uint64_t)gridDim.x * 4;
    for (uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += W) {
        const float* src = staged + i * dim;
        const uint64_t d = dest[i];
        if (d & INGEST_SKIP) continue;
        unsigned char* dst = ((d & INGEST_SIDE) ? side_out : rows_out) +
                             (d & ~(INGEST_SIDE | INGEST_SKIP)) * row_vecs * 16;
        if (dtype == VS_F32) {
            float* o = reinterpret_cast<float*>(dst);
            for (uint32_t e = lane; e < row_vecs * 4; e += 64) o[e] = e < dim ? src[e] : 0.0f;
        } else if (dtype == VS_BF16) {
            uint16_t* o = reinterpret_cast<uint16_t*>(dst);
            for (uint32_t e = lane; e < row_vecs * 8; e += 64)
                o[e] = e < dim ? (uint16_t)f32_to_bf16_rne(src[e]) : (uint16_t)0;
        } else {
            uint64_t* o = reinterpret_cast<uint64_t*>(dst);
            for (uint32_t e0 = 0; e0 < row_vecs * 128; e0 += 64) {
                const uint32_t e = e0 + lane;
                const uint64_t bits = __ballot(e < dim && src[e] > 0.0f);
                if (lane == 0) o[e0 >> 6] = bits;
            }
        }
    }
}

// ---- roaring -> dense mask ----------------------------------------------------------

// One workgroup per container; `words` was zeroed by the caller.
__global__ __launch_bounds__(256) void roaring_expand_kernel(const uint8_t* bytes,
                                                             const RoaringContainerDesc* desc,
                                                             uint32_t docid_base, uint64_t* words,
                                                             uint64_t n_words) {
    const RoaringContainerDesc d = desc[blockIdx.x];
    const int64_t word0 = ((int64_t)d.key << 10) - (int64_t)(docid_base >> 6);
    const uint8_t* p = bytes + d.offset;
    auto in_range = [&](int64_t wi) { return wi >= 0 && (uint64_t)wi < n_words; };
    if (d.kind == 1) {  // 1024 little-endian u64 words
        for (uint32_t i = threadIdx.x; i < 1024; i += blockDim.x) {
            uint64_t v = 0;
            for (int b = 0; b < 8; ++b) v |= (uint64_t)p[i * 8 + b] << (8 * b);
            if (v && in_range(word0 + i)) atomicOr((unsigned long long*)&words[word0 + i], v);
        }
    } else if (d.kind == 0) {  // sorted u16 values
        for (uint32_t i = threadIdx.x; i < d.count; i += blockDim.x) {
            const uint32_t v = (uint32_t)p[2 * i] | ((uint32_t)p[2 * i + 1] << 8);
            const int64_t wi = word0 + (v >> 6);
            if (in_range(wi)) atomicOr((unsigned long long*)&words[wi], 1ull << (v & 63));
        }
    } else {  // runs: (start, length - 1) u16 pairs
        for (uint32_t i = threadIdx.x; i < d.count; i += blockDim.x) {
            const uint32_t s = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8);
            const uint32_t e = s + ((uint32_t)p[4 * i + 2] | ((uint32_t)p[4 * i + 3] << 8));
            for (uint32_t wq = s >> 6; wq <= (e >> 6) && wq < 1024; ++wq) {
                const uint32_t lo = wq == (s >> 6) ? (s & 63) : 0;
                const uint32_t hi = wq == (e >> 6) ? (e & 63) : 63;
                const uint64_t m = (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1)) & ~((1ull << lo) - 1);
                if (in_range(word0 + wq)) atomicOr((unsigned long long*)&words[word0 + wq], m);
            }
        }
    }
}

// ---- launchers ------------------------------------------------------------------------

// Default grid of the Hamming scan (and of the float scan over rows that one wave per SIMD streams
// at full rate, below), measured on 50M x 1536-bit and 10M x 768 f32 rows: an unfiltered scan is
// fastest with one workgroup per CU (one wave per SIMD); with a candidate mask two workgroups per
// CU win, because one wave's mask decoding then overlaps another's row loads.
static uint32_t light_grid_blocks(const SlabView& s, const ScanTuning& t, bool masked) {
    if (t.blocks > 0) return (uint32_t)t.blocks;
    const uint64_t want = (s.n_slots + 1023) / 1024;  // >= 256 slots per wave
    const uint64_t cap = masked ? 2 * NUM_CUS : NUM_CUS;
    return (uint32_t)(want < 1 ? 1 : want > cap ? cap : want);
}

// Default grid of the scans in which several queries share a pass (scan_multi_kernel,
// scan_hamming_multi_kernel): every row costs four or eight queries' arithmetic, so one wave per
// SIMD waits on itself; two workgroups per CU with or without a mask (10M x 768 f32, eight queries
// per pass: 6.17 ms with 256 workgroups, 4.84 with 512, 4.91 with 1024; 384-d f32 5.40 / 3.56 /
// 3.68; 768-d bf16, four per pass, 5.09 / 3.37 / 3.47).
static uint32_t shared_grid_blocks(const SlabView& s, const ScanTuning& t) {
    return light_grid_blocks(s, t, true);
}

// Whether ONE wave per SIMD can keep HBM busy with the single-query float scan over rows of this
// shape. A row costs its loads plus a fixed tail (reduction, distance, list test), so short rows
// and bf16 rows (two conversions per pair of elements) are more instructions per byte, and one wave
// then waits on its own arithmetic. Measured dense, one against two waves per SIMD, TB/s:
// 3 KiB rows 7.19 / 7.00 (768 f32) and 7.09 / 6.94 (1536 bf16); 2 KiB 7.29 / 6.87 (512 f32) but
// 6.63 / 6.87 (1024 bf16); 1.5 KiB 5.79 / 6.90 (384 f32) and 5.15 / 6.83 (768 bf16).
static bool float_rows_stream_with_one_wave(const SlabView& s) {
    return s.row_vecs >= 192 || (s.dtype == VS_F32 && s.row_vecs >= 128);
}

// Default grid of a single-query scan. Float rows that one wave per SIMD streams at full rate
// take the grid above (and pair their workgroups under a mask, see scan_float_rows); other float
// rows take two workgroups per CU with or without a mask, and four from 1 KiB per row down
// (768-byte rows: 2.7 / 4.9 / 6.4 / 5.8 TB/s with 1 / 2 / 4 / 8 workgroups per CU, 512-byte rows
// 2.0 / 3.7 / 5.4 / 5.5; the same order under a 10 % filter).
uint32_t scan_grid_blocks(const SlabView& s, const ScanTuning& t, bool masked) {
    if (t.blocks > 0) return (uint32_t)t.blocks;
    if ((s.dtype != VS_F32 && s.dtype != VS_BF16) || float_rows_stream_with_one_wave(s))
        return light_grid_blocks(s, t, masked);
    const uint64_t want = (s.n_slots + 1023) / 1024;  // >= 256 slots per wave
    const uint64_t cap = (s.row_vecs <= 64 ? 4 : 2) * NUM_CUS;
    return (uint32_t)(want < 1 ? 1 : want > cap ? cap : want);
}

void set_device_cus(uint32_t n) {
    if (n >= 1 && n <= 1024) g_num_cus = n;
}
uint32_t device_cus() { return g_num_cus; }

uint64_t scan_partial_bytes(const SlabView& s, uint32_t k, const ScanTuning& t) {
    uint32_t m = 0;
    for (int masked = 0; masked < 2; ++masked) {
        const uint32_t a = scan_grid_blocks(s, t, masked != 0), b = shared_grid_blocks(s, t);
        m = a > m ? a : m;
        m = b > m ? b : m;
    }
    return (uint64_t)m * k * sizeof(uint64_t);
}

uint64_t scan_workspace_bytes(const SlabView& s, uint32_t k) {
    const uint64_t dflt = scan_partial_bytes(s, k, ScanTuning{0, 1});
    const uint64_t tuned = (uint64_t)MAX_TUNED_BLOCKS * k * sizeof(uint64_t);
    return dflt > tuned ? dflt : tuned;
}

template <int ELEM, int NCH, int U>
static hipError_t launch_scan_t(const ScanArgs& a, uint32_t blocks, uint32_t lanes, size_t lds, const ScanTuning& t,
                                hipStream_t stream) {
    const dim3 grid(blocks, lanes), block(WAVES * 64);
    if (a.metric == VS_EUCLIDEAN)
        hipLaunchKernelGGL((scan_euclidean_kernel<ELEM, NCH, U>), grid, block, lds, stream, a);
    else if (a.metric == VS_DOT)
        hipLaunchKernelGGL((scan_dot_kernel<ELEM, NCH, U>), grid, block, lds, stream, a);
    else if (t.nt)
        hipLaunchKernelGGL((scan_angular_kernel<ELEM, NCH, U, true>), grid, block, lds, stream, a);
    else
        hipLaunchKernelGGL((scan_angular_kernel<ELEM, NCH, U, false>), grid, block, lds, stream, a);
    return hipGetLastError();
}

template <int ELEM>
static hipError_t launch_scan_e(const ScanArgs& a, uint32_t blocks, uint32_t lanes, size_t lds, const ScanTuning& t,
                                hipStream_t stream) {
    const uint32_t nch = (a.row_vecs + 63) / 64;
    if (nch <= 1) return launch_scan_t<ELEM, 1, 8>(a, blocks, lanes, lds, t, stream);
    if (nch <= 2) {
        if (a.metric == VS_ANGULAR) {  // schedule probes, angular only
            if (t.rows_in_flight == 4) return launch_scan_t<ELEM, 2, 4>(a, blocks, lanes, lds, t, stream);
            if (t.rows_in_flight == 16) return launch_scan_t<ELEM, 2, 16>(a, blocks, lanes, lds, t, stream);
        }
        return launch_scan_t<ELEM, 2, 8>(a, blocks, lanes, lds, t, stream);
    }
    if (nch <= 3) {
        if (a.metric == VS_ANGULAR) {  // schedule probes, angular only
            if (t.rows_in_flight == 2) return launch_scan_t<ELEM, 3, 2>(a, blocks, lanes, lds, t, stream);
            if (t.rows_in_flight == 8) return launch_scan_t<ELEM, 3, 8>(a, blocks, lanes, lds, t, stream);
        }
        return launch_scan_t<ELEM, 3, 4>(a, blocks, lanes, lds, t, stream);
    }
    if (nch <= 4) return launch_scan_t<ELEM, 4, 4>(a, blocks, lanes, lds, t, stream);
    if (nch <= 6) return launch_scan_t<ELEM, 6, 2>(a, blocks, lanes, lds, t, stream);
    if (nch <= 8) return launch_scan_t<ELEM, 8, 2>(a, blocks, lanes, lds, t, stream);
    if (nch <= 12) return launch_scan_t<ELEM, 12, 1>(a, blocks, lanes, lds, t, stream);
    if (nch <= 16) return launch_scan_t<ELEM, 16, 1>(a, blocks, lanes, lds, t, stream);
    return hipErrorInvalidValue;
}

// Query lanes of a many-query launch (gridDim.y, see split_queries): four by default, which the
// workgroups themselves reduce to one where the filter is dense.
static uint32_t scan_query_lanes(uint32_t n_q, const ScanTuning& t) {
    const uint32_t lanes = t.query_lanes > 0 ? (uint32_t)t.query_lanes : 4u;
    return lanes < n_q ? lanes : n_q;
}

hipError_t launch_scan_float_queries(const SlabView& s, const float* d_queries, uint32_t query_stride,
                                     uint32_t n_q, const uint64_t* cand, const ScanJob* d_jobs,
                                     uint32_t k, uint64_t* partial, const ScanTuning& t, hipStream_t stream) {
    if (k == 0 || k > VS_MAX_K || (s.n_slots & 63) || s.n_slots == 0 || n_q == 0) return hipErrorInvalidValue;
    const bool masked = cand != nullptr || d_jobs != nullptr;
    ScanArgs a;
    a.rows = reinterpret_cast<const u32x4*>(s.rows);
    a.norms = s.norms;
    a.present = s.present;
    a.cand = cand;
    a.query = d_queries;
    a.partial = partial;
    a.n_slots = s.n_slots;
    a.row_vecs = s.row_vecs;
    a.dim = s.dim;
    a.docid_base = s.docid_base;
    a.k = k;
    a.metric = s.metric;
    if (s.metric != VS_ANGULAR && s.metric != VS_EUCLIDEAN && s.metric != VS_DOT)
        return hipErrorInvalidValue;
    if (s.metric == VS_ANGULAR && !s.norms) return hipErrorInvalidValue;
    const uint32_t blocks = scan_grid_blocks(s, t, masked);
    // the default grid under a filter is two workgroups per CU: let them pair up (see scan_float_rows)
    a.pair = (masked && t.blocks <= 0 && blocks >= 2 && !(blocks & 1) && float_rows_stream_with_one_wave(s))
                 ? 1u
                 : 0u;
    const uint32_t lanes = scan_query_lanes(n_q, t);
    a.n_q = n_q;
    a.query_stride = query_stride;
    a.partial_stride = (uint64_t)blocks * k;
    a.jobs = d_jobs;
    a.list_bufs = (n_q > 1 && k <= (uint32_t)WaveTopK::REG_K) ? 2u : 1u;
    const size_t lds = (size_t)a.list_bufs * WAVES * k * 8 + (size_t)WAVES * QCAP * 4 +
                       (WaveTopK::needs_park_buffer(k) ? (size_t)WAVES * 128 * 8 : 0);
    if (s.dtype == VS_F32) return launch_scan_e<VS_F32>(a, blocks, lanes, lds, t, stream);
    if (s.dtype == VS_BF16) return launch_scan_e<VS_BF16>(a, blocks, lanes, lds, t, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_scan_float(const SlabView& s, const float* d_query, const uint64_t* cand,
                               uint32_t k, uint64_t* partial, uint64_t* out_keys,
                               uint32_t* out_ids, float* out_dist, uint32_t* out_n,
                               const ScanTuning& t, hipStream_t stream) {
    const hipError_t e = launch_scan_float_queries(s, d_query, 0, 1, cand, nullptr, k, partial, t, stream);
    if (e != hipSuccess) return e;
    if (!out_keys && !out_ids) return hipSuccess;  // partial lists only: the caller merges
    const uint32_t blocks = scan_grid_blocks(s, t, cand != nullptr);
    return launch_merge(partial, blocks, 1, blocks, k, s.metric, out_keys, out_ids, out_dist,
                        out_n, stream);
}

template <int NI, int U>
static hipError_t launch_hamming_t(const HammingArgs& a, uint32_t blocks, uint32_t lanes, size_t lds,
                                   const ScanTuning& t, hipStream_t stream) {
    const dim3 grid(blocks, lanes), block(WAVES * 64);
    if (t.nt)
        hipLaunchKernelGGL((scan_hamming_kernel<NI, U, true>), grid, block, lds, stream, a);
    else
        hipLaunchKernelGGL((scan_hamming_kernel<NI, U, false>), grid, block, lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_scan_hamming_queries(const SlabView& s, const void* d_queries, uint32_t query_stride,
                                       uint32_t n_q, const uint64_t* cand, const ScanJob* d_jobs,
                                       uint32_t k, uint64_t* partial, const ScanTuning& t, hipStream_t stream) {
    if (k == 0 || k > VS_MAX_K || (s.n_slots & 63) || s.n_slots == 0 || s.dtype != VS_BIN1 || n_q == 0)
        return hipErrorInvalidValue;
    const bool masked = cand != nullptr || d_jobs != nullptr;
    HammingArgs a;
    a.rows = reinterpret_cast<const u32x4*>(s.rows);
    a.present = s.present;
    a.cand = cand;
    a.query = reinterpret_cast<const u32x4*>(d_queries);
    a.partial = partial;
    a.n_slots = s.n_slots;
    a.row_vecs = s.row_vecs;
    a.docid_base = s.docid_base;
    a.k = k;
    // top-100 under a filter: an insert at a time was 30 to 40 us of every launch (50 M x 1536 bit)
    // (lists beyond the registers, k > 128, cannot take a key at a time at all)
    a.park = k > 32 ? 1u : 0u;
    const uint32_t blocks = scan_grid_blocks(s, t, masked);
    // the default grid under a filter is two workgroups per CU: let them pair up where the rows
    // have the whole-word path (1536 bits)
    a.pair = (masked && t.blocks <= 0 && blocks >= 2 && !(blocks & 1) && s.row_vecs == 12) ? 1u : 0u;
    const uint32_t lanes = scan_query_lanes(n_q, t);
    a.n_q = n_q;
    a.query_stride = query_stride;
    a.partial_stride = (uint64_t)blocks * k;
    a.jobs = d_jobs;
    a.list_bufs = (n_q > 1 && k <= (uint32_t)WaveTopK::REG_K) ? 2u : 1u;
    const size_t lds = (size_t)a.list_bufs * WAVES * k * 8 + (size_t)WAVES * QCAP * 4 + (a.park ? (size_t)WAVES * 128 * 8 : 0);
    const uint32_t ni = (s.row_vecs + 3) / 4;
    if (ni <= 1) return launch_hamming_t<1, 4>(a, blocks, lanes, lds, t, stream);
    if (ni <= 2) return launch_hamming_t<2, 4>(a, blocks, lanes, lds, t, stream);
    if (ni <= 3) {
        if (t.rows_in_flight == 2) return launch_hamming_t<3, 2>(a, blocks, lanes, lds, t, stream);
        if (t.rows_in_flight == 8) return launch_hamming_t<3, 8>(a, blocks, lanes, lds, t, stream);
        return launch_hamming_t<3, 4>(a, blocks, lanes, lds, t, stream);
    }
    if (ni <= 4) return launch_hamming_t<4, 2>(a, blocks, lanes, lds, t, stream);
    if (ni <= 8) return launch_hamming_t<8, 2>(a, blocks, lanes, lds, t, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_scan_hamming(const SlabView& s, const void* d_query, const uint64_t* cand,
                               uint32_t k, uint64_t* partial, uint64_t* out_keys,
                               uint32_t* out_ids, float* out_dist, uint32_t* out_n,
                               const ScanTuning& t, hipStream_t stream) {
    const hipError_t e = launch_scan_hamming_queries(s, d_query, 0, 1, cand, nullptr, k, partial, t, stream);
    if (e != hipSuccess) return e;
    if (!out_keys && !out_ids) return hipSuccess;  // partial lists only: the caller merges
    const uint32_t blocks = scan_grid_blocks(s, t, cand != nullptr);
    return launch_merge(partial, blocks, 1, blocks, k, VS_HAMMING, out_keys, out_ids, out_dist,
                        out_n, stream);
}

bool scan_multi_serves(const SlabView& s, uint32_t k) {
    if (s.metric != VS_ANGULAR && s.metric != VS_EUCLIDEAN && s.metric != VS_DOT) return false;
    const uint32_t nch = (s.row_vecs + 63) / 64;
    return k >= 1 && k <= (uint32_t)WaveTopK::REG_K && nch <= (s.dtype == VS_F32 ? 6u : 3u);
}

// Queries one pass takes: eight for f32 rows of up to 768 dimensions (the query slices of
// eight still fit the register file), four otherwise.
uint32_t scan_multi_queries(const SlabView& s) {
    return s.dtype == VS_F32 && (s.row_vecs + 63) / 64 <= 3 ? 8u : 4u;
}

template <int METRIC, int ELEM>
static hipError_t launch_multi_e(const MultiArgs& m, uint32_t blocks, size_t lds, bool eight,
                                 hipStream_t stream) {
    const dim3 grid(blocks), block(WAVES * 64);
    const uint32_t nch = (m.s.row_vecs + 63) / 64;
    if (ELEM == VS_F32 && eight && nch <= 3) {
        if (nch <= 1) hipLaunchKernelGGL((scan_multi_kernel<METRIC, VS_F32, 1, 4, 8>), grid, block, lds, stream, m);
        else if (nch <= 2) hipLaunchKernelGGL((scan_multi_kernel<METRIC, VS_F32, 2, 4, 8>), grid, block, lds, stream, m);
        else hipLaunchKernelGGL((scan_multi_kernel<METRIC, VS_F32, 3, 4, 8>), grid, block, lds, stream, m);
        return hipGetLastError();
    }
    if (nch <= 1) hipLaunchKernelGGL((scan_multi_kernel<METRIC, ELEM, 1, 4, 4>), grid, block, lds, stream, m);
    else if (nch <= 2) hipLaunchKernelGGL((scan_multi_kernel<METRIC, ELEM, 2, 4, 4>), grid, block, lds, stream, m);
    else if (nch <= 3) hipLaunchKernelGGL((scan_multi_kernel<METRIC, ELEM, 3, 4, 4>), grid, block, lds, stream, m);
    else if (ELEM == VS_F32 && nch <= 4)
        hipLaunchKernelGGL((scan_multi_kernel<METRIC, VS_F32, 4, 2, 4>), grid, block, lds, stream, m);
    else if (ELEM == VS_F32 && nch <= 6)
        hipLaunchKernelGGL((scan_multi_kernel<METRIC, VS_F32, 6, 2, 4>), grid, block, lds, stream, m);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

template <int METRIC>
static hipError_t launch_multi_m(const MultiArgs& m, int32_t dtype, uint32_t blocks, size_t lds,
                                 bool eight, hipStream_t stream) {
    if (dtype == VS_F32) return launch_multi_e<METRIC, VS_F32>(m, blocks, lds, eight, stream);
    if (dtype == VS_BF16) return launch_multi_e<METRIC, VS_BF16>(m, blocks, lds, false, stream);
    return hipErrorInvalidValue;
}

hipError_t launch_scan_multi(const SlabView& s, const float* d_queries, uint32_t query_stride,
                             uint32_t n_q, const uint64_t* cand, uint32_t k, uint64_t* partial,
                             uint64_t* out_keys, uint32_t* out_ids, float* out_dist, uint32_t* out_n,
                             const ScanTuning& t, hipStream_t stream) {
    if (!scan_multi_serves(s, k) || n_q == 0 || n_q > scan_multi_queries(s) || (s.n_slots & 63) ||
        s.n_slots == 0 || (s.metric == VS_ANGULAR && !s.norms))
        return hipErrorInvalidValue;
    const bool eight = n_q > 4;  // a group of up to four runs the lighter kernel
    const uint32_t lists = eight ? 8 : 4;
    const uint32_t blocks = shared_grid_blocks(s, t);
    MultiArgs m;
    m.s.rows = reinterpret_cast<const u32x4*>(s.rows);
    m.s.norms = s.norms;
    m.s.present = s.present;
    m.s.cand = cand;
    m.s.query = d_queries;
    m.s.partial = partial;
    m.s.n_slots = s.n_slots;
    m.s.row_vecs = s.row_vecs;
    m.s.dim = s.dim;
    m.s.docid_base = s.docid_base;
    m.s.k = k;
    m.s.metric = s.metric;
    m.query_stride = query_stride;
    m.n_q = n_q;
    m.partial_stride = (uint64_t)blocks * k;
    const size_t lds = (size_t)lists * WAVES * k * 8 + (size_t)WAVES * QCAP * 4;
    hipError_t e;
    if (s.metric == VS_ANGULAR) e = launch_multi_m<VS_ANGULAR>(m, s.dtype, blocks, lds, eight, stream);
    else if (s.metric == VS_EUCLIDEAN) e = launch_multi_m<VS_EUCLIDEAN>(m, s.dtype, blocks, lds, eight, stream);
    else e = launch_multi_m<VS_DOT>(m, s.dtype, blocks, lds, eight, stream);
    if (e != hipSuccess) return e;
    // list r of workgroup b of query q: partial[q * blocks * k + r * blocks + b]
    return launch_merge_batch(partial, blocks, 1, blocks, n_q, (uint64_t)blocks * k, k, s.metric, out_keys,
                              out_ids, out_dist, out_n, stream);
}

bool scan_mq32_serves(const SlabView& s, uint32_t k) {
    return s.dtype == VS_F32 && (s.metric == VS_ANGULAR || s.metric == VS_DOT) && k >= 1 &&
           k <= MQ_MAX_K && s.row_vecs <= MQ_MAX_VECS;
}
uint32_t scan_mq32_blocks(const SlabView& s) {
    const uint64_t want = (s.n_slots + 1023) / 1024;
    return (uint32_t)(want < 1 ? 1 : want > NUM_CUS ? NUM_CUS : want);  // the query block fills a CU's LDS
}
// Stores large enough that a bound from 1/128 of the words pays for a pass of its own: with it the
// main pass inserts next to nothing (without, every tile's epilogue runs the divergent insertion).
static bool mq32_two_passes(const SlabView& s) { return s.n_slots >= (1u << 20); }
uint64_t scan_mq32_partial_bytes(const SlabView& s, uint32_t k) {
    const uint64_t lists = (uint64_t)MQ * (scan_mq32_blocks(s) + 1) * k;  // main pass (+ the sample's list)
    const uint64_t sample = mq32_two_passes(s) ? (uint64_t)MQ * scan_mq32_blocks(s) * k + (uint64_t)MQ * k : 0;
    return (lists + sample) * sizeof(uint64_t);
}

hipError_t launch_scan_mq32(const SlabView& s, const float* d_queries, uint32_t query_stride,
                            uint32_t n_q, const uint64_t* cand, uint32_t k, uint64_t* partial,
                            uint64_t* out_keys, uint32_t* out_ids, float* out_dist, uint32_t* out_n,
                            hipStream_t stream) {
    if (!scan_mq32_serves(s, k) || n_q == 0 || n_q > (uint32_t)MQ || (s.n_slots & 63) || s.n_slots == 0 ||
        (s.metric == VS_ANGULAR && !s.norms))
        return hipErrorInvalidValue;
    const uint32_t blocks = scan_mq32_blocks(s);
    const bool two = mq32_two_passes(s);
    Mq32Args m;
    m.s.rows = reinterpret_cast<const u32x4*>(s.rows);
    m.s.norms = s.norms;
    m.s.present = s.present;
    m.s.cand = cand;
    m.s.query = d_queries;
    m.s.n_slots = s.n_slots;
    m.s.row_vecs = s.row_vecs;
    m.s.dim = s.dim;
    m.s.docid_base = s.docid_base;
    m.s.k = k;
    m.s.metric = s.metric;
    m.query_stride = query_stride;
    m.n_q = n_q;
    const uint32_t nw = mq32_waves(s.row_vecs, k);
    const size_t lds = mq32_lds_bytes(s.row_vecs, k, nw);
    const bool ang = s.metric == VS_ANGULAR;
    void (*sample_kernel)(Mq32Args);
    void (*kernel)(Mq32Args);
    if (nw == 8) {
        sample_kernel = ang ? scan_mq32_kernel<VS_ANGULAR, 1, 8> : scan_mq32_kernel<VS_DOT, 1, 8>;
        kernel = two ? (ang ? scan_mq32_kernel<VS_ANGULAR, 2, 8> : scan_mq32_kernel<VS_DOT, 2, 8>)
                     : (ang ? scan_mq32_kernel<VS_ANGULAR, 0, 8> : scan_mq32_kernel<VS_DOT, 0, 8>);
    } else {
        sample_kernel = ang ? scan_mq32_kernel<VS_ANGULAR, 1, 4> : scan_mq32_kernel<VS_DOT, 1, 4>;
        kernel = two ? (ang ? scan_mq32_kernel<VS_ANGULAR, 2, 4> : scan_mq32_kernel<VS_DOT, 2, 4>)
                     : (ang ? scan_mq32_kernel<VS_ANGULAR, 0, 4> : scan_mq32_kernel<VS_DOT, 0, 4>);
    }
    hipError_t e = hipSuccess;
    for (auto* f : {sample_kernel, kernel}) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // workspace: main lists [MQ][blocks + 1][k] | sample lists [MQ][blocks][k] | sample top-k [MQ][k]
    uint64_t* main_lists = partial;
    uint64_t* sample_lists = partial + (uint64_t)MQ * (blocks + 1) * k;
    uint64_t* sample_top = sample_lists + (uint64_t)MQ * blocks * k;
    if (two) {
        m.s.partial = sample_lists;
        m.partial_stride = (uint64_t)blocks * k;
        m.bound = nullptr;
        m.words = 1;
        hipLaunchKernelGGL(sample_kernel, dim3(blocks), dim3(nw * 64), lds, stream, m);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        e = launch_merge_batch(sample_lists, blocks, k, 1, n_q, (uint64_t)blocks * k, k, s.metric, sample_top,
                               nullptr, nullptr, nullptr, stream);
        if (e != hipSuccess) return e;
    }
    m.s.partial = main_lists;
    m.partial_stride = (uint64_t)(blocks + 1) * k;
    m.bound = two ? sample_top : nullptr;
    m.words = two ? 2 : 0;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(nw * 64), lds, stream, m);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return launch_merge_batch(main_lists, blocks + (two ? 1 : 0), k, 1, n_q, (uint64_t)(blocks + 1) * k, k,
                              s.metric, out_keys, out_ids, out_dist, out_n, stream);
}

bool scan_hamming_multi_serves(const SlabView& s, uint32_t k) {
    return s.dtype == VS_BIN1 && k >= 1 && k <= (uint32_t)WaveTopK::REG_K && s.row_vecs <= 16;
}

hipError_t launch_scan_hamming_multi(const SlabView& s, const void* d_queries, uint32_t query_stride,
                                     uint32_t n_q, const uint64_t* cand, uint32_t k, uint64_t* partial,
                                     uint64_t* out_keys, uint32_t* out_ids, float* out_dist,
                                     uint32_t* out_n, const ScanTuning& t, hipStream_t stream) {
    if (!scan_hamming_multi_serves(s, k) || n_q == 0 || n_q > (uint32_t)HQ || (s.n_slots & 63) ||
        s.n_slots == 0)
        return hipErrorInvalidValue;
    const uint32_t blocks = shared_grid_blocks(s, t);
    HammingMultiArgs m;
    m.s.rows = reinterpret_cast<const u32x4*>(s.rows);
    m.s.present = s.present;
    m.s.cand = cand;
    m.s.query = reinterpret_cast<const u32x4*>(d_queries);
    m.s.partial = partial;
    m.s.n_slots = s.n_slots;
    m.s.row_vecs = s.row_vecs;
    m.s.docid_base = s.docid_base;
    m.s.k = k;
    m.query_stride = query_stride;
    m.n_q = n_q;
    m.partial_stride = (uint64_t)blocks * k;
    // (four lists per wave leave no registers for batched insertion: tried, the kernel spilled and a dense
    // pass took twice as long)
    const size_t lds = (size_t)HQ * WAVES * k * 8 + (size_t)WAVES * QCAP * 4;
    const dim3 grid(blocks), block(WAVES * 64);
    const uint32_t ni = (s.row_vecs + 3) / 4;
    if (ni <= 1) hipLaunchKernelGGL((scan_hamming_multi_kernel<1, 4>), grid, block, lds, stream, m);
    else if (ni <= 2) hipLaunchKernelGGL((scan_hamming_multi_kernel<2, 4>), grid, block, lds, stream, m);
    else if (ni <= 3) hipLaunchKernelGGL((scan_hamming_multi_kernel<3, 4>), grid, block, lds, stream, m);
    else hipLaunchKernelGGL((scan_hamming_multi_kernel<4, 2>), grid, block, lds, stream, m);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_merge_batch(partial, blocks, 1, blocks, n_q, (uint64_t)blocks * k, k, VS_HAMMING, out_keys,
                              out_ids, out_dist, out_n, stream);
}

hipError_t launch_merge(const uint64_t* keys, uint32_t n_lists, uint64_t list_stride,
                        uint64_t rank_stride, uint32_t k, int32_t metric, uint64_t* out_keys,
                        uint32_t* out_ids, float* out_dist, uint32_t* out_n, hipStream_t stream) {
    return launch_merge_batch(keys, n_lists, list_stride, rank_stride, 1, 0, k, metric, out_keys,
                              out_ids, out_dist, out_n, stream);
}

hipError_t launch_merge_batch(const uint64_t* keys, uint32_t n_lists, uint64_t list_stride,
                              uint64_t rank_stride, uint32_t n_queries, uint64_t query_stride,
                              uint32_t k, int32_t metric, uint64_t* out_keys, uint32_t* out_ids,
                              float* out_dist, uint32_t* out_n, hipStream_t stream, const uint32_t* gate) {
    if (k == 0 || k > VS_MAX_K || n_lists == 0 || n_queries == 0) return hipErrorInvalidValue;
    uint32_t n_ranks = k;
    // Few keys in one contiguous run (lists back to back, or the rank-major
    // partial buffer): treat every key as a list of its own, so that all of them
    // are requested at once instead of rank after rank.
    const uint64_t total = (uint64_t)n_lists * k;
    if (total <= 4096 &&
        ((rank_stride == 1 && list_stride == k) || (list_stride == 1 && rank_stride == n_lists))) {
        n_lists = (uint32_t)total;
        list_stride = 1;
        rank_stride = 0;
        n_ranks = 1;
    }
    const size_t lds = (size_t)4 * k * 8 + (WaveTopK::needs_park_buffer(k) ? (size_t)4 * 128 * 8 : 0);
    hipLaunchKernelGGL(merge_kernel, dim3(n_queries), dim3(256), lds, stream, keys,
                       n_lists, list_stride, rank_stride, n_ranks, query_stride, k, metric, out_keys,
                       out_ids, out_dist, out_n, gate);
    return hipGetLastError();
}

hipError_t launch_norms(const SlabView& s, float* norms, uint64_t slot0, uint64_t n,
                        hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint64_t want = (n + 3) / 4;
    const uint32_t blocks = (uint32_t)(want > 4096 ? 4096 : want);
    const u32x4* rows = reinterpret_cast<const u32x4*>(s.rows);
    if (s.dtype == VS_F32)
        hipLaunchKernelGGL(norms_kernel<VS_F32>, dim3(blocks), dim3(256), 0, stream, rows, s.row_vecs,
                           norms, slot0, n);
    else if (s.dtype == VS_BF16)
        hipLaunchKernelGGL(norms_kernel<VS_BF16>, dim3(blocks), dim3(256), 0, stream, rows,
                           s.row_vecs, norms, slot0, n);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_ingest_f32(const SlabView& s, void* rows_out, void* side_out, const float* staged,
                             const uint64_t* dest, uint64_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint64_t want = (n + 3) / 4;
    const uint32_t blocks = (uint32_t)(want > 4096 ? 4096 : want);
    hipLaunchKernelGGL(ingest_f32_kernel, dim3(blocks), dim3(256), 0, stream,
                       reinterpret_cast<unsigned char*>(rows_out), reinterpret_cast<unsigned char*>(side_out),
                       s.row_vecs, s.dim, s.dtype, staged, dest, n);
    return hipGetLastError();
}

hipError_t launch_scatter_rows(const SlabView& s, void* rows_out, const void* staged, const uint64_t* d_slots,
                               uint64_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint64_t want = (n + 3) / 4;
    const uint32_t blocks = (uint32_t)(want > 4096 ? 4096 : want);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(blocks), dim3(256), 0, stream,
                       reinterpret_cast<u32x4*>(rows_out), s.row_vecs, reinterpret_cast<const u32x4*>(staged),
                       d_slots, n);
    return hipGetLastError();
}

hipError_t launch_norms_slots(const SlabView& s, float* norms, const uint64_t* d_slots, uint64_t n,
                              hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint64_t want = (n + 3) / 4;
    const uint32_t blocks = (uint32_t)(want > 4096 ? 4096 : want);
    const u32x4* rows = reinterpret_cast<const u32x4*>(s.rows);
    if (s.dtype == VS_F32)
        hipLaunchKernelGGL(norms_slots_kernel<VS_F32>, dim3(blocks), dim3(256), 0, stream, rows, s.row_vecs,
                           norms, d_slots, n);
    else if (s.dtype == VS_BF16)
        hipLaunchKernelGGL(norms_slots_kernel<VS_BF16>, dim3(blocks), dim3(256), 0, stream, rows, s.row_vecs,
                           norms, d_slots, n);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_roaring_expand(const uint8_t* d_bytes, const RoaringContainerDesc* d_desc,
                                 uint32_t n_containers, uint32_t docid_base, uint64_t* words,
                                 uint64_t n_words, hipStream_t stream) {
    if (n_containers == 0) return hipSuccess;
    hipLaunchKernelGGL(roaring_expand_kernel, dim3(n_containers), dim3(256), 0, stream, d_bytes,
                       d_desc, docid_base, words, n_words);
    return hipGetLastError();
}

}  // namespace vs
