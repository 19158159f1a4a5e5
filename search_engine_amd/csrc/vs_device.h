// Device-side helpers shared by the scan kernels: wave-wide reductions on DPP,
// the 64-bit (distance, docid) sort key, and the per-wave sorted top-k list
// that lives in LDS. gfx950 only: a wavefront is 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr uint64_t KEY_NONE = 0xFFFFFFFFFFFFFFFFull;

// ---- cross-lane ------------------------------------------------------------

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
    // lanes whose row is not in ROW_MASK, or whose source lane is invalid, read 0
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}

// Sum of `v` over the 64 lanes, returned wave-uniform. Fixed summation tree:
// pairs, quads, 8, 16 inside each row of 16 lanes, then rows 0+1 and 2+3, then
// the two halves; the total is read from lane 63.
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_f32<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_f32<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_f32<0x140, 0xF>(v);  // row_mirror
    v += dpp_f32<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_f32<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += dpp_u32<0xB1, 0xF>(v);
    v += dpp_u32<0x4E, 0xF>(v);
    v += dpp_u32<0x141, 0xF>(v);
    v += dpp_u32<0x140, 0xF>(v);
    v += dpp_u32<0x142, 0xA>(v);
    v += dpp_u32<0x143, 0xC>(v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    const uint32_t lo = uniform_u32((uint32_t)v), hi = uniform_u32((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}

// Stops the compiler from moving LDS accesses of one wave across this point.
// The hardware returns one wave's LDS operations in issue order, so a wave that
// stores and then loads through this fence reads what its own lanes stored.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- sort key ----------------------------------------------------------------

// Monotone map f32 -> u32 for the total order OrderedFloat defines: -inf < ...
// < -0 == +0 < ... < +inf < NaN (every NaN equal). arroy orders neighbours by
// (OrderedFloat(distance), item id).
__device__ __forceinline__ uint32_t f32_orderable(float d) {
    if (d != d) return 0xFFC00000u;
    d += 0.0f;  // -0.0 -> +0.0
    const uint32_t u = __float_as_uint(d);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_orderable(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
__device__ __forceinline__ uint64_t make_key(uint32_t ordered_distance, uint32_t docid) {
    return ((uint64_t)ordered_distance << 32) | docid;
}

// ---- per-wave sorted list in LDS ---------------------------------------------

// (the merge of a sorted batch into an LDS list is WaveTopK::merge_parked_lds, below)

// lane i receives lane i-1's value (lane 0 receives `fill`): DPP wave_shr:1
__device__ __forceinline__ uint64_t wave_shift_up_u64(uint64_t v, uint64_t fill) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)fill, (int)(uint32_t)v,
                                                              0x138, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(
        (int)(uint32_t)(fill >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, false);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask, 64);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t min_u64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t max_u64(uint64_t a, uint64_t b) { return a < b ? b : a; }

// 128 keys per wave, element i in `lo` of lane i (i < 64) or `hi` of lane i - 64.
// A bitonic sequence becomes ascending: the last phase of a bitonic sort.
__device__ __forceinline__ void bitonic_merge128(uint64_t& lo, uint64_t& hi, int lane) {
    const uint64_t a = min_u64(lo, hi), b = max_u64(lo, hi);  // distance 64: inside the lane
    lo = a;
    hi = b;
#pragma unroll
    for (int j = 32; j >= 1; j >>= 1) {
        const uint64_t plo = shfl_xor_u64(lo, j), phi = shfl_xor_u64(hi, j);
        const bool upper = lane & j;
        lo = upper ? max_u64(lo, plo) : min_u64(lo, plo);
        hi = upper ? max_u64(hi, phi) : min_u64(hi, phi);
    }
}
// Any 128 keys become ascending (27 compare-exchange phases).
__device__ __forceinline__ void bitonic_sort128(uint64_t& lo, uint64_t& hi, int lane) {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int j = size >> 1; j >= 1; j >>= 1) {
            const uint64_t plo = shfl_xor_u64(lo, j), phi = shfl_xor_u64(hi, j);
            const bool upper = lane & j;
            const bool desc_lo = lane & size, desc_hi = (64 + lane) & size;  // blocks alternate direction
            lo = (upper != desc_lo) ? max_u64(lo, plo) : min_u64(lo, plo);
            hi = (upper != desc_hi) ? max_u64(hi, phi) : min_u64(hi, phi);
        }
    }
    bitonic_merge128(lo, hi, lane);  // lo ascending, hi descending: bitonic
}

// The n_parked keys of `parked` (a wave's 128-entry buffer in LDS) merged into the wave's ascending
// LDS list of k keys, in place; returns the list's new last entry. The batch is sorted in registers
// (entry `lane` and entry 64 + lane) and goes back to the buffer; every key's place in the merged list
// is its own index plus the number of keys of the OTHER sequence below it (keys are distinct: they
// carry the docid), found by binary search: each lane places two parked keys (11 probes of the list)
// and, 256 list entries at a time from the top down, four list entries (8 probes of the sorted batch).
// Entries only move up, and a group is read completely before any of it is written, so the list is
// rewritten where it lies. About 20 cycles per parked key at k = 1000, against some 1 600 for an insert
// that shifts the list. Not inlined: one copy per kernel, entered once per 128 candidates or so.
__device__ __attribute__((noinline)) inline uint64_t merge_parked_into_lds_list(uint64_t* lds, uint64_t* parked,
                                                                                uint32_t n_parked, int k) {
    const int lane = (int)(threadIdx.x & 63);
    uint64_t a = (uint32_t)lane < n_parked ? parked[lane] : KEY_NONE;
    uint64_t b = (uint32_t)lane + 64 < n_parked ? parked[lane + 64] : KEY_NONE;
    wave_lds_fence();
    bitonic_sort128(a, b, lane);
    parked[lane] = a;
    parked[lane + 64] = b;
    wave_lds_fence();
    auto below_in_list = [&](uint64_t key) {  // number of list entries < key
        int pos = 0;
#pragma unroll
        for (int s = 512; s >= 1; s >>= 1)
            if (pos + s - 1 < k && lds[pos + s - 1] < key) pos += s;
        if (pos < k && lds[pos] < key) ++pos;
        return pos;
    };
    const int pa = lane + below_in_list(a), pb = 64 + lane + below_in_list(b);
    // list entries below the smallest parked key stay where they are (lane 0 holds the smallest)
    const int first_base = ((int)uniform_u32((uint32_t)pa) >> 6) << 6;
    for (int top_base = ((k - 1) >> 6) << 6; top_base >= first_base; top_base -= 256) {
        uint64_t v[4];
        int at[4], np[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            at[u] = top_base - 64 * u + lane;
            const bool have = at[u] >= first_base && at[u] < k;
            v[u] = have ? lds[at[u]] : KEY_NONE;
            if (!have) at[u] = k;  // nowhere
            np[u] = 0;
        }
#pragma unroll
        for (int s = 64; s >= 1; s >>= 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (parked[np[u] + s - 1] < v[u]) np[u] += s;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (parked[np[u]] < v[u]) ++np[u];
        wave_lds_fence();
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (at[u] + np[u] < k && np[u] > 0) lds[at[u] + np[u]] = v[u];
        wave_lds_fence();
    }
    if (pa < k) lds[pa] = a;
    if (pb < k) lds[pb] = b;
    wave_lds_fence();
    return uniform_u64(lds[k - 1]);
}

// The k smallest keys one wave has seen, ascending. For k <= 128 the list lives in
// two register pairs (lane i holds the i-th and the (64+i)-th smallest) and an
// insert is two compares, two ballots and a cross-lane shift. Larger k (a request
// deep into its pagination: limit = offset + limit, src/search/search.rs:115) keep
// the list in LDS, and keys are never inserted one at a time there: they are parked
// in a 128-entry buffer and merged into the list a batch at a time (park_in must
// have been called). `tau` is the k-th smallest so far (KEY_NONE while fewer than k
// keys are held): only keys below it may be inserted.
// LDS_LISTS = false: for kernels that only ever run with k <= 128 (no code for the LDS list).
template <bool LDS_LISTS>
struct WaveTopKT {
    static constexpr int REG_K = 128;
    uint64_t* lds;  // k entries owned by this wave
    int k, lane;
    bool in_regs;
    __device__ __forceinline__ bool regs() const { return !LDS_LISTS || in_regs; }
    uint64_t lo, hi;  // entries lane and 64 + lane
    uint64_t tau;
    // Batched insertion (register lists with a larger k, where one insert costs a few hundred cycles
    // and a scan under a filter makes hundreds of them per wave: a wave that sees n candidates inserts
    // about k (1 + ln(n / k)) times): candidates below tau are parked in a 128-entry LDS buffer and,
    // 65 or more at a time, sorted and merged into the list in one go. tau moves only then, so a few
    // more candidates get parked than an insert at a time would have admitted; the list is the same.
    uint64_t* parked = nullptr;
    uint32_t n_parked = 0;  // wave-uniform

    __device__ __forceinline__ void init(uint64_t* lds_, int k_, int lane_) {
        lds = lds_;
        k = k_;
        lane = lane_;
        in_regs = !LDS_LISTS || k_ <= REG_K;
        lo = hi = KEY_NONE;
        tau = KEY_NONE;
        parked = nullptr;
        n_parked = 0;
        if (!regs()) {
            for (int i = lane; i < k; i += 64) lds[i] = KEY_NONE;
            wave_lds_fence();
        }
    }
    // whether a list of k entries needs a park buffer (128 keys of LDS per wave) to work at all
    static __host__ __device__ constexpr bool needs_park_buffer(uint32_t k_) { return k_ > (uint32_t)REG_K; }
    // `key` is wave-uniform and < tau
    __device__ __forceinline__ void insert(uint64_t key) {
        if (regs()) {
            const int pos = __popcll(__ballot(lo < key)) + __popcll(__ballot(hi < key));
            const uint64_t carry = readlane_u64(lo, 63);
            const uint64_t up_lo = wave_shift_up_u64(lo, KEY_NONE);
            const uint64_t up_hi = wave_shift_up_u64(hi, carry);
            lo = lane < pos ? lo : (lane == pos ? key : up_lo);
            hi = lane + 64 < pos ? hi : (lane + 64 == pos ? key : up_hi);
            if (lane >= k) lo = KEY_NONE;
            if (lane + 64 >= k) hi = KEY_NONE;
            tau = k <= 64 ? readlane_u64(lo, k - 1) : readlane_u64(hi, k - 65);
        } else {
            // parked; the list and tau move at the caller's next make_room()
            if (lane == 0) parked[n_parked] = key;
            ++n_parked;
        }
    }
    // LDS lists: call before up to `n` insert()s in a row (n <= 64), at a point where few registers
    // are live: this is where the parked keys are merged into the list (a function call).
    __device__ __forceinline__ void make_room(uint32_t n) {
        if (!regs() && n_parked + n > 128) merge_parked();
    }
    __device__ __forceinline__ void park_in(uint64_t* buffer128) {
        parked = buffer128;
        n_parked = 0;
    }
    __device__ __forceinline__ bool parks() const { return parked != nullptr; }
    // Every lane calls; lanes with `mine` hold a candidate `key` < tau.
    __device__ __forceinline__ void park(bool mine, uint64_t key) {
        const uint64_t m = __ballot(mine);
        if (!m) return;
        if (n_parked > 64) merge_parked();  // (keys parked one at a time by insert() may have filled it further)
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
        if (mine) parked[n_parked + below] = key;  // at most 64 parked before, at most 64 new: 128 fit
        n_parked += (uint32_t)__popcll(m);
        wave_lds_fence();
        if (regs() && n_parked > 64) merge_parked();  // (LDS lists merge at the next call: fewer, fuller batches)
    }
    __device__ __forceinline__ void merge_parked() {
        if (!n_parked) return;
        if (!regs()) {
            wave_lds_fence();
            tau = merge_parked_into_lds_list(lds, parked, n_parked, k);
            n_parked = 0;
            return;
        }
        uint64_t a = (uint32_t)lane < n_parked ? parked[lane] : KEY_NONE;
        uint64_t b = (uint32_t)lane + 64 < n_parked ? parked[lane + 64] : KEY_NONE;
        wave_lds_fence();
        n_parked = 0;
        bitonic_sort128(a, b, lane);
        // C[i] = min(list[i], parked[127 - i]) holds the 128 smallest of both and is bitonic (as in
        // absorb): entry 127 - i is `b` of lane 63 - i for i < 64, `a` of lane 127 - i beyond
        const int from = (63 - lane) * 4;
        const uint64_t rb = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)(b >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)b);
        const uint64_t ra = ((uint64_t)(uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)(a >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)(uint32_t)a);
        lo = min_u64(lo, rb);
        hi = min_u64(hi, ra);
        bitonic_merge128(lo, hi, lane);
        if (lane >= k) lo = KEY_NONE;
        if (lane + 64 >= k) hi = KEY_NONE;
        tau = k <= 64 ? readlane_u64(lo, k - 1) : readlane_u64(hi, k - 65);
    }
    // Register lists only, list still empty: takes 128 arbitrary keys (two per lane, KEY_NONE =
    // none) at the cost of one sort instead of up to 128 inserts.
    __device__ __forceinline__ void fill128(uint64_t a, uint64_t b) {
        lo = a;
        hi = b;
        bitonic_sort128(lo, hi, lane);
        if (lane >= k) lo = KEY_NONE;
        if (lane + 64 >= k) hi = KEY_NONE;
        tau = k <= 64 ? readlane_u64(lo, k - 1) : readlane_u64(hi, k - 65);
    }
    // after this the list can be read from `lds` (by this wave, or by others
    // after a workgroup barrier)
    __device__ __forceinline__ void spill() {
        if (!regs()) merge_parked();
        if (regs()) {
            if (lane < k) lds[lane] = lo;
            if (lane + 64 < k) lds[lane + 64] = hi;
            wave_lds_fence();
        }
    }
    // inserts every entry below tau of another wave's spilled list
    __device__ __forceinline__ void absorb(uint64_t* other) {
        if (regs()) {
            // Two ascending lists A (mine) and B: C[i] = min(A[i], B[127 - i]) holds the 128 smallest
            // of both and is bitonic; one merge phase sorts it. A handful of instructions against
            // one insert per entry.
            const int ia = 127 - lane, ib = 63 - lane;
            lo = min_u64(lo, ia < k ? other[ia] : KEY_NONE);
            hi = min_u64(hi, ib < k ? other[ib] : KEY_NONE);
            bitonic_merge128(lo, hi, lane);
            if (lane >= k) lo = KEY_NONE;
            if (lane + 64 >= k) hi = KEY_NONE;
            tau = k <= 64 ? readlane_u64(lo, k - 1) : readlane_u64(hi, k - 65);
            return;
        }
        // the other list is ascending: 64 entries at a time while any of them is below tau
        for (int base = 0; base < k; base += 64) {
            const uint64_t key = base + lane < k ? other[base + lane] : KEY_NONE;
            const bool mine = key < tau;
            if (!__ballot(mine)) break;
            park(mine, key);
        }
        merge_parked();
    }
    // i-th smallest for i = base + lane (base a multiple of 64), KEY_NONE for i >= k
    __device__ __forceinline__ uint64_t at_lane(int base) const {
        const int i = base + lane;
        if (i >= k) return KEY_NONE;
        if (regs()) return base == 0 ? lo : hi;
        return lds[i];
    }
};
using WaveTopK = WaveTopKT<true>;
using WaveTopKRegs = WaveTopKT<false>;

}  // namespace vs
