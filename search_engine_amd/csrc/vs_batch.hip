// Batched scan: up to 256 queries against every candidate row of a bf16 store in
// one pass over HBM. No counterpart in the reference (it searches one vector per
// request); it is the batch form of the same arroy Angular nns_by_vector result.
//
// Shape of the work: scores = Docs[N x D] * Queries[256 x D]^T is a real dense
// contraction (2*N*D*256 flop for N*D*2 bytes: ~256 flop/byte), so it runs on the
// matrix cores: v_mfma_f32_32x32x16_bf16, f32 accumulation. The N x 256 score
// matrix is never written: every 128-document tile is reduced on the spot into
// per-query top-k lists that live in LDS for the whole launch.
//
// One workgroup = 8 waves (two per SIMD) = one 128-document x 256-query tile at a
// time (wave tile 64 x 64 = 2 x 2 MFMA tiles, 64 accumulator registers);
// workgroups are persistent and walk the tiles round-robin. K is walked in steps of 64: the
// next step's 48 KiB (documents from HBM, queries from L2) are in flight in
// registers while the current step's MFMAs run out of an XOR-swizzled LDS image.
#include "vs_kernels.h"

#include "../../include/vs_abi.h"
#include "vs_device.h"

namespace vs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int BM = 128;      // documents per tile
constexpr int BN = 256;      // queries per launch (padded)
constexpr int BK = 64;       // bf16 elements per K step = 8 vectors of 16 bytes

// LDS carve (bytes). After a tile's K loop the A/B images are dead and the first
// 32 KiB of them hold one 32-row chunk of scores, f32 [32][256].
constexpr int OFF_A = 0;                       // [128][64] bf16, swizzled
constexpr int OFF_B = OFF_A + BM * BK * 2;     // [256][64] bf16, swizzled
constexpr int OFF_NORM = OFF_B + BN * BK * 2;  // f32[128] norms of the tile's rows
constexpr int OFF_LISTS = OFF_NORM + BM * 4;   // u64[k][256], rank-major
constexpr int OFF_SCORES = OFF_A;

struct BatchArgs {
    const u32x4* rows;
    const float* norms;
    const uint64_t* present;
    const uint64_t* cand;
    const u32x4* qrows;   // [256][ksteps * 8] bf16 vectors, zero padded
    const float* qnorm;   // [256]
    uint64_t* partial;    // [n_q][gridDim.x][k]
    uint64_t n_slots;
    uint32_t row_vecs;
    uint32_t ksteps;
    uint32_t docid_base;
    uint32_t k;
    uint32_t n_q;
};

// byte offset of 16-byte slot `slot` (0..7) of row `row` in a [rows][64] bf16 tile
__device__ __forceinline__ uint32_t swz(uint32_t row, uint32_t slot) {
    return row * 128u + ((slot ^ (row & 7u)) << 4);
}

constexpr int BTHREADS = 512;

__global__ __launch_bounds__(BTHREADS) void scan_batch_bf16_kernel(BatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wv = tid >> 6;
    const uint32_t wm = wv >> 2, wn = wv & 3;  // wave tile origin: rows wm*64, cols wn*64
    const uint32_t r32 = lane & 31, half = lane >> 5;
    const int k = (int)a.k;

    float* snorm = reinterpret_cast<float*>(smem + OFF_NORM);
    float* scores = reinterpret_cast<float*>(smem + OFF_SCORES);
    volatile uint64_t* lists = reinterpret_cast<volatile uint64_t*>(smem + OFF_LISTS);

    // thread `tid` < 256 owns query `tid`: its list, its k-th best (tau) and the
    // matching pre-test threshold cthr
    const bool owner = tid < a.n_q;
    const float my_qn = owner ? a.qnorm[tid] : 0.0f;
    const float my_marg = my_qn * 1e-6f;
    uint64_t my_tau = KEY_NONE;
    float my_cthr = -__builtin_inff();
    if (tid < (uint32_t)BN)
        for (int r = 0; r < k; ++r) lists[r * BN + tid] = KEY_NONE;
    __syncthreads();

    const uint64_t n_words = a.n_slots >> 6;  // n_slots is a multiple of 64: the last tile may be half
    const uint32_t n_tiles = (uint32_t)((n_words + 1) / 2);
    const uint32_t KS = a.ksteps;

    // Document vectors are requested TWO K steps ahead of their use and query vectors one
    // step ahead, across tile boundaries: (tile, ks) is one flat stream per workgroup.
    auto load_a = [&](uint32_t tile, uint32_t ks, u32x4 (&dst)[2]) {
        const uint64_t row0 = (uint64_t)tile * BM;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const uint32_t id = (uint32_t)it * BTHREADS + tid, row = id >> 3, slot = id & 7;
            const uint32_t v = ks * 8 + slot;
            dst[it] = (tile < n_tiles && v < a.row_vecs && row0 + row < a.n_slots)
                          ? __builtin_nontemporal_load(a.rows + (row0 + row) * a.row_vecs + v)
                          : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto load_b = [&](uint32_t ks, u32x4 (&dst)[4]) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t id = (uint32_t)it * BTHREADS + tid, row = id >> 3, slot = id & 7;
            dst[it] = a.qrows[(uint64_t)row * (KS * 8) + ks * 8 + slot];
        }
    };
    // position p of the stream = K step (p % KS) of tile blockIdx.x + (p / KS) * gridDim.x
    auto load_a_at = [&](uint32_t tile, uint32_t ks, uint32_t ahead, u32x4 (&dst)[2]) {
        uint32_t k2 = ks + ahead, t2 = tile;
        while (k2 >= KS) {
            k2 -= KS;
            t2 += gridDim.x;
        }
        load_a(t2, k2, dst);
    };

    u32x4 ga0[2], ga1[2], gb[4];
    load_a_at(blockIdx.x, 0, 0, ga0);
    load_a_at(blockIdx.x, 0, 1, ga1);
    load_b(0, gb);

    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t row0 = (uint64_t)tile * BM;
        const bool has1 = 2 * (uint64_t)tile + 1 < n_words;
        uint64_t w0 = a.present[2 * (uint64_t)tile], w1 = has1 ? a.present[2 * (uint64_t)tile + 1] : 0;
        if (a.cand) {
            w0 &= a.cand[2 * (uint64_t)tile];
            if (has1) w1 &= a.cand[2 * (uint64_t)tile + 1];
        }
        w0 = uniform_u64(w0);
        w1 = uniform_u64(w1);
        if ((w0 | w1) == 0) {  // no candidate in this tile: restart the stream at the next one
            load_a_at(tile + gridDim.x, 0, 0, ga0);
            load_a_at(tile + gridDim.x, 0, 1, ga1);
            continue;
        }

        f32x16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;
        if (tid < BM) snorm[tid] = row0 + tid < a.n_slots ? a.norms[row0 + tid] : 0.0f;

        auto k_step = [&](uint32_t ks, u32x4 (&ga)[2]) {
            __syncthreads();  // everyone is done reading the previous step's image
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const uint32_t id = (uint32_t)it * BTHREADS + tid;
                *reinterpret_cast<u32x4*>(smem + OFF_A + swz(id >> 3, id & 7)) = ga[it];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const uint32_t id = (uint32_t)it * BTHREADS + tid;
                *reinterpret_cast<u32x4*>(smem + OFF_B + swz(id >> 3, id & 7)) = gb[it];
            }
            __syncthreads();
            load_a_at(tile, ks, 2, ga);                   // two steps ahead, in flight under the MFMAs
            load_b(ks + 1 < KS ? ks + 1 : 0, gb);         // one step ahead (step 0 again for the next tile)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const uint32_t slot = (uint32_t)kk * 2 + half;
                bf16x8 fa[2], fb[2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    fa[m] = __builtin_bit_cast(
                        bf16x8, *reinterpret_cast<const u32x4*>(smem + OFF_A + swz(wm * 64 + m * 32 + r32, slot)));
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    fb[n] = __builtin_bit_cast(
                        bf16x8, *reinterpret_cast<const u32x4*>(smem + OFF_B + swz(wn * 64 + n * 32 + r32, slot)));
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
            }
        };
        // Even steps consume ga0, odd steps ga1. With an odd step count the roles of the two
        // registers swap at every tile; `flip` keeps track.
        for (uint32_t ks = 0; ks < KS; ks += 2) {
            k_step(ks, ga0);
            if (ks + 1 < KS) k_step(ks + 1, ga1);
        }
        if (KS & 1) {  // odd: the next tile's step 0 sits in ga1, its step 1 in ga0 -> swap
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const u32x4 t = ga0[it];
                ga0[it] = ga1[it];
                ga1[it] = t;
            }
        }

        // ---- epilogue: which of this tile's 128 x 256 scores beat a query's k-th best? ----
        // The scores go through LDS in four chunks of 32 rows; the thread that owns a query
        // walks its column. Pre-test without a division: d < dthr  <=>  pq > (1 - 2 dthr) * qn * pn,
        // taken with a margin so that rounding can only let too many through; what passes is
        // judged exactly (arroy's formula, full (distance, docid) key) and inserted.
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            __syncthreads();  // the K loop, or the previous chunk's readers, are done with this LDS
            if (wm == (uint32_t)(c >> 1)) {
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const uint32_t row = (uint32_t)(e & 3) + 8u * (uint32_t)(e >> 2) + 4u * half;
                        scores[row * BN + wn * 64 + (uint32_t)n * 32 + r32] = acc[c & 1][n][e];
                    }
            }
            __syncthreads();
            if (owner) {
                const uint32_t bits = (uint32_t)((c < 2 ? w0 : w1) >> ((c & 1) * 32));
                // all loads first, then one 32-bit mask of the rows that may beat tau
                uint32_t pass = 0;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4v pn4 = *reinterpret_cast<const f32x4v*>(&snorm[c * 32 + g * 4]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = scores[(g * 4 + j) * BN + tid];
                        if (!(v < my_cthr * pn4[j] - pn4[j] * my_marg)) pass |= 1u << (g * 4 + j);
                    }
                }
                pass &= bits;
                while (pass) {  // rare after the first tiles
                    const uint32_t r = (uint32_t)__builtin_ctz(pass);
                    pass &= pass - 1;
                    const float v = scores[r * BN + tid];
                    const float pn = snorm[c * 32 + r];
                    if (v < my_cthr * pn - pn * my_marg) continue;  // tau moved since the mask was taken
                    const float pnqn = pn * my_qn;
                    float d = 0.0f;
                    if (pnqn != 0.0f) d = (1.0f - v / pnqn) / 2.0f;
                    const uint64_t key = make_key(f32_orderable(d), a.docid_base + (uint32_t)(row0 + c * 32 + r));
                    if (key < my_tau) {
                        int pos = k - 1;
                        while (pos > 0 && lists[(pos - 1) * BN + tid] > key) {
                            lists[pos * BN + tid] = lists[(pos - 1) * BN + tid];
                            --pos;
                        }
                        lists[pos * BN + tid] = key;
                        my_tau = lists[(k - 1) * BN + tid];
                        if (my_tau != KEY_NONE)
                            my_cthr = (1.0f - 2.0f * f32_from_orderable((uint32_t)(my_tau >> 32))) * my_qn;
                    }
                }
            }
        }
    }

    if (tid < a.n_q)
        for (int r = 0; r < k; ++r)
            a.partial[((uint64_t)tid * gridDim.x + blockIdx.x) * k + r] = lists[r * BN + tid];
}

// queries f32 [n_q][dim] -> bf16 rows [256][kvecs * 8] (round to nearest even, zero
// padded) and the norm of each ROUNDED query; one wave per query row
__global__ __launch_bounds__(256) void prep_queries_bf16_kernel(const float* queries, uint32_t n_q,
                                                                uint32_t dim, uint32_t kelems,
                                                                uint16_t* qrows, float* qnorm) {
    const int lane = threadIdx.x & 63;
    const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= (uint32_t)BN) return;
    float acc = 0.0f;
    for (uint32_t e = lane; e < kelems; e += 64) {
        uint16_t h = 0;
        if (j < n_q && e < dim) {
            const float f = queries[(uint64_t)j * dim + e];
            uint32_t u = __float_as_uint(f);
            h = (f != f) ? (uint16_t)0x7FC0 : (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
            const float r = __uint_as_float((uint32_t)h << 16);
            acc = __builtin_fmaf(r, r, acc);
        }
        qrows[(uint64_t)j * kelems + e] = h;
    }
    const float s = wave_sum_f32(acc);
    if (lane == 0) qnorm[j] = sqrtf(s);
}

uint32_t batch_grid_blocks(const SlabView& s) {
    const uint64_t tiles = (s.n_slots + BM - 1) / BM;
    return (uint32_t)(tiles < 1 ? 1 : tiles > 256 ? 256 : tiles);
}

uint64_t batch_workspace_bytes(const SlabView& s, uint32_t k) {
    const uint32_t ksteps = (s.row_vecs + 7) / 8;
    return (uint64_t)BN * ksteps * BK * 2 + (uint64_t)BN * 4 +
           (uint64_t)BN * batch_grid_blocks(s) * k * 8 + 256;
}

hipError_t launch_scan_batch_bf16(const SlabView& s, const float* d_queries, uint32_t n_q,
                                  const uint64_t* cand, uint32_t k, void* workspace,
                                  uint64_t* out_keys, uint32_t* out_ids, float* out_dist,
                                  uint32_t* out_n, hipStream_t stream) {
    if (s.dtype != VS_BF16 || k == 0 || k > VS_BATCH_MAX_K || n_q == 0 || n_q > (uint32_t)BN ||
        (s.n_slots & 63) != 0 || s.n_slots == 0)
        return hipErrorInvalidValue;
    const uint32_t ksteps = (s.row_vecs + 7) / 8;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    uint16_t* qrows = reinterpret_cast<uint16_t*>(ws);
    float* qnorm = reinterpret_cast<float*>(ws + (size_t)BN * ksteps * BK * 2);
    uint64_t* partial = reinterpret_cast<uint64_t*>(ws + (size_t)BN * ksteps * BK * 2 + (size_t)BN * 4 + 128);
    hipLaunchKernelGGL(prep_queries_bf16_kernel, dim3(BN / 4), dim3(256), 0, stream, d_queries, n_q, s.dim,
                       ksteps * BK, qrows, qnorm);
    BatchArgs a;
    a.rows = reinterpret_cast<const u32x4*>(s.rows);
    a.norms = s.norms;
    a.present = s.present;
    a.cand = cand;
    a.qrows = reinterpret_cast<const u32x4*>(qrows);
    a.qnorm = qnorm;
    a.partial = partial;
    a.n_slots = s.n_slots;
    a.row_vecs = s.row_vecs;
    a.ksteps = ksteps;
    a.docid_base = s.docid_base;
    a.k = k;
    a.n_q = n_q;
    const uint32_t blocks = batch_grid_blocks(s);
    const size_t lds = (size_t)OFF_LISTS + (size_t)k * BN * 8;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(scan_batch_bf16_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(scan_batch_bf16_kernel, dim3(blocks), dim3(BTHREADS), lds, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_merge_batch(partial, blocks, k, 1, n_q, (uint64_t)blocks * k, k, VS_ANGULAR, out_keys,
                              out_ids, out_dist, out_n, stream);
}

}  // namespace vs
