// Batched scan: up to 256 queries against every candidate row of a bf16 store in
// one pass over HBM. No counterpart in the reference (it searches one vector per
// request); it is the batch form of the same arroy Angular nns_by_vector result.
//
// Shape of the work: scores = Docs[N x D] * Queries[256 x D]^T is a real dense
// contraction (2*N*D*256 flop for N*D*2 bytes: 256 flop/byte), so it runs on the
// matrix cores: v_mfma_f32_32x32x16_bf16 (16x16x32 in the streaming pass), f32 accumulation. The N x 256 score
// matrix is never written: every 256-document tile is reduced on the spot into
// per-query top-k lists that live in LDS for the whole launch.
//
// One workgroup = 8 waves (two per SIMD) = one 256-document x 256-query tile at a
// time (wave tile 64 x 128 = 2 x 4 MFMA tiles, 128 accumulator registers); a
// workgroup walks its tiles in ascending order. K is walked in steps of 64: the
// next step's 64 KiB (documents from HBM, queries from L2) are in flight while
// the current step's MFMAs run out of an XOR-swizzled LDS image. A 256-row tile
// re-reads the queries half as often as a 128-row tile would; that re-read and
// the documents themselves, not the matrix cores, are what bound the kernel:
// at 256 queries the contraction has 256 flop per document byte, below the
// machine's ratio of matrix-core rate to HBM rate.
//
// Two builds of the kernel differ in how a step reaches LDS. DMA = true (k <= 12):
// buffer loads write straight into one of TWO images (no staging registers, one
// barrier per step, the next step's loads fly during the whole MFMA phase).
// DMA = false (larger k: the lists need the LDS of the second image): the step
// is staged in 32 registers per thread and stored into the one image between
// two barriers.
//
// Selection. A launch may be given, per query, an upper bound on its k-th best
// key (`bound_keys`): then almost no score survives, and the test runs on the
// accumulator registers themselves -- t = score - |row| * c2[query] >= 0, where
// c2 is the query's bound turned into a dot-product threshold, with a margin so
// that rounding can only let too many through. Survivors go through a small LDS
// queue to the thread that owns the query, which judges them exactly (arroy's
// formula, full (distance, docid) key) and inserts them into its sorted list.
// Without a bound, or when a tile overflows the queue, the tile's scores go
// through LDS in four chunks of 64 rows and every owner walks its column.
//
// The launcher runs two passes: 256 tiles spread evenly over the store without a
// bound, a merge that turns their lists into the per-query k-th best of that
// sample, and the rest of the store with it as the bound. With a candidate mask and rows of
// three steps or more the tiles are not tiles of the store but of the list of candidates: the
// streaming pass fetches each tile's 256 rows from wherever they lie (its gathered build), so a
// filter costs what its candidates cost.
#include "vs_kernels.h"

#include <stdlib.h>

#include "../../include/vs_abi.h"
#include "vs_device.h"

namespace vs {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int BM = 256;      // documents per tile
constexpr int BN = 256;      // queries per launch (padded)
constexpr int BK = 64;       // bf16 elements per K step = 8 vectors of 16 bytes
constexpr int BTHREADS = 512;
constexpr int QCAP = 256;    // survivors one tile may queue before it falls back to the chunked walk
constexpr int SAMPLE_TILES = 256;  // tiles of the first (unbounded) pass

// LDS carve (bytes): `images` images of 64 KiB (A [256][64] bf16 at +0, B [256][64] bf16 at
// +32 KiB, both swizzled), then the small arrays, then the lists. After a tile's K loop the
// image it used last is dead and holds scores: one 64-row chunk f32 [64][256], or 8 KiB per wave.
constexpr int IMG_BYTES = (BM + BN) * BK * 2;
constexpr int IMG_B = BM * BK * 2;
constexpr int OFF_NORM = 0;                       // f32[256] norms of the tile's rows (NaN: not a candidate)
constexpr int OFF_C2 = OFF_NORM + BM * 4;         // f32[256] per-query pre-test threshold
constexpr int OFF_QUEUE = OFF_C2 + BN * 4;        // u64[QCAP]: score bits << 32 | query << 8 | row
constexpr int OFF_QCOUNT = OFF_QUEUE + QCAP * 8;  // u32[2], used by alternate tiles (16 bytes reserved)
constexpr int OFF_QNORM = OFF_QCOUNT + 16;        // f32[256] query norms
constexpr int OFF_BOUND = OFF_QNORM + BN * 4;     // u64[256] the bound each query started with
constexpr int OFF_LISTS = OFF_BOUND + BN * 8;     // u64[k][256], rank-major
constexpr uint32_t DMA_MAX_K = 12;

constexpr size_t batch_lds_bytes(bool dma, uint32_t k) {
    return (size_t)(dma ? 2 : 1) * IMG_BYTES + OFF_LISTS + (size_t)k * BN * 8;
}
static_assert(batch_lds_bytes(true, DMA_MAX_K) <= 160 * 1024 && batch_lds_bytes(false, VS_BATCH_MAX_K) <= 160 * 1024,
              "LDS carve exceeds 160 KiB");

struct BatchArgs {
    const u32x4* rows;
    const float* norms;
    const uint64_t* present;
    const uint64_t* cand;
    const u32x4* qrows;          // [256][ksteps * 8] bf16 vectors, zero padded
    const float* qnorm;          // [256]
    const uint64_t* bound_keys;  // nullable: [n_q][k] ascending keys of an earlier pass over OTHER rows
    uint64_t* partial;           // [n_q][lists][k]; list b = this launch's workgroup b
    uint64_t n_slots;
    uint32_t row_vecs;
    uint32_t ksteps;
    uint32_t docid_base;
    uint32_t k;
    uint32_t n_q;
    // Tiles of 256 rows. The sample = tiles 0, stride, 2*stride, ... (sample_count of them).
    // pass 1: workgroup b scans sample tile b (gridDim.x == sample_count); pass 2: workgroup b
    // scans the non-sample tiles among b, b + gridDim.x, ...
    uint32_t n_tiles, sample_stride, sample_count, pass;
    uint32_t lists;            // gridDim.x, + 1 when bound_keys is carried along as a list of its own
    const uint32_t* gate;      // nullable: the launch does nothing unless *gate != 0 (the rare-case second pass)
};

// Byte offset of 16-byte slot `slot` (0..7) of row `row` in a [rows][64] bf16 image.
// ds_read_b128 serves lanes {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32) together:
// with the slot XORed by (row >> 1) & 7 the 16 rows of each group hit 16 different
// 16-byte bank groups, and the 8 lanes that store one row still cover its 8 slots.
__device__ __forceinline__ uint32_t swz(uint32_t row, uint32_t slot) {
    return row * 128u + ((slot ^ ((row >> 1) & 7u)) << 4);
}

// Returns x, but the compiler cannot see that: whatever is computed from the result is
// computed where it is used, instead of being kept in a register across the K loop.
__device__ __forceinline__ uint32_t opaque(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}

// FIRST = true is the build for the unbounded first pass: gridDim.x == number of tiles, one tile
// per workgroup, so a query's list only exists after the K loop and lives in its owner's
// registers (insertion is a chain of compare-exchanges, no LDS round trips). FIRST = false keeps
// the lists in LDS across tiles; there insertions are rare.
template <bool DMA, bool FIRST>
__global__ __launch_bounds__(BTHREADS) void scan_batch_bf16_kernel(BatchArgs a) {
    constexpr int KMAX = DMA ? (int)DMA_MAX_K : (int)VS_BATCH_MAX_K;  // k <= KMAX in this build
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
    if (a.gate && *a.gate == 0) return;  // wave-uniform
    unsigned char* const smem = lds_all + (DMA ? 2 : 1) * IMG_BYTES;  // the small arrays and the lists
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wv = uniform_u32(tid >> 6);
    const uint32_t wm = wv >> 1, wn = wv & 1;  // wave tile origin: rows wm*64, cols wn*128
    const uint32_t r32 = lane & 31, half = lane >> 5;
    const int k = (int)a.k;

    float* snorm = reinterpret_cast<float*>(smem + OFF_NORM);
    float* c2 = reinterpret_cast<float*>(smem + OFF_C2);
    uint64_t* queue = reinterpret_cast<uint64_t*>(smem + OFF_QUEUE);
    uint32_t* qcounts = reinterpret_cast<uint32_t*>(smem + OFF_QCOUNT);
    // (plain pointers: a volatile one would be demoted to flat, system-scope accesses)
    uint64_t* lists = reinterpret_cast<uint64_t*>(smem + OFF_LISTS);

    // Thread `tid` < n_q owns query `tid`: its list, the key an entry must beat (tau =
    // min(bound, k-th best so far)) and the matching pre-test threshold.
    const bool owner = tid < a.n_q;
    // (the norm and the bound are kept in LDS and re-read where they are needed: registers are
    // dear in the K loop)
    float* sqn = reinterpret_cast<float*>(smem + OFF_QNORM);
    uint64_t* sbound = reinterpret_cast<uint64_t*>(smem + OFF_BOUND);
    uint64_t my_tau = (owner && a.bound_keys) ? a.bound_keys[(uint64_t)tid * k + (k - 1)] : KEY_NONE;
    uint32_t tile_docid0 = 0;  // first docid of the tile being judged (owners only need it for a zero query)
    int my_cnt = 0;  // entries in my list
    // pq >= c2 * pn is necessary for a row to beat tau: d < dthr <=> pq > (1 - 2 dthr) qn pn.
    auto threshold = [&](float qn) -> float {
        if (my_tau == KEY_NONE) return -3.0e38f;
        // a zero query is at distance 0 from everything: once tau is set only a lower docid beats it
        if (qn == 0.0f) return tile_docid0 > (uint32_t)my_tau ? 3.0e38f : -3.0e38f;
        const float c = (1.0f - 2.0f * f32_from_orderable((uint32_t)(my_tau >> 32))) * qn - qn * 1e-6f;
        return c == c ? c : -3.0e38f;
    };
    float my_c2;
    bool bad_query;
    {
        const float qn = owner ? a.qnorm[tid] : 0.0f;
        if (tid < (uint32_t)BN) {
            sqn[tid] = qn;
            sbound[tid] = my_tau;
        }
        my_c2 = threshold(qn);
        bad_query = owner && !(qn < __builtin_inff());
    }
    if (tid < (uint32_t)BN) {
        if constexpr (!FIRST)
            for (int r = 0; r < k; ++r) lists[r * BN + tid] = KEY_NONE;
        c2[tid] = owner ? my_c2 : 3.0e38f;
    }
    // FIRST: ascending, slots >= k hold 0 so that nothing ever sinks into them
    uint64_t reg_list[FIRST ? KMAX : 1];
#pragma unroll
    for (int r = 0; r < (FIRST ? KMAX : 1); ++r) reg_list[r] = r < k ? KEY_NONE : 0;
    // a query with a non-finite norm defeats the register test: every tile then takes the chunked walk
    const uint32_t qcount0 = __syncthreads_or(bad_query) ? (uint32_t)QCAP + 1u : 0u;
    if (tid < 2) qcounts[tid] = qcount0;
    uint32_t par = 0;  // which of the two survivor counters this tile uses; the other one is being reset

    const uint64_t n_words = a.n_slots >> 6;
    const uint32_t KS = a.ksteps;
    const bool col_ok[4] = {wn * 128 + 0 * 32 + r32 < a.n_q, wn * 128 + 1 * 32 + r32 < a.n_q,
                            wn * 128 + 2 * 32 + r32 < a.n_q, wn * 128 + 3 * 32 + r32 < a.n_q};

    // judge one surviving score exactly and insert it into this thread's list
    auto consider = [&](uint64_t row0, uint32_t row, float v) {
        const float pn = snorm[row];
        const float my_qn = sqn[tid];
        const float pnqn = pn * my_qn;
        float d = 0.0f;
        if (pnqn != 0.0f) d = (1.0f - v / pnqn) / 2.0f;
        const uint64_t key = make_key(f32_orderable(d), a.docid_base + (uint32_t)(row0 + row));
        if (key < my_tau) {
            uint64_t kth = KEY_NONE;
            if constexpr (FIRST) {
                uint64_t sink = key;  // sinks to its place, pushing every larger entry one slot down
#pragma unroll
                for (int r = 0; r < KMAX; ++r) {
                    const bool lt = sink < reg_list[r];
                    const uint64_t lo = lt ? sink : reg_list[r], hi = lt ? reg_list[r] : sink;
                    reg_list[r] = lo;
                    sink = hi;
                    if (r == k - 1) kth = lo;
                }
            } else {
                int pos = my_cnt < k ? my_cnt : k - 1;  // slots past my_cnt are empty
                my_cnt = pos + 1;
                while (pos > 0 && lists[(pos - 1) * BN + tid] > key) {
                    lists[pos * BN + tid] = lists[(pos - 1) * BN + tid];
                    --pos;
                }
                lists[pos * BN + tid] = key;
                if (my_cnt < k) return;  // the list is not full: its k-th entry, hence tau, has not moved
                kth = lists[(k - 1) * BN + tid];
            }
            const uint64_t my_bound = sbound[tid];
            my_tau = kth < my_bound ? kth : my_bound;
            my_c2 = threshold(my_qn);
            c2[tid] = my_c2;
        }
    };

    // One K step of one tile = 64 KiB: slot (16 bytes) s of row r goes to byte swz(r, s) of its
    // image. Buffer loads: a wave-uniform descriptor and scalar offset plus a 32-bit per-thread
    // offset. A tile past the end gets an empty descriptor (reads zeros, touches nothing); the K
    // tail of the last step is pushed out of range the same way. Rows past n_slots inside the
    // last tile lie in the slab's padding: whatever they hold, they are not candidates.
    const uint32_t row_bytes = a.row_vecs * 16u;
    const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4*>(a.qrows), 0, (int)((uint32_t)BN * KS * 128u), 0x00020000);
    auto doc_rsrc = [&](uint32_t tile) {
        const bool live = tile < a.n_tiles;
        return __builtin_amdgcn_make_buffer_rsrc(
            const_cast<u32x4*>(a.rows + (live ? (uint64_t)tile * BM * a.row_vecs : 0)), 0,
            live ? (int)((uint32_t)BM * row_bytes) : 0, 0x00020000);
    };
    // DMA = false: thread t stages slot (t & 7) of rows (t >> 3) + 64 * it, it = 0..3, of both images
    u32x4 ga[DMA ? 1 : 4], gb[DMA ? 1 : 4];
    const uint32_t ld_row = tid >> 3, ld_slot = tid & 7;
    const uint32_t voff_a = ld_row * row_bytes + ld_slot * 16u;
    const uint32_t voff_b = ld_row * (KS * 128u) + ld_slot * 16u;
    const uint32_t st_off = swz(ld_row, ld_slot);  // + it * 64 rows * 128 bytes: the swizzle repeats every 16 rows
    // DMA = true: one load instruction fills 1 KiB of LDS in lane order = 8 rows of an image; wave
    // w issues pieces 4w..4w+3 of each image. Lane l therefore fetches, for row (l >> 3) of the
    // piece, the slot that belongs at physical position (l & 7): (l & 7) ^ ((row >> 1) & 7), which
    // for piece 4w+i is (l & 7) ^ ((i & 1) * 4 + (l >> 4)).
    const uint32_t dslot[2] = {(lane & 7) ^ (lane >> 4), (lane & 7) ^ (4 + (lane >> 4))};
    const uint32_t dva[2] = {(lane >> 3) * row_bytes + dslot[0] * 16u, (lane >> 3) * row_bytes + dslot[1] * 16u};
    const uint32_t dvb[2] = {(lane >> 3) * (KS * 128u) + dslot[0] * 16u, (lane >> 3) * (KS * 128u) + dslot[1] * 16u};
    auto load_step = [&](uint32_t tile, uint32_t ks, uint32_t image) {
        const __amdgpu_buffer_rsrc_t ares = doc_rsrc(tile);
        if constexpr (DMA) {
            typedef __attribute__((address_space(3))) void* lds_ptr;
            unsigned char* img = lds_all + image * IMG_BYTES + wv * 4096;
            const uint32_t kill[2] = {ks * 8 + dslot[0] < a.row_vecs ? 0u : 0x40000000u,
                                      ks * 8 + dslot[1] < a.row_vecs ? 0u : 0x40000000u};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(qres, (lds_ptr)(img + IMG_B + i * 1024), 16, (int)dvb[i & 1],
                                                         (int)(ks * 128u + (wv * 4 + i) * 8u * (KS * 128u)), 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)  // nt: every document byte is read once per launch
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ares, (lds_ptr)(img + i * 1024), 16, (int)(dva[i & 1] | kill[i & 1]),
                                                         (int)(ks * 128u + (wv * 4 + i) * 8u * row_bytes), 0, 2);
        } else {
            const uint32_t va = voff_a | (ks * 8 + ld_slot < a.row_vecs ? 0u : 0x40000000u);
#pragma unroll
            for (int it = 0; it < 4; ++it)
                gb[it] = __builtin_amdgcn_raw_buffer_load_b128(qres, (int)voff_b,
                                                               (int)(ks * 128u + (uint32_t)it * 64u * (KS * 128u)), 0);
#pragma unroll
            for (int it = 0; it < 4; ++it)  // nt: every document byte is read once per launch
                ga[it] = __builtin_amdgcn_raw_buffer_load_b128(ares, (int)va,
                                                               (int)(ks * 128u + (uint32_t)it * 64u * row_bytes), 2);
        }
    };

    const uint32_t frag_a = swz(wm * 64 + r32, half), frag_b = IMG_B + swz(wn * 128 + r32, half);
    uint32_t image = 0;  // DMA: the image the step about to be computed lands in
    bool ahead = false;  // DMA: step 1 of the coming tile is already in flight (into the other image)

    // this workgroup's tiles, ascending
    auto skip_sample = [&](uint32_t t) {
        while (t < a.n_tiles && t % a.sample_stride == 0 && t / a.sample_stride < a.sample_count) t += gridDim.x;
        return t;
    };
    auto tile_after = [&](uint32_t t) { return a.pass == 1 ? a.n_tiles : skip_sample(t + gridDim.x); };
    uint32_t tile = a.pass == 1 ? blockIdx.x * a.sample_stride : skip_sample(blockIdx.x);
    load_step(tile, 0, image);
    // lane j < 4: candidate word j of tile t (present & filter), 0 past the end
    auto load_words = [&](uint32_t t) -> uint64_t {
        const uint64_t wi = 4 * (uint64_t)t + (lane & 3);
        uint64_t x = 0;
        if (t < a.n_tiles && wi < n_words) {
            x = a.present[wi];
            if (a.cand) x &= a.cand[wi];
        }
        return x;
    };
    uint64_t w_ahead = load_words(tile);

    for (; tile < a.n_tiles;) {
        const uint32_t next_tile = tile_after(tile);
        const uint64_t row0 = (uint64_t)tile * BM;
        if (owner && sqn[tid] == 0.0f) {  // a zero query's threshold depends on where the tile lies
            tile_docid0 = a.docid_base + (uint32_t)row0;
            const float c = threshold(0.0f);
            if (c != my_c2) c2[tid] = my_c2 = c;  // (read again only after this tile's barriers)
        }
        // The tile's four candidate words (a wave's 64 rows are exactly one of them) were
        // requested a tile ago by lanes 0..3; the next tile's are requested now and arrive
        // under this tile's K loop.
        uint64_t w[4];
        uint64_t any = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w[j] = readlane_u64(w_ahead, j);
            any |= w[j];
        }
        w_ahead = load_words(next_tile);
        if (any == 0) {  // no candidate in this tile: restart the stream at the next one
            if constexpr (DMA) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the same image is about to be refilled
            ahead = false;
            load_step(next_tile, 0, image);
            tile = next_tile;
            continue;
        }
        uint32_t* qcount = qcounts + par;

        f32x16 acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;

        for (uint32_t ks = 0; ks < KS; ++ks) {
            if constexpr (DMA) {  // my pieces of this step are in LDS (step 1's eight may still be on their way)
                if (ks == 0 && ahead)
                    __builtin_amdgcn_s_waitcnt(0x0F78);  // vmcnt(8)
                else
                    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            }
            // everyone's pieces are; the previous step's readers, or the previous tile's epilogue, are done
            __syncthreads();
            if (ks == 0) {
                if (tid == 0) qcounts[par ^ 1] = qcount0;  // for the next tile; nobody reads it before then
                if (tid < (uint32_t)BM) {
                    // Rows that are not candidates get a NaN norm: their t is NaN, which the running
                    // maximum ignores, and the chunked walk masks them by the candidate bits anyway.
                    const bool is_cand = (w[tid >> 6] >> (tid & 63)) & 1;  // w[] index is wave-uniform
                    const float pn = is_cand ? a.norms[row0 + tid] : 0.0f;
                    snorm[tid] = is_cand ? pn : __builtin_nanf("");
                    if (!(pn < __builtin_inff())) atomicAdd(qcount, (uint32_t)QCAP + 1u);  // as for a non-finite query
                }
            }
            const unsigned char* img = lds_all;
            if constexpr (DMA) {
                img += image * IMG_BYTES;
                image ^= 1u;
                if (ks == 0 && ahead)
                    ahead = false;  // this tile's step 1 was requested during the previous tile's drain
                else if (ks + 1 < KS)
                    load_step(tile, ks + 1, image);
                else
                    load_step(next_tile, 0, image);
            } else {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    *reinterpret_cast<u32x4*>(lds_all + it * 64 * 128 + st_off) = ga[it];
                    *reinterpret_cast<u32x4*>(lds_all + IMG_B + it * 64 * 128 + st_off) = gb[it];
                }
                __syncthreads();
                if (ks + 1 < KS)
                    load_step(tile, ks + 1, 0);
                else
                    load_step(next_tile, 0, 0);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                // slot kk*2 + half of a row: the swizzled offset of slot `half`, XOR kk*32 bytes
                uint32_t fo_a = frag_a ^ ((uint32_t)kk << 5), fo_b = frag_b ^ ((uint32_t)kk << 5);
                asm volatile("" : "+v"(fo_a), "+v"(fo_b));  // recompute per step, do not keep 8 addresses live
                bf16x8 fa[2], fb[4];
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    fa[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + m * 32 * 128 + fo_a));
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    fb[n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + n * 32 * 128 + fo_b));
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
            }
        }
        par ^= 1u;
        // the image the last step used is dead now (after a barrier) and serves the epilogue
        unsigned char* const epi = lds_all + (DMA ? (image ^ 1u) * IMG_BYTES : 0u);

        // ---- register test: t = score - |row| * c2[query], per query column the largest t ----
        // accumulator element e of MFMA tile (m, n) is row m*32 + (e&3) + 8*(e>>2) + 4*half,
        // column n*32 + r32 of the wave tile
        const bool reg_test = a.bound_keys != nullptr;  // without a bound everything would survive
        if (reg_test) {
            const float* c2_mine = reinterpret_cast<const float*>(smem + OFF_C2 + opaque((wn * 128 + r32) * 4));
            const float* norm_mine = reinterpret_cast<const float*>(smem + OFF_NORM + opaque((wm * 64 + 4 * half) * 4));
            float c2v[4], tmax[2][4];  // per MFMA tile (m, n): the largest t of this lane's 16 rows
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                c2v[n] = c2_mine[n * 32];
                tmax[0][n] = tmax[1][n] = -__builtin_inff();
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4v pn4 = *reinterpret_cast<const f32x4v*>(norm_mine + m * 32 + 8 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            tmax[m][n] = __builtin_fmaxf(tmax[m][n], __builtin_fmaf(-pn4[j], c2v[n], acc[m][n][g * 4 + j]));
                }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const bool lane_hit = tmax[m][n] >= 0.0f && col_ok[n];
                    if (__ballot(lane_hit) == 0) continue;  // wave-uniform
                    // rare: which of this lane's 16 rows (bit e), then queue them
                    uint32_t hit = 0;
                    if (lane_hit && __hip_atomic_load(qcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= (uint32_t)QCAP) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float t = __builtin_fmaf(-norm_mine[m * 32 + (e & 3) + 8 * (e >> 2)], c2v[n], acc[m][n][e]);
                            hit |= (t >= 0.0f) ? (1u << e) : 0u;
                        }
                    }
                    while (hit) {
                        const uint32_t e = (uint32_t)__builtin_ctz(hit);
                        hit &= hit - 1;
                        float v = acc[m][n][0];
#pragma unroll
                        for (int j = 1; j < 16; ++j) v = (e == (uint32_t)j) ? acc[m][n][j] : v;
                        const uint32_t wrow = (uint32_t)m * 32 + (e & 3) + 8u * (e >> 2) + 4u * half;
                        const uint32_t idx = atomicAdd(qcount, 1u);
                        if (idx < (uint32_t)QCAP)
                            queue[idx] = ((uint64_t)__float_as_uint(v) << 32) |
                                         ((wn * 128 + (uint32_t)n * 32 + r32) << 8) | (wm * 64 + wrow);
                    }
                }
        }
        __syncthreads();
        const uint32_t n_queued = reg_test ? *qcount : (uint32_t)QCAP + 1u;
        if (n_queued <= (uint32_t)QCAP) {
            if constexpr (DMA && !FIRST) {
                // Nobody reads an image any more (the walk below would): the one the last step used
                // can already take step 1 of the coming tile while the survivors are drained.
                if (KS > 1) {
                    load_step(next_tile, 1, image ^ 1u);
                    ahead = true;
                }
            }
            if (owner) {
#pragma unroll 1
                for (uint32_t i = 0; i < n_queued; i += 4) {  // (entries past n_queued are stale, never out of bounds)
                    uint64_t e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = queue[(i + j) & (QCAP - 1)];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (i + j < n_queued && (((uint32_t)e[j] >> 8) & 0xFFFFFFu) == tid)
                            consider(row0, (uint32_t)e[j] & 0xFFu, __uint_as_float((uint32_t)(e[j] >> 32)));
                }
            }
            tile = next_tile;
            continue;  // the next tile's first barrier closes this epilogue
        }

        // ---- chunked walk: the scores go through LDS, 64 rows at a time; the thread that owns
        // a query walks its column, pre-tests without a division and judges what passes ----
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c) __syncthreads();  // the previous chunk's readers are done with this LDS
            if (wm == (uint32_t)c) {
                // element (m, n, e) -> scores[m*32 + (e&3) + 8*(e>>2) + 4*half][wn*128 + n*32 + r32]
                float* dst = reinterpret_cast<float*>(epi + opaque((4 * half * BN + wn * 128 + r32) * 4));
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            dst[(m * 32 + (e & 3) + 8 * (e >> 2)) * BN + n * 32] = acc[m][n][e];
            }
            __syncthreads();
            if (owner) {
                const float* col = reinterpret_cast<const float*>(epi + opaque(tid * 4));
                const float* norm_c = reinterpret_cast<const float*>(smem + OFF_NORM + opaque((uint32_t)c * 64 * 4));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t bits = (uint32_t)(w[c] >> (h * 32));
                    // all loads first, then one 32-bit mask of the rows that may beat tau
                    uint32_t pass = 0;
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const f32x4v pn4 = *reinterpret_cast<const f32x4v*>(norm_c + h * 32 + g * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = col[(h * 32 + g * 4 + j) * BN];
                            if (!(v < my_c2 * pn4[j])) pass |= 1u << (g * 4 + j);
                        }
                    }
                    pass &= bits;
                    while (pass) {
                        const uint32_t r = (uint32_t)h * 32 + (uint32_t)__builtin_ctz(pass);
                        pass &= pass - 1;
                        const float v = col[r * BN];
                        if (v < my_c2 * snorm[c * 64 + r]) continue;  // tau moved since the mask was taken
                        consider(row0, (uint32_t)c * 64 + r, v);
                    }
                }
            }
        }
        if constexpr (FIRST) break;  // one tile per workgroup
        tile = next_tile;
    }

    if (tid < a.n_q) {
        uint64_t* out = a.partial + (uint64_t)tid * a.lists * k;
        if constexpr (FIRST) {
#pragma unroll
            for (int r = 0; r < KMAX; ++r)
                if (r < k) out[(uint64_t)blockIdx.x * k + r] = reg_list[r];
        } else {
            for (int r = 0; r < k; ++r) out[(uint64_t)blockIdx.x * k + r] = lists[r * BN + tid];
        }
        if (a.bound_keys && blockIdx.x == 0)  // the earlier pass's list rides along as list gridDim.x
            for (int r = 0; r < k; ++r) out[(uint64_t)gridDim.x * k + r] = a.bound_keys[(uint64_t)tid * k + r];
    }
}

// ---- the streaming second pass ------------------------------------------------------------------
//
// The second pass does not need the generality of the kernel above: every query has a bound from
// the sample, so almost no score survives; the bound does not tighten inside one workgroup's
// 1/256 of the rows; and survivors can be judged later. Its K loop is the one above (64-element
// steps, two 64 KiB images filled by LDS-DMA, one barrier per step); what goes is everything that
// made the stream stand still between tiles:
//  * no per-query lists in LDS, no owner threads, no drain: a survivor (score, query, row) goes to
//    a small LDS queue, and wave 0 appends the PREVIOUS tile's queue to the workgroup's own region
//    in global memory while the others multiply (it keeps the region's fill in a register: plain
//    stores, no atomics). Later launches sort the regions into per-query buckets
//    (batch_bucket_kernel), judge the buckets exactly and merge them with the sample's list
//    (batch_select_kernel);
//  * nothing in the loop returns a value from vector memory, so the compiler never waits for
//    loads that are still wanted in flight: the next tile's norms arrive by LDS-DMA, tile ids and
//    candidate words through the scalar cache (constant address space);
//  * the tiles to visit (a candidate outside the sample) come as a list made by a tiny launch, so a
//    workgroup always knows its next tile and the first steps of tile j + 1 are requested during
//    the last steps of tile j without any test.
// Anything the scheme cannot hold — a query without a bound or with a zero / non-finite norm, a
// non-finite row norm, a full queue or region — raises `overflow`, and the launcher's gated rerun
// of the general kernel produces the result instead.
//
// Tried and measured (10 M x 768, 256 queries; in-kernel stamps, cycles per 256-row tile): a ring
// of 32-element sub-step images (6 document, 3 query buffers: 80 KiB of documents in flight across
// barriers and tiles, counted vmcnt, raw barriers) ran its K loop in 40 k cycles, no better than the
// 38.5 k of the plain loop, and at a lower clock (1.4 against 1.6 GHz); the same with the two
// wave groups staggered by a phase (one reads and issues while the other multiplies) 50 k. With
// the loads removed altogether the ring's loop took as long; with the MFMAs removed it still ran at
// only 3.5 TB/s. A sub-step of 16 MFMAs per wave is too short to amortise its barrier, its LDS
// latency and the 100-200 cycles each LDS-DMA instruction takes to issue.
// On the plain loop with its barrier BETWEEN two steps (cycles per tile and the clock; 3.93 ms for the
// launch): whole kernel 46 k at 1.8 GHz; loads and barriers alone 45 k at 2.4 GHz; LDS reads and
// barriers alone 13 k; matrix instructions and barriers alone 29 k (24.6 k of them the instructions);
// matrix instructions + LDS reads 34 k WITHOUT the barrier and 49 k with it. So it was neither memory
// nor LDS bandwidth but the barrier's place: behind it all eight waves first waited for their
// fragments with the matrix pipe idle. Giving memory more time changed nothing while that stood
// (fragments of sub-step kk + 1 requested before the matrix instructions of kk: 3.96 ms; a step's
// lines touched three steps ahead so that the LDS-DMA hits L2: 4.22 ms; rows through two alternating
// register sets two steps ahead: 3.98 ms, although the loads alone then took 38 k instead of 45 k).
// Rotated as below: 39 k cycles at 1.62 GHz, 3.64 ms -- the clock gives back part of what the cycles
// gain (the chip is at its power limit with the matrix pipe 63 % busy). The same loop on
// v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (same wave tile, same LDS bytes, same flop per cycle):
// 40.8 k cycles at 1.75 GHz, 3.45 ms in a same-box A/B -- the chip holds a higher clock on that shape.
constexpr int S_OFF_NORM = 2 * IMG_BYTES;             // f32[2][256], by tile parity
constexpr int S_OFF_QUEUE = S_OFF_NORM + 2 * BM * 4;  // u64[2][QCAP]
constexpr int S_OFF_QCOUNT = S_OFF_QUEUE + 2 * QCAP * 8;
constexpr int S_OFF_MAP = S_OFF_QCOUNT + 16;  // u32[2][BM]: slots of the tile's rows (gathered form only)
constexpr int S_LDS = S_OFF_MAP + 2 * BM * 4;
constexpr uint32_t GATHER_SPAN = 0x40000000u;  // bytes a gathered tile's rows may lie apart
static_assert(S_LDS <= 160 * 1024, "LDS carve exceeds 160 KiB");
constexpr uint32_t TILE_NONE = 0xFFFFFFFFu;

struct StreamArgs {
    const u32x4* rows;
    const float* norms;
    const uint64_t* present;
    const uint64_t* cand;
    const u32x4* qrows;          // [256][ksteps * 8] bf16 vectors, zero padded
    const float* qnorm;          // [256]
    const uint64_t* bound_keys;  // [n_q][k] the sample's lists
    const uint32_t* tiles;       // tiles with a candidate outside the sample, any order
    const uint32_t* tile_count;
    u32x4* regions;              // [gridDim.x][cap]: {score bits, docid, query, 0}
    uint32_t* region_count;      // [gridDim.x]
    uint32_t* overflow;
    uint64_t n_slots;
    uint32_t row_vecs, ksteps, docid_base, k, n_q, cap;
    // GATHER build only: the candidates' slots in slot order (padded to whole tiles with a valid
    // slot), and head[0] = their number. `norms` is then indexed like `map`, `tiles` holds pairs
    // (tile of the candidate list, slot of its first row), and the survivors' docid field is the
    // position in the candidate list.
    const uint32_t* map;
    const uint32_t* head;
};

// atomicAdd on an LDS word, written out: before the builtin's ds_add the compiler waits for every
// outstanding vector-memory operation (an LDS-DMA might write the word), which would drain the
// step in flight for every survivor. No DMA ever targets the queue's counters.
__device__ __forceinline__ uint32_t lds_add_return(uint32_t* word, uint32_t v) {
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)word;
    uint32_t old;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(v) : "memory");
    return old;
}

template <int V>
struct Int {
    static constexpr int value = V;
};

typedef const __attribute__((address_space(4))) uint32_t* const_u32;
typedef const __attribute__((address_space(4))) uint64_t* const_u64;

// NARROW: a batch of 128 queries or fewer. The image and the loads keep their 256-row shape, but
// all eight waves work on the first 128 query columns -- 64 per wave, half the matrix instructions
// and half the query fragments -- and nobody fetches query rows 128 .. 255. (With the full-width
// mapping waves s and s + 4, which share SIMD s, own the same 128 columns: absent queries would
// leave two SIMDs in four multiplying zeros while the other two carry the tile.)
template <bool GATHER, bool NARROW>
__global__ __launch_bounds__(BTHREADS) void scan_batch_stream_kernel(StreamArgs a) {
    constexpr uint32_t QW = NARROW ? 64 : 128;  // query columns per wave
    constexpr int NQ16 = QW / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wv = uniform_u32(tid >> 6);
    const uint32_t wm = wv >> 1, wn = wv & 1;
    const uint32_t r16 = lane & 15, quarter = lane >> 4;
    float* snorm = reinterpret_cast<float*>(lds + S_OFF_NORM);
    uint64_t* queue = reinterpret_cast<uint64_t*>(lds + S_OFF_QUEUE);
    uint32_t* qcount = reinterpret_cast<uint32_t*>(lds + S_OFF_QCOUNT);
    const int k = (int)a.k;
    const uint32_t KS = a.ksteps;

    // This lane's eight query columns: the dot-product threshold a row's score must reach to beat
    // the query's bound (pq >= c2 * |row| is necessary for d < bound; the margin lets rounding only
    // admit too many). Constant for the launch.
    float c2v[8];
    bool col_ok[8];
    bool bad = false;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const uint32_t q = wn * QW + (uint32_t)n * 16 + r16;
        col_ok[n] = n < NQ16 && q < a.n_q;
        c2v[n] = 3.0e38f;
        if (col_ok[n]) {
            const uint64_t tau = a.bound_keys[(uint64_t)q * k + (k - 1)];
            const float qn = a.qnorm[q];
            const float c = (1.0f - 2.0f * f32_from_orderable((uint32_t)(tau >> 32))) * qn - qn * 1e-6f;
            if (tau == KEY_NONE || !(qn > 0.0f) || !(qn < __builtin_inff()) || !(c == c)) bad = true;
            c2v[n] = c;
        }
    }
    if (tid < 2) qcount[tid] = 0;
    if (__syncthreads_or(bad)) {  // (no DMA is in flight yet: the plain barrier is fine)
        if (tid == 0) *a.overflow = 1u;
        return;
    }

    const_u32 tiles = (const_u32)(uintptr_t)a.tiles;
    const_u64 present = (const_u64)(uintptr_t)a.present;
    const_u64 cand = (const_u64)(uintptr_t)a.cand;
    const uint32_t count = *(const_u32)(uintptr_t)a.tile_count;
    const uint32_t G = gridDim.x, b = blockIdx.x;
    const uint32_t n_mine = count > b ? (count - b + G - 1) / G : 0;
    if (n_mine == 0) return;
    // rows the tiles index: the store's slots, or the candidate list
    const uint64_t n_rows = GATHER ? (uint64_t) * (const_u32)(uintptr_t)a.head : a.n_slots;
    auto tile_at = [&](uint32_t j) -> uint32_t {
        return j < n_mine ? tiles[(GATHER ? 2u : 1u) * (b + j * G)] : TILE_NONE;
    };
    // GATHER: the slot of the tile's first row (the base address of its loads)
    auto slot_at = [&](uint32_t j) -> uint32_t { return GATHER && j < n_mine ? tiles[2u * (b + j * G) + 1u] : 0u; };
    const uint64_t n_words = a.n_slots >> 6;
    auto words_of = [&](uint32_t t, uint64_t (&w)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint64_t x = 0;
            if constexpr (GATHER) {  // every row of the list is a candidate
                const uint64_t first = (uint64_t)t * BM + 64u * (uint32_t)i;
                if (t != TILE_NONE && first < n_rows) x = n_rows - first >= 64 ? ~0ull : (1ull << (n_rows - first)) - 1ull;
            } else {
                const uint64_t wi = 4 * (uint64_t)t + (uint64_t)i;
                if (t != TILE_NONE && wi < n_words) {
                    x = present[wi];
                    if (a.cand) x &= cand[wi];
                }
            }
            w[i] = x;
        }
    };

    // One K step of one tile = 64 KiB, laid out and fetched exactly as in the kernel above.
    const uint32_t row_bytes = a.row_vecs * 16u;
    const __amdgpu_buffer_rsrc_t qres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4*>(a.qrows), 0, (int)((uint32_t)BN * KS * 128u), 0x00020000);
    // GATHER: the tile's rows lie anywhere above its first one, less than GATHER_SPAN bytes away
    // (the launcher's tile list guarantees it); offsets with bit 30 set read zeros as before
    auto doc_rsrc = [&](uint32_t t, uint32_t slot0) {
        const bool live = t != TILE_NONE;
        if constexpr (GATHER)
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a.rows + (live ? (uint64_t)slot0 * a.row_vecs : 0)),
                                                     0, live ? (int)GATHER_SPAN : 0, 0x00020000);
        else
            return __builtin_amdgcn_make_buffer_rsrc(
                const_cast<u32x4*>(a.rows + (live ? (uint64_t)t * BM * a.row_vecs : 0)), 0,
                live ? (int)((uint32_t)BM * row_bytes) : 0, 0x00020000);
    };
    const uint32_t dslot[2] = {(lane & 7) ^ (lane >> 4), (lane & 7) ^ (4 + (lane >> 4))};
    const uint32_t dva[2] = {(lane >> 3) * row_bytes + dslot[0] * 16u, (lane >> 3) * row_bytes + dslot[1] * 16u};
    // GATHER: byte offset of this lane's row in piece i (row (wv * 4 + i) * 8 + (lane >> 3) of the
    // tile) above the tile's first row, plus its slot. One tile's worth: the loads of a tile's
    // first two steps go out during the last two steps of the tile before, and that is where the
    // offsets are replaced.
    uint32_t gva[4] = {0, 0, 0, 0};
    uint32_t* smap = reinterpret_cast<uint32_t*>(lds + S_OFF_MAP);
    auto offsets_from = [&](const uint32_t* m, uint32_t slot0) {  // m: the tile's 256 slots
        const uint32_t r = wv * 32 + (opaque(lane) >> 3);  // (taken afresh: not worth a register across the K loop)
#pragma unroll
        for (int i = 0; i < 4; ++i) gva[i] = (m[r + 8 * (uint32_t)i] - slot0) * row_bytes + dslot[i & 1] * 16u;
    };
    const uint32_t dvb[2] = {(lane >> 3) * (KS * 128u) + dslot[0] * 16u, (lane >> 3) * (KS * 128u) + dslot[1] * 16u};
    // A wave's eight pieces of a step go out one by one, spread over the step (documents first, they
    // have furthest to come; then queries): 64 pieces issued by all waves at
    // once queue up behind the CU's one load path (16 cycles a piece), and the waves stand in that
    // queue with the matrix pipe idle.
    auto load_piece = [&](uint32_t t, uint32_t slot0, uint32_t ks, uint32_t image, int pc) {  // pc 0..3: documents, 4..7: queries
        unsigned char* img = lds + image * IMG_BYTES + wv * 4096;
        if (pc < 4) {
            const int i = pc;
            const __amdgpu_buffer_rsrc_t ares = doc_rsrc(t, slot0);
            const uint32_t kill = ks * 8 + dslot[i & 1] < a.row_vecs ? 0u : 0x40000000u;
            // nt: every document byte is read once per launch
            if constexpr (GATHER)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ares, (lds_ptr)(img + i * 1024), 16, (int)(gva[i] | kill),
                                                         (int)(ks * 128u), 0, 2);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ares, (lds_ptr)(img + i * 1024), 16, (int)(dva[i & 1] | kill),
                                                         (int)(ks * 128u + (wv * 4 + i) * 8u * row_bytes), 0, 2);
        } else {
            const int i = pc - 4;
            if (NARROW && wv >= 4) return;  // query rows 128 .. 255: nobody reads them
            // eight query rows the batch does not hold: their columns' scores are never looked at
            // (col_ok), so whatever the image holds there may be multiplied
            // (narrow build only: in the full-width one the branches cost more than the pieces)
            if (NARROW && (wv * 4 + (uint32_t)i) * 8u >= a.n_q) return;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(qres, (lds_ptr)(img + IMG_B + i * 1024), 16, (int)dvb[i & 1],
                                                     (int)(ks * 128u + (wv * 4 + i) * 8u * (KS * 128u)), 0, 0);
        }
    };
    auto load_step = [&](uint32_t t, uint32_t slot0, uint32_t ks, uint32_t image) {
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) load_piece(t, slot0, ks, image, pc);
    };
    // the norms of tile t's rows into snorm[par] (one wave; rows past the end read 0)
    auto issue_norms = [&](uint32_t t, uint32_t par) {
        const bool live = t != TILE_NONE;
        const uint64_t row0 = live ? (uint64_t)t * BM : 0;
        const uint64_t left = live && n_rows > row0 ? n_rows - row0 : 0;
        const __amdgpu_buffer_rsrc_t nres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.norms + row0), 0, (int)((left < (uint64_t)BM ? left : (uint64_t)BM) * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(nres, (lds_ptr)(lds + S_OFF_NORM + par * BM * 4 + i * 256), 4,
                                                     (int)(lane * 4u), i * 256, 0, 0);
    };

    // GATHER: the slots of tile t's rows into smap[par] (one wave; the list is padded to whole tiles)
    auto issue_map = [&](uint32_t t, uint32_t par) {
        const bool live = t != TILE_NONE;
        const __amdgpu_buffer_rsrc_t mres = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t*>(a.map + (live ? (uint64_t)t * BM : 0)), 0, live ? BM * 4 : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(mres, (lds_ptr)(lds + S_OFF_MAP + par * BM * 4 + i * 256), 4,
                                                     (int)(lane * 4u), i * 256, 0, 0);
    };

    uint32_t t0 = tile_at(0), t1 = tile_at(1);
    uint32_t s0 = slot_at(0), s1 = slot_at(1);
    uint64_t w0[4];
    words_of(t0, w0);
    if constexpr (GATHER) offsets_from(a.map + (uint64_t)t0 * BM, s0);  // (no load is in flight yet: waiting is free)
    if (wv == 7) issue_norms(t0, 0);
    uint32_t image = 0;  // the image the step about to be multiplied lies in
    load_step(t0, s0, 0, image);

    const uint32_t frag_a = swz(wm * 64 + r16, quarter), frag_b = IMG_B + swz(wn * QW + r16, quarter);
    uint32_t t_prev = TILE_NONE;

    // The matrix instruction is v_mfma_f32_16x16x32_bf16: a 64-element step is two sub-steps of 32, a
    // wave's 64 x 128 tile 4 x 8 products of 16 x 16. Per flop it takes the cycles and the LDS bytes of
    // the 32x32x16 form; the chip, which lowers its clock under this loop, holds a higher one with it.
    // Lane l holds row l & 15, elements 8 * (l >> 4) .. + 8 of a sub-step: one 16-byte slot, and the
    // image's swizzle is conflict-free for these reads as well.
    //
    // A step is walked in four phases of 16 matrix instructions: (sub-step, query half) = (0,0) (0,1)
    // (1,0) (1,1). The loop is rotated by one phase: a step's one barrier stands before the matrix
    // instructions of its LAST phase, whose fragments are in registers by then. Behind the barrier
    // a wave requests the first fragments of the NEXT step (its image is complete: that is what the
    // barrier says) and the rows of the step after that (into the image everybody has just finished
    // reading), and then still has sixteen matrix instructions to issue while those reads fly. With the
    // barrier between two steps instead, every step began with all eight waves waiting for LDS and
    // the matrix pipe idle: 49 k cycles per tile against 34 k without any barrier (loads left out).
    bf16x8 fa[2][4], fb[2][4];
    auto read_a = [&](const unsigned char* img, int kk, bf16x8 (&xa)[4]) {
        uint32_t fo = frag_a ^ ((uint32_t)kk << 6);
        asm volatile("" : "+v"(fo));  // recompute per phase, do not keep the addresses live
#pragma unroll
        for (int m = 0; m < 4; ++m)
#ifdef VS_ABL_NOLDSREAD
            xa[m] = __builtin_bit_cast(bf16x8, u32x4{fo, 1u, 2u, 3u});
#else
            xa[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + m * 16 * 128 + fo));
#endif
    };
    auto read_b = [&](const unsigned char* img, int kk, int h, bf16x8 (&xb)[4]) {
        if (NARROW && h == 1) return;
        uint32_t fo = frag_b ^ ((uint32_t)kk << 6);
        asm volatile("" : "+v"(fo));
#pragma unroll
        for (int n = 0; n < 4; ++n)
#ifdef VS_ABL_NOLDSREAD
            xb[n] = __builtin_bit_cast(bf16x8, u32x4{fo, 1u, 2u, 3u});
#else
            xb[n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + (h * 4 + n) * 16 * 128 + fo));
#endif
    };
    // step 0 is complete and visible; its first fragments and the rows of step 1 are requested
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    read_a(lds, 0, fa[0]);
    read_b(lds, 0, 0, fb[0]);
    {
        const uint32_t t_1 = 1 < KS ? t0 : t1, k_1 = 1 < KS ? 1u : 0u, s_1 = 1 < KS ? s0 : s1;
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) load_piece(t_1, s_1, k_1, 1, pc);
    }

    // appends the queue of tile `t` (parity pp) to this workgroup's region; wave 0 only, which
    // keeps the region's fill in `n_out`
    uint32_t n_out = 0;
    u32x4* const region = a.regions + (uint64_t)b * a.cap;
    auto flush = [&](uint32_t t, uint32_t pp) {
        uint32_t n = uniform_u32(qcount[pp]);
        if (n > (uint32_t)QCAP || n_out + n > a.cap) {
            if (lane == 0) *a.overflow = 1u;
            n = 0;
        }
        // (the lane number is taken afresh: hoisted out of the tile loop, this loop's addresses were
        // the first thing the register allocator spilled)
        for (uint32_t i = opaque(lane); i < n; i += 64) {
            const uint64_t e = queue[pp * QCAP + i];
            u32x4 out;
            out[0] = (uint32_t)(e >> 32);
            out[1] = a.docid_base + t * (uint32_t)BM + ((uint32_t)e & 0xFFu);
            out[2] = ((uint32_t)e >> 8) & 0xFFu;
            out[3] = 0;
            region[n_out + i] = out;
        }
        n_out += n;
        if (lane == 0) qcount[pp] = 0;
    };

#ifdef VS_STAMP  // diagnostic build: cycles per tile and the clock they ran at
    const uint64_t stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // (Measured and dropped: `s_setprio 1` for waves 4 .. 7, the half that loses issue arbitration to
    // its older SIMD partners. Same-box A/B, steady-state launches on 10 M x 768: 3.53 ms with, 3.51 ms
    // without. The loop is not held up by who issues first.)
    for (uint32_t j = 0; j < n_mine; ++j) {
        const uint32_t par = j & 1;
        // tile ids two ahead and the next tile's words are requested now and used a tile later
        const uint32_t t2 = tile_at(j + 2), s2 = slot_at(j + 2);
        uint64_t w1[4];
        words_of(t1, w1);

        f32x4v acc[4][8];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 8; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[m][n][e] = 0.0f;

        // the tile's own barrier: the previous tile's survivors are queued, this tile's norms have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < (uint32_t)BM) {
            // rows that are not candidates get a NaN norm: their t is NaN, which the running
            // maximum ignores
            const bool is_cand = (w0[tid >> 6] >> (tid & 63)) & 1;  // w0[] index is wave-uniform
            const uint32_t me = opaque(tid);  // (address taken afresh per tile, or it is kept in a spilled register)
            const float pn = snorm[par * BM + me];
            if (is_cand && !(pn < __builtin_inff())) *a.overflow = 1u;
            if (!is_cand) snorm[par * BM + me] = __builtin_nanf("");
        }
        if (wv == 0 && j > 0) flush(t_prev, par ^ 1u);
        if (wv == 7) {
            issue_norms(t1, par ^ 1u);
            if constexpr (GATHER) issue_map(t1, par ^ 1u);
        }

        for (uint32_t ks = 0; ks < KS; ++ks) {
            const unsigned char* img = lds + image * IMG_BYTES;
            const uint32_t other = image ^ 1u;
            // (tile, step) of the next step and of the one after it (past the last tile: zeros)
            const uint32_t t_1 = ks + 1 < KS ? t0 : t1, k_1 = ks + 1 < KS ? ks + 1 : 0;
            const uint32_t s_1 = ks + 1 < KS ? s0 : s1;
            uint32_t t_2 = t0, k_2 = ks + 2, s_2 = s0;
            if (k_2 >= KS) {
                k_2 -= KS;
                t_2 = t1;
                s_2 = s1;
                if (k_2 >= KS) {  // one-step tiles (never in the GATHER build: its launcher wants three steps)
                    k_2 -= KS;
                    t_2 = t2;
                    s_2 = s2;
                }
            }
            auto multiply = [&](auto H, const bf16x8 (&xa)[4], const bf16x8 (&xb)[4]) {
                constexpr int h = decltype(H)::value;
                if (NARROW && h == 1) return;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#ifdef VS_ABL_NOMFMA
                        asm volatile("" ::"v"(xa[m]), "v"(xb[n]));
#else
                        acc[m][h * 4 + n] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[m], xb[n], acc[m][h * 4 + n], 0, 0, 0);
#endif
            };
            // phases 0 .. 2: the fragments of the next one first (the scheduler would sink them to
            // their first use), the next step's query pieces in between
            // (this step's first fragments, requested sixteen matrix instructions ago: waiting for them
            // here, before anything else is requested, keeps the compiler from waiting for everything later)
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            read_b(img, 0, 1, fb[1]);
            __builtin_amdgcn_sched_barrier(0);
#ifndef VS_ABL_NOLOAD
            load_piece(t_1, s_1, k_1, other, 4);
            load_piece(t_1, s_1, k_1, other, 5);
#endif
            multiply(Int<0>{}, fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            read_a(img, 1, fa[1]);
            read_b(img, 1, 0, fb[0]);
            __builtin_amdgcn_sched_barrier(0);
#ifndef VS_ABL_NOLOAD
            load_piece(t_1, s_1, k_1, other, 6);
            load_piece(t_1, s_1, k_1, other, 7);
#endif
            multiply(Int<1>{}, fa[0], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
            read_b(img, 1, 1, fb[1]);
            __builtin_amdgcn_sched_barrier(0);
            multiply(Int<0>{}, fa[1], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            // my pieces of the next step are in LDS, my reads of this one are done ...
            // (Measured again on this rotated loop and dropped: one LDS-DMA dword per line three or five steps
            // ahead, kept out of the wait by vmcnt(1), so that the step's own requests hit L2 -- 3.92 / 3.99 ms
            // against 3.52 ms in a same-box A/B. The loop does not wait for HBM; one more memory instruction
            // per wave and step costs more than the earlier arrival gains.)
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // ... everyone's are
#ifndef VS_ABL_NOBARRIER
            __builtin_amdgcn_s_barrier();
#endif
            read_a(lds + other * IMG_BYTES, 0, fa[0]);
            read_b(lds + other * IMG_BYTES, 0, 0, fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (GATHER) {
                // the rows about to be requested are the NEXT tile's from here on: its slots were
                // staged at this tile's top and are visible since the first step's barrier
                if (ks + 2 == KS) offsets_from(smap + (par ^ 1u) * BM, s1);
            }
#ifndef VS_ABL_NOLOAD
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) load_piece(t_2, s_2, k_2, image, pc);
#endif
            multiply(Int<1>{}, fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
            image = other;
        }

        // ---- register test: t = score - |row| * c2[query]; per query column the largest t.
        // accumulator element e of product (m, n) is row m*16 + 4*quarter + e, column n*16 + r16
        // of the wave tile
        {
            const float* norm_mine = snorm + par * BM + wm * 64 + 4 * quarter;
            f32x4v pn4[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) pn4[m] = *reinterpret_cast<const f32x4v*>(norm_mine + m * 16);
            float tmax[8];
#pragma unroll
            for (int n = 0; n < NQ16; ++n) {
                tmax[n] = -__builtin_inff();
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        tmax[n] = __builtin_fmaxf(tmax[n], __builtin_fmaf(-pn4[m][e], c2v[n], acc[m][n][e]));
            }
#pragma unroll
            for (int n = 0; n < NQ16; ++n) {
#ifdef VS_ABL_NOHITS
                const bool lane_hit = tmax[n] >= 3.0e38f;
#else
                const bool lane_hit = tmax[n] >= 0.0f && col_ok[n];
#endif
                if (__ballot(lane_hit) == 0) continue;  // wave-uniform, the common case
                uint32_t hit = 0;
                if (lane_hit) {
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = __builtin_fmaf(-pn4[m][e], c2v[n], acc[m][n][e]);
                            hit |= (t >= 0.0f) ? (1u << (m * 4 + e)) : 0u;
                        }
                }
                while (hit) {
                    const uint32_t e = (uint32_t)__builtin_ctz(hit);
                    hit &= hit - 1;
                    float v = acc[0][n][0];
#pragma unroll
                    for (int jj = 1; jj < 16; ++jj) v = (e == (uint32_t)jj) ? acc[jj >> 2][n][jj & 3] : v;
                    const uint32_t wrow = (e >> 2) * 16u + 4u * quarter + (e & 3u);
                    const uint32_t idx = lds_add_return(qcount + par, 1u);
                    if (idx < (uint32_t)QCAP)
                        queue[par * QCAP + idx] = ((uint64_t)__float_as_uint(v) << 32) |
                                                  ((wn * QW + (uint32_t)n * 16 + r16) << 8) | (wm * 64 + wrow);
                }
            }
        }
        t_prev = t0;
        t0 = t1;
        t1 = t2;
        s0 = s1;
        s1 = s2;
#pragma unroll
        for (int i = 0; i < 4; ++i) w0[i] = w1[i];
    }
#ifdef VS_STAMP
    if (b == 7 && tid == 0) {
        const uint64_t c = __builtin_amdgcn_s_memtime() - stamp_c0, r = __builtin_amdgcn_s_memrealtime() - stamp_r0;
        printf("stream: %u tiles, %llu cycles per tile, %.0f MHz\n", n_mine, (unsigned long long)(c / n_mine),
               (double)c / (double)r * 100.0);
    }
#endif
    // the last tile's survivors, once every wave has tested its scores
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wv == 0) {
        flush(t_prev, (n_mine - 1) & 1u);
        if (lane == 0) a.region_count[b] = n_out;
    }
}

// The tiles the streaming pass has to visit: every tile outside the sample with at least one
// candidate. One thread per tile, one atomic per wave; the order of the list does not matter.
__global__ __launch_bounds__(256) void batch_tiles_kernel(const uint64_t* present, const uint64_t* cand,
                                                          uint64_t n_words, uint32_t n_tiles, uint32_t sample_stride,
                                                          uint32_t sample_count, uint32_t* tiles, uint32_t* tile_count) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool keep = false;
    if (t < n_tiles && !(t % sample_stride == 0 && t / sample_stride < sample_count)) {
        uint64_t any = 0;
        for (int i = 0; i < 4; ++i) {
            const uint64_t wi = 4 * (uint64_t)t + (uint64_t)i;
            if (wi < n_words) any |= cand ? (present[wi] & cand[wi]) : present[wi];
        }
        keep = any != 0;
    }
    const uint64_t m = __ballot(keep);
    if (m == 0) return;
    uint32_t base = 0;
    if (lane == __builtin_ctzll(m)) base = atomicAdd(tile_count, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
    if (keep) tiles[base + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = t;
}

// Sorts the workgroups' survivor regions into per-query buckets: one workgroup per region. The
// region is counted per query in LDS first, so that a bucket's fill is bumped once per workgroup
// and query instead of once per survivor (256 workgroups hammering 256 counters took 190 us).
__global__ __launch_bounds__(256) void batch_bucket_kernel(const u32x4* regions, const uint32_t* region_count,
                                                           uint32_t cap, uint64_t* buckets, uint32_t* bucket_count,
                                                           uint32_t* overflow) {
    __shared__ uint32_t cnt[BN], base[BN];
    uint32_t n = region_count[blockIdx.x];
    if (n > cap) n = cap;
    const u32x4* mine = regions + (uint64_t)blockIdx.x * cap;
    cnt[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256) atomicAdd(&cnt[mine[i][2] & 0xFFu], 1u);
    __syncthreads();
    {
        const uint32_t c = cnt[threadIdx.x];
        base[threadIdx.x] = c ? atomicAdd(bucket_count + threadIdx.x, c) : 0;
        cnt[threadIdx.x] = 0;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const u32x4 e = mine[i];
        const uint32_t q = e[2] & 0xFFu;
        const uint32_t idx = base[q] + atomicAdd(&cnt[q], 1u);
        if (idx < cap)
            buckets[(uint64_t)q * cap + idx] = ((uint64_t)e[0] << 32) | e[1];
        else
            *overflow = 1u;
    }
}

// Judges the survivors of the streaming pass exactly (arroy's formula on the score, the row's norm
// and the query's norm) and merges them with the sample's list: one workgroup per query.
__global__ __launch_bounds__(256) void batch_select_kernel(const uint64_t* buckets, const uint32_t* bucket_count,
                                                           uint32_t cap, const uint64_t* sample, const float* norms,
                                                           const float* qnorm, uint32_t docid_base, uint32_t k_,
                                                           uint64_t* out_keys, uint32_t* out_ids, float* out_dist,
                                                           uint32_t* out_n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t wv = uniform_u32(threadIdx.x >> 6);
    const int k = (int)k_;
    uint32_t n = bucket_count[q];
    if (n > cap) n = cap;
    const float qn = qnorm[q];
    const uint64_t* mine = buckets + (uint64_t)q * cap;
    WaveTopK top;
    top.init(reinterpret_cast<uint64_t*>(smem) + (size_t)wv * k, k, lane);
    auto offer = [&](uint64_t key) {
        uint64_t m = __ballot(key < top.tau);
        while (m) {
            const int src = __builtin_ctzll(m);
            top.insert(readlane_u64(key, src));
            m &= m - 1;
            m &= __ballot(key < top.tau);
        }
    };
    for (uint32_t base = wv * 64; base < n; base += 256) {
        const uint32_t i = base + (uint32_t)lane;
        uint64_t key = KEY_NONE;
        if (i < n) {
            const uint64_t e = mine[i];
            const uint32_t docid = (uint32_t)e;
            const float v = __uint_as_float((uint32_t)(e >> 32));
            const float pnqn = norms[docid - docid_base] * qn;
            float d = 0.0f;
            if (pnqn != 0.0f) d = (1.0f - v / pnqn) / 2.0f;
            key = make_key(f32_orderable(d), docid);
        }
        offer(key);
    }
    if (wv == 0)  // the sample's rows are not among the streamed ones: its list joins as it is
        for (int base = 0; base < k; base += 64)
            offer(base + lane < k ? sample[(uint64_t)q * k + base + lane] : KEY_NONE);
    top.spill();
    __syncthreads();
    if (wv == 0) {
        for (int v = 1; v < 4; ++v) top.absorb(reinterpret_cast<uint64_t*>(smem) + (size_t)v * k);
        uint32_t cnt = 0;
        for (int base = 0; base < k; base += 64) {
            const int i = base + lane;
            const uint64_t key = top.at_lane(base);
            if (i < k) {
                if (out_keys) out_keys[(uint64_t)q * k + i] = key;
                if (out_ids) {
                    out_ids[(uint64_t)q * k + i] = (uint32_t)key;
                    out_dist[(uint64_t)q * k + i] = key != KEY_NONE ? f32_from_orderable((uint32_t)(key >> 32)) : 0.0f;
                }
            }
            cnt += (uint32_t)__popcll(__ballot(key != KEY_NONE));
        }
        if (out_n && lane == 0) out_n[q] = cnt;
    }
}

// queries f32 [n_q][dim] -> bf16 rows [256][kvecs * 8] (round to nearest even, zero
// padded) and the norm of each ROUNDED query; one wave per query row
__global__ __launch_bounds__(256) void prep_queries_bf16_kernel(const float* queries, uint32_t n_q,
                                                                uint32_t dim, uint32_t kelems,
                                                                uint16_t* qrows, float* qnorm,
                                                                uint32_t* counters, uint32_t n_counters) {
    const int lane = threadIdx.x & 63;
    // the streaming pass's counters (tile count, overflow flag, bucket fill) start at zero
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_counters; i += gridDim.x * 256) counters[i] = 0;
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

// ---- selective filters: the candidates gathered into a dense slab ----------------------------
//
// The matrix-core scan multiplies whole 256-row tiles, so under a filter that keeps one row in ten
// it would still read and multiply all of them. Below a quarter of the rows the candidates are
// copied once into a dense slab of their own (rows, norms, an all-ones present bitmap), in slot
// order, the two passes above run on that slab, and the compact row numbers in their result are
// turned back into docids. Slot order is kept, so ties still break by docid. Cost: the candidate
// bytes are read twice and written once instead of every row being read once.

namespace {

constexpr uint32_t GATHER_BLOCK_WORDS = 256;  // mask words per workgroup of the counting kernels: one per thread

__device__ __forceinline__ uint64_t candidate_word(const uint64_t* present, const uint64_t* cand, uint64_t i,
                                                   uint64_t n_words) {
    return i < n_words ? present[i] & cand[i] : 0ull;
}

// exclusive prefix sum over the 256 threads of a workgroup; *total = the sum
__device__ __forceinline__ uint32_t block_exclusive_256(uint32_t v, uint32_t* wave_sums, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= (uint32_t)d) inc += o;
    }
    if (lane == 63) wave_sums[wv] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        const uint32_t x = wave_sums[i];
        if (i < wv) before += x;
        all += x;
    }
    __syncthreads();
    *total = all;
    return before + inc - v;
}

}  // namespace

// candidates per block of 256 mask words
__global__ __launch_bounds__(256) void gather_count_kernel(const uint64_t* present, const uint64_t* cand,
                                                           uint64_t n_words, uint32_t* block_sums) {
    __shared__ uint32_t wave_sums[4];
    const uint64_t wi = (uint64_t)blockIdx.x * GATHER_BLOCK_WORDS + threadIdx.x;
    const uint32_t c = (uint32_t)__builtin_popcountll(candidate_word(present, cand, wi, n_words));
    uint32_t total;
    (void)block_exclusive_256(c, wave_sums, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// The sample of a candidate list of `tiles` tiles of 256: tiles 0, stride, 2 * stride, ... (`count`
// of them), as plan_batch spreads the sample over a store.
__host__ __device__ inline void sample_of(uint32_t tiles, uint32_t* stride, uint32_t* count) {
    *count = tiles < (uint32_t)SAMPLE_TILES ? tiles : (uint32_t)SAMPLE_TILES;
    uint32_t st = tiles / (uint32_t)SAMPLE_TILES;
    if (st < 1) st = 1;
    if (!(st & 1)) st -= 1;
    *stride = st;
}

// block_sums -> exclusive offsets in place; head[0] = min(number of candidates, cap), head[1] = tiles
// of 256 they fill, head[2], head[3] = the sample's stride and tile count. One workgroup.
__global__ __launch_bounds__(256) void gather_scan_kernel(uint32_t* block_sums, uint32_t n_blocks, uint32_t cap,
                                                          uint32_t* head) {
    __shared__ uint32_t wave_sums[4];
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sums[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_256(v, wave_sums, &total);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        const uint32_t n = carry < cap ? carry : cap;
        const uint32_t tiles = (uint32_t)(((uint64_t)n + BM - 1) / BM);
        head[0] = n;
        head[1] = tiles;
        sample_of(tiles, head + 2, head + 3);
    }
}

// map[i] = slot of the i-th candidate, in slot order (and its norm, if asked). A lane expands its
// own word into a per-wave LDS list (as many steps as the fullest word has bits), and the wave
// copies the list out with contiguous writes.
__global__ __launch_bounds__(256) void gather_map_kernel(const uint64_t* present, const uint64_t* cand,
                                                         uint64_t n_words, const uint32_t* block_offsets,
                                                         uint32_t cap, uint32_t* map, const float* norms,
                                                         float* norms_c /* nullable: norms in list order */) {
    __shared__ uint16_t stage[4][4096];  // bit positions within the wave's 64 words
    __shared__ uint32_t wave_sums[4];
    const uint32_t lane = threadIdx.x & 63, wv = uniform_u32(threadIdx.x >> 6);
    const uint64_t wave_word0 = (uint64_t)blockIdx.x * GATHER_BLOCK_WORDS + (uint64_t)wv * 64;
    uint64_t x = candidate_word(present, cand, wave_word0 + lane, n_words);
    const uint32_t c = (uint32_t)__builtin_popcountll(x);
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= (uint32_t)d) inc += o;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    const bool all_rows = total == 4096;  // wave-uniform: 64 full words need no list, row i is candidate i
    uint32_t o = inc - c;
    while (x && !all_rows) {
        stage[wv][o++] = (uint16_t)(lane * 64 + (uint32_t)__builtin_ctzll(x));
        x &= x - 1;
    }
    if (lane == 0) wave_sums[wv] = total;
    __syncthreads();
    uint32_t at = block_offsets[blockIdx.x];
    for (uint32_t i = 0; i < wv; ++i) at += wave_sums[i];
    for (uint32_t i = lane; i < total; i += 64) {
        const uint32_t pos = at + i;
        if (pos < cap) {
            const uint32_t slot = (uint32_t)(wave_word0 * 64) + (all_rows ? i : (uint32_t)stage[wv][i]);
            map[pos] = slot;
            if (norms_c) norms_c[pos] = norms[slot];
        }
    }
}

// rows, norms and the present words of the dense slab; one thread per 16-byte vector
__global__ __launch_bounds__(256) void gather_rows_kernel(const u32x4* rows, const float* norms, uint32_t row_vecs,
                                                          const uint32_t* map, const uint32_t* head, u32x4* rows_c,
                                                          float* norms_c, uint64_t* present_c) {
    const uint32_t n = head[0];
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t i = (uint32_t)(g / row_vecs), v = (uint32_t)(g - (uint64_t)i * row_vecs);
    if (i >= n) return;
    const uint32_t slot = map[i];
    rows_c[g] = __builtin_nontemporal_load(rows + (uint64_t)slot * row_vecs + v);
    if (v == 0) {
        norms_c[i] = norms[slot];
        if ((i & 63u) == 0) present_c[i >> 6] = n - i >= 64 ? ~0ull : (1ull << (n - i)) - 1ull;
    }
}

// compact row numbers in the result back to docids
__global__ __launch_bounds__(256) void gather_translate_kernel(const uint32_t* map, uint32_t docid_base, uint32_t total,
                                                               uint64_t* keys, uint32_t* ids, const uint32_t* counts,
                                                               uint32_t k, const uint32_t* skip_if_set) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total || (skip_if_set && *skip_if_set)) return;
    if (keys) {
        const uint64_t key = keys[i];
        if (key != KEY_NONE) keys[i] = (key & 0xFFFFFFFF00000000ull) | (uint64_t)(docid_base + map[(uint32_t)key]);
    }
    if (ids && i % k < counts[i / k]) ids[i] = docid_base + map[ids[i]];
}

// ---- the same, without the copy: the streaming pass fetches the candidates where they lie -------
//
// scan_batch_stream_kernel<true> takes a tile's 256 rows from anywhere (one slot per row, staged in
// LDS a tile ahead), so under a filter it reads exactly the candidates' bytes, once. Only the sample
// (at most 65 536 rows) is copied into a dense slab, for the general kernel's unbounded first pass;
// everything downstream works on positions in the candidate list, which the last launch turns
// into docids. Needs three steps per tile or more (the offsets are swapped two steps before a tile
// ends) and a tile's rows within GATHER_SPAN bytes of each other; a tile that is not raises
// `overflow`, and then, as for anything else the streaming pass cannot hold, a gated unbounded scan
// of the store itself produces the result.

// pads the slot list to whole tiles with its last slot (rows that are not candidates, but valid addresses)
__global__ __launch_bounds__(256) void gather_pad_kernel(uint32_t* map, const uint32_t* head) {
    const uint32_t n = head[0];
    const uint32_t i = n + threadIdx.x;
    if (n != 0 && i < (n + BM - 1) / BM * BM) map[i] = map[n - 1];
}

// the sample tiles of the candidate list as a dense slab: rows, norms, present words
__global__ __launch_bounds__(256) void gather_sample_kernel(const u32x4* rows, const float* norms, uint32_t row_vecs,
                                                            const uint32_t* map, const uint32_t* head, u32x4* rows_s,
                                                            float* norms_s, uint64_t* present_s) {
    const uint32_t n = head[0], stride = head[2], count = head[3];
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t i = (uint32_t)(g / row_vecs), v = (uint32_t)(g - (uint64_t)i * row_vecs);  // row i of the sample slab
    const uint32_t j = i / (uint32_t)BM;
    if (j >= count) return;
    const uint64_t at = (uint64_t)j * stride * BM + (i & (uint32_t)(BM - 1));  // its place in the candidate list
    if (at >= n) return;
    const uint32_t slot = map[at];
    rows_s[g] = __builtin_nontemporal_load(rows + (uint64_t)slot * row_vecs + v);
    if (v == 0) {
        norms_s[i] = norms[slot];
        if ((i & 63u) == 0) present_s[i >> 6] = n - at >= 64 ? ~0ull : (1ull << (n - at)) - 1ull;
    }
}

// sample slab row numbers in the sample's lists -> positions in the candidate list
__global__ __launch_bounds__(256) void gather_sample_keys_kernel(uint64_t* keys, uint32_t total, const uint32_t* head) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint64_t key = keys[i];
    if (key == KEY_NONE) return;
    const uint32_t row = (uint32_t)key;
    keys[i] = (key & 0xFFFFFFFF00000000ull) | (uint64_t)((row / (uint32_t)BM) * head[2] * (uint32_t)BM + (row & (uint32_t)(BM - 1)));
}

// The tiles of the candidate list outside the sample, each with the slot of its first row. A tile
// whose rows lie further apart than the streaming pass can address is left out and raises `overflow`.
__global__ __launch_bounds__(256) void batch_tiles_gather_kernel(const uint32_t* map, const uint32_t* head,
                                                                 uint32_t row_bytes, uint32_t* tiles,
                                                                 uint32_t* tile_count, uint32_t* overflow) {
    const uint32_t n = head[0], n_tiles = head[1], stride = head[2], count = head[3];
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tiles || (t % stride == 0 && t / stride < count)) return;
    const uint64_t first = (uint64_t)t * BM;
    const uint64_t last = first + BM <= n ? first + BM - 1 : (uint64_t)n - 1;
    const uint32_t s0 = map[first];
    if ((uint64_t)(map[last] - s0 + 1) * row_bytes > (uint64_t)GATHER_SPAN) {
        *overflow = 1u;
        return;
    }
    const uint32_t at = atomicAdd(tile_count, 1u);
    tiles[2 * at] = t;
    tiles[2 * at + 1] = s0;
}

namespace {

struct BatchPlan {
    uint32_t ksteps, n_tiles, sample_tiles, sample_stride, blocks1, blocks2, cap;
    size_t off_qnorm, off_sample, off_partial1, off_partial2, off_counters, off_tiles, off_buckets, off_regions, total;
    // the filtered form that fetches the candidates in place (three steps per tile or more)
    bool in_place;
    uint32_t fallback_blocks, gather_blocks;
    size_t off_ghead, off_gsums, off_map, off_normsc, off_spresent, off_snorms, off_srows;
};
// counters: [0] tile count, [1] overflow flag, [2 .. 2 + 256) bucket fill, [258 .. 258 + 256) region fill
constexpr uint32_t STREAM_GRID = 256;
constexpr uint32_t N_COUNTERS = 2 + BN + STREAM_GRID;

BatchPlan plan_batch(const SlabView& s, uint32_t k) {
    BatchPlan p;
    p.ksteps = (s.row_vecs + 7) / 8;
    p.n_tiles = (uint32_t)((s.n_slots + BM - 1) / BM);
    p.sample_tiles = p.n_tiles < (uint32_t)SAMPLE_TILES ? p.n_tiles : (uint32_t)SAMPLE_TILES;
    // Sample tiles are spread over the whole store (a filter may select one docid range only).
    // An odd stride: the tiles of a workgroup of the second pass are 256 apart, so each loses
    // at most one tile to the sample and the passes stay balanced.
    p.sample_stride = p.n_tiles / (uint32_t)SAMPLE_TILES;
    if (p.sample_stride < 1) p.sample_stride = 1;
    if (!(p.sample_stride & 1)) p.sample_stride -= 1;
    p.blocks1 = p.sample_tiles < 1 ? 1 : p.sample_tiles;
    const uint32_t rest = p.n_tiles - p.sample_tiles;
    p.blocks2 = rest > device_cus() ? device_cus() : rest;  // one persistent workgroup per CU
    size_t off = (size_t)BN * p.ksteps * BK * 2;
    p.off_qnorm = off;
    off += (size_t)BN * 4 + 128;
    p.off_sample = off;
    off += (size_t)BN * k * 8;
    p.off_partial1 = off;
    off += (size_t)BN * p.blocks1 * k * 8;
    p.off_partial2 = off;
    p.in_place = p.ksteps >= 3;
    p.fallback_blocks = p.n_tiles < device_cus() ? p.n_tiles : device_cus();
    {
        const uint32_t lists = p.in_place && p.fallback_blocks > p.blocks2 + 1 ? p.fallback_blocks : p.blocks2 + 1;
        off += (size_t)BN * lists * k * 8;
    }
    // streaming second pass: a query expects k * rows / sample rows survivors (its bound is the k-th
    // best of the sample); buckets hold four times that, a power of two between 1024 and 65536
    const uint64_t sample_rows = (uint64_t)p.sample_tiles * BM;
    const uint64_t expect = sample_rows ? (uint64_t)k * s.n_slots / sample_rows : 0;
    uint32_t cap = 1024;
    while (cap < 65536 && cap < 4 * expect) cap *= 2;
    p.cap = cap;
    off = (off + 255) / 256 * 256;
    p.off_counters = off;
    off += (size_t)N_COUNTERS * 4 + 256;
    p.off_tiles = off;
    off += ((size_t)p.n_tiles * 8 + 255) / 256 * 256;
    p.off_buckets = off;
    off += (size_t)BN * cap * 8;
    p.off_regions = off;  // a workgroup's share of the survivors is a query's share: the same capacity
    off += (size_t)STREAM_GRID * cap * 16;
    off = (off + 255) / 256 * 256;
    p.gather_blocks = (uint32_t)(((s.n_slots >> 6) + GATHER_BLOCK_WORDS - 1) / GATHER_BLOCK_WORDS);
    if (p.in_place) {
        auto up = [](size_t x) { return (x + 255) / 256 * 256; };
        p.off_ghead = off;
        off += 256;
        p.off_gsums = off;
        off += up((size_t)p.gather_blocks * 4);
        p.off_map = off;
        off += up(((size_t)s.n_slots + BM) * 4);
        p.off_normsc = off;
        off += up(((size_t)s.n_slots + BM) * 4);
        p.off_spresent = off;
        off += up((size_t)p.sample_tiles * BM / 8 + 64);
        p.off_snorms = off;
        off += up((size_t)p.sample_tiles * BM * 4);
        p.off_srows = off;
        off += up((size_t)p.sample_tiles * BM * s.row_vecs * 16);
    }
    p.total = off + 256;
    return p;
}

}  // namespace

uint64_t batch_workspace_bytes(const SlabView& s, uint32_t k) { return plan_batch(s, k).total; }

bool batch_filters_in_place(const SlabView& s) {
    static const bool allowed = getenv("VS_BATCH_NO_GATHER") == nullptr;
    return allowed && (s.row_vecs + 7) / 8 >= 3 && s.n_slots < 0xFFFFFF00ull;
}

hipError_t launch_scan_batch_bf16(const SlabView& s, const float* d_queries, uint32_t n_q,
                                  const uint64_t* cand, uint32_t k, void* workspace,
                                  uint64_t* out_keys, uint32_t* out_ids, float* out_dist,
                                  uint32_t* out_n, hipStream_t stream, bool whole_tiles,
                                  const uint32_t** d_candidates) {
    if (d_candidates) *d_candidates = nullptr;
    if (s.dtype != VS_BF16 || s.metric != VS_ANGULAR || k == 0 || k > VS_BATCH_MAX_K || n_q == 0 || n_q > (uint32_t)BN ||
        (s.n_slots & 63) != 0 || s.n_slots == 0)
        return hipErrorInvalidValue;
    const BatchPlan p = plan_batch(s, k);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    uint16_t* qrows = reinterpret_cast<uint16_t*>(ws);
    float* qnorm = reinterpret_cast<float*>(ws + p.off_qnorm);
    uint64_t* sample = reinterpret_cast<uint64_t*>(ws + p.off_sample);
    uint64_t* partial1 = reinterpret_cast<uint64_t*>(ws + p.off_partial1);
    uint64_t* partial2 = reinterpret_cast<uint64_t*>(ws + p.off_partial2);
    uint32_t* counters = reinterpret_cast<uint32_t*>(ws + p.off_counters);
    uint32_t* tiles = reinterpret_cast<uint32_t*>(ws + p.off_tiles);
    uint64_t* buckets = reinterpret_cast<uint64_t*>(ws + p.off_buckets);
    u32x4* regions = reinterpret_cast<u32x4*>(ws + p.off_regions);
    hipLaunchKernelGGL(prep_queries_bf16_kernel, dim3(BN / 4), dim3(256), 0, stream, d_queries, n_q, s.dim,
                       p.ksteps * BK, qrows, qnorm, counters, N_COUNTERS);
    const bool dma = k <= DMA_MAX_K;
    const size_t lds = batch_lds_bytes(dma, k);
    // (k > 12: a register list would spill; the general build serves both passes)
    void (*first)(BatchArgs) = dma ? scan_batch_bf16_kernel<true, true> : scan_batch_bf16_kernel<false, false>;
    void (*kernel)(BatchArgs) = dma ? scan_batch_bf16_kernel<true, false> : scan_batch_bf16_kernel<false, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(first),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    BatchArgs a;
    a.rows = reinterpret_cast<const u32x4*>(s.rows);
    a.norms = s.norms;
    a.present = s.present;
    a.cand = cand;
    a.qrows = reinterpret_cast<const u32x4*>(qrows);
    a.qnorm = qnorm;
    a.n_slots = s.n_slots;
    a.row_vecs = s.row_vecs;
    a.ksteps = p.ksteps;
    a.docid_base = s.docid_base;
    a.k = k;
    a.n_q = n_q;
    a.gate = nullptr;

    if (cand && batch_filters_in_place(s) && whole_tiles) {
        // the caller knows that the filter excludes next to nothing: the tiles of the store, below;
        // the candidates are counted all the same, so that the caller sees when that changes
        uint32_t* head = reinterpret_cast<uint32_t*>(ws + p.off_ghead);
        uint32_t* gsums = reinterpret_cast<uint32_t*>(ws + p.off_gsums);
        hipLaunchKernelGGL(gather_count_kernel, dim3(p.gather_blocks), dim3(256), 0, stream, s.present, cand,
                           s.n_slots >> 6, gsums);
        hipLaunchKernelGGL(gather_scan_kernel, dim3(1), dim3(256), 0, stream, gsums, p.gather_blocks, (uint32_t)s.n_slots,
                           head);
        if (d_candidates) *d_candidates = head;
    } else if (cand && batch_filters_in_place(s)) {
        // ---- under a filter: the candidates where they lie (see batch_tiles_gather_kernel) ----
        uint32_t* head = reinterpret_cast<uint32_t*>(ws + p.off_ghead);
        if (d_candidates) *d_candidates = head;
        uint32_t* gsums = reinterpret_cast<uint32_t*>(ws + p.off_gsums);
        uint32_t* map = reinterpret_cast<uint32_t*>(ws + p.off_map);
        float* norms_c = reinterpret_cast<float*>(ws + p.off_normsc);
        uint64_t* present_s = reinterpret_cast<uint64_t*>(ws + p.off_spresent);
        float* norms_s = reinterpret_cast<float*>(ws + p.off_snorms);
        u32x4* rows_s = reinterpret_cast<u32x4*>(ws + p.off_srows);
        const uint64_t n_words = s.n_slots >> 6;
        // sample slab rows past the candidates are not present, and their norms are defined
        e = hipMemsetAsync(ws + p.off_spresent, 0, p.off_srows - p.off_spresent, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gather_count_kernel, dim3(p.gather_blocks), dim3(256), 0, stream, s.present, cand, n_words, gsums);
        hipLaunchKernelGGL(gather_scan_kernel, dim3(1), dim3(256), 0, stream, gsums, p.gather_blocks, (uint32_t)s.n_slots,
                           head);
        hipLaunchKernelGGL(gather_map_kernel, dim3(p.gather_blocks), dim3(256), 0, stream, s.present, cand, n_words, gsums,
                           (uint32_t)s.n_slots, map, s.norms, norms_c);
        hipLaunchKernelGGL(gather_pad_kernel, dim3(1), dim3(256), 0, stream, map, head);
        const uint64_t sample_vecs = (uint64_t)p.sample_tiles * BM * s.row_vecs;
        hipLaunchKernelGGL(gather_sample_kernel, dim3((uint32_t)((sample_vecs + 255) / 256)), dim3(256), 0, stream, a.rows,
                           s.norms, s.row_vecs, map, head, rows_s, norms_s, present_s);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        // the unbounded pass over the sample slab: lists of sample-slab row numbers -> list positions
        BatchArgs f = a;
        f.rows = rows_s;
        f.norms = norms_s;
        f.present = present_s;
        f.cand = nullptr;
        f.n_slots = (uint64_t)p.sample_tiles * BM;
        f.docid_base = 0;
        f.bound_keys = nullptr;
        f.partial = partial1;
        f.n_tiles = p.sample_tiles;
        f.sample_stride = 1;
        f.sample_count = p.sample_tiles;
        f.pass = 1;
        f.lists = p.blocks1;
        hipLaunchKernelGGL(first, dim3(p.blocks1), dim3(BTHREADS), lds, stream, f);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = launch_merge_batch(partial1, p.blocks1, k, 1, n_q, (uint64_t)p.blocks1 * k, k, VS_ANGULAR, sample, nullptr,
                               nullptr, nullptr, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gather_sample_keys_kernel, dim3((n_q * k + 255) / 256), dim3(256), 0, stream, sample, n_q * k, head);
        // the streaming pass over the rest of the list
        void (*gather_kernel)(StreamArgs) =
            n_q <= 128 ? scan_batch_stream_kernel<true, true> : scan_batch_stream_kernel<true, false>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(gather_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                S_LDS);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(batch_tiles_gather_kernel, dim3((p.n_tiles + 255) / 256), dim3(256), 0, stream, map, head,
                           s.row_vecs * 16u, tiles, counters, counters + 1);
        StreamArgs sa;
        sa.rows = a.rows;
        sa.norms = norms_c;
        sa.present = s.present;
        sa.cand = cand;
        sa.qrows = a.qrows;
        sa.qnorm = qnorm;
        sa.bound_keys = sample;
        sa.tiles = tiles;
        sa.tile_count = counters;
        sa.regions = regions;
        sa.region_count = counters + 2 + BN;
        sa.overflow = counters + 1;
        sa.n_slots = s.n_slots;
        sa.row_vecs = s.row_vecs;
        sa.ksteps = p.ksteps;
        sa.docid_base = 0;
        sa.k = k;
        sa.n_q = n_q;
        sa.cap = p.cap;
        sa.map = map;
        sa.head = head;
        hipLaunchKernelGGL(gather_kernel, dim3(STREAM_GRID), dim3(BTHREADS), S_LDS, stream, sa);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(batch_bucket_kernel, dim3(STREAM_GRID), dim3(256), 0, stream, regions, counters + 2 + BN,
                           p.cap, buckets, counters + 2, counters + 1);
        hipLaunchKernelGGL(batch_select_kernel, dim3(n_q), dim3(256), (size_t)4 * k * 8, stream, buckets, counters + 2,
                           p.cap, sample, norms_c, qnorm, 0u, k, out_keys, out_ids, out_dist, out_n);
        hipLaunchKernelGGL(gather_translate_kernel, dim3((n_q * k + 255) / 256), dim3(256), 0, stream, map, s.docid_base,
                           n_q * k, out_keys, out_ids, out_n, k, counters + 1);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        // ... and, only where something raised `overflow`, one unbounded scan of the store itself
        a.gate = counters + 1;
        a.bound_keys = nullptr;
        a.partial = partial2;
        a.n_tiles = p.n_tiles;
        a.sample_stride = 1;
        a.sample_count = 0;
        a.pass = 2;
        a.lists = p.fallback_blocks;
        hipLaunchKernelGGL(kernel, dim3(p.fallback_blocks), dim3(BTHREADS), lds, stream, a);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        return launch_merge_batch(partial2, p.fallback_blocks, k, 1, n_q, (uint64_t)p.fallback_blocks * k, k, VS_ANGULAR,
                                  out_keys, out_ids, out_dist, out_n, stream, a.gate);
    }

    // pass 1: the first tiles, no bound
    const bool two = p.blocks2 > 0;
    a.bound_keys = nullptr;
    a.partial = partial1;
    a.n_tiles = p.n_tiles;
    a.sample_stride = p.sample_stride;
    a.sample_count = p.sample_tiles;
    a.pass = 1;
    a.lists = p.blocks1;
    hipLaunchKernelGGL(first, dim3(p.blocks1), dim3(BTHREADS), lds, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (!two)
        return launch_merge_batch(partial1, p.blocks1, k, 1, n_q, (uint64_t)p.blocks1 * k, k, VS_ANGULAR, out_keys,
                                  out_ids, out_dist, out_n, stream);
    e = launch_merge_batch(partial1, p.blocks1, k, 1, n_q, (uint64_t)p.blocks1 * k, k, VS_ANGULAR, sample, nullptr,
                           nullptr, nullptr, stream);
    if (e != hipSuccess) return e;
    // pass 2: everything else, bounded by the sample's k-th best. The streaming kernel first ...
    static const bool use_stream = getenv("VS_BATCH_NO_STREAM") == nullptr;  // measurement switch
    if (use_stream) {
        StreamArgs sa;
        void (*stream_kernel)(StreamArgs) =
            n_q <= 128 ? scan_batch_stream_kernel<false, true> : scan_batch_stream_kernel<false, false>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(batch_tiles_kernel, dim3((p.n_tiles + 255) / 256), dim3(256), 0, stream, s.present, cand,
                           s.n_slots >> 6, p.n_tiles, p.sample_stride, p.sample_tiles, tiles, counters);
        sa.rows = a.rows;
        sa.norms = s.norms;
        sa.present = s.present;
        sa.cand = cand;
        sa.qrows = a.qrows;
        sa.qnorm = qnorm;
        sa.bound_keys = sample;
        sa.tiles = tiles;
        sa.tile_count = counters;
        sa.regions = regions;
        sa.region_count = counters + 2 + BN;
        sa.overflow = counters + 1;
        sa.n_slots = s.n_slots;
        sa.row_vecs = s.row_vecs;
        sa.ksteps = p.ksteps;
        sa.docid_base = s.docid_base;
        sa.k = k;
        sa.n_q = n_q;
        sa.cap = p.cap;
        sa.map = nullptr;
        sa.head = nullptr;
        hipLaunchKernelGGL(stream_kernel, dim3(STREAM_GRID), dim3(BTHREADS), S_LDS, stream, sa);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(batch_bucket_kernel, dim3(STREAM_GRID), dim3(256), 0, stream, regions, counters + 2 + BN,
                           p.cap, buckets, counters + 2, counters + 1);
        hipLaunchKernelGGL(batch_select_kernel, dim3(n_q), dim3(256), (size_t)4 * k * 8, stream, buckets, counters + 2,
                           p.cap, sample, s.norms, qnorm, s.docid_base, k, out_keys, out_ids, out_dist, out_n);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        a.gate = counters + 1;  // ... and the general kernel only where it raised `overflow`
    }
    a.bound_keys = sample;
    a.partial = partial2;
    a.pass = 2;
    a.lists = p.blocks2 + 1;
    hipLaunchKernelGGL(kernel, dim3(p.blocks2), dim3(BTHREADS), lds, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_merge_batch(partial2, p.blocks2 + 1, k, 1, n_q, (uint64_t)(p.blocks2 + 1) * k, k, VS_ANGULAR,
                              out_keys, out_ids, out_dist, out_n, stream, a.gate);
}


namespace {

struct GatherPlan {
    uint64_t cap;  // rows of the dense slab: the candidate count rounded up to whole tiles
    uint32_t n_blocks;
    size_t off_head, off_map, off_present, off_norms, off_rows, off_inner, total;
};

GatherPlan plan_gather(const SlabView& s, uint64_t max_candidates, uint32_t k) {
    GatherPlan g;
    g.cap = (max_candidates + BM - 1) / BM * BM;
    if (g.cap == 0) g.cap = BM;
    g.n_blocks = (uint32_t)(((s.n_slots >> 6) + GATHER_BLOCK_WORDS - 1) / GATHER_BLOCK_WORDS);
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t off = up((size_t)g.n_blocks * 4);  // block sums / offsets at +0
    g.off_head = off;
    off += 256;
    g.off_map = off;
    off += up((size_t)g.cap * 4);
    g.off_present = off;
    off += up((size_t)g.cap / 8 + 64);
    g.off_norms = off;
    off += up((size_t)g.cap * 4);
    g.off_rows = off;
    off += up((size_t)g.cap * s.row_vecs * 16);
    g.off_inner = off;
    SlabView c = s;
    c.n_slots = g.cap;
    g.total = off + (size_t)plan_batch(c, k).total;
    return g;
}

}  // namespace

uint64_t batch_gather_workspace_bytes(const SlabView& s, uint64_t max_candidates, uint32_t k) {
    return plan_gather(s, max_candidates, k).total;
}

hipError_t launch_count_candidates(const SlabView& s, const uint64_t* cand, uint32_t* d_block_sums,
                                   uint32_t* d_count, hipStream_t stream) {
    const uint32_t n_blocks = (uint32_t)(((s.n_slots >> 6) + GATHER_BLOCK_WORDS - 1) / GATHER_BLOCK_WORDS);
    hipLaunchKernelGGL(gather_count_kernel, dim3(n_blocks), dim3(256), 0, stream, s.present, cand, s.n_slots >> 6,
                       d_block_sums);
    hipLaunchKernelGGL(gather_scan_kernel, dim3(1), dim3(256), 0, stream, d_block_sums, n_blocks, 0xFFFFFFFFu, d_count);
    return hipGetLastError();
}

uint64_t count_candidates_workspace_bytes(const SlabView& s) {
    return (((s.n_slots >> 6) + GATHER_BLOCK_WORDS - 1) / GATHER_BLOCK_WORDS) * 4 + 256;
}

hipError_t launch_scan_batch_bf16_gathered(const SlabView& s, const float* d_queries, uint32_t n_q,
                                           const uint64_t* cand, uint64_t max_candidates, uint32_t k,
                                           void* workspace, uint64_t* out_keys, uint32_t* out_ids, float* out_dist,
                                           uint32_t* out_n, hipStream_t stream) {
    if (!cand || max_candidates == 0 || max_candidates > 0xFFFFFF00ull || (s.n_slots & 63) != 0 || s.n_slots == 0 ||
        (!out_keys && !(out_ids && out_n)))
        return hipErrorInvalidValue;
    const GatherPlan g = plan_gather(s, max_candidates, k);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    uint32_t* block_sums = reinterpret_cast<uint32_t*>(ws);
    uint32_t* head = reinterpret_cast<uint32_t*>(ws + g.off_head);
    uint32_t* map = reinterpret_cast<uint32_t*>(ws + g.off_map);
    uint64_t* present_c = reinterpret_cast<uint64_t*>(ws + g.off_present);
    float* norms_c = reinterpret_cast<float*>(ws + g.off_norms);
    u32x4* rows_c = reinterpret_cast<u32x4*>(ws + g.off_rows);
    const uint64_t n_words = s.n_slots >> 6;
    // rows past the candidates are not present; their norms and the map entries the translation may
    // never reach still have to be defined
    hipError_t e = hipMemsetAsync(ws + g.off_map, 0, g.off_rows - g.off_map, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gather_count_kernel, dim3(g.n_blocks), dim3(256), 0, stream, s.present, cand, n_words, block_sums);
    hipLaunchKernelGGL(gather_scan_kernel, dim3(1), dim3(256), 0, stream, block_sums, g.n_blocks, (uint32_t)g.cap, head);
    hipLaunchKernelGGL(gather_map_kernel, dim3(g.n_blocks), dim3(256), 0, stream, s.present, cand, n_words, block_sums,
                       (uint32_t)g.cap, map, s.norms, static_cast<float*>(nullptr));
    const uint64_t vecs = g.cap * s.row_vecs;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)((vecs + 255) / 256)), dim3(256), 0, stream,
                       reinterpret_cast<const u32x4*>(s.rows), s.norms, s.row_vecs, map, head, rows_c, norms_c, present_c);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    SlabView c = s;
    c.rows = rows_c;
    c.norms = norms_c;
    c.present = present_c;
    c.n_slots = g.cap;
    c.docid_base = 0;
    e = launch_scan_batch_bf16(c, d_queries, n_q, nullptr, k, ws + g.off_inner, out_keys, out_ids, out_dist, out_n, stream);
    if (e != hipSuccess) return e;
    const uint32_t total = n_q * k;
    hipLaunchKernelGGL(gather_translate_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, map, s.docid_base, total,
                       out_keys, out_ids, out_n, k, static_cast<const uint32_t*>(nullptr));
    return hipGetLastError();
}

}  // namespace vs
