// What does it cost to pick scattered 192-byte rows out of HBM? A 1536-bit binary row is one and a half
// 128-byte lines, so a picked row touches two lines whatever its neighbours do. Three layouts of the same
// 192 payload bytes per row, rows picked at random with a given probability, four lanes per row as the
// filtered Hamming scan reads them (a quad loads 64 contiguous bytes per instruction):
//   stride 192   rows back to back (the store's layout)
//   stride 256   every row at the start of its own 256 bytes (two lines of its own, the second half empty)
//   split        bytes 0..127 of row i in one slab (stride 128: one whole line), bytes 128..191 in a second
//                slab (stride 64: the line is shared with a neighbour's tail)
// Prints the time per pass and payload GB/s; run once more under `rocprofv3 --pmc FETCH_SIZE` to see the
// bytes each variant really moves (kernel names say which is which).
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/row_gather_probe tools/probes/row_gather_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            return 1;                                                             \
        }                                                                         \
    } while (0)

// LAYOUT 0: stride 192, 1: stride 256, 2: split slabs. `picked` lists the rows to read.
template <int LAYOUT>
__global__ __launch_bounds__(256) void gather_kernel(const u32x4* a, const u32x4* b, const uint32_t* picked,
                                                     uint32_t n_picked, uint32_t* out) {
    const uint32_t lane = threadIdx.x & 63, sub = lane & 3;
    const uint32_t quads = gridDim.x * 64;  // 16 quads per wave, 4 waves per workgroup
    uint32_t acc = 0;
    for (uint32_t at = blockIdx.x * 64 + (threadIdx.x >> 2); at < n_picked; at += quads * 4) {
        u32x4 x[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // four rows in flight per quad
            const uint32_t i = at + u * quads < n_picked ? at + u * quads : at;
            const uint32_t row = picked[i];
#pragma unroll
            for (int v = 0; v < 3; ++v) {  // vectors sub, sub + 4, sub + 8 of the row's twelve
                const uint32_t p = v * 4 + sub;
                const u32x4* src = LAYOUT == 0   ? a + (uint64_t)row * 12 + p
                                   : LAYOUT == 1 ? a + (uint64_t)row * 16 + p
                                   : p < 8       ? a + (uint64_t)row * 8 + p
                                                 : b + (uint64_t)row * 4 + (p - 8);
                x[u][v] = __builtin_nontemporal_load(src);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc += __builtin_popcount(x[u][v][e]);
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;  // keep the loads
}

int main(int argc, char** argv) {
    const uint64_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 50000000ull;
    void *a = nullptr, *b = nullptr;
    uint32_t *d_picked = nullptr, *d_out = nullptr;
    CHECK(hipMalloc(&a, rows * 256));  // large enough for every layout
    CHECK(hipMalloc(&b, rows * 64));
    CHECK(hipMemset(a, 0x5A, rows * 256));
    CHECK(hipMemset(b, 0xA5, rows * 64));
    CHECK(hipMalloc(&d_out, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int percent : {10, 1}) {
        std::vector<uint32_t> picked;
        uint32_t s = 12345u;
        for (uint64_t i = 0; i < rows; ++i) {
            s ^= s << 13;
            s ^= s >> 17;
            s ^= s << 5;
            if (s % 100 < (uint32_t)percent) picked.push_back((uint32_t)i);
        }
        const uint32_t n = (uint32_t)picked.size();
        if (d_picked) CHECK(hipFree(d_picked));
        CHECK(hipMalloc(&d_picked, (size_t)n * 4));
        CHECK(hipMemcpy(d_picked, picked.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        for (int layout = 0; layout < 3; ++layout) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHECK(hipEventRecord(e0));
                const dim3 grid(512), block(256);
                const u32x4* pa = static_cast<const u32x4*>(a);
                const u32x4* pb = static_cast<const u32x4*>(b);
                if (layout == 0) hipLaunchKernelGGL(gather_kernel<0>, grid, block, 0, 0, pa, pb, d_picked, n, d_out);
                else if (layout == 1) hipLaunchKernelGGL(gather_kernel<1>, grid, block, 0, 0, pa, pb, d_picked, n, d_out);
                else hipLaunchKernelGGL(gather_kernel<2>, grid, block, 0, 0, pa, pb, d_picked, n, d_out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const char* names[3] = {"stride 192", "stride 256", "split 128 + 64"};
            printf("%2d %% of %llu rows, %-15s: %8.1f us per pass, %7.1f GB/s of payload (192 B per picked row, %u rows)\n",
                   percent, (unsigned long long)rows, names[layout], best * 1e3, (double)n * 192 / (best * 1e-3) / 1e9, n);
        }
    }
    return 0;
}
