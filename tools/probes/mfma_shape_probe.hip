// Bare matrix-instruction loops on random register operands, two waves per SIMD on every CU: does the chip
// hold a different clock on v_mfma_f32_16x16x4_f32 than on v_mfma_f32_32x32x2_f32 (the same flop per cycle)?
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_shape_probe tools/probes/mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void loop_kernel(const float* in, float* out, uint32_t iters, uint64_t* stamps) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = in[(tid * 16 + i) & 0xFFFFF];
        b[i] = in[(tid * 16 + 8 + i) & 0xFFFFF];
    }
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.0f;
    if (SHAPE == 32) {
        f32x16 acc0 = {0}, acc1 = {0};
        for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i + 1], b[i + 1], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc0[e] + acc1[e];
    } else {
        f32x4 acc[4] = {{0}, {0}, {0}, {0}};
        for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) {  // 16 instructions of half the size = the same flop as above
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i + (j & 1)], b[i + (j >> 1)], acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += acc[j][e];
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[tid] = sum;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int main() {
    const int blocks = 256, threads = 512;
    float *in, *out;
    uint64_t* stamps;
    hipMalloc(&in, (1 << 20) * 4);
    hipMalloc(&out, blocks * threads * 4);
    hipMalloc(&stamps, blocks * 16);
    float* h = (float*)malloc((1 << 20) * 4);
    srand(1);
    for (int i = 0; i < (1 << 20); ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, (1 << 20) * 4, hipMemcpyHostToDevice);
    const uint32_t iters = 40000;  // 8 (or 16) instructions of 64 (32) cycles each: 512 cycles per iteration and wave
    for (int round = 0; round < 3; ++round)
        for (int shape = 0; shape < 2; ++shape) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int w = 0; w < 20; ++w) {  // about 0.4 s of warm-up per arm
                if (shape == 0) hipLaunchKernelGGL(loop_kernel<32>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, stamps);
                else hipLaunchKernelGGL(loop_kernel<16>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, stamps);
            }
            hipEventRecord(e0, 0);
            for (int w = 0; w < 10; ++w) {
                if (shape == 0) hipLaunchKernelGGL(loop_kernel<32>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, stamps);
                else hipLaunchKernelGGL(loop_kernel<16>, dim3(blocks), dim3(threads), 0, 0, in, out, iters, stamps);
            }
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            uint64_t hs[2 * 256];
            hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost);
            double clk = 0;
            for (int i = 0; i < blocks; ++i) clk += (double)hs[2 * i] / (double)hs[2 * i + 1] * 100.0;
            // flop per launch: waves * iters * 8 instructions * 32*32*2*2
            const double flop = (double)blocks * 8 * iters * 8 * 4096.0 * 10;
            printf("%s: %.3f ms per launch, %.1f TFLOP/s, in-kernel clock %.0f MHz, cycles per iteration %.0f\n",
                   shape == 0 ? "32x32x2 " : "16x16x4 ", ms / 10, flop / ms / 1e9, clk / blocks,
                   (double)hs[0] / iters);
        }
    return 0;
}
