// Micro-benchmark: sustained fp16 MFMA throughput with operands re-read from LDS at the conv
// kernel's ratio (4 KiB of ds_read_b128 per 32co x 32px x K32 block), 32x32x16 vs 16x16x32.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int RE = 1>
__global__ void __launch_bounds__(512) k(const f16* __restrict__ src, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 16; i += 512) ((uint4*)smem)[i] = ((const uint4*)src)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = smem + wave * 4096 + lane * 16;
    if (SHAPE == 32) {
        f32x16 acc0 = {0}, acc1 = {0};
        for (int it = 0; it < iters / RE; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int o = ((it + u) & 7) * 4096 * 0 + u * 1024 * 0;   // same 4 KiB window, different slots
                f16x8 a0 = *(const f16x8*)(base + o + 0), a1 = *(const f16x8*)(base + o + 1024);
                f16x8 b0 = *(const f16x8*)(base + o + 2048), b1 = *(const f16x8*)(base + o + 3072);
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
#pragma unroll
                for (int r = 0; r < RE; ++r) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc1, 0, 0, 0);
                }
            }
        }
        float s = 0;
        for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        f32x4 acc[4] = {{0}, {0}, {0}, {0}};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                f16x8 a0 = *(const f16x8*)(base + 0), a1 = *(const f16x8*)(base + 1024);
                f16x8 b0 = *(const f16x8*)(base + 2048), b1 = *(const f16x8*)(base + 3072);
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[3], 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 4; ++i) s += acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

int main() {
    f16* src; float* out;
    hipMalloc(&src, 65536); hipMalloc(&out, 256 * 512 * 4);
    f16* h = (f16*)malloc(65536);
    unsigned s = 1;
    for (int i = 0; i < 32768; ++i) { s = s * 1664525u + 1013904223u; h[i] = (f16)(((int)(s >> 16) % 2001 - 1000) / 1000.0f); }
    hipMemcpy(src, h, 65536, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<32, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int iters = 40000;   // x8 unrolled blocks of 32co x 32px x K32 per wave
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {32, 16, 322, 328}) {
            hipEventRecord(e0);
            if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(256), dim3(512), 65536, 0, src, out, iters);
            else if (shape == 322) hipLaunchKernelGGL((k<32, 2>), dim3(256), dim3(512), 65536, 0, src, out, iters);
            else if (shape == 328) hipLaunchKernelGGL((k<32, 8>), dim3(256), dim3(512), 65536, 0, src, out, iters);
            else hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 65536, 0, src, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 256.0 * 8 * iters * 8 * (2.0 * 32 * 32 * 32);
            printf("shape %dx%d: %.1f ms  %.0f TFLOP/s\n", shape, shape, ms, flop / ms / 1e9);
        }
    return 0;
}
