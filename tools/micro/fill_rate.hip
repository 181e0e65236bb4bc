// Micro-benchmark (r04): how fast can ONE workgroup (4 waves) pull a once-read block from global memory into LDS --
// (a) LDS-DMA (global_load_lds_dwordx4, what the conv rings use) against (b) global_load_dwordx4 into VGPRs + ds_write_b128 --
// with few (16) and all (256) workgroups active, the block L2-cold (each workgroup its own region, read once) -- the regime of a
// single-tile launch (one 8x32 patch per CU: 80..240 KiB per workgroup and launch, nothing re-read).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/fill_rate.hip -o tools/micro/fill_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// each wave moves `kib_per_wave` KiB, 1 KiB per instruction, DEPTH instructions in flight (register path) / all in flight (DMA)
template <int MODE, int DEPTH>
__global__ void __launch_bounds__(256) fill(const char* __restrict__ src, size_t wg_stride, int kib_per_wave, unsigned long long* __restrict__ t,
                                            float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + (size_t)blockIdx.x * wg_stride + (size_t)wave * kib_per_wave * 1024 + lane * 16;
    const char* ub = src + (size_t)blockIdx.x * wg_stride + (size_t)wave * kib_per_wave * 1024;                 // wave-uniform part of the address
    const uint64_t uv = (uint64_t)ub;
    const char* ubase = (const char*)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(uv >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)uv));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem + (uint32_t)wave * 32768;      // 32 KiB of LDS per wave, reused round robin
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 0) {
        for (int i = 0; i < kib_per_wave; ++i) {
            const uint32_t m = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(i & 31) * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"((uint32_t)(i * 1024 + lane * 16)), "s"(ubase), "s"(m) : "memory");
            if ((i & 31) == 31) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");              // the ring: at most 48 KiB ahead of itself
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        u32x4 r[DEPTH];
        int issued = 0;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d, ++issued) r[d] = *(const u32x4*)(base + (size_t)issued * 1024);
        for (int i = 0; i < kib_per_wave; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                *(u32x4*)(smem + wave * 32768 + ((i + d) & 31) * 1024 + lane * 16) = r[d];     // compiler waits for r[d] only (vmcnt counts down in order)
                if (issued < kib_per_wave) r[d] = *(const u32x4*)(base + (size_t)issued * 1024);
                ++issued;
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = t0; t[blockIdx.x * 2 + 1] = t1; }
    if (sink) sink[blockIdx.x * 256 + threadIdx.x] = ((const float*)smem)[threadIdx.x];       // keep the LDS contents observable
}

int main() {
    const size_t region = 1 << 20;                       // 1 MiB per workgroup: never the same line twice
    char* src; unsigned long long* t; float* sink;
    (void)hipMalloc(&src, 256 * region * 8); (void)hipMalloc(&t, 256 * 16); (void)hipMalloc(&sink, 256 * 256 * 4);
    (void)hipMemset(src, 1, 256 * region * 8);
    std::vector<unsigned long long> ht(512);
    printf("one workgroup = 4 waves; KiB per workgroup; us = median over workgroups of (last wave done - start); GB/s per workgroup\n");
    for (int wgs : {16, 256})
        for (int kib_wg : {80, 160, 240})
            for (int mode = 0; mode < 3; ++mode) {
                const int kpw = kib_wg / 4;
                double best = 1e9;
                for (int rep = 0; rep < 5; ++rep) {
                    const char* s = src + (size_t)(rep % 8) * 256 * region;      // a region not touched by the previous launch
                    if (mode == 0) hipLaunchKernelGGL((fill<0, 1>), dim3(wgs), dim3(256), 131072, 0, s, region, kpw, t, sink);
                    if (mode == 1) hipLaunchKernelGGL((fill<1, 4>), dim3(wgs), dim3(256), 131072, 0, s, region, kpw, t, sink);
                    if (mode == 2) hipLaunchKernelGGL((fill<1, 10>), dim3(wgs), dim3(256), 131072, 0, s, region, kpw, t, sink);
                    (void)hipDeviceSynchronize();
                    (void)hipMemcpy(ht.data(), t, wgs * 16, hipMemcpyDeviceToHost);
                    std::vector<double> d;
                    for (int i = 0; i < wgs; ++i) d.push_back((ht[2 * i + 1] - ht[2 * i]) * 0.01);
                    std::sort(d.begin(), d.end());
                    if (rep > 0) best = std::min(best, d[d.size() / 2]);
                }
                printf("wgs %3d  %3d KiB  %-28s %6.2f us  %6.1f GB/s\n", wgs, kib_wg,
                       mode == 0 ? "LDS-DMA" : mode == 1 ? "global_load x4 + ds_write" : "global_load x10 + ds_write", best, kib_wg * 1024 / best / 1e3);
            }
    return 0;
}
