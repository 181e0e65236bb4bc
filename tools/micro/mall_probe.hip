// Probe of the Infinity Cache (256 MB, memory side) on MI355X: (A) streaming-read bandwidth against the size of the buffer that is read
// over and over -- where the cache stops serving it; (B) are freshly WRITTEN bytes retained: a buffer is written by one kernel and then read
// once by another, over and over -- the read's bandwidth against (A) for the same size; (C) the control: the same with 512 MB of other
// writes between the write and the read.  Decides whether a schedule that keeps a launch group's dense tensor (100 - 200 MB, written layer by
// layer) cache-resident can be fed at the rate profiles/r05_mfma_ceiling.txt measured from a read-only source.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mall_probe.hip -o /tmp/mall_probe && /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) read_kernel(const uint4* __restrict__ src, size_t n16, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {            // four independent 16-B loads in flight per lane
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const uint4 a = src[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) sink[0] = acc;                      // (keeps the loads alive)
}

__global__ void __launch_bounds__(256) write_kernel(uint4* __restrict__ dst, size_t n16, uint32_t seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint32_t v = (uint32_t)i * 2654435761u + seed;
        dst[i] = make_uint4(v, v ^ 0x9e3779b9u, v + 0x7f4a7c15u, ~v);
    }
}

int main() {
    const size_t MB = 1 << 20, cap = 1024 * MB;
    char *buf, *other;
    uint32_t* sink;
    CHECK(hipMalloc(&buf, cap));
    CHECK(hipMalloc(&other, 512 * MB));
    CHECK(hipMalloc(&sink, 4));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grid = 256 * 8;      // 8 workgroups of 256 per CU
    hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, st, (uint4*)buf, cap / 16, 1u);
    hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, st, (uint4*)other, 512 * MB / 16, 2u);
    CHECK(hipStreamSynchronize(st));

    const size_t sizes[] = {8, 16, 32, 64, 100, 150, 200, 230, 256, 300, 400, 1024};
    printf("A: the same buffer read over and over (GB/s by size)\n");
    for (size_t s : sizes) {
        const size_t n16 = s * MB / 16;
        const int reps = (int)(20000 / s) + 4;                 // ~20 GB per measurement
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, st, (const uint4*)buf, n16, sink);
        CHECK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, st, (const uint4*)buf, n16, sink);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %5zu MB  %8.1f GB/s  (%d passes, %.1f us each)\n", s, (double)s * MB * reps / (ms * 1e-3) / 1e9, reps, ms * 1e3 / reps);
        fflush(stdout);
    }

    const size_t wsizes[] = {32, 100, 200, 400};
    for (int variant = -1; variant < 2; ++variant) {
        printf(variant < 0 ? "A1: read, then read again, each launch timed alone (the protocol of B and C; first column the second read)\n"
               : variant == 0 ? "B: written by one kernel, then read once by another; read bandwidth (GB/s), write bandwidth (GB/s)\n"
                              : "C: the same with 512 MB of other writes between the write and the read\n");
        for (size_t s : wsizes) {
            const size_t n16 = s * MB / 16;
            const int reps = (int)(8000 / s) + 4;
            double rd_ms = 0, wr_ms = 0;
            for (int r = -2; r < reps; ++r) {
                CHECK(hipEventRecord(e0, st));
                if (variant < 0) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, st, (const uint4*)buf, n16, sink);
                else hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, st, (uint4*)buf, n16, (uint32_t)r);
                CHECK(hipEventRecord(e1, st));
                CHECK(hipStreamSynchronize(st));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 0) wr_ms += ms;
                if (variant == 1) hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, st, (uint4*)other, 512 * MB / 16, (uint32_t)r);
                CHECK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, st, (const uint4*)buf, n16, sink);
                CHECK(hipEventRecord(e1, st));
                CHECK(hipStreamSynchronize(st));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 0) rd_ms += ms;
            }
            printf("  %5zu MB  read %8.1f GB/s   %s %8.1f GB/s\n", s, (double)s * MB * reps / (rd_ms * 1e-3) / 1e9, variant < 0 ? "first read" : "write", (double)s * MB * reps / (wr_ms * 1e-3) / 1e9);
            fflush(stdout);
        }
    }
    return 0;
}
