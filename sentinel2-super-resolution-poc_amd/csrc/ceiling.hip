// Measured ceilings for the fp16 trunk kernel's roofline claim (bench.py `secondary.mfma_ceiling`).
//
// The 2.5 PFLOP/s dense fp16 peak is 1024 FLOP per clock and SIMD at 2.4 GHz; under a dense MFMA stream the part holds its
// 1400 W socket cap by lowering the clock, so what a kernel can reach is set by how much power its operand traffic takes next to
// the matrix pipe.  Three loops with the MFMA count, the LDS operand reads and the LDS-DMA fill of conv_trunk_f16's conv1-4 form
// (conv_trunk.hip: per 32x32-pixel patch and 16-channel plane 288 v_mfma_f32_32x32x16_f16, 216 ds_read_b128 = 0.75 KiB per MFMA,
// 46 KiB of global_load_lds_dwordx4), and nothing else -- no epilogue, no stores, no bias, no patch boundaries:
//   mode 0  bare      operands stay in registers
//   mode 1  lds       + the operand reads from LDS (0.75 KiB per MFMA)
//   mode 2  lds+dma   + the LDS ring refilled by LDS-DMA from a buffer that streams from HBM (48 KiB per 288 MFMAs, 3-deep ring,
//                     counted vmcnt + one barrier per stage like the kernel)
//   mode 3            as 2 with half the fill (24 KiB per 288 MFMAs): what a schedule that moved half the bytes per FLOP would get
//   mode 4            as 2 from an 8-MB source that stays in L2 / MALL: the LDS-DMA issue and fill without the HBM side
//   mode 5  the kernel's own mix: 36 KiB of the 48 streamed from HBM (the slab plane), 12 KiB from the cached source (the weights:
//           every workgroup fetches the same ones), and 8 KiB stored per stage (the launch's output: 32 channels of fp16 per pixel)
//           -- per launch of 16 images 264 MB read + 59 MB written against the 243 + 67 MB the counters see for conv1-4
//   mode 6  conv5's mix (conv_trunk_f16<2, ...>: 64 output channels, so 576 MFMAs per 36-KiB slab plane and 18 KiB of weights, 0.44 LDS
//           reads per MFMA, and per patch 192 KiB of residual read (fp16 x + its e4m3 lo plane) and 192 KiB stored), scaled to this ring's
//           48-KiB stage: 384 MFMAs per stage, 32 KiB streamed from HBM (21.3 slab + 10.7 residual, taken through the ring as well),
//           16 KiB from the cached source (12 weights + the halo the neighbours' L2 lines serve), 12 KiB stored: 117 B of HBM traffic
//           per MFMA against 114 algorithmic (99 in the counters) -- what a loop with conv5's bytes per FLOP and no epilogue sustains
//   mode 7 / 8  as 2 from a 100-MB / 200-MB source: past the L2s (8 x 4 MB), inside the 256-MB Infinity Cache -- the dense tensor of a launch
//           group of 4 / 8 images: what a schedule whose working set stayed cache-resident would be fed at
// Random fp16 operands in (-1, 1) (toggle rates, and with them power, depend on the data).  Two waves per SIMD so that LDS latency
// hides without hand scheduling: these are ceilings, the occupancy is free to choose.  Diagnostic entry; nothing of the product
// calls it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

#include "s2sr_internal.h"

namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int STAGE = 48 * 1024;     // LDS bytes of one pipeline stage (the kernel's 46 KiB rounded up to 6 KiB per wave)
constexpr int RING = 3;              // stages in the ring (conv1-4 form: 3 x 46 KiB)
constexpr int WAVES = 8;
constexpr int PW_FULL = STAGE / 1024 / WAVES;  // LDS-DMA instructions per wave and stage (48 KiB per stage)
constexpr int STEPS = 12;            // per wave and stage: 12 steps of 3 MFMAs = 36 (x 8 waves = the kernel's 288 per stage)
// ds_read_b128 per wave and stage: 27 = 0.75 per MFMA (12 B fragments, 15 A fragments)
constexpr int STEPS5 = 16;           // conv5's mix: 16 steps of 3 = 48 per wave and stage (x 8 = 384), 21 reads = 0.44 per MFMA

__device__ __forceinline__ void mfma(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void glds16(const char* base, uint32_t voff, uint32_t lds_addr) {
    // base and LDS address are wave-uniform by construction; say so (the mix mode derives them from the wave index)
    const uint64_t v = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    base = (const char*)(((uint64_t)hi << 32) | lo);
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ f16x8 lds16(const char* smem, uint32_t off) { return *(const f16x8*)(smem + off); }

__global__ void fill_random_f16(uint32_t* dst, size_t n_words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t s = (uint32_t)i * 2654435761u + 0x9e3779b9u;
        s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; s *= 3266489917u; s ^= s >> 16;
        const f16 lo = (f16)(((int)(s & 0xffff) - 32768) / 32768.0f), hi = (f16)(((int)(s >> 16) - 32768) / 32768.0f);
        dst[i] = (uint32_t)__builtin_bit_cast(uint16_t, lo) | ((uint32_t)__builtin_bit_cast(uint16_t, hi) << 16);
    }
}

template <int MODE, int PW, int MIX = 0, int NSTEPS = STEPS>   // MODE 0 bare, 1 + LDS reads, 2 + LDS-DMA of PW KiB per wave and stage; MIX 1: mode 5, 2: mode 6
__global__ void __launch_bounds__(512) mfma_ceiling_kernel(const char* __restrict__ src, uint32_t nchunks, float* __restrict__ sink, int stages,
                                                           char* __restrict__ store, uint32_t store_kib) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the ring starts full of operand data (all modes take their first fragments from it)
    {
        const uint4* s4 = (const uint4*)(src + (size_t)(blockIdx.x % nchunks) * STAGE);
        for (int i = threadIdx.x; i < RING * STAGE / 16; i += 512) ((uint4*)smem)[i] = s4[i % (STAGE / 16)];
    }
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const uint32_t lane_off = (uint32_t)lane * 16;
    f16x8 a0 = lds16(smem, lane_off), a1 = lds16(smem, 1024 + lane_off), a2 = lds16(smem, 2048 + lane_off), b = lds16(smem, 3072 + lane_off);

    // LDS-DMA of stage s: this wave's PW KiB of chunk (s * gridDim.x + blockIdx.x) % nchunks into ring slot s % RING
    constexpr uint32_t kCached = 160;      // chunks at the front of the source that MIX's "weight" pieces keep re-reading (7.5 MB)
    auto dma = [&](int s, int p) {
        uint32_t chunk = ((uint32_t)s * gridDim.x + blockIdx.x) % nchunks;
        if (MIX) {
            // waves 0-3: pieces 0-4 streamed, piece 5 cached; waves 4-7: pieces 0-3 streamed, 4-5 cached -> 36 KiB / 12 KiB per stage
            // (MIX 2: pieces 0-3 streamed, 4-5 cached in every wave -> 32 KiB / 16 KiB)
            const bool cached = p >= (MIX == 2 ? 4 : (wave < 4 ? 5 : 4));
            chunk = cached ? chunk % kCached : kCached + chunk % (nchunks - kCached);
        }
        const uint32_t piece = (uint32_t)(wave * PW_FULL + p) * 1024;
        glds16(src + (size_t)chunk * STAGE, piece + lane_off, (uint32_t)(s % RING) * STAGE + piece);
    };
    if (MODE == 2) {
#pragma unroll
        for (int p = 0; p < PW; ++p) dma(0, p);
#pragma unroll
        for (int p = 0; p < PW; ++p) dma(1, p);
    }
    for (int s = 0; s < stages; ++s) {
        const uint32_t slot = (uint32_t)(s % RING) * STAGE;
        if (MODE == 2) {
            // my pieces of stage s have landed (all but the PW of stage s + 1 are done), my reads of the slot about to be refilled
            // have returned; past the barrier both hold for every wave
            // (MIX: + the one store of the stage before, issued in front of that stage's pieces)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PW * (RING - 2) + MIX) : "memory");
            if (MIX) {      // the stage's share of the launch's output: 1 KiB per wave, streaming through a 64-MB region
                const uint32_t kib = (((uint32_t)s * gridDim.x + blockIdx.x) * WAVES + (uint32_t)wave) % store_kib;
                float* q = (float*)(store + (size_t)kib * 1024 + lane_off);
                const f32x4 v4 = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]};      // (a fixed accumulator: a run-time index would put them all in scratch)
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(q), "v"(v4) : "memory");
                if (MIX == 2) {      // conv5 writes 64 channels and their lo plane: half a KiB more per wave and stage (12 KiB per stage)
                    const uint32_t kib2 = (kib + store_kib / 2) % store_kib;
                    float* q2 = (float*)(store + (size_t)kib2 * 1024 + (uint32_t)lane * 8);
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 v2 = {acc[1][0], acc[1][1]};
                    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(q2), "v"(v2) : "memory");
                }
            }
        } else if (MODE == 1) {
            asm volatile("" ::: "memory");
        }
        int rd = 0;      // fragments read so far in this stage: READS = 27 in all, 2,2,2,3 per four steps
#pragma unroll
        for (int st = 0; st < NSTEPS; ++st) {
            f16x8 na0 = a0, na1 = a1, nb = b;
            if (MODE >= 1) {
                // next step's fragments while this step's MFMAs run: B every step, A 15 times per stage
                // (27 distinct KiB of the stage's 48 per wave, starting at the wave's own pieces: nothing for the compiler to merge)
                // (MIX 2: B every step, A every third: 21 reads for 48 MFMAs -- an A fragment serves both output-channel tiles)
                auto piece = [&](int k) { return slot + (((uint32_t)wave * PW_FULL + (uint32_t)k) % (STAGE / 1024)) * 1024 + lane_off; };
                nb = lds16(smem, piece(rd++));
                if (MIX == 2) {
                    if (st % 3 == 2) na0 = lds16(smem, piece(rd++));
                } else {
                    na0 = lds16(smem, piece(rd++));
                    if (st % 4 == 3) na1 = lds16(smem, piece(rd++));
                }
            }
            if (MODE == 2 && st < PW) dma(s + 2, st);
            mfma(acc[(3 * st + 0) & 3], a0, b);
            mfma(acc[(3 * st + 1) & 3], a1, b);
            mfma(acc[(3 * st + 2) & 3], a2, b);
            a0 = na0; a1 = na1; b = nb;
        }
    }
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing in flight into LDS when the workgroup ends
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += acc[i][j];
    sink[(size_t)blockIdx.x * 512 + threadIdx.x] = sum;
}

}  // namespace

namespace s2sr {

// src: >= nchunks * 48 KiB of operand data (filled here on first use when `fill`), sink: gridDim * 512 floats.  Returns the
// launch error; FLOP of one launch = grid * stages * 8 waves * 36 MFMAs * 32768.
hipError_t launch_mfma_ceiling(int mode, char* d_src, size_t src_bytes, bool fill, float* d_sink, int grid, int stages, char* d_store,
                               size_t store_bytes, hipStream_t st) {
    if ((mode == 5 || mode == 6) && (!d_store || store_bytes < (1u << 20))) return hipErrorInvalidValue;
    if (mode < 0 || mode > 8 || grid <= 0 || stages <= 0 || src_bytes < (size_t)STAGE) return hipErrorInvalidValue;
    if (fill) hipLaunchKernelGGL(fill_random_f16, dim3(2048), dim3(256), 0, st, (uint32_t*)d_src, src_bytes / 4);
    uint32_t nchunks = (uint32_t)(src_bytes / STAGE);
    const size_t lds = (size_t)RING * STAGE;
    typedef void (*K)(const char*, uint32_t, float*, int, char*, uint32_t);
    static const K kern[7] = {mfma_ceiling_kernel<0, PW_FULL>, mfma_ceiling_kernel<1, PW_FULL>, mfma_ceiling_kernel<2, PW_FULL>,
                              mfma_ceiling_kernel<2, PW_FULL / 2>, mfma_ceiling_kernel<2, PW_FULL>, mfma_ceiling_kernel<2, PW_FULL, 1>,
                              mfma_ceiling_kernel<2, PW_FULL, 2, STEPS5>};
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [&] {
        for (int m = 0; m < 7 && attr_err == hipSuccess; ++m)
            attr_err = hipFuncSetAttribute((const void*)kern[m], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (attr_err != hipSuccess) return attr_err;
    if (mode == 4) nchunks = nchunks < 160 ? nchunks : 160;       // 7.5 MB of source: resident in L2 / MALL
    if (mode == 7) nchunks = nchunks < 2133 ? nchunks : 2133;     // 100 MB: the Infinity Cache, not the L2s
    if (mode == 8) nchunks = nchunks < 4266 ? nchunks : 4266;     // 200 MB
    hipLaunchKernelGGL(kern[mode >= 7 ? 2 : mode], dim3(grid), dim3(512), lds, st, d_src, nchunks, d_sink, stages, d_store, (uint32_t)(store_bytes >> 10));
    return hipGetLastError();
}

double mfma_ceiling_flop_per_launch(int mode, int grid, int stages) {
    return (double)grid * stages * WAVES * (3.0 * (mode == 6 ? STEPS5 : STEPS)) * 32768.0;
}
double mfma_ceiling_dma_bytes_per_launch(int mode, int grid, int stages) {
    return mode < 2 ? 0.0 : (double)grid * stages * (mode == 3 ? STAGE / 2 : STAGE);      // (mode 5: 36 of the 48 KiB come from HBM)
}

}  // namespace s2sr
