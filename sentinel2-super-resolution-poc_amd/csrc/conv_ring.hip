// conv3x3 implicit GEMM, version 2: PERSISTENT workgroups + an R-deep LDS ring fed by LDS-DMA.
//
// Same math, data layouts and GEMM orientation as conv_mfma.hip (read its header first).  What
// changes is the schedule, driven by in-kernel stamps of v1 (profiles/r01_notes.md): a v1
// workgroup spent ~25 % of its life waiting for its FIRST chunk, ~25 % waiting for the next
// chunk at every barrier (one chunk in flight cannot cover ~2 us of loaded-HBM latency) and,
// in the conv5 form, 40 % in a serialised residual read-modify-write epilogue.  Here:
//   * one workgroup per CU walks a list of output patches; the (patch, 16-channel half-chunk)
//     pairs form ONE stream of stages, so the loads of the next patch are already in flight
//     while the current patch finishes and runs its epilogue;
//   * a stage = one 16-channel slab plane (TH+2 x 34 px x 32 B) + its weights (9 x CT KiB),
//     R stages of LDS (R = 5 for one cout tile, 4 for two) -> R-2 stages always in flight
//     behind a COUNTED s_waitcnt vmcnt(N) and a raw s_barrier (never vmcnt(0) in steady state);
//   * every wave issues the same number of LDS-DMA instructions per stage (padding slots
//     re-load the last piece), which is what makes the vmcnt count exact;
//   * the residual operands of the conv5 epilogues are loaded in one burst, not one by one.
#include <stdlib.h>

#include "s2sr_internal.h"

namespace s2sr {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int WAVES_, int NP_, int CT_, int R_>
struct RingGeom {
    static constexpr int WAVES = WAVES_, NP = NP_, CT = CT_, R = R_;
    static constexpr int TH = WAVES * NP, TW = 32;
    static constexpr int SW = TW + 2, SH = TH + 2, SPX = SH * SW;
    static constexpr int PLANE = ((SPX * 32 + 1023) / 1024) * 1024;
    static constexpr int PI = PLANE / 1024;            // slab LDS-DMA instructions per stage
    static constexpr int WI = 9 * CT;                  // weight LDS-DMA instructions per stage
    static constexpr int NSTI = PI + WI;
    static constexpr int PW = (NSTI + WAVES - 1) / WAVES;   // instructions per wave per stage (uniform)
    static constexpr int STAGE_BYTES = NSTI * 1024;
    static constexpr int BIAS_OFF = R * STAGE_BYTES;
    static constexpr int LDS_BYTES = BIAS_OFF + 256;
};

// LDS-DMA, 16 B per lane: LDS[m0 + lane*16] <- global[base + voff].  Written as inline asm on
// purpose: hipcc's waitcnt pass then neither counts these loads nor drains them (with the
// builtin it put s_waitcnt vmcnt(0) in front of every ds_read of the ring and of every stage
// issue, i.e. no load ever stayed in flight).  All waits for them are the counted ones in
// wait_vm_barrier().  M0 is written in the same statement that uses it and restored;
// s_nop 4 covers a base/offset SGPR produced by v_readfirstlane just before.
__device__ __forceinline__ void glds16r(const char* base /*wave-uniform*/, uint32_t voff, uint32_t lds_addr /*wave-uniform*/) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 4\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(base), "s"(lds_addr)
        : "memory");
}
// 16-B global load hidden from hipcc's waitcnt bookkeeping (same reason as glds16r: any load
// hipcc knows about inside the ring loop makes it guard later register reuse with vmcnt(0)).
// Result is valid only after asm_wait_loads().
template <int OFF>
__device__ __forceinline__ f32x4 asm_load16(const float* addr) {
    f32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
__device__ __forceinline__ float lrelu_r(float v) { return v > 0.f ? v : __fmul_rn(v, 0.2f); }

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    // counted wait for this wave's own LDS-DMA, then the workgroup barrier: after it every
    // wave's pieces of the awaited stage are in LDS.  One asm statement with a memory clobber
    // so that no LDS read is scheduled above it.
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int CT, int NP, int WAVES, int EPI, bool UP, int R>
__global__ void __launch_bounds__(WAVES * 64) conv3x3_f16_ring(const ConvParams p) {
    using G = RingGeom<WAVES, NP, CT, R>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pcol = lane & 31, hh = lane >> 5;

    // ---- my patches.  Round `it` of the grid covers tiles [it*nwg, (it+1)*nwg); inside a round
    // the workgroups that share an XCD (same blockIdx % 8) take one contiguous run of tiles.
    const int nwg = gridDim.x;   // multiple of 8 (host)
    const int slot_in_round = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
    const int tpi = p.tilesX * p.tilesY;
    const int ntiles = tpi * p.N;
    const int my_tiles = (ntiles - slot_in_round + nwg - 1) / nwg;   // >= 0
    const int NH = 2 * p.nchunks;                                    // stages per patch
    const int S = my_tiles * NH;

    if (tid < CT * 32) ((float*)(smem + G::BIAS_OFF))[tid] = (EPI == EPI_FIRST) ? 0.f : p.bias[tid];

    // ---- per-lane pieces of this wave's PW load slots (patch independent)
    uint32_t lpix[G::PW], lkb[G::PW];
#pragma unroll
    for (int s = 0; s < G::PW; ++s) {
        int j = wave + s * WAVES;
        if (j > G::NSTI - 1) j = G::NSTI - 1;
        lpix[s] = 0;
        lkb[s] = 0;
        if (j < G::PI) {
            const int i = j * 64 + lane;
            int q = i >> 1;
            const int sl = i & 1;
            if (q >= G::SPX) q = 0;
            const int h2 = sl ^ ((q >> 3) & 1);
            const int ry = q / G::SW, rx = q - ry * G::SW;
            int ly, lx;   // offset from the patch's source origin (padded coordinates)
            if (UP) {
                ly = ((ry - 1) >> 1) + 1;
                lx = ((rx - 1) >> 1) + 1;
            } else {
                ly = ry;
                lx = rx;
            }
            lpix[s] = (uint32_t)(ly * p.sWp + lx);
            lkb[s] = (uint32_t)(h2 * 16);
        }
    }
    const size_t simg = (size_t)p.sHp * p.sWp;

    // issue cursor (tile iteration, half-chunk) and the uniform bases of its patch
    int it_i = 0, hc_i = 0;
    const char* pb0 = nullptr;   // src0 + image + patch origin
    const char* pb1 = nullptr;
    auto patch_bases = [&](int it) {
        const int tile = it * nwg + slot_in_round;
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int ty = trem / p.tilesX, tx = trem - ty * p.tilesX;
        const int y0 = ty * G::TH, x0 = tx * G::TW;
        const size_t opix = UP ? (size_t)(y0 >> 1) * p.sWp + (x0 >> 1) : (size_t)y0 * p.sWp + x0;
        pb0 = p.src0 + ((size_t)n * simg + opix) * p.rec0;
        pb1 = p.src1 + ((size_t)n * simg + opix) * p.rec1;
    };

    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    auto issue = [&](uint32_t sl_off) {   // stage (it_i, hc_i) -> LDS slot at byte offset sl_off; advances the cursor
        if (hc_i == 0) patch_bases(it_i);
        const int c = hc_i >> 1;
        const bool first = c < p.split;
        const char* sb = (first ? pb0 + c * 64 : pb1 + (c - p.split) * 64) + (hc_i & 1) * 32;
        const uint32_t rec = first ? p.rec0 : p.rec1;
        const char* wb = (const char*)p.wpack + (size_t)hc_i * (G::WI * 1024);
#pragma unroll
        for (int s = 0; s < G::PW; ++s) {
            int j = wave + s * WAVES;
            if (j > G::NSTI - 1) j = G::NSTI - 1;   // padding slot: same piece again
            const uint32_t dst = lds0 + sl_off + (uint32_t)j * 1024;
            if (j < G::PI) glds16r(sb, lpix[s] * rec + lkb[s], dst);
            else glds16r(wb, (uint32_t)((j - G::PI) * 1024 + lane * 16), dst);
        }
        if (++hc_i == NH) { hc_i = 0; ++it_i; }
    };

    // ---- B-fragment addresses inside a slab plane
    uint32_t baddr[NP][9];
#pragma unroll
    for (int np = 0; np < NP; ++np)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int q = (wave * NP + np + t / 3) * G::SW + pcol + (t % 3);
            baddr[np][t] = (uint32_t)(q * 32 + 16 * (hh ^ ((q >> 3) & 1)));
        }
    const uint32_t aaddr = G::PLANE + lane * 16;

    f32x16 acc[CT][NP];
    auto init_acc = [&]() {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            f32x16 b;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *(const f32x4*)(smem + G::BIAS_OFF + (ct * 32 + 8 * g + 4 * hh) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) b[4 * g + i] = v[i];
            }
#pragma unroll
            for (int np = 0; np < NP; ++np) acc[ct][np] = b;
        }
    };

    auto compute = [&](const char* buf) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            f16x8 a[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) a[ct] = *(const f16x8*)(buf + aaddr + (t * CT + ct) * 1024);
#pragma unroll
            for (int np = 0; np < NP; ++np) {
                const f16x8 b = *(const f16x8*)(buf + baddr[np][t]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct][np] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ct], b, acc[ct][np], 0, 0, 0);
            }
        }
    };

    // ---- epilogue of the patch at tile iteration `it` (lane owns 16 couts of one pixel per tile)
    auto epilogue = [&](int it) {
        const int tile = it * nwg + slot_in_round;
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int ty = trem / p.tilesX, tx = trem - ty * p.tilesX;
        const int y0 = ty * G::TH, x0 = tx * G::TW;
        const int x = x0 + pcol;
        bool ok[NP];
        size_t opix[NP];
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            const int y = y0 + wave * NP + np;
            ok[np] = (y < p.H) && (x < p.W);
            opix[np] = ((size_t)n * p.Hp + (y + 1)) * p.Wp + (x + 1);
        }
        // burst-load every residual operand first (independent loads, one latency)
        f32x4 res0[CT][NP][4], res1[CT][NP][4];
        if (EPI == EPI_RDB5 || EPI == EPI_RDB5_RRDB || EPI == EPI_BODY) {
            const float* P0 = (EPI == EPI_BODY) ? p.F : p.T;
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // unconditional: the padded planes make every address valid, and a load
                        // whose use is skipped would leave hipcc a "maybe pending" register that it
                        // guards with s_waitcnt vmcnt(0) inside the ring loop
                        const int cb = ct * 32 + 8 * g + 4 * hh;
                        res0[ct][np][g] = *(const f32x4*)(P0 + opix[np] * 64 + cb);
                        if (EPI == EPI_RDB5_RRDB) res1[ct][np][g] = *(const f32x4*)(p.R + opix[np] * 64 + cb);
                    }
        }
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            const int y = y0 + wave * NP + np;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cb = ct * 32 + 8 * g + 4 * hh;
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[ct][np][4 * g + i];
                    if (EPI == EPI_LRELU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = lrelu_r(v[i]);
                    } else if (EPI == EPI_RDB5) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(__fmul_rn(v[i], 0.2f), res0[ct][np][g][i]);
                        if (ok[np]) *(f32x4*)(p.T + opix[np] * 64 + cb) = v;
                    } else if (EPI == EPI_RDB5_RRDB) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            v[i] = __fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(v[i], 0.2f), res0[ct][np][g][i]), 0.2f),
                                             res1[ct][np][g][i]);
                        if (ok[np]) {
                            *(f32x4*)(p.T + opix[np] * 64 + cb) = v;
                            *(f32x4*)(p.R + opix[np] * 64 + cb) = v;
                        }
                    } else if (EPI == EPI_FIRST) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(__fmul_rn(v[i], p.in_scale), p.bias[cb + i]);
                        if (ok[np]) {
                            *(f32x4*)(p.T + opix[np] * 64 + cb) = v;
                            *(f32x4*)(p.R + opix[np] * 64 + cb) = v;
                            *(f32x4*)(p.F + opix[np] * 64 + cb) = v;
                        }
                    } else if (EPI == EPI_BODY) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(res0[ct][np][g][i], v[i]);
                    }
                    if (EPI == EPI_LAST || EPI == EPI_DEBUG) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int co = cb + i;
                            if (co >= p.cout || !ok[np]) continue;
                            float o = v[i];
                            if (EPI == EPI_DEBUG && p.act) o = lrelu_r(o);
                            if (p.out_f32) p.out_f32[(((size_t)n * p.cout + co) * p.H + y) * p.W + x] = o;
                            if (EPI == EPI_LAST && p.out_u8) {
                                // (out*255).clip(0,255).astype(uint8): truncation (cnn_super_resolution.py:232)
                                const float q = fminf(fmaxf(__fmul_rn(o, 255.0f), 0.f), 255.f);
                                p.out_u8[(((size_t)n * p.H + y) * p.W + x) * 3 + co] = (uint8_t)(int)q;
                            }
                        }
                    } else {
                        f16x4 hv;
#pragma unroll
                        for (int i = 0; i < 4; ++i) hv[i] = (f16)v[i];
                        if (ok[np]) *(f16x4*)(p.dst + opix[np] * p.dst_rec + p.dst_coff + cb * 2) = hv;
                    }
                }
            }
        }
    };

    // ---- prologue: R-1 stages in flight
#pragma unroll
    for (int r = 0; r < R - 1; ++r)
        if (r < S) issue((uint32_t)(r * G::STAGE_BYTES));
    __syncthreads();   // bias visible in LDS
    init_acc();

    int k = 0, it_c = 0, hc_c = 0;
    // one ring revolution per loop trip; every condition below is workgroup-uniform
    while (k < S) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (k < S) {
                // stage k must have landed; stages k+1 .. k+R-2 may stay in flight
                if (k + (R - 2) < S) wait_vm_barrier<G::PW*(R - 2)>();
                else wait_vm_barrier<0>();
                if (k + (R - 1) < S) issue((uint32_t)(((r + R - 1) % R) * G::STAGE_BYTES));
                compute(smem + r * G::STAGE_BYTES);
                if (++hc_c == NH) {
                    epilogue(it_c);
                    init_acc();
                    hc_c = 0;
                    ++it_c;
                }
                ++k;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------
static int env_int_r(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

template <int CT, int EPI, bool UP, int WAVES, int NP, int R>
static hipError_t launch_ring_t(const ConvParams& p, hipStream_t st) {
    using G = RingGeom<WAVES, NP, CT, R>;
    static_assert(G::LDS_BYTES <= 160 * 1024, "LDS ring does not fit");
    auto kern = conv3x3_f16_ring<CT, NP, WAVES, EPI, UP, R>;
    static bool attr_set = false;
    static int ncu = 256;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        attr_set = true;
    }
    ConvParams q = p;
    q.tilesX = (p.W + G::TW - 1) / G::TW;
    q.tilesY = (p.H + G::TH - 1) / G::TH;
    const int ntiles = q.tilesX * q.tilesY * p.N;
    int grid = ncu & ~7;                      // one persistent workgroup per CU
    if (ntiles < grid) grid = (ntiles + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), G::LDS_BYTES, st, q);
    return hipGetLastError();
}

hipError_t launch_conv_f16_ring(const ConvParams& p, int ct, int epi, bool up, hipStream_t st) {
    if (ct == 1) {
        if (epi == EPI_LRELU && !up) return launch_ring_t<1, EPI_LRELU, false, 8, 2, 5>(p, st);
        if (epi == EPI_LAST && !up) return launch_ring_t<1, EPI_LAST, false, 8, 2, 5>(p, st);
        if (epi == EPI_DEBUG)
            return up ? launch_ring_t<1, EPI_DEBUG, true, 8, 2, 5>(p, st) : launch_ring_t<1, EPI_DEBUG, false, 8, 2, 5>(p, st);
    } else if (ct == 2) {
        if (epi == EPI_LRELU)
            return up ? launch_ring_t<2, EPI_LRELU, true, 8, 2, 4>(p, st) : launch_ring_t<2, EPI_LRELU, false, 8, 2, 4>(p, st);
        if (epi == EPI_RDB5 && !up) return launch_ring_t<2, EPI_RDB5, false, 8, 2, 4>(p, st);
        if (epi == EPI_RDB5_RRDB && !up) return launch_ring_t<2, EPI_RDB5_RRDB, false, 8, 2, 4>(p, st);
        if (epi == EPI_FIRST && !up) return launch_ring_t<2, EPI_FIRST, false, 8, 2, 4>(p, st);
        if (epi == EPI_BODY && !up) return launch_ring_t<2, EPI_BODY, false, 8, 2, 4>(p, st);
        if (epi == EPI_DEBUG)
            return up ? launch_ring_t<2, EPI_DEBUG, true, 8, 2, 4>(p, st) : launch_ring_t<2, EPI_DEBUG, false, 8, 2, 4>(p, st);
    }
    return hipErrorInvalidValue;
}

// Weight layout of the ring kernel: [half-chunk hc][tap][ct][lane 0..63][j 0..7] fp16 with
//   value = W[cout = ct*32 + (lane&31)][cin = hc*16 + 8*(lane>>5) + j][tap/3][tap%3] * wscale
// -> each stage's weights are 9*CT contiguous KiB, each KiB one A fragment.
void pack_conv_weights_ring(const float* w, int cin, int cout, float wscale, void* dst_host) {
    const int nh = 2 * ((cin + 31) / 32), CT = (cout + 31) / 32;
    f16* d = (f16*)dst_host;
    for (int hc = 0; hc < nh; ++hc)
        for (int t = 0; t < 9; ++t)
            for (int ct = 0; ct < CT; ++ct)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int co = ct * 32 + (l & 31);
                        const int ci = hc * 16 + 8 * (l >> 5) + j;
                        float v = 0.f;
                        if (co < cout && ci < cin) v = w[((size_t)co * cin + ci) * 9 + t] * wscale;
                        *d++ = (f16)v;
                    }
}

}  // namespace s2sr
