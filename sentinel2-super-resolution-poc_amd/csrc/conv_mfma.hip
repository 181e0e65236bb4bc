// 3x3 / stride 1 / zero-pad 1 convolution as an im2col-free implicit GEMM on the gfx950
// matrix cores -- the kernel behind every conv of RRDBNet.forward
// (reference server/app/cnn_super_resolution.py:85-91,103-107,140-158).
//
// GEMM orientation:  D[Cout x pixels] = Wt[Cout x K] * X[K x pixels],  K = 9 taps x Cin.
//   v_mfma_f32_32x32x16_f16:  A = weights (row = cout, lane = cout + 32*(k/8)),
//                             B = activations (col = pixel, lane = pixel + 32*(k/8)),
//                             D: lane = pixel column, 16 registers = 16 couts
//   so every lane ends up owning 16 output channels of ONE pixel: bias, LeakyReLU, the
//   x0.2 residual adds of the RDB / RRDB and the fp16 NHWC store are all lane-local.
//
// Workgroup = WAVES waves; output patch = (WAVES*NP) rows x 32 columns of one image; wave w
// owns rows [w*NP, w*NP+NP) x all CT*32 couts, i.e. CT*NP accumulator tiles of 32x32.
// K loop: 32 input channels ("chunk") at a time.  Per chunk the workgroup streams into LDS
//   * the (TH+2) x 34 pixel slab of those 32 channels, split into two 16-channel planes
//     (plane ks = channels [16ks,16ks+16) : 32 B per pixel, 16-B halves XOR-swizzled with
//      bit 3 of the pixel index -> every ds_read_b128 of a B fragment is bank-conflict free)
//   * the chunk's weights, pre-packed on the host in exact A-fragment order
// both with LDS-DMA (global_load_lds_dwordx4: no VGPR round trip), double buffered: the loads
// of chunk c+1 are in flight while chunk c feeds 18*CT*NP MFMAs per wave.
// Activations have a physical zero halo (s2sr_internal.h), so the loader has no bounds checks;
// nearest-2x upsampling (cnn_super_resolution.py:146-154) is folded into the loader's source
// address (>>1), the upsampled tensor is never materialised.
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

template <int WAVES_, int NP_, int CT_>
struct ConvGeom {
    static constexpr int WAVES = WAVES_, NP = NP_, CT = CT_;
    static constexpr int TH = WAVES * NP, TW = 32;
    static constexpr int SW = TW + 2, SH = TH + 2, SPX = SH * SW;
    static constexpr int PLANE_BYTES = ((SPX * 32 + 1023) / 1024) * 1024;
    static constexpr int PI = PLANE_BYTES / 1024;   // LDS-DMA wave-instructions per plane
    static constexpr int NSI = 2 * PI;              // slab instructions per chunk
    static constexpr int NWI = 18 * CT;             // weight instructions per chunk (1 KiB each)
    static constexpr int W_BYTES = NWI * 1024;
    static constexpr int BUF_BYTES = 2 * PLANE_BYTES + W_BYTES;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
    static constexpr int NSL = (NSI + WAVES - 1) / WAVES;
    static constexpr int NWL = (NWI + WAVES - 1) / WAVES;
};

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : __fmul_rn(v, 0.2f); }

#define S2SR_STAMP(k)                                                              \
    do {                                                                           \
        if (TRACE && p.trace && tid == 0 && (k) < 24)                              \
            p.trace[(size_t)blockIdx.x * 24 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

template <int CT, int NP, int WAVES, int EPI, bool UP, bool TRACE = false>
__global__ void __launch_bounds__(WAVES * 64) conv3x3_f16_kernel(const ConvParams p) {
    using G = ConvGeom<WAVES, NP, CT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    S2SR_STAMP(0);

    // ---- tile decode; blockIdx -> tile is XCD-aware: the 8 XCDs (blockIdx % 8 labels the
    // blocks that share one) each take a contiguous run of tiles, so neighbouring patches
    // (shared halos) and one image's planes stay in one XCD's L2.  Bijective for any grid.
    int tile;
    {
        const int bid = blockIdx.x, nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tpi = p.tilesX * p.tilesY;
    const int n = tile / tpi;
    const int trem = tile - n * tpi;
    const int ty = trem / p.tilesX;
    const int tx = trem - ty * p.tilesX;
    const int y0 = ty * G::TH, x0 = tx * G::TW;   // logical coords of the patch origin

    // ---- per-lane source offsets of this wave's slab pieces (identical for every chunk)
    uint32_t spix[G::NSL], skb[G::NSL];
#pragma unroll
    for (int s = 0; s < G::NSL; ++s) {
        const int j = wave + s * WAVES;
        const int ks = (j >= G::PI) ? 1 : 0;
        const int i = (j - ks * G::PI) * 64 + lane;   // 16-B piece index inside the plane
        int q = i >> 1;
        const int sl = i & 1;
        if (q >= G::SPX) q = 0;                        // tail pieces land in the plane's pad
        const int hh = sl ^ ((q >> 3) & 1);
        const int ry = q / G::SW, rx = q - ry * G::SW;
        int py, px;                                     // padded source coordinates
        if (UP) {
            py = ((y0 + ry - 1) >> 1) + 1;
            px = ((x0 + rx - 1) >> 1) + 1;
        } else {
            py = y0 + ry;
            px = x0 + rx;
        }
        spix[s] = (uint32_t)(py * p.sWp + px);
        skb[s] = (uint32_t)(ks * 32 + hh * 16);
    }
    const size_t simg = (size_t)p.sHp * p.sWp;
    const char* src0n = p.src0 + (size_t)n * simg * p.rec0;
    const char* src1n = p.src1 + (size_t)n * simg * p.rec1;
    const char* wsrc = (const char*)p.wpack + lane * 16;

    auto stage = [&](int c, char* buf) {
        const bool first = c < p.split;
        const char* g = first ? (src0n + c * 64) : (src1n + (c - p.split) * 64);
        const uint32_t rec = first ? p.rec0 : p.rec1;   // wave-uniform
#pragma unroll
        for (int s = 0; s < G::NSL; ++s) {
            const int j = wave + s * WAVES;
            if (G::NSI % WAVES == 0 || j < G::NSI) glds16(g + (spix[s] * rec + skb[s]), buf + j * 1024);
        }
        const char* wg = wsrc + (size_t)c * G::W_BYTES;
#pragma unroll
        for (int s = 0; s < G::NWL; ++s) {
            const int j = wave + s * WAVES;
            if (G::NWI % WAVES == 0 || j < G::NWI) glds16(wg + j * 1024, buf + 2 * G::PLANE_BYTES + j * 1024);
        }
    };

    // ---- B-fragment LDS addresses: lane = (pixel column, k half)
    const int pcol = lane & 31, hh = lane >> 5;
    uint32_t baddr[NP][9];
#pragma unroll
    for (int np = 0; np < NP; ++np)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int q = (wave * NP + np + t / 3) * G::SW + pcol + (t % 3);
            baddr[np][t] = (uint32_t)(q * 32 + 16 * (hh ^ ((q >> 3) & 1)));
        }
    const uint32_t aaddr = 2 * G::PLANE_BYTES + lane * 16;

    // ---- accumulators start from the bias (EPI_FIRST adds it after the 1/255 scale)
    f32x16 acc[CT][NP];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        f32x16 b;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            b[r] = (EPI == EPI_FIRST) ? 0.f : p.bias[ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
#pragma unroll
        for (int np = 0; np < NP; ++np) acc[ct][np] = b;
    }

    auto compute = [&](const char* buf) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 a[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    a[ct] = *(const f16x8*)(buf + aaddr + ((t * 2 + ks) * CT + ct) * 1024);
#pragma unroll
                for (int np = 0; np < NP; ++np) {
                    const f16x8 b = *(const f16x8*)(buf + baddr[np][t] + ks * G::PLANE_BYTES);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[ct][np] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ct], b, acc[ct][np], 0, 0, 0);
                }
            }
        }
    };

    // ---- K loop, 2 LDS buffers, one barrier per chunk
    char* buf0 = smem;
    char* buf1 = smem + G::BUF_BYTES;
    const int nch = p.nchunks;
    S2SR_STAMP(1);
    stage(0, buf0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    S2SR_STAMP(2);
    for (int c = 0;;) {
        if (c + 1 < nch) stage(c + 1, buf1);
        compute(buf0);
        S2SR_STAMP(3 + 2 * c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        S2SR_STAMP(4 + 2 * c);
        if (++c >= nch) break;
        if (c + 1 < nch) stage(c + 1, buf0);
        compute(buf1);
        S2SR_STAMP(3 + 2 * c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        S2SR_STAMP(4 + 2 * c);
        if (++c >= nch) break;
    }

    // ---- epilogue: lane owns couts {ct*32 + 8g + 4hh + i} of pixel (row, pcol)
    const int x = x0 + pcol;
#pragma unroll
    for (int np = 0; np < NP; ++np) {
        const int y = y0 + wave * NP + np;
        if (y >= p.H || x >= p.W) continue;
        const size_t opix = ((size_t)n * p.Hp + (y + 1)) * p.Wp + (x + 1);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cb = ct * 32 + 8 * g + 4 * hh;   // first of 4 consecutive couts
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[ct][np][4 * g + i];

                if (EPI == EPI_LRELU) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = lrelu(v[i]);
                } else if (EPI == EPI_RDB5) {
                    f32x4* tp = (f32x4*)(p.T + opix * 64 + cb);
                    const f32x4 t = *tp;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(__fmul_rn(v[i], 0.2f), t[i]);
                    *tp = v;
                } else if (EPI == EPI_RDB5_RRDB) {
                    f32x4* tp = (f32x4*)(p.T + opix * 64 + cb);
                    f32x4* rp = (f32x4*)(p.R + opix * 64 + cb);
                    const f32x4 t = *tp;
                    const f32x4 r = *rp;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        v[i] = __fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(v[i], 0.2f), t[i]), 0.2f), r[i]);
                    *tp = v;
                    *rp = v;
                } else if (EPI == EPI_FIRST) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(__fmul_rn(v[i], p.in_scale), p.bias[cb + i]);
                    *(f32x4*)(p.T + opix * 64 + cb) = v;
                    *(f32x4*)(p.R + opix * 64 + cb) = v;
                    *(f32x4*)(p.F + opix * 64 + cb) = v;
                } else if (EPI == EPI_BODY) {
                    const f32x4 f = *(const f32x4*)(p.F + opix * 64 + cb);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(f[i], v[i]);
                }

                if (EPI == EPI_LAST || EPI == EPI_DEBUG) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int co = cb + i;
                        if (co >= p.cout) continue;
                        float o = v[i];
                        if (EPI == EPI_DEBUG && p.act) o = lrelu(o);
                        if (p.out_f32) p.out_f32[(((size_t)n * p.cout + co) * p.H + y) * p.W + x] = o;
                        if (EPI == EPI_LAST && p.out_u8) {
                            // (out*255).clip(0,255).astype(uint8): truncation (cnn_super_resolution.py:232)
                            float q = fminf(fmaxf(__fmul_rn(o, 255.0f), 0.f), 255.f);
                            p.out_u8[(((size_t)n * p.H + y) * p.W + x) * 3 + co] = (uint8_t)(int)q;
                        }
                    }
                } else {
                    f16x4 hv;
#pragma unroll
                    for (int i = 0; i < 4; ++i) hv[i] = (f16)v[i];
                    *(f16x4*)(p.dst + opix * p.dst_rec + p.dst_coff + cb * 2) = hv;
                }
            }
        }
    }
    S2SR_STAMP(23);
}

// ------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------
// Workgroup shapes: 8 waves x 2 rows (16 x 32 px patch, 1 workgroup / CU) or 4 waves x 2 rows
// (8 x 32 px patch; with one cout tile the LDS footprint is exactly 80 KiB -> 2 workgroups / CU).
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

template <int CT, int EPI, bool UP, int WAVES, int NP>
static hipError_t launch_t(const ConvParams& p, hipStream_t st) {
    using G = ConvGeom<WAVES, NP, CT>;
    auto kern = conv3x3_f16_kernel<CT, NP, WAVES, EPI, UP>;
    static bool attr_set = false;   // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    ConvParams q = p;
    q.tilesX = (p.W + G::TW - 1) / G::TW;
    q.tilesY = (p.H + G::TH - 1) / G::TH;
    const int grid = q.tilesX * q.tilesY * p.N;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), G::LDS_BYTES, st, q);
    return hipGetLastError();
}

hipError_t launch_conv_f16_trace(const ConvParams& p, int ct, hipStream_t st) {
    if (ct == 1) {
        using G = ConvGeom<8, 2, 1>;
        auto kern = conv3x3_f16_kernel<1, 2, 8, EPI_LRELU, false, true>;
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) return e;
        ConvParams q = p;
        q.tilesX = (p.W + G::TW - 1) / G::TW; q.tilesY = (p.H + G::TH - 1) / G::TH;
        hipLaunchKernelGGL(kern, dim3(q.tilesX * q.tilesY * p.N), dim3(512), G::LDS_BYTES, st, q);
    } else {
        using G = ConvGeom<8, 2, 2>;
        auto kern = conv3x3_f16_kernel<2, 2, 8, EPI_RDB5, false, true>;
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
        if (e != hipSuccess) return e;
        ConvParams q = p;
        q.tilesX = (p.W + G::TW - 1) / G::TW; q.tilesY = (p.H + G::TH - 1) / G::TH;
        hipLaunchKernelGGL(kern, dim3(q.tilesX * q.tilesY * p.N), dim3(512), G::LDS_BYTES, st, q);
    }
    return hipGetLastError();
}

template <int CT, int EPI, bool UP>
static hipError_t launch_w(const ConvParams& p, hipStream_t st) {
    static const int small_ct1 = env_int("S2SR_CT1_WAVES", 8);
    if (CT == 1 && EPI == EPI_LRELU && !UP && small_ct1 == 4) return launch_t<CT, EPI, UP, 4, 2>(p, st);
    return launch_t<CT, EPI, UP, 8, 2>(p, st);
}

hipError_t launch_conv_f16(const ConvParams& p, int ct, int epi, bool up, hipStream_t st) {
    if (ct == 1) {
        if (epi == EPI_LRELU && !up) return launch_w<1, EPI_LRELU, false>(p, st);
        if (epi == EPI_LAST && !up) return launch_w<1, EPI_LAST, false>(p, st);
        if (epi == EPI_DEBUG) return up ? launch_w<1, EPI_DEBUG, true>(p, st) : launch_w<1, EPI_DEBUG, false>(p, st);
    } else if (ct == 2) {
        if (epi == EPI_LRELU) return up ? launch_w<2, EPI_LRELU, true>(p, st) : launch_w<2, EPI_LRELU, false>(p, st);
        if (epi == EPI_RDB5 && !up) return launch_w<2, EPI_RDB5, false>(p, st);
        if (epi == EPI_RDB5_RRDB && !up) return launch_w<2, EPI_RDB5_RRDB, false>(p, st);
        if (epi == EPI_FIRST && !up) return launch_w<2, EPI_FIRST, false>(p, st);
        if (epi == EPI_BODY && !up) return launch_w<2, EPI_BODY, false>(p, st);
        if (epi == EPI_DEBUG) return up ? launch_w<2, EPI_DEBUG, true>(p, st) : launch_w<2, EPI_DEBUG, false>(p, st);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// host-side weight repack.  Layout: [chunk][tap][ks][ct][lane 0..63][j 0..7] fp16 with
//   value = W[cout = ct*32 + (lane&31)][cin = chunk*32 + ks*16 + 8*(lane>>5) + j][tap/3][tap%3] * wscale
// i.e. the 1 KiB an A fragment read (ds_read_b128 at lane*16) wants, so the LDS image is a
// straight copy of global memory.  Missing couts / cins are zero.
// ------------------------------------------------------------------------------------------
size_t conv_wpack_bytes(int cin, int cout) {
    const int nch = (cin + 31) / 32, ct = (cout + 31) / 32;
    return (size_t)nch * 18 * ct * 1024;
}

void pack_conv_weights(const float* w, int cin, int cout, float wscale, void* dst_host) {
    const int nch = (cin + 31) / 32, CT = (cout + 31) / 32;
    f16* d = (f16*)dst_host;
    for (int c = 0; c < nch; ++c)
        for (int t = 0; t < 9; ++t)
            for (int ks = 0; ks < 2; ++ks)
                for (int ct = 0; ct < CT; ++ct)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int co = ct * 32 + (l & 31);
                            const int ci = c * 32 + ks * 16 + 8 * (l >> 5) + j;
                            float v = 0.f;
                            if (co < cout && ci < cin) v = w[((size_t)co * cin + ci) * 9 + t] * wscale;
                            *d++ = (f16)v;
                        }
}

}  // namespace s2sr
