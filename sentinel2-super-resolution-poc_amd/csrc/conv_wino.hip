// RDB conv1..4 (ResidualDenseBlock, reference server/app/cnn_super_resolution.py:78-81,86-89) in ROW-WINOGRAD form:
// F(2,3) along image rows, direct along columns.
//
// Why (profiles/r03_winograd.txt): at the 1400 W socket cap the fp16 trunk's time follows its energy, and the L2-miss path
// that feeds the CUs runs on the clock the MFMA power leaves it.  With one MFMA in three not issued (S2SR_DIAG_SKIPDY2) the
// shipped direct kernel runs conv1-4 at 61.6 instead of 70.2 us: the ceiling of what fewer MFMAs can buy.  The 1-D form gets
// that third -- 12 MFMAs per (2 output rows x 32 px x 16 ch x 32 couts) instead of 18 -- with an input transform of 4 packed
// fp16 adds per MFMA, computed in registers from the same slab rows the direct kernel reads (no LDS re-layout, no extra LDS
// traffic: a 2-D F(2x2,3x3) form needs 8 adds per MFMA and twice the LDS reads).
//
//   y[2t+j] = sum_dy w[dy] * d[2t+j+dy]          (per column shift dx and input channel; d = 4 consecutive slab rows)
//   U = G w :  U0 = w0, U1 = (w0+w1+w2)/2, U2 = (w0-w1+w2)/2, U3 = w2             (packed once per weight load, fp16)
//   V = B^T d: V0 = d0-d2, V1 = d1+d2, V2 = d2-d1, V3 = d1-d3                       (v_pk_add_f16, one rounding each)
//   M_xi += U_xi * V_xi over dx and channels (4 fp32 accumulators per tile row: 16 x 16 = all 256 AGPRs for 8 rows)
//   y[2t] = M0+M1+M2,  y[2t+1] = M1-M2-M3                                           (epilogue, fp32)
//
// Everything else -- blocked-16 HBM layout with the zero halo, XCD-aware persistent workgroups, the LDS-DMA stage ring with
// counted vmcnt waits and one barrier per stage, one wave per SIMD with asm MFMAs on pinned AGPR accumulators, unconditional
// epilogue stores -- is conv_trunk.hip's (read its header first).  The stage loop is written as 3*NT steps (column shift dx,
// tile row t), each = 4 MFMAs; a step also computes the NEXT step's V (VALU results are never consumed by the MFMA right
// behind them: hipcc pads nothing around an asm MFMA), issues the LDS reads of the step after that, and carries its share
// of the stage's DMA instructions.  The barrier sits two steps before the end of a stage: everything read behind it comes
// from the next ring slot.
//
// Numerics: tools/emulate_r03.py (CPU emulation of exactly this arithmetic on the 23-block golden): the trunk's share of
// the error goes 2.0e-5 -> 2.6e-5 (9.4e-5 -> 1.1e-4 on the stress weights); per layer: tests/test_gpu_trunk.py (form 3).
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "s2sr_internal.h"

#ifndef S2SR_WINO_DIAG_NOXFORM
#define S2SR_WINO_DIAG_NOXFORM 0
#endif

namespace s2sr {

namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int NP_, int R_>
struct WG {
    static constexpr int NP = NP_, R = R_, WAVES = 4, NT = NP / 2;       // NT tile rows (2 output rows each) per wave
    static constexpr int TH = WAVES * NP, TW = 32, SW = TW + 2, SH = TH + 2, SPX = SH * SW;
    static constexpr int ROWB = SW * 32;
    static constexpr int PLANE = ((SPX * 32 + 1023) / 1024) * 1024;
    static constexpr int PI = PLANE / 1024, WI = 12, NSTI = PI + WI;      // 12 weight fragments per stage: [dx][xi]
    static constexpr int PW = (NSTI + WAVES - 1) / WAVES;                 // LDS-DMA instructions per wave and stage
    static constexpr int STAGE_BYTES = NSTI * 1024;
    static constexpr int RING_BYTES = R * STAGE_BYTES;
    static constexpr int BIAS_OFF = RING_BYTES;
    static constexpr int LDS_BYTES = BIAS_OFF + 256;
    static constexpr int NSTEP = 3 * NT;                                  // steps per stage
    static constexpr int BAR = NSTEP - 2;                                 // the barrier opens this step
    static constexpr int NW = PW * (R - 2);                               // DMA instructions that may stay in flight at a barrier
    static constexpr int NST = 2 * NP;                                    // epilogue stores per wave
    static_assert(NP % 2 == 0 && NT >= 2, "whole tile rows, and the A prefetch needs two steps per column shift");
    static_assert(NW + NST < 64, "vmcnt field is 6 bits");
};

__device__ __forceinline__ void glds16(const char* base, uint32_t voff, uint32_t lds_addr) {
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
template <typename T>
__device__ __forceinline__ void asm_land(T& r) { asm volatile("" : "+a"(r)); }

// The MFMAs and the transform VALU are ALL `asm volatile`: volatile statements keep their program order, which is the only
// way to pin pure VALU between the MFMAs.  (hipcc places compiler-visible VALU freely -- sched_barrier orders the machine
// scheduler, not instruction selection: as plain C++ all 16 transform instructions of a step ended up in front of its four
// MFMAs and the matrix pipe idled behind them; operand ties through the MFMA statements work too but cost an s_nop each.)
__device__ __forceinline__ void mfma_acc(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_first(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}
// a - b / a + b in packed fp16, one rounding per element.  (A plain vector subtraction -- and fma(b, -1, a), which LLVM folds
// back into it -- compiles to v_sub_f16 + v_sub_f16_sdwa + v_pack_b32_f16 per dword: three instructions instead of one.)
__device__ __forceinline__ f16x8 pk_sub(const f16x8& a, const f16x8& b) {
    const u32x4 x = __builtin_bit_cast(u32x4, a), y = __builtin_bit_cast(u32x4, b);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t r;
        asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x[i]), "v"(y[i]));
        o[i] = r;
    }
    return __builtin_bit_cast(f16x8, o);
}
__device__ __forceinline__ f16x8 pk_add(const f16x8& a, const f16x8& b) {
    const u32x4 x = __builtin_bit_cast(u32x4, a), y = __builtin_bit_cast(u32x4, b);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t r;
        asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(x[i]), "v"(y[i]));
        o[i] = r;
    }
    return __builtin_bit_cast(f16x8, o);
}
template <int N>
__device__ __forceinline__ void wait_release_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

template <int NP, int R>
__global__ void __launch_bounds__(256, 1) conv_wino_f16(const ConvParams p) {
    using G = WG<NP, R>;
    constexpr int NT = G::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pcol = lane & 31, hh = lane >> 5;

    // ---- my patches (XCD-aware round-robin, as in conv_trunk.hip)
    const int nwg = gridDim.x;
    const int slot_in_round = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
    const int tpi = p.tilesX * p.tilesY;
    const int ntiles = tpi * p.N;
    const int my_tiles = (ntiles - slot_in_round + nwg - 1) / nwg;
    if (my_tiles <= 0) return;                                   // workgroup-uniform
    const int NS = p.nstage;
    const uint32_t sblk = (uint32_t)p.sHp * p.sWp * 32;
    const size_t oblk = (size_t)p.Hp * p.Wp * 32;

    if (tid < 32) ((float*)(smem + G::BIAS_OFF))[tid] = p.bias[tid];
    if (p.trace && tid == 0) {                                   // diagnostic (tools/wino_anatomy.py): whole-kernel clock stamps
        p.trace[(size_t)blockIdx.x * 24 + 20] = __builtin_amdgcn_s_memrealtime();
        p.trace[(size_t)blockIdx.x * 24 + 22] = __builtin_amdgcn_s_memtime();
    }

    // ---- per-lane global offsets of this wave's PW DMA pieces (patch independent)
    uint32_t loff[G::PW];
#pragma unroll
    for (int sl = 0; sl < G::PW; ++sl) {
        int j = wave + sl * 4;
        if (j > G::NSTI - 1) j = G::NSTI - 1;                    // padding slot: the last piece again
        if (j < G::PI) {
            const int i = j * 64 + lane;                         // 16-B piece of the slab plane in LDS order
            int q = i >> 1;
            if (q >= G::SPX) q = 0;                              // tail pieces land in the plane's pad
            const int ry = q / G::SW, rx = q - ry * G::SW;
            const int h2 = (i & 1) ^ ((rx >> 3) & 1);            // swizzle on bit 3 of the COLUMN
            loff[sl] = (uint32_t)((ry * p.sWp + rx) * 32 + h2 * 16);
        } else {
            loff[sl] = (uint32_t)((j - G::PI) * 1024 + lane * 16);
        }
    }

    // ---- DMA issue cursor: (tile iteration, stage in patch); stays on the very last stage once it gets there
    int it_i = 0, st_i = 0;
    const char* pbase = nullptr;
    const char* sb_i = nullptr;
    const char* wb_i = nullptr;
    auto cursor_next = [&]() __attribute__((always_inline)) {
        if (st_i == 0) {
            const int tile = it_i * nwg + slot_in_round;
            const int n = tile / tpi;
            const int trem = tile - n * tpi;
            const int ty = trem / p.tilesX, tx = trem - ty * p.tilesX;
            pbase = p.src + (size_t)n * p.src_img + ((size_t)(ty * G::TH) * p.sWp + tx * G::TW) * 32;
        }
        sb_i = pbase + (size_t)st_i * sblk;
        wb_i = (const char*)p.wpack + (size_t)st_i * (G::WI * 1024);
        if (++st_i == NS) {
            if (it_i + 1 < my_tiles) { st_i = 0; ++it_i; }
            else st_i = NS - 1;                                   // clamp: re-load the last stage (into a free slot)
        }
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    auto dma_piece = [&](int sl, uint32_t slot_off) __attribute__((always_inline)) {
        int j = wave + sl * 4;
        if (j > G::NSTI - 1) j = G::NSTI - 1;
        glds16(j < G::PI ? sb_i : wb_i, loff[sl], lds0 + slot_off + (uint32_t)j * 1024);
    };

    // ---- fragment addresses inside a slot: per-lane base + immediate
    uint32_t bbase[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int c = pcol + dx;
        bbase[dx] = (uint32_t)((wave * NP) * G::ROWB + c * 32 + 16 * (hh ^ ((c >> 3) & 1)));
    }
    const uint32_t abase = (uint32_t)(G::PLANE + lane * 16);

    char* const trash = p.trash + (size_t)tid * 16;

    f32x16 acc[NT][4];        // M_xi of tile row t: ALL the accumulation registers at NP = 8
    f16x8 b[NP + 2];          // slab rows of the column shift being worked on (rows 4.. may already be the next shift's)
    f16x8 bq[4];              // rows 0..3 of the NEXT column shift: they arrive while rows 2, 3 of this one are still being transformed
    f16x8 ac[4], an[4];       // U fragments of this / the next column shift
    f16x8 vc[4], vn[4];       // V of this / the next step

    // ---- prologue: R-1 stages in flight, then the registers the first step expects
#pragma unroll
    for (int r = 0; r < R - 1; ++r) {
        cursor_next();
#pragma unroll
        for (int sl = 0; sl < G::PW; ++sl) dma_piece(sl, (uint32_t)(r * G::STAGE_BYTES));
    }
    uint32_t cur_off = 0;
    wait_release_barrier<G::NW>();                                // stage 0 has landed (and the bias is visible)
    {
        const char* sb = smem;
#pragma unroll
        for (int r = 0; r < 6; ++r) b[r] = *(const f16x8*)(sb + bbase[0] + r * G::ROWB);
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) ac[xi] = *(const f16x8*)(sb + abase + xi * 1024);
        vc[0] = pk_sub(b[0], b[2]); vc[1] = pk_add(b[1], b[2]); vc[2] = pk_sub(b[2], b[1]); vc[3] = pk_sub(b[1], b[3]);
    }

    // One stage = 3 * NT steps u = (dx, t).  Step u:
    //   * MFMAs of tile (dx, t) from vc / ac;
    //   * the transform of the NEXT tile into vn -- tile (dx, t+1), or (dx+1, 0) / the next stage's (0, 0) behind t = NT-1;
    //   * LDS reads for the tile after that: its new slab rows (rows 0..3 for a tile row 0, else 2t'+2, 2t'+3), and in the last
    //     two steps of a column shift the next shift's four U fragments;
    //   * DMA pieces of the stage R-1 ahead, spread over the steps in front of the barrier.
    auto stage = [&](auto first_tag, bool first_patch) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const uint32_t next_off = (cur_off + G::STAGE_BYTES == (uint32_t)G::RING_BYTES) ? 0u : cur_off + G::STAGE_BYTES;
        const uint32_t dma_off = (cur_off == 0) ? (uint32_t)(G::RING_BYTES - G::STAGE_BYTES) : cur_off - G::STAGE_BYTES;
        const char* sb = smem + cur_off;
        const char* sn = smem + next_off;
        cursor_next();                                            // the stage R-1 ahead: its DMA rides on this stage
#pragma unroll
        for (int u = 0; u < G::NSTEP; ++u) {
            const int dx = u / NT, t = u % NT;
            if (u == G::BAR) {
                // next stage landed + this slot released; everything read below comes from the NEXT slot
                constexpr int NEPI = G::NW + G::NST;
                if (FIRST && !first_patch) wait_release_barrier<NEPI>();   // the previous patch's epilogue stores may still be in flight
                else wait_release_barrier<G::NW>();
            }
            // ---- LDS reads for the tile two steps ahead
            {
                const int u2 = u + 2;
                const bool nxt = u2 >= G::NSTEP;                  // in the next stage (u >= BAR: behind the barrier)
                const int dx2 = (nxt ? u2 - G::NSTEP : u2) / NT, t2 = (nxt ? u2 - G::NSTEP : u2) % NT;
                const char* s2 = nxt ? sn : sb;
                if (t2 == 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bq[r] = *(const f16x8*)(s2 + bbase[dx2] + r * G::ROWB);
                } else {
                    b[2 * t2 + 2] = *(const f16x8*)(s2 + bbase[dx2] + (2 * t2 + 2) * G::ROWB);
                    b[2 * t2 + 3] = *(const f16x8*)(s2 + bbase[dx2] + (2 * t2 + 3) * G::ROWB);
                }
            }
            if (t >= NT - 2) {                                    // U fragments of the next column shift, two per step
                const bool nxt = dx == 2;
                const char* s2 = nxt ? sn : sb;
                const int dxn = nxt ? 0 : dx + 1, k0 = (t - (NT - 2)) * 2;
                an[k0] = *(const f16x8*)(s2 + abase + (dxn * 4 + k0) * 1024);
                an[k0 + 1] = *(const f16x8*)(s2 + abase + (dxn * 4 + k0 + 1) * 1024);
            }
            // the next tile's four slab rows: rows 2t+2 .. 2t+5 of this column shift, or rows 0..3 of the next one
            const bool wrap = t + 1 == NT;
            const f16x8 d0 = wrap ? bq[0] : b[wrap ? 0 : 2 * t + 2], d1 = wrap ? bq[1] : b[wrap ? 0 : 2 * t + 3],
                        d2 = wrap ? bq[2] : b[wrap ? 0 : 2 * t + 4], d3 = wrap ? bq[3] : b[wrap ? 0 : 2 * t + 5];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                if (FIRST && dx == 0) mfma_first(acc[t][xi], ac[xi], vc[xi]);
                else mfma_acc(acc[t][xi], ac[xi], vc[xi]);
                // behind each MFMA: one V of the next tile (4 instructions), and this slot's share of the DMA
#if S2SR_WINO_DIAG_NOXFORM
                // timing diagnostic only (wrong results): no transform instructions -- what the 4 packed adds per MFMA cost
                if (xi == 0) vn[0] = d0;
                if (xi == 1) vn[1] = d1;
                if (xi == 2) vn[2] = d2;
                if (xi == 3) vn[3] = d3;
#else
                if (xi == 0) vn[0] = pk_sub(d0, d2);
                if (xi == 1) vn[1] = pk_add(d1, d2);
                if (xi == 2) vn[2] = pk_sub(d2, d1);
                if (xi == 3) vn[3] = pk_sub(d1, d3);
#endif
#pragma unroll
                for (int sl = 0; sl < G::PW; ++sl)
                    if ((sl * (G::BAR * 4)) / G::PW == u * 4 + xi) dma_piece(sl, dma_off);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) vc[xi] = vn[xi];
            if (t == NT - 1) {
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) ac[xi] = an[xi];
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = bq[r];
            }
        }
        cur_off = next_off;
    };

    // ---- epilogue of the patch at tile iteration `it`: output transform, bias, LeakyReLU, fp16 stores
    auto epilogue = [&](int it) __attribute__((always_inline)) {
        // the MFMA results must have left the matrix pipe before the VALU reads them (hipcc does not know these asm
        // statements are MFMAs); the wait is tied to the data: every accumulator passes through a "+a" statement behind it
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) asm_land(acc[t][xi]);
        const int tile = it * nwg + slot_in_round;
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int ty = trem / p.tilesX, tx = trem - ty * p.tilesX;
        const int y0 = ty * G::TH, x0 = tx * G::TW;
        const int x = x0 + pcol;
        const PatchLive pl = patch_live(p, y0, x0);
        f32x16 bv;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *(const f32x4*)(smem + G::BIAS_OFF + (8 * g + 4 * hh) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[4 * g + i] = v[i];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int np = 2 * t + j;
                const int y = y0 + wave * NP + np;
                const bool ok = px_live(p, pl, y0, x0, y, x);
                const size_t opix = (size_t)(y + 1) * p.Wp + (x + 1);
                u32x2 hpk[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = 4 * g + i;
                        // y[2t] = M0 + M1 + M2, y[2t+1] = M1 - M2 - M3 (fp32), then bias and LeakyReLU as in the direct form
                        const float s = (j == 0) ? __fadd_rn(__fadd_rn(acc[t][0][e], acc[t][1][e]), acc[t][2][e])
                                                 : __fsub_rn(__fsub_rn(acc[t][1][e], acc[t][2][e]), acc[t][3][e]);
                        const float w = __fadd_rn(s, bv[e]);
                        v[i] = fmaxf(w, __fmul_rn(w, 0.2f));
                    }
                    f32x2 v01, v23;
                    v01[0] = v[0]; v01[1] = v[1]; v23[0] = v[2]; v23[1] = v[3];
                    hpk[g][0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v01, f16x2));      // v_cvt_pk_f16_f32
                    hpk[g][1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v23, f16x2));
                }
                // pair the half-waves: one 16-B store per 16-channel block, 1 KiB contiguous per wave-instruction
#pragma unroll
                for (int bk = 0; bk < 2; ++bk) {
                    u32x2 lo = hpk[2 * bk], hi = hpk[2 * bk + 1];
                    const auto r0 = __builtin_amdgcn_permlane32_swap(lo[0], hi[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(lo[1], hi[1], false, false);
                    u32x4 o;
                    o[0] = r0[0]; o[1] = r1[0]; o[2] = r0[1]; o[3] = r1[1];
                    *(u32x4*)(ok ? p.dst + (size_t)n * p.dst_img + (size_t)bk * oblk + opix * 32 + hh * 16 : trash) = o;
                }
            }
        }
    };

    using std::integral_constant;
    for (int it = 0; it < my_tiles; ++it) {
        stage(integral_constant<bool, true>{}, it == 0);
        for (int st = 1; st < NS; ++st) stage(integral_constant<bool, false>{}, false);
        epilogue(it);
    }
    // nothing may still be on its way into this workgroup's LDS when it ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.trace && tid == 0) {
        p.trace[(size_t)blockIdx.x * 24 + 21] = __builtin_amdgcn_s_memrealtime();
        p.trace[(size_t)blockIdx.x * 24 + 23] = __builtin_amdgcn_s_memtime();
    }
}

template <int NP, int R>
hipError_t launch_wino_t(const ConvParams& p, hipStream_t st) {
    using G = WG<NP, R>;
    static_assert(G::LDS_BYTES <= 160 * 1024, "LDS ring does not fit");
    auto kern = conv_wino_f16<NP, R>;
    static std::mutex attr_mu;
    static bool attr_set[64] = {false};
    static int ncu_dev[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    int ncu;
    {
        std::lock_guard<std::mutex> lk(attr_mu);
        if (!attr_set[dev]) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
            if (e != hipSuccess) return e;
            int n = 256;
            (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
            ncu_dev[dev] = n;
            attr_set[dev] = true;
        }
        ncu = ncu_dev[dev];
    }
    // operand shapes the kernel's indexing assumes (a violation would read or write outside the tensors)
    if (p.nstage < 1 || p.nstage > 12 || p.sHp != p.Hp || p.sWp != p.Wp || p.Hp < p.H + 2 || p.Wp < p.W + 2) return hipErrorInvalidValue;
    if (p.Hp < ((p.H + G::TH - 1) / G::TH) * G::TH + 2 || p.Wp < ((p.W + 31) / 32) * 32 + 2) return hipErrorInvalidValue;   // slabs of edge patches stay inside the plane
    if (!p.src || !p.dst || !p.wpack || !p.bias || !p.trash) return hipErrorInvalidValue;
    ConvParams q = p;
    q.tilesX = (p.W + G::TW - 1) / G::TW;
    q.tilesY = (p.H + G::TH - 1) / G::TH;
    const int ntiles = q.tilesX * q.tilesY * p.N;
    int grid = ncu & ~7;
    if (ntiles < grid) grid = (ntiles + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G::LDS_BYTES, st, q);
    return hipGetLastError();
}

// out[stage][dx][xi][lane][8] = fp16(U_xi[co = lane & 31][ci = stage*16 + 8*(lane>>5) + j][dx]),  U = G w over dy
__global__ void pack_trunk_wino_kernel(const float* __restrict__ w, int cin, int cout, int ns, f16* __restrict__ out) {
    const size_t total = (size_t)ns * 12 * 64 * 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
        size_t r = i >> 9;
        const int xi = (int)(r & 3); r >>= 2;
        const int dx = (int)(r % 3);
        const int st = (int)(r / 3);
        const int co = l & 31, ci = st * 16 + 8 * (l >> 5) + j;
        float v = 0.f;
        if (co < cout && ci < cin) {
            const float* q = w + ((size_t)co * cin + ci) * 9 + dx;   // q[0], q[3], q[6] = the kernel column's dy = 0, 1, 2
            const float w0 = q[0], w1 = q[3], w2 = q[6];
            v = xi == 0 ? w0 : xi == 3 ? w2 : xi == 1 ? __fmul_rn(__fadd_rn(__fadd_rn(w0, w1), w2), 0.5f)
                                                      : __fmul_rn(__fadd_rn(__fsub_rn(w0, w1), w2), 0.5f);
        }
        out[i] = (f16)v;
    }
}

}  // namespace

size_t conv_wpack_bytes_wino(int cin, int cout) {
    (void)cout;
    return (size_t)((cin + 15) / 16) * 12 * 1024;
}

hipError_t launch_pack_trunk_wino(const float* d_w, int cin, int cout, void* d_out, hipStream_t st) {
    if (cout > 32) return hipErrorNotSupported;
    const int ns = (cin + 15) / 16;
    const size_t total = (size_t)ns * 12 * 512;
    hipLaunchKernelGGL(pack_trunk_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_w, cin, cout, ns, (f16*)d_out);
    return hipGetLastError();
}

// conv1..4 only (32 couts, EPI_LRELU).  32x32 patches (4 tile rows per wave) unless that leaves most CUs without a patch:
// then 16x32 patches (2 tile rows per wave, 4-deep ring), as the direct kernel does.
hipError_t launch_conv_trunk_wino(const ConvParams& p, hipStream_t st) {
    const long n32 = (long)((p.W + 31) / 32) * ((p.H + 31) / 32) * p.N;
    if (n32 < 192) return launch_wino_t<4, 4>(p, st);
    return launch_wino_t<8, 3>(p, st);
}

}  // namespace s2sr
