// The 345 convs of the RRDB trunk (ResidualDenseBlock conv1..5, reference
// server/app/cnn_super_resolution.py:85-91; RRDB :103-107) as ONE-WAVE-PER-SIMD workgroups.
//
// Same GEMM orientation, HBM layout, LDS-DMA ring and epilogue arithmetic as conv3x3.hip (read its
// header first); what changes is who issues the MFMAs and how the stream around them is laid out:
//
//   * 4 waves per workgroup, one per SIMD, 512 registers each: the accumulators (8 rows x 32 couts or
//     4 rows x 64 couts per wave = 128 registers) live in AGPRs for the whole patch, the MFMAs are
//     inline asm with "+a" operands.  (With the builtin hipcc moves every finished accumulator
//     AGPR -> VGPR -> AGPR once per stage, ~7 v_accvgpr moves per MFMA: 51-67 cycles per MFMA instead
//     of 32.)  Two MFMA-bound waves on one SIMD do not add up either: measured in the 8-wave form the
//     older wave issues at 57 cycles per MFMA and the younger mostly waits for it (46 per MFMA for
//     the pair, profiles/r02_wave_anatomy.txt) -- an in-order wave cannot use an MFMA-pipe gap
//     shorter than one MFMA.
//   * the K loop of a stage walks dx -> slab row -> dy, so a B fragment (slab row, column shift) is read
//     once and feeds the three kernel rows; B fragments are requested three steps ahead, the A
//     fragments of a kernel column one column ahead (9*CT fragment registers).
//   * the workgroup barrier sits THREE STEPS BEFORE THE END of a stage: behind it the wave first
//     requests the next stage's first fragments (other ring slot), then issues the 6*CT MFMAs it still
//     owes the current stage -- LDS latency and barrier skew hide under them.  The barrier is at the
//     same time the release of the current slot (every read of it has been issued before, and
//     lgkmcnt(0) in the same statement completes them), so the refill DMA of that slot starts with
//     the next stage.
//   * every stage issues exactly PW LDS-DMA instructions per wave, unconditionally (at the end of a
//     workgroup's work the cursor stays on its last stage: two redundant stage loads per launch), so
//     the loop has no branches around DMA and every vmcnt is a compile-time constant.
//   * the slab swizzle is keyed on the COLUMN's bit 3 (not the pixel index's): conflict-free for
//     ds_read_b128 all the same, and it makes every fragment address "per-lane base + immediate".
//   * the first MFMA of an accumulator in a patch takes C = 0, the bias is added in the epilogue
//     (no 128 v_accvgpr_write per patch) -- or, conv1-4 forms, takes the bias as its C operand.
//   * the trunk is carried as a pair: fp16 hi (the MFMA operand of the next RDB) + a lo half that conv5 reads and
//     writes as e4m3(lo * 2^lo_exp) planes of 32 channels -- 4 significant bits of lo keep the net inside 3e-4 of the
//     fp32 reference at half the lo traffic.
//   * what binds these kernels is the memory side and the clock the socket power leaves it (profiles/
//     r02_trunk_anatomy.txt section 8: with the MFMAs compiled out, -DS2SR_DIAG_NOMFMA=1, 70-86 % of the time remains;
//     per clock the shipped kernels are at 0.8-1.0 of that rate).  Bytes and instructions (joules) are what count.
//
// conv_trunk_f8 (second half of the file): the same schedule on e4m3 planes of 32 channels with the K = 64
// block-scaled MFMA; its conv1-4 form runs with a fifth, load-only wave (PROD).
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "s2sr_internal.h"

#ifndef S2SR_F16_BIASC
#define S2SR_F16_BIASC 1   // conv_trunk_f16 conv1-4: bias as the first MFMA's C operand + packed LeakyReLU (0: bias add in the epilogue)
#endif
#ifndef S2SR_F16_ACCV
#define S2SR_F16_ACCV 1    // conv_trunk_f16 conv1-4: accumulators (and the bias C operand) in architectural VGPRs instead of AGPRs (76 + 128 + 16
                           // registers fit the 256): the epilogue reads them in place, 128 v_accvgpr_read per patch and wave less.  Same bytes out;
                           // measured (tools/ab_macro.sh, A/B/A/B on one box) conv1-4 73.4 -> 72.0 us per launch, step 85.9 -> 85.4 ms
#endif
#ifndef S2SR_F16_LOENC
#define S2SR_F16_LOENC 1   // conv_trunk_f16 conv5: the short form of the lo encoding (v_fma_mix_f32 + v_cvt_scalef32_pk_fp8_f32), see the epilogue
#endif
#ifndef S2SR_SMALL_PL
#define S2SR_SMALL_PL 2         // conv_trunk_f16, the single-tile forms (8x32 patches): planes per pipeline stage (1: as until r04's first half)
#endif
#ifndef S2SR_F16_EARLYBIAS
#define S2SR_F16_EARLYBIAS 1    // conv_trunk_f16: 1 = the bias is requested (inline-asm loads) before the ring fill and consumed behind the first
                                // wait; 0 = plain C++ loads in front of the first DMA instruction, as until r03 (two dependent round trips)
#endif
#ifndef S2SR_DIAG_NOLO
#define S2SR_DIAG_NOLO 0        // numerics diagnostic (tools/nolo_probe.sh): the fp16 trunk carried WITHOUT its lo half.  Measured: max-abs
                                // 2.2e-3 .. 3.4e-3 instead of 7e-5 .. 1.8e-4: the pair is what the 1e-3 costs (its lo half needs only e4m3, see the epilogue)
#endif
#ifndef S2SR_DIAG_SKIPDY2
#define S2SR_DIAG_SKIPDY2 0     // timing diagnostic, conv_trunk_f16: 1 = the dy == 2 MFMAs are not issued (results wrong): 2/3 of the MFMA work
#endif
#ifndef S2SR_DIAG_NOMFMA
#define S2SR_DIAG_NOMFMA 0      // timing diagnostic, both kernels: 1 = issue no MFMA (results are wrong).  What is left is the memory
                                // side of the kernel: profiles/r02_trunk_anatomy.txt section 8
#endif

namespace s2sr {

namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// WGL (r04 A/B; measured 3.4 % SLOWER, profiles/r04_trunk_ab.txt: experimental library only): the weights do NOT travel through the LDS-DMA ring.  Every wave fetches the 9 * CT A
// fragments of the NEXT stage with global_load_dwordx4 straight into AGPRs (two register sets, ping-pong; the accumulators of
// the conv1-4 forms live in architectural VGPRs, so the AGPRs are free), one stage ahead.  A stage in LDS is then the slab plane
// alone: 20 % fewer LDS-DMA bytes and instructions, 23 % fewer LDS reads, and room for one more ring slot.
// PL (r04, the single-tile forms): planes per pipeline stage.  A stage of an 8x32 patch is 0.7 us for 0.28 us of MFMA work -- its
// barrier, its LDS round trips and its DMA issue are per-stage costs (profiles/r04_latency_anatomy.txt) -- so the small forms take
// TWO 16-channel planes (and their two weight blocks) per stage: half the barriers per patch, the same MFMAs in the same order.
template <int CT_, int NP_, int R_, int WGL_ = 0, int PL_ = 1>
struct TG {
    static constexpr int CT = CT_, NP = NP_, R = R_, WAVES = 4, WGL = WGL_, PL = PL_;
    static constexpr int TH = WAVES * NP, TW = 32, SW = TW + 2, SH = TH + 2, SPX = SH * SW;
    static constexpr int ROWB = SW * 32;                       // bytes of one slab row
    static constexpr int PLANE = ((SPX * 32 + 1023) / 1024) * 1024;
    static constexpr int PI1 = PLANE / 1024, PI = PL * PI1;    // slab DMA pieces per plane / per stage
    static constexpr int WI1 = 9 * CT, WI = PL * WI1;          // weight pieces (KiB) per plane / per stage
    static constexpr int NSTI = PI + (WGL ? 0 : WI);
    static constexpr int WOFF = PL * PLANE;                    // the stage's weight blocks sit behind its slab planes
    static constexpr int PW = (NSTI + WAVES - 1) / WAVES;      // LDS-DMA instructions per wave and stage
    static constexpr int STAGE_BYTES = NSTI * 1024;
    static constexpr int RING_BYTES = R * STAGE_BYTES;
    static constexpr int BIAS_OFF = RING_BYTES;
    static constexpr int LDS_BYTES = BIAS_OFF + CT * 128;
    static constexpr int T = 3 * (NP + 2);                     // B fragments (steps) per plane
    static constexpr int TS = PL * T;                          // ... per stage
    // steps over which a stage's DMA instructions are spread.  Deeper rings issue them up to the barrier step (the awaited stage was
    // issued stages ago); a DOUBLE buffer of big stages (the 64x32 form) awaits the very stage it is issuing: its pieces go out in the first third
    static constexpr int ISS = (R_ == 2 && NP_ > 8) ? 12 : TS - 3;     // (12 steps for 20 pieces: two per step at most)
    static constexpr int PV = PW + (WGL ? WI : 0);             // vector-memory instructions per wave and stage (WGL: + the A-fragment loads)
    // ... that may stay in flight at a barrier.  WGL: the A fragments of the next stage were requested at the start of this one
    // and must have landed: only this stage's own DMA pieces, issued behind them, may still fly
    static constexpr int NW = WGL ? PW : PV * (R - 2);
    static constexpr int NW0 = WGL ? PW * (R - 2) : NW;        // the prologue's wait: stage 0 (and its A fragments, requested first) landed
    static_assert((NP + 2) % 2 == 0, "the 6-deep B ring needs T % 6 == 0");
    static constexpr int AK = (3 * CT + NP + 1) / (NP + 2);    // A fragments fetched per step: the 3 * CT of the next kernel column must fit the NP + 2 steps of this one
    static_assert(3 * CT <= AK * (NP + 2), "the A fragments of a kernel column must fit its steps");
    static_assert(ISS <= TS - 3 && PW <= 2 * ISS, "DMA slots must fit in front of the barrier step");
    static_assert(PL == 1 || WGL == 0, "");
};

// LDS-DMA, 16 B per lane (conv3x3.hip glds16).  FORCE_UNIFORM: the stamped diagnostic build's divergent stamp
// branches make hipcc keep the uniform base / LDS address in VGPRs; re-derive them through v_readfirstlane.
template <bool FORCE_UNIFORM>
__device__ __forceinline__ void glds16(const char* base, uint32_t voff, uint32_t lds_addr) {
    if (FORCE_UNIFORM) {
        const uint64_t v = (uint64_t)base;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        base = (const char*)(((uint64_t)hi << 32) | lo);
        lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory");
    } else {
        // the slot offset also feeds VALU address arithmetic, and hipcc then keeps it in a VGPR, which an "s"
        // operand does not legalise: pin it to an SGPR (folds away when it already is one)
        lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory");
    }
}
// hidden global loads into AGPRs + landing tie: see conv3x3.hip (asm_load16)
__device__ __forceinline__ f32x4 asm_load16(const char* addr) {
    f32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ u32x2 asm_load8(const char* addr) {
    u32x2 r;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=a"(r) : "v"(addr) : "memory");
    return r;
}
template <typename T>
__device__ __forceinline__ void asm_land(T& r) { asm volatile("" : "+a"(r)); }

__device__ __forceinline__ f32x4 half4_to_float(u32x2 h) {
    const f16x4 v = __builtin_bit_cast(f16x4, h);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (float)v[i];
    return o;
}
__device__ __forceinline__ float lrelu(float v) { return fmaxf(v, __fmul_rn(v, 0.2f)); }

// v_fma_mix_f32 with ONE fp16 operand read in place from a packed register (HI: its upper half): the compiler prefers
// v_cvt_f32_f16 + v_pk_fma_f32 for these, one instruction more per value
template <bool HI>
__device__ __forceinline__ float fma_f32_f32_h(float a, float b, uint32_t c16) {    // a * b + f16(c16.half)
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c16));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c16));
    return r;
}
typedef f16 f16x2 __attribute__((ext_vector_type(2)));

// acc (AGPRs) += A x B, or = A x B for the first MFMA of a patch
__device__ __forceinline__ void mfma_acc(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_first(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
}

// first MFMA of a patch with the bias vector (AGPRs, constant for the whole kernel) as C: the bias costs no instruction
__device__ __forceinline__ void mfma_first_bias(f32x16& acc, const f16x8& a, const f16x8& b, const f32x16& bias) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&a"(acc) : "v"(a), "v"(b), "a"(bias));
}
// the same three with C / D in ARCHITECTURAL VGPRs (S2SR_F16_ACCV, conv1-4 form: 76 + 128 + 16 registers fit the 256):
// the epilogue's VALU then reads the results in place instead of through 128 v_accvgpr_read per patch and wave
__device__ __forceinline__ void mfma_acc_v(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_first_v(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_first_bias_v(f32x16& acc, const f16x8& a, const f16x8& b, const f32x16& bias) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(bias));
}
// ... and with the A operand in AGPRs (WGL: the weights are loaded there straight from global memory; gfx90a+ MFMAs take A / B
// from either file)
__device__ __forceinline__ void mfma_acc_va(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void mfma_first_bias_va(f32x16& acc, const f16x8& a, const f16x8& b, const f32x16& bias) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "a"(a), "v"(b), "v"(bias));
}
template <typename T>
__device__ __forceinline__ void asm_land_v(T& r) { asm volatile("" : "+v"(r)); }
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void wait_release_barrier() {
    // my DMA pieces of the awaited stage have landed (all but the N youngest vector-memory ops are done),
    // my LDS reads of the slot being released have returned; past the barrier both hold for every wave
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

__device__ __forceinline__ f32x4 asm_load16v(const char* addr) {   // as asm_load16, destination in architectural VGPRs
    f32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ uint32_t asm_load4v(const char* addr) {
    uint32_t r;
    asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(addr) : "memory");
    return r;
}

template <int EPI, int CT, int NP>
struct TrunkStores {   // epilogue stores per wave (all unconditional, see conv3x3.hip EpiStores)
    static constexpr int value = (EPI == EPI_LRELU) ? CT * 2 * NP : CT * 3 * NP;     // conv5: two fp16 blocks of x + one e4m3 plane of lo per 32 couts
};

// PROD = 1 (conv1-4 form only: its 208 registers fit two waves on a SIMD): a FIFTH wave issues every LDS-DMA instruction of the
// workgroup and nothing else, as in conv_trunk_f8.  An LDS-DMA instruction holds its wave's issue port for ~60 cycles (12 per
// stage and wave: a quarter of a stage with the matrix pipe starving behind an in-order wave); a wave that only loads can sit
// in that stall for free.  One barrier per stage for everybody: the loader arrives when the NEXT stage has landed.
// FULL: the launch has no ragged edge and no mosaic separators (H % TH == 0, W % 32 == 0, mos_py == 0 -- the 256x256 tile
// batches): every pixel of every patch is live, so the epilogue carries no px_live arithmetic, no trash-line selects and none
// of the SGPR spills they cost (conv1-4, 32x32 form: 2544 -> 1880 instructions, 142 -> 6 v_readlane; 71.1 -> 68.8 us per launch).
// FULL == 3: ragged launches without mosaics (any image that is no multiple of the patch): the extent test alone.
// FULL == 2: mosaics of the reference's default windows (256 + 2 x 10 = 276 pixels, period 277, at the trunk's scale): the separator
// test on compile-time constants -- the 8 scalars of the runtime geometry are what pushes the generic form over its SGPR budget.
// LOE: conv5's lo encoding in its short form (the shipped one) or its long form (experimental library: the byte-identity test
// of the two, tests/test_gpu_trunk.py::test_f16_conv5_lo_encoding_forms_agree).
template <int CT, int NP, int R, int EPI, bool TRACE, int PROD = 0, int FULL = 0, int WGL = 0, int LOE = S2SR_F16_LOENC, int PL = 1>
__global__ void __launch_bounds__(PROD ? 320 : 256, 1) conv_trunk_f16(const ConvParams p) {
    using G = TG<CT, NP, R, WGL, PL>;
    static_assert(PL == 1 || (!TRACE && PROD == 0), "two-plane stages: plain forms only");
    static_assert(WGL == 0 || (EPI == EPI_LRELU && !TRACE && PROD == 0), "weights-from-global form: conv1-4 only");
    constexpr bool kTrunk = (EPI == EPI_RDB5 || EPI == EPI_RDB5_RRDB);
    static_assert(EPI == EPI_LRELU || kTrunk, "trunk kernel: conv1-4 (LRELU) and conv5 (RDB5 / RDB5_RRDB) only");
    static_assert(PROD == 0 || (EPI == EPI_LRELU && !TRACE), "loader wave: conv1-4 form only");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pcol = lane & 31, hh = lane >> 5;

    // ---- my patches (XCD-aware round-robin, as in conv3x3.hip)
    const int nwg = gridDim.x;
    const int slot_in_round = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
    const int tpi = p.tilesX * p.tilesY;
    const int ntiles = tpi * p.N;
    const int my_tiles = (ntiles - slot_in_round + nwg - 1) / nwg;
    if (my_tiles <= 0) return;                                   // workgroup-uniform
    const int NS = p.nstage / PL;                                // stages per patch (of PL planes each)
    // dbg bit 5 (32): launch anatomy of a small launch (tools/launch_anatomy.py), wave 0 lane 0: [0] entry (s_memrealtime), [1] entry,
    // [2] prologue DMA issued, [3] first barrier passed (stage 0 landed), [4] last stage of the last patch done, [5] its epilogue's
    // stores issued, [6] exit after vmcnt(0) (all s_memtime), [7] exit (s_memrealtime)
    const bool kAnat = TRACE && p.trace && (p.dbg & 32) && tid == 0;
    if (kAnat) {
        p.trace[(size_t)blockIdx.x * 24 + 0] = __builtin_amdgcn_s_memrealtime();
        p.trace[(size_t)blockIdx.x * 24 + 1] = __builtin_amdgcn_s_memtime();
    }
    if (TRACE && p.trace && !(p.dbg & 52) && tid == 0) {         // whole-kernel clock stamps (tools/trunk_anatomy.py)
        p.trace[(size_t)blockIdx.x * 24 + 20] = __builtin_amdgcn_s_memrealtime();
        p.trace[(size_t)blockIdx.x * 24 + 22] = __builtin_amdgcn_s_memtime();
    }
    const uint32_t sblk = (uint32_t)p.sHp * p.sWp * 32;
    const size_t oblk = (size_t)p.Hp * p.Wp * 32;

    constexpr bool kBiasC = (EPI == EPI_LRELU) && S2SR_F16_BIASC;
    constexpr bool kAccV = (EPI == EPI_LRELU) && S2SR_F16_ACCV && !TRACE;
    // rows whose accumulators live in architectural VGPRs (the rest in AGPRs): all of them up to 8 rows per wave; the 16-row form
    // (64x32 patches) splits 8 + 8 -- 256 accumulator registers do not fit one file next to the fragments
    constexpr int NPV = kAccV ? (NP > 8 ? 8 : NP) : 0;
    constexpr bool kSplit = NPV > 0 && NPV < NP;

    // ---- per-lane global offsets of this wave's PW DMA pieces (patch independent)
    uint32_t loff[G::PW];
#pragma unroll
    for (int sl = 0; sl < G::PW; ++sl) {
        int j = wave + sl * 4;
        if (j > G::NSTI - 1) j = G::NSTI - 1;                    // padding slot: the last piece again
        if (j < G::PI) {
            const int pl = j / G::PI1;                           // plane of the stage, piece inside the plane
            const int i = (j - pl * G::PI1) * 64 + lane;         // 16-B piece of the slab plane in LDS order
            int q = i >> 1;
            if (q >= G::SPX) q = 0;                              // tail pieces land in the plane's pad
            const int ry = q / G::SW, rx = q - ry * G::SW;
            const int h2 = (i & 1) ^ ((rx >> 3) & 1);            // swizzle on bit 3 of the COLUMN
            loff[sl] = (uint32_t)((ry * p.sWp + rx) * 32 + h2 * 16) + (uint32_t)pl * sblk;
        } else {
            loff[sl] = (uint32_t)((j - G::PI) * 1024 + lane * 16);
        }
    }

    // ---- DMA issue cursor: (tile iteration, stage in patch); stays on the very last stage once it gets there
    int it_i = 0, st_i = 0;
    const char* pbase = nullptr;
    const char* sb_i = nullptr;          // slab source of the stage under the cursor
    const char* wb_i = nullptr;          // its weights
    auto cursor_next = [&]() __attribute__((always_inline)) {
        if (st_i == 0) {
            const int tile = it_i * nwg + slot_in_round;
            const int n = tile / tpi;
            const int trem = tile - n * tpi;
            const int ty = trem / p.tilesX, tx = trem - ty * p.tilesX;
            pbase = p.src + (size_t)n * p.src_img + ((size_t)(ty * G::TH) * p.sWp + tx * G::TW) * 32;
        }
        sb_i = pbase + (size_t)st_i * PL * sblk;
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
        if (TRACE && (p.dbg & 8)) return;                         // ablation (results wrong): no DMA instruction at all
        glds16<TRACE>(j < G::PI ? sb_i : wb_i, loff[sl], lds0 + slot_off + (uint32_t)j * 1024);
    };

    // ---- the loader wave (PROD): the whole workgroup's DMA schedule                          [role-branch]
    // (tests/test_abi_cpu.py moves the block between the [hidden-bias-requests] markers up to this line in a scratch copy -- r04's
    // faulting placement -- and expects tools/check_asm_loads.py --cfg to report it)
    if (PROD && wave == 4) {
        uint32_t loffP[G::PI];                                   // my 16 bytes of slab piece j
#pragma unroll
        for (int j = 0; j < G::PI; ++j) {
            const int i = j * 64 + lane;
            int q = i >> 1;
            if (q >= G::SPX) q = 0;
            const int ry = q / G::SW, rx = q - ry * G::SW;
            const int h2 = (i & 1) ^ ((rx >> 3) & 1);
            loffP[j] = (uint32_t)((ry * p.sWp + rx) * 32 + h2 * 16);
        }
        auto issue_stage = [&](uint32_t slot_off) __attribute__((always_inline)) {   // the stage under the cursor -> ring slot
            cursor_next();
#pragma unroll
            for (int j = 0; j < G::PI; ++j) glds16<false>(sb_i, loffP[j], lds0 + slot_off + (uint32_t)j * 1024);
#pragma unroll
            for (int j = 0; j < G::WI; ++j) glds16<false>(wb_i, (uint32_t)(j * 1024 + lane * 16), lds0 + slot_off + (uint32_t)(G::PI + j) * 1024);
        };
        constexpr int NLAND = PROD ? G::NSTI * (R - 2) : 0;      // pieces that may still be in flight when the awaited stage has landed
        static_assert(NLAND < 64, "vmcnt field is 6 bits");
#pragma unroll
        for (int r = 0; r < R - 1; ++r) issue_stage((uint32_t)(r * G::STAGE_BYTES));
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NLAND) : "memory");          // stage 0 has landed
        const int total = my_tiles * NS;
        uint32_t slot = 0;
        for (int k = 0; k < total; ++k) {
            // stage k + R - 1 goes where stage k - 1 was: the barrier of stage k - 1 (passed) released that slot
            issue_stage(slot == 0 ? (uint32_t)(G::RING_BYTES - G::STAGE_BYTES) : slot - G::STAGE_BYTES);
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NLAND) : "memory");      // stage k + 1 has landed: the barrier inside stage k
            slot = (slot + G::STAGE_BYTES == (uint32_t)G::RING_BYTES) ? 0u : slot + G::STAGE_BYTES;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // The bias is REQUESTED here and consumed behind the first wait of the ring (r04 launch anatomy: as plain C++ the two
    // dependent round trips -- bias to LDS, bias to the C operand -- sat in front of the first DMA instruction: 1.5 of the
    // 2 us prologue of a 7-8 us single-tile launch).  Inline asm: the compiler's own wait would drain the ring.
    // BEHIND the loader wave's branch: a hidden load whose destination is dead on some path lands in registers the compiler has
    // handed to something else there (the loader's DMA offsets: a memory fault, r04).
    // [hidden-bias-requests begin]
    constexpr bool kEarly = S2SR_F16_EARLYBIAS != 0;
    uint32_t bias_l = 0;
    if (!kEarly && tid < CT * 32) ((float*)(smem + G::BIAS_OFF))[tid] = p.bias[tid];
    if (kEarly && !kBiasC) bias_l = asm_load4v((const char*)(p.bias + (tid < CT * 32 ? tid : 0)));
    f32x4 bq[kBiasC ? CT : 1][4];
    if (kEarly && kBiasC) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const char* a = (const char*)(p.bias + ct * 32 + 8 * g + 4 * hh);
                bq[kBiasC ? ct : 0][g] = kAccV ? asm_load16v(a) : asm_load16(a);
            }
    }
    // [hidden-bias-requests end]

    // ---- fragment addresses inside a slot: per-lane base + immediate
    uint32_t bbase[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int c = pcol + dx;
        bbase[dx] = (uint32_t)((wave * NP) * G::ROWB + c * 32 + 16 * (hh ^ ((c >> 3) & 1)));
    }
    const uint32_t abase = (uint32_t)(G::WOFF + lane * 16);
    // conv5: x of the patch's own pixels (channels 8g+4hh.. of block r) is picked out of the slab while stages 0..3 are in LDS
    const uint32_t cbase = (uint32_t)((wave * NP + 1) * G::ROWB + (pcol + 1) * 32 + 8 * hh + 16 * (((pcol + 1) >> 3) & 1));
    u32x2 hi_cap[kTrunk ? 4 : 1][kTrunk ? NP : 1][2];

    char* const trash = p.trash + (size_t)tid * 16;

    f32x16 acc[CT][NP];
    // conv1-4: the bias rides in as the C operand of each accumulator's first MFMA (16 AGPRs per cout tile, loaded once);
    // conv5 keeps adding it in the epilogue (its AGPRs are spoken for by the residual operands)
    f32x16 bacc[kBiasC ? CT : 1];
    f32x16 bacc_a[(kBiasC && kSplit) ? CT : 1];                  // ... and its AGPR copy for the rows that accumulate there (C and D share a file)
    if (!kEarly && kBiasC) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) bacc[kBiasC ? ct : 0][4 * g + i] = p.bias[ct * 32 + 8 * g + 4 * hh + i];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if (kAccV) asm volatile("" : "+v"(bacc[kBiasC ? ct : 0]));
            else asm volatile("" : "+a"(bacc[kBiasC ? ct : 0]));
        }
    }
    f16x8 acol[3][3][CT];     // A fragments: [kernel column dx][kernel row dy][cout tile]
    f16x8 breg[6];            // B fragments of steps t, t+1, t+2, t+3 (ring indexed by step % 6)
    // WGL: the A fragments of two stages in AGPRs, [set][tap * CT + ct]; set = stage parity inside the patch (NS is even)
    f16x8 aw[WGL ? 2 : 1][WGL ? 9 * CT : 1];
    const char* const wlane = (const char*)p.wpack + (size_t)lane * 16;
    auto issue_a = [&](auto set_tag, int st) __attribute__((always_inline)) {     // stage st of a patch -> register set
        constexpr int SET = decltype(set_tag)::value;
        if constexpr (WGL != 0) {
            const char* a = wlane + (size_t)st * (G::WI * 1024);
#pragma unroll
            for (int f = 0; f < 9 * CT; ++f)
                aw[SET][f] = __builtin_bit_cast(f16x8, asm_load16(a + (size_t)f * 1024));     // (through the helper: an asm operand that names a captured array inside a generic lambda does not compile)
        }
    };

    // ---- prologue: R-1 stages in flight, then the first fragments of stage 0
    // (r04, measured and dropped -- tools/ab_latency.sh, profiles/r04_latency_anatomy.txt: issuing exactly as many stage loads
    // as a workgroup runs stages, instead of letting the cursor park on the last stage and re-load it R-1 times, with waits that
    // count the loads really younger: one tile 4.00 -> 4.20 ms, 64x64 2.83 -> 3.10 ms, the 32-tile step 82.2 -> 83.7 ms.  The
    // redundant loads cost nothing measurable; the run-time wait selection and its scalar state do.)
    if (WGL) issue_a(std::integral_constant<int, 0>{}, 0);          // the first stage's weights, in front of everything
#pragma unroll
    for (int r = 0; r < (PROD ? 0 : R - 1); ++r) {
        cursor_next();
#pragma unroll
        for (int sl = 0; sl < G::PW; ++sl) dma_piece(sl, (uint32_t)(r * G::STAGE_BYTES));
    }
    uint32_t cur_off = 0;                                         // LDS offset of the slot of the stage being computed
    if (kAnat) p.trace[(size_t)blockIdx.x * 24 + 2] = __builtin_amdgcn_s_memtime();
    if (PROD) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the loader waited for the data (vmcnt: my bias request)
    else wait_release_barrier<G::NW0>();                          // stage 0 has landed
    if (kAnat) p.trace[(size_t)blockIdx.x * 24 + 3] = __builtin_amdgcn_s_memtime();
    // the bias requests are older than every DMA instruction: they have landed too.  conv5 reads it from LDS in its epilogue
    // (every stage barrier lies in between), conv1-4 feed it to the first MFMA of each accumulator as C.
    if (kEarly && !kBiasC) {
        asm volatile("" : "+v"(bias_l));
        if (tid < CT * 32) ((uint32_t*)(smem + G::BIAS_OFF))[tid] = bias_l;
    }
    if (kEarly && kBiasC) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (kAccV) asm_land_v(bq[kBiasC ? ct : 0][g]);
                else asm_land(bq[kBiasC ? ct : 0][g]);
#pragma unroll
                for (int i = 0; i < 4; ++i) bacc[kBiasC ? ct : 0][4 * g + i] = bq[kBiasC ? ct : 0][g][i];
            }
            if (kAccV) asm volatile("" : "+v"(bacc[kBiasC ? ct : 0]));
            else asm volatile("" : "+a"(bacc[kBiasC ? ct : 0]));
            if (kSplit) {
                bacc_a[kSplit ? ct : 0] = bacc[kBiasC ? ct : 0];
                asm volatile("" : "+a"(bacc_a[kSplit ? ct : 0]));
            }
        }
    }
    {
        const char* sb = smem;
        if (!WGL) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acol[0][dy][ct] = *(const f16x8*)(sb + abase + ((dy * 3) * CT + ct) * 1024);
        }
#pragma unroll
        for (int v = 0; v < 3; ++v) breg[v] = *(const f16x8*)(sb + bbase[0] + v * G::ROWB);
    }

    int kglob = 0;   // TRACE only

    // conv5: the trunk-lo operands of the patch's own pixels are requested at the START of the patch's last stage, straight
    // into AGPRs, and arrive under its MFMAs instead of stalling the epilogue for an HBM round trip
    // The lo half is an e4m3 plane pair (32 channels per plane, 32 bytes per pixel: e4m3(lo * 2^lo_exp)); a lane fetches 16
    // bytes of its pixel (half-wave 0: channels 0-15 of the plane, half-wave 1: 16-31) and two v_permlane32_swap hand each
    // lane the four dwords that match its accumulator registers -- the store path of the fp8 kernel run backwards.
    f32x4 lo_old[kTrunk ? CT : 1][kTrunk ? NP : 1];
    // rdb3: the RRDB's input (fp16 hi blocks + e4m3 lo plane) per output row, double-buffered: row 0 is requested with the
    // lo prefetch, row np + 1 while row np is worked on (a wait on fresh loads would also wait for every DMA issued
    // before them, and four serial round trips per patch were 15 % of this kernel)
    constexpr bool kRR = (EPI == EPI_RDB5_RRDB);
    u32x2 rhi[kRR ? 2 : 1][kRR ? CT : 1][4];
    f32x4 rlo[kRR ? 2 : 1][kRR ? CT : 1];
    auto load_skip = [&](int buf, size_t sn, size_t ln, size_t opix) __attribute__((always_inline)) {   // CT * 5 loads
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                rhi[kRR ? buf : 0][kRR ? ct : 0][g] = asm_load8(p.xh_skip + sn + (size_t)(ct * 2 + (g >> 1)) * oblk + opix * 32 + (g & 1) * 16 + hh * 8);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
            rlo[kRR ? buf : 0][kRR ? ct : 0] = asm_load16(p.lo_skip + ln + (size_t)ct * oblk + opix * 32 + hh * 16);
    };
    auto prefetch_lo = [&](int it) __attribute__((always_inline)) {
        const int tile = it * nwg + slot_in_round;
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int ty = trem / p.tilesX, tx = trem - ty * p.tilesX;
        const size_t ln = (size_t)n * 2 * oblk;
        if (kRR) load_skip(0, (size_t)n * p.xh_img, ln, (size_t)(ty * G::TH + wave * NP + 1) * p.Wp + (tx * G::TW + pcol + 1));
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            const size_t opix = (size_t)(ty * G::TH + wave * NP + np + 1) * p.Wp + (tx * G::TW + pcol + 1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                lo_old[kTrunk ? ct : 0][kTrunk ? np : 0] = asm_load16(p.xh_in + ln + (size_t)ct * oblk + opix * 32 + hh * 16);   // xh_in: the trunk lo coming in
        }
    };

    // One stage.  FIRST: first stage of a patch (accumulators start from C = 0; the barrier inside it may also
    // leave the previous patch's epilogue stores in flight).  CAP >= 0: capture x block CAP for the trunk epilogue.
    auto stage = [&](auto first_tag, auto cap_tag, bool first_patch, auto set_tag, int st_next) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int CAP = decltype(cap_tag)::value;
        constexpr int SET = decltype(set_tag)::value;             // WGL: the register set this stage's A fragments sit in
        static_assert(!WGL || (kAccV && kBiasC), "WGL: accumulators in VGPRs, bias as C");
        if (WGL) issue_a(std::integral_constant<int, SET ^ 1>{}, st_next);     // the next stage's, one stage ahead (other set)
        const uint32_t next_off = (cur_off + G::STAGE_BYTES == (uint32_t)G::RING_BYTES) ? 0u : cur_off + G::STAGE_BYTES;
        const uint32_t dma_off = (cur_off == 0) ? (uint32_t)(G::RING_BYTES - G::STAGE_BYTES) : cur_off - G::STAGE_BYTES;
        const char* sb = smem + cur_off;
        const char* sn = smem + next_off;
        if (!PROD) cursor_next();                                 // the stage R-1 ahead: its DMA rides on this stage
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {                         // the stage's planes, one after the other; the barrier sits in the last
        const bool last = pl == PL - 1;
        const char* sbp = sb + pl * G::PLANE;                     // this plane's slab ...
        const uint32_t ab = abase + (uint32_t)pl * (G::WI1 * 1024);   // ... and weight block
        // what follows this plane: the stage's next plane (already landed with it), or plane 0 of the next stage (behind the barrier)
        const char* nbp = last ? sn : sb + (pl + 1) * G::PLANE;
        const char* nap = last ? sn + abase : sb + ab + G::WI1 * 1024;
#pragma unroll
        for (int t = 0; t < G::T; ++t) {
            const int dx = t / (NP + 2), s = t % (NP + 2);
            if (last && t == G::T - 3) {
                // next stage landed + this slot released; everything below reads the NEXT slot
                constexpr int NST = TrunkStores<EPI, CT, NP>::value;
                // (WGL: the A fragments requested at the top of this stage are YOUNGER than the previous epilogue's stores and must
                // have landed: no allowance for the stores)
                // (R == 2, a double buffer: the awaited stage's DMA pieces are issued in THIS stage, behind the previous epilogue's
                // stores -- an allowance for the stores would leave that many of the awaited pieces in flight: r04, found by
                // test_window_mosaics_give_the_same_bytes on the two-plane conv5 form with several patches per workgroup)
                constexpr int NEPI = (WGL || R == 2) ? G::NW : ((G::NW + NST < 63) ? G::NW + NST : 63);
                if (PROD) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // my only vector-memory traffic are stores
                else if (FIRST && !first_patch) wait_release_barrier<NEPI>();
                else wait_release_barrier<G::NW>();
                if (TRACE && p.trace && (p.dbg & 4) && lane == 0) {
                    if (kglob == 4) p.trace[(size_t)blockIdx.x * 24 + 0 * 8 + wave] = __builtin_amdgcn_s_memtime();
                    if (kglob == 5) p.trace[(size_t)blockIdx.x * 24 + 1 * 8 + wave] = __builtin_amdgcn_s_memtime();
                }
                if (TRACE && p.trace && (p.dbg & 16) && lane == 0) {   // patch-boundary anatomy (tools/trace_boundary.py)
                    if (kglob == 2 * NS - 2) p.trace[(size_t)blockIdx.x * 24 + 20 + wave] = __builtin_amdgcn_s_memtime();   // barrier of patch 1's last-but-one stage
                    if (kglob == 2 * NS - 1) p.trace[(size_t)blockIdx.x * 24 + 4 + wave] = __builtin_amdgcn_s_memtime();    // ... of its last stage
                    if (kglob == 2 * NS) p.trace[(size_t)blockIdx.x * 24 + 16 + wave] = __builtin_amdgcn_s_memtime();       // ... of patch 2's first stage
                    if (kglob == 2 * NS + 1) p.trace[(size_t)blockIdx.x * 24 + 12 + wave] = __builtin_amdgcn_s_memtime();   // ... of its second stage
                }
            }
            // B fragment of step t+3
            {
                const int u = t + 3;
                if (u < G::T) breg[u % 6] = *(const f16x8*)(sbp + bbase[u / (NP + 2)] + (u % (NP + 2)) * G::ROWB);
                else breg[u % 6] = *(const f16x8*)(nbp + bbase[0] + (u - G::T) * G::ROWB);
            }
            // A fragments: the next kernel column's, one per step; behind the barrier the next stage's column 0
            if (dx < 2 && !WGL) {
#pragma unroll
                for (int k = 0; k < G::AK; ++k) {             // AK = 1 in every form but the 8x32-patch conv5 (6 fragments, 4 steps)
                    const int f = s * G::AK + k;
                    if (f < 3 * CT) {
                        const int dy = f / CT, ct = f % CT;
                        acol[dx + 1][dy][ct] = *(const f16x8*)(sb + ab + ((dy * 3 + dx + 1) * CT + ct) * 1024);
                    }
                }
            }
            if (t >= G::T - 3 && !WGL) {
                const int dy = t - (G::T - 3);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acol[0][dy][ct] = *(const f16x8*)(nap + ((dy * 3) * CT + ct) * 1024);
            }
            if constexpr (kTrunk && CAP >= 0) {                   // CAP: the x block of the stage's first plane
                if (t >= 1 && t <= 2 * NP) {
                    const int np = (t - 1) >> 1, half = (t - 1) & 1;
                    hi_cap[CAP + pl][np][half] = *(const u32x2*)(sbp + (cbase ^ (half * 16)) + np * G::ROWB);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // LDS-DMA of the stage R-1 ahead, spread over the steps in front of the barrier
#pragma unroll
            for (int sl = 0; sl < G::PW; ++sl)
                if ((sl * G::ISS) / G::PW == pl * G::T + t && !PROD) dma_piece(sl, dma_off);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int np = s - dy;
                if (np < 0 || np >= NP) continue;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
#if S2SR_DIAG_NOMFMA
                    // timing diagnostic only (wrong results): the fragment reads stay alive, no MFMA is issued
                    if (FIRST && pl == 0 && dx == 0 && dy == 0) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[ct][np][i] = 0.0f;
                    }
                    asm volatile("" : "+a"(acc[ct][np]) : "v"(acol[dx][dy][ct]), "v"(breg[t % 6]));
                    continue;
#endif
#if S2SR_DIAG_SKIPDY2
                    // timing diagnostic only (wrong results): one MFMA in three is not issued (its operand reads stay) -- the MFMA
                    // count of a row-Winograd F(2,3) form without its transform cost: an upper bound for what that form can buy
                    if (dy == 2) {
                        asm volatile("" : "+a"(acc[ct][np]) : "v"(acol[dx][dy][ct]), "v"(breg[t % 6]));
                        continue;
                    }
#endif
                    if (WGL) {
                        const f16x8& af = aw[WGL ? SET : 0][WGL ? (dy * 3 + dx) * CT + ct : 0];
                        if (FIRST && pl == 0 && dx == 0 && dy == 0) mfma_first_bias_va(acc[ct][np], af, breg[t % 6], bacc[kBiasC ? ct : 0]);
                        else mfma_acc_va(acc[ct][np], af, breg[t % 6]);
                        continue;
                    }
                    if (kAccV && np >= NPV) {                     // the 16-row form's upper rows: AGPR accumulators, AGPR bias
                        if (FIRST && pl == 0 && dx == 0 && dy == 0) mfma_first_bias(acc[ct][np], acol[dx][dy][ct], breg[t % 6], bacc_a[kSplit ? ct : 0]);
                        else mfma_acc(acc[ct][np], acol[dx][dy][ct], breg[t % 6]);
                        continue;
                    }
                    if (kAccV) {
                        if (FIRST && pl == 0 && dx == 0 && dy == 0) {
                            if (kBiasC) mfma_first_bias_v(acc[ct][np], acol[dx][dy][ct], breg[t % 6], bacc[kBiasC ? ct : 0]);
                            else mfma_first_v(acc[ct][np], acol[dx][dy][ct], breg[t % 6]);
                        } else mfma_acc_v(acc[ct][np], acol[dx][dy][ct], breg[t % 6]);
                        continue;
                    }
                    if (FIRST && pl == 0 && dx == 0 && dy == 0) {
                        if (kBiasC) mfma_first_bias(acc[ct][np], acol[dx][dy][ct], breg[t % 6], bacc[kBiasC ? ct : 0]);
                        else mfma_first(acc[ct][np], acol[dx][dy][ct], breg[t % 6]);
                    } else mfma_acc(acc[ct][np], acol[dx][dy][ct], breg[t % 6]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (TRACE && p.trace && (p.dbg & 4) && lane == 0 && kglob == 5)
            p.trace[(size_t)blockIdx.x * 24 + 2 * 8 + wave] = __builtin_amdgcn_s_memtime();
        if (TRACE) ++kglob;
        cur_off = next_off;
    };

    // ---- epilogue of the patch at tile iteration `it` (arithmetic as in conv3x3.hip; the bias joins here)
    auto epilogue = [&](int it) __attribute__((always_inline)) {
        // the MFMA results must have left the matrix pipe before the VALU reads them; hipcc does not know these
        // asm statements are MFMAs, so the wait states are spelled out (16-pass MFMA: 18 needed)
        if (TRACE && p.trace && (p.dbg & 16) && lane == 0 && it == 1) p.trace[(size_t)blockIdx.x * 24 + 0 + wave] = __builtin_amdgcn_s_memtime();
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        // ... and tied to the data: every accumulator passes through an empty "+a" statement BEHIND the nops, so no read of
        // it (hipcc hoisted 16-17 v_accvgpr_read above the nops before this; tools/check_asm_loads.py now counts the wait
        // states between an inline MFMA and the first non-MFMA read of its destination) can be scheduled in front of them
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int np = 0; np < NP; ++np) {
                if (kAccV && np < NPV) asm_land_v(acc[ct][np]);
                else asm_land(acc[ct][np]);
            }
        const int tile = it * nwg + slot_in_round;
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int ty = trem / p.tilesX, tx = trem - ty * p.tilesX;
        const int y0 = ty * G::TH, x0 = tx * G::TW;
        const int x = x0 + pcol;
        f32x16 bv[CT];
        if constexpr (!(kBiasC && S2SR_F16_EARLYBIAS)) {          // (bias as the MFMA's C operand: nothing in LDS, the 64x32 form has no room for it)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *(const f32x4*)(smem + G::BIAS_OFF + (ct * 32 + 8 * g + 4 * hh) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[ct][4 * g + i] = kTrunk ? v[i] * 0.2f : v[i];
            }
        }
        const PatchLive pl = FULL ? PatchLive{0, 0, false} : patch_live(p, y0, x0);
        bool ok[NP];
        size_t opix[NP];
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            const int y = y0 + wave * NP + np;
            ok[np] = FULL == 1 ? true
                   : FULL == 2 ? (y < p.H) && (x < p.W) && (y % 277 < 276) && (x % 277 < 276)
                   : FULL == 3 ? (y < p.H) && (x < p.W)
                               : px_live(p, pl, y0, x0, y, x);
            opix[np] = (size_t)(y + 1) * p.Wp + (x + 1);
        }
        const size_t ln = (size_t)n * 2 * oblk;   // image offset inside the e4m3 lo tensors, bytes
        const float lo_dec = __builtin_ldexpf(1.0f, -p.lo_exp), lo_enc = __builtin_ldexpf(1.0f, p.lo_exp);   // e4m3 lo planes hold lo * 2^lo_exp
        const float lo_lim = __builtin_ldexpf(448.0f, -p.lo_exp);                                            // ... of |lo| up to 448 * 2^-lo_exp
        (void)lo_enc; (void)lo_lim;
        // the four dwords of a lane's channel groups out of the 16 bytes it fetched (q[g] = channels 8g+4hh.. of the plane)
        auto unswap = [&](const f32x4& o, uint32_t (&q)[4]) __attribute__((always_inline)) {
            const u32x4 u = __builtin_bit_cast(u32x4, o);
            const auto s0 = __builtin_amdgcn_permlane32_swap(u[0], u[1], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(u[2], u[3], false, false);
            q[0] = s0[0]; q[2] = s0[1]; q[1] = s1[0]; q[3] = s1[1];
        };
        const size_t sn = (size_t)n * p.xh_img;   // image offset inside the skip-hi tensor, bytes
        if (kTrunk) {
            // the prefetched trunk lo: only the last stage's DMA instructions are younger
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::PV) : "memory");
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    asm_land(lo_old[ct][np]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            if (kRR) {
                // row np's skip was requested one row ago (row 0: with the lo prefetch, already waited for above); the request
                // of row np + 1 and the stores of row np - 1 are younger and stay in flight
                constexpr int LR = CT * 5, SR = CT * 3;        // loads / stores per row and wave
                if (np + 1 < NP) load_skip((np + 1) & 1, sn, ln, opix[np + 1]);
                if (np > 0) {
                    if (np + 1 < NP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LR + SR) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SR) : "memory");
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int g = 0; g < 4; ++g) asm_land(rhi[kRR ? np & 1 : 0][kRR ? ct : 0][g]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) asm_land(rlo[kRR ? np & 1 : 0][kRR ? ct : 0]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                u32x2 hpk[4];
                uint32_t lq8[4], loq[4], rlq[4];
                if (kTrunk) unswap(lo_old[kTrunk ? ct : 0][kTrunk ? np : 0], loq);
                if (kRR) unswap(rlo[kRR ? np & 1 : 0][kRR ? ct : 0], rlq);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
                    if (EPI == EPI_LRELU && !kBiasC) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = lrelu(__fadd_rn(acc[ct][np][4 * g + i], bv[ct][4 * g + i]));
                    } else if (EPI == EPI_LRELU) {
                        // LeakyReLU on pairs (v_pk_mul_f32 / v_pk_max_f32): the bias is already in the accumulator
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            f32x2 x;
                            x[0] = acc[ct][np][4 * g + 2 * h2]; x[1] = acc[ct][np][4 * g + 2 * h2 + 1];
                            const f32x2 t = x * 0.2f;
                            // one v_max_f32 per value: the builtin max first canonicalises an operand it cannot see the
                            // origin of (the accumulator comes out of an asm): a second v_max per value, 128 per patch
                            asm("v_max_f32 %0, %1, %2" : "=v"(v[2 * h2]) : "v"(x[0]), "v"(t[0]));
                            asm("v_max_f32 %0, %1, %2" : "=v"(v[2 * h2 + 1]) : "v"(x[1]), "v"(t[1]));
                        }
                    } else {
                        // v = 0.2 * (acc + bias) + t [then 0.2 * v + skip], t = hi + lo: on pairs, fused -- v_pk_fma_f32 for the 0.2
                        // scalings, v_fma_mix_f32 where an operand is still fp16 / the lo decode scale rides along (bv holds
                        // 0.2 * bias here).  HP parity is a tolerance, not the oracle's rounding order: the fused forms round less.
                        const u32x2 th16 = hi_cap[kTrunk ? (ct * 2 + (g >> 1)) & 3 : 0][kTrunk ? np : 0][g & 1];   // 4 fp16, two per word
                        const f32x2 tl01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)loq[g], false), tl23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)loq[g], true);
                        f32x2 rl01, rl23;
                        u32x2 rh16;
                        if (EPI == EPI_RDB5_RRDB) {
                            rl01 = __builtin_amdgcn_cvt_pk_f32_fp8((int)rlq[g], false); rl23 = __builtin_amdgcn_cvt_pk_f32_fp8((int)rlq[g], true);
                            rh16 = rhi[kRR ? np & 1 : 0][kRR ? ct : 0][g];
                        }
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const f32x2 tl = h2 ? tl23 : tl01;
                            f32x2 a, t, b2;
                            a[0] = acc[ct][np][4 * g + 2 * h2]; a[1] = acc[ct][np][4 * g + 2 * h2 + 1];
                            b2[0] = bv[ct][4 * g + 2 * h2]; b2[1] = bv[ct][4 * g + 2 * h2 + 1];
                            t[0] = fma_f32_f32_h<false>(tl[0], lo_dec, th16[h2]);          // hi + lo, exact in fp32
                            t[1] = fma_f32_f32_h<true>(tl[1], lo_dec, th16[h2]);
                            f32x2 w2 = __builtin_elementwise_fma(a, (f32x2){0.2f, 0.2f}, t + b2);
                            if (EPI == EPI_RDB5_RRDB) {
                                const f32x2 rl = h2 ? rl23 : rl01;
                                f32x2 rs;
                                rs[0] = fma_f32_f32_h<false>(rl[0], lo_dec, rh16[h2]);       // the trunk at the RRDB's input
                                rs[1] = fma_f32_f32_h<true>(rl[1], lo_dec, rh16[h2]);
                                w2 = __builtin_elementwise_fma(w2, (f32x2){0.2f, 0.2f}, rs);
                            }
                            v[2 * h2] = w2[0]; v[2 * h2 + 1] = w2[1];
                        }
                    }
                    {
                        f32x2 v01, v23;
                        v01[0] = v[0]; v01[1] = v[1]; v23[0] = v[2]; v23[1] = v[3];
                        hpk[g][0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v01, f16x2));      // v_cvt_pk_f16_f32
                        hpk[g][1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v23, f16x2));
                    }
                    if (kTrunk) {
                        // lo = v - fp16(v), kept as e4m3(lo * 2^lo_exp): 4 significant bits of it are what the 1e-3 needs
                        // (measured: max-abs 8.8e-5 .. 1.85e-4 against 7.0e-5 .. 1.8e-4 with an fp16 lo, 2.2e-3 .. 3.4e-3 without one)
                        // Two forms with the same bytes (test_f16_conv5_lo_encoding_forms_agree, experimental library, stress weights).
                        // (r02 tried another short form -- fma(fp16(v), -2^lo_exp, v * 2^lo_exp) as one asm v_fma_mix_f32 -- that measured
                        // 1.4e-3 instead of 1.3e-4 inside this kernel although it was bit-identical in isolation; never explained, and
                        // not this form: here the fma computes v - fp16(v), exact in fp32, and the scale rides in the conversion.)
                        if constexpr (LOE != 0) {
                        // r03: 2.5 instead of 4.5 instructions per value (the conv5 epilogue was 1/3 lo encoding): v - fp16(v) in ONE
                        // v_fma_mix_f32 that reads the packed half in place (exact: the difference of a float and its own fp16 rounding),
                        // the clamp on the unscaled value, and the 2^lo_exp inside v_cvt_scalef32_pk_fp8_f32 (it divides by the power
                        // of two of its scale operand and, like the plain conversion, turns overflow into NaN -- hence the clamp:
                        // tools/micro/cvt_scale_probe.hip).  Same bytes as the long form.
                        float q[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float d;
                            if (i & 1) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpk[g][i >> 1]), "v"(v[i]));
                            else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpk[g][i >> 1]), "v"(v[i]));
                            q[i] = S2SR_DIAG_NOLO ? 0.0f : __builtin_amdgcn_fmed3f(d, -lo_lim, lo_lim);
                        }
                        typedef short v2s __attribute__((ext_vector_type(2)));
                        v2s w8 = {0, 0};
                        w8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w8, q[0], q[1], lo_dec, false);
                        w8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w8, q[2], q[3], lo_dec, true);
                        lq8[g] = __builtin_bit_cast(uint32_t, w8);
                        } else {
                        float q[4];
                        {
                            const f16x4 hv4 = __builtin_bit_cast(f16x4, hpk[g]);
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                q[i] = S2SR_DIAG_NOLO ? 0.0f : __builtin_amdgcn_fmed3f(__fmul_rn(__fsub_rn(v[i], (float)hv4[i]), lo_enc), -448.0f, 448.0f);
                        }
                        int w8 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
                        w8 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w8, true);
                        lq8[g] = (uint32_t)w8;
                        }
                    }
                }
                // pair the half-waves: one 16-B store per 16-channel block, 1 KiB contiguous per wave-instruction
#pragma unroll
                for (int bk = 0; bk < 2; ++bk) {
                    u32x2 lo = hpk[2 * bk], hi = hpk[2 * bk + 1];
                    const auto r0 = __builtin_amdgcn_permlane32_swap(lo[0], hi[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(lo[1], hi[1], false, false);
                    u32x4 o;
                    o[0] = r0[0]; o[1] = r1[0]; o[2] = r0[1]; o[3] = r1[1];
                    *(u32x4*)(ok[np] ? p.dst + (size_t)n * p.dst_img + (size_t)(ct * 2 + bk) * oblk + opix[np] * 32 + hh * 16 : trash) = o;
                }
                if (kTrunk) {   // the e4m3 lo plane: after the swaps half-wave 0 holds dwords 0-3 of the pixel, half-wave 1 dwords 4-7
                    const auto q0 = __builtin_amdgcn_permlane32_swap(lq8[0], lq8[2], false, false);
                    const auto q1 = __builtin_amdgcn_permlane32_swap(lq8[1], lq8[3], false, false);
                    u32x4 ol;
                    ol[0] = q0[0]; ol[1] = q0[1]; ol[2] = q1[0]; ol[3] = q1[1];
                    *(u32x4*)(ok[np] ? (char*)p.T + ln + (size_t)ct * oblk + opix[np] * 32 + hh * 16 : trash) = ol;
                }
            }
        }
    };

    using std::integral_constant;
    for (int it = 0; it < my_tiles; ++it) {
        const bool first_patch = it == 0;
        if (TRACE && p.trace && (p.dbg & 16) && lane == 0 && it == 2) p.trace[(size_t)blockIdx.x * 24 + 8 + wave] = __builtin_amdgcn_s_memtime();   // epilogue of patch 1 left
        if (kTrunk && PL == 2) {   // two planes per stage: stages 0 and 1 are the 64 channels of x (NS >= 3, host)
            constexpr integral_constant<int, 0> S0{};
            stage(integral_constant<bool, true>{}, integral_constant<int, 0>{}, first_patch, S0, 0);
            stage(integral_constant<bool, false>{}, integral_constant<int, 2>{}, false, S0, 0);
            for (int st = 2; st < NS - 1; ++st) stage(integral_constant<bool, false>{}, integral_constant<int, -1>{}, false, S0, 0);
            prefetch_lo(it);
            stage(integral_constant<bool, false>{}, integral_constant<int, -1>{}, false, S0, 0);
        } else if (kTrunk) {   // NS >= 4 (host): stages 0..3 are the 64 channels of x
            constexpr integral_constant<int, 0> S0{};
            stage(integral_constant<bool, true>{}, integral_constant<int, 0>{}, first_patch, S0, 0);
            if constexpr (PL == 1) {
            stage(integral_constant<bool, false>{}, integral_constant<int, 1>{}, false, S0, 0);
            stage(integral_constant<bool, false>{}, integral_constant<int, 2>{}, false, S0, 0);
            stage(integral_constant<bool, false>{}, integral_constant<int, 3>{}, false, S0, 0);
            }
            for (int st = 4; st < NS - 1; ++st) stage(integral_constant<bool, false>{}, integral_constant<int, -1>{}, false, S0, 0);
            prefetch_lo(it);                                      // NS >= 5 (host): the last stage is peeled, the loads ride on it
            stage(integral_constant<bool, false>{}, integral_constant<int, -1>{}, false, S0, 0);
        } else if (WGL) {
            // the register set is the stage's parity inside the patch (NS is even: every patch starts on set 0)
            constexpr integral_constant<int, 0> S0{};
            constexpr integral_constant<int, 1> S1{};
            constexpr integral_constant<int, -1> NC{};
            stage(integral_constant<bool, true>{}, NC, first_patch, S0, 1);
            stage(integral_constant<bool, false>{}, NC, false, S1, 2 == NS ? 0 : 2);
            for (int st = 2; st < NS; st += 2) {
                stage(integral_constant<bool, false>{}, NC, false, S0, st + 1);
                stage(integral_constant<bool, false>{}, NC, false, S1, st + 2 == NS ? 0 : st + 2);
            }
        } else {
            constexpr integral_constant<int, 0> S0{};
            stage(integral_constant<bool, true>{}, integral_constant<int, -1>{}, first_patch, S0, 0);
            for (int st = 1; st < NS; ++st) stage(integral_constant<bool, false>{}, integral_constant<int, -1>{}, false, S0, 0);
        }
        if (kAnat && it == my_tiles - 1) p.trace[(size_t)blockIdx.x * 24 + 4] = __builtin_amdgcn_s_memtime();
        epilogue(it);
        if (kAnat && it == my_tiles - 1) p.trace[(size_t)blockIdx.x * 24 + 5] = __builtin_amdgcn_s_memtime();
    }
    // nothing may still be on its way into this workgroup's LDS when it ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (kAnat) {
        p.trace[(size_t)blockIdx.x * 24 + 6] = __builtin_amdgcn_s_memtime();
        p.trace[(size_t)blockIdx.x * 24 + 7] = __builtin_amdgcn_s_memrealtime();
    }
    if (TRACE && p.trace && !(p.dbg & 52) && tid == 0) {
        p.trace[(size_t)blockIdx.x * 24 + 21] = __builtin_amdgcn_s_memrealtime();
        p.trace[(size_t)blockIdx.x * 24 + 23] = __builtin_amdgcn_s_memtime();
    }
}

template <int CT, int NP, int R, int EPI, bool TRACE, int PROD = 0, int FULL = 0, int WGL = 0, int LOE = S2SR_F16_LOENC, int PL = 1>
hipError_t launch_trunk_t(const ConvParams& p, hipStream_t st) {
    using G = TG<CT, NP, R, WGL, PL>;
    constexpr bool kNoLdsBias = (EPI == EPI_LRELU) && S2SR_F16_BIASC && S2SR_F16_EARLYBIAS;    // bias as the C operand: no LDS copy
    constexpr int LDSB = kNoLdsBias ? G::RING_BYTES : G::LDS_BYTES;
    static_assert(LDSB <= 160 * 1024, "LDS ring does not fit");
    static_assert(G::NW < 64, "vmcnt field is 6 bits");
    if (FULL == 1 && (p.mos_py != 0 || p.H % G::TH != 0 || p.W % 32 != 0)) return hipErrorInvalidValue;
    if (WGL && (p.nstage & 1)) return hipErrorInvalidValue;       // the A-fragment register sets alternate with the stage's parity
    if (p.nstage % PL != 0) return hipErrorInvalidValue;          // whole stages of PL planes
    if (FULL == 3 && p.mos_py != 0) return hipErrorInvalidValue;
    if (FULL == 2 && (p.mos_py != 277 || p.mos_ry != 276 || p.mos_px != 277 || p.mos_rx != 276)) return hipErrorInvalidValue;
    auto kern = conv_trunk_f16<CT, NP, R, EPI, TRACE, PROD, FULL, WGL, LOE, PL>;
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
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
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
    if ((EPI == EPI_RDB5 || EPI == EPI_RDB5_RRDB) && (p.nstage < 5 * PL || !p.T || !p.xh_in || (EPI == EPI_RDB5_RRDB && (!p.xh_skip || !p.lo_skip))))
        return hipErrorInvalidValue;   // T = trunk lo out, xh_in = trunk lo in (e4m3 planes), (xh_skip, lo_skip) = the RRDB's input as an (fp16, e4m3) pair
    if ((EPI == EPI_RDB5 || EPI == EPI_RDB5_RRDB) && (p.lo_exp < 6 || p.lo_exp > 18)) return hipErrorInvalidValue;
    if (p.Hp < ((p.H + G::TH - 1) / G::TH) * G::TH + 2 || p.Wp < ((p.W + 31) / 32) * 32 + 2) return hipErrorInvalidValue;   // slabs of edge patches stay inside the plane
    if (!p.src || !p.dst || !p.wpack || !p.bias || !p.trash) return hipErrorInvalidValue;
    ConvParams q = p;
    q.tilesX = (p.W + G::TW - 1) / G::TW;
    q.tilesY = (p.H + G::TH - 1) / G::TH;
    const int ntiles = q.tilesX * q.tilesY * p.N;
    int grid = ncu & ~7;
    if (ntiles < grid) grid = (ntiles + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PROD ? 320 : 256), LDSB, st, q);
    return hipGetLastError();
}


// ==========================================================================================
// fp8 trunk (S2SR_PREC_FP8): the same convs on e4m3 operands with the block-scaled MFMA
// v_mfma_scale_f32_32x32x64_f8f6f4 -- K = 64 per instruction = two planes of 32 channels (lanes 0-31
// take their 32 K bytes from plane A, lanes 32-63 from plane B), 64 cycles per MFMA, twice the fp16
// rate on half the bytes.  A 64-cycle MFMA also covers the ~60 cycles an LDS-DMA instruction holds the
// wave's issue port, which the one-wave-per-SIMD form cannot hide behind 32-cycle fp16 MFMAs.
//   * a PAIR-STEP consumes two planes.  LDS: a ring of RS slab slots (one plane each) + a 2-slot ring of
//     weight blocks (the two planes' 9*CT KiB each): slabs arrive RS/2-1 pair-steps ahead, the L2-hot
//     weights one pair-step ahead, weights issued first (vmcnt completes in issue order).
//   * the MFMA's hardware scales do the dequantisation: scale_a (per lane = per cout row) carries the
//     per-output-channel weight scale 2^-k_co, scale_b the activation scale 2^-x_exp / 2^-g_exp.
//   * both 16-B halves of a pixel are needed by the same lane, so the two ds_read_b128 of a B fragment
//     fetch "logical half 0" and "logical half 1" (different physical halves for swizzled columns):
//     conflict-free, and the bytes arrive in channel order without a select.
//   * odd plane counts (Cin 96, 160) are padded with a phantom plane: plane 0 again, against zero weights.
//   * the trunk itself stays fp16 (xh_in / xh_skip / xh_out): the e4m3 rounding of the conv operands
//     (2^-4 relative) dominates everything a second fp16 "lo" half could add.
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma8_acc(f32x16& acc, const v8i& a, const v8i& b, int sa, int sb) {
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
__device__ __forceinline__ void mfma8_first(f32x16& acc, const v8i& a, const v8i& b, int sa, int sb) {
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %4 op_sel_hi:[0,0,0]" : "=a"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}

__device__ __forceinline__ void mfma8_first_bias(f32x16& acc, const v8i& a, const v8i& b, int sa, int sb, const f32x16& bias) {
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %5, %3, %4 op_sel_hi:[0,0,0]" : "=&a"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb), "a"(bias));
}

template <int CT_, int NP_, int RS_, int WV_ = 4, int NPL_ = 0>
struct TG8 {
    static constexpr int CT = CT_, NP = NP_, RS = RS_, WAVES = WV_;
    // NPL > 0: ALL of the conv's weight planes (up to NPL, 9*CT KiB each) are loaded into LDS once per workgroup and stay:
    // no weight DMA in the loop (+1 % on conv1-3, see launch_conv_trunk_f8).
    static constexpr int NPL = NPL_;
    static constexpr bool WRES = NPL_ > 0;
    static constexpr int TH = WAVES * NP, TW = 32, SW = TW + 2, SH = TH + 2, SPX = SH * SW;
    static constexpr int ROWB = SW * 32;
    static constexpr int PLANE = ((SPX * 32 + 1023) / 1024) * 1024;
    static constexpr int PI = PLANE / 1024;                     // slab DMA pieces per plane
    static constexpr int WI = 9 * CT;                           // weight DMA pieces per plane
    static constexpr bool CTMAP = (PI % WAVES == 0);            // piece -> plane mapping is a compile-time property of the slot
    static constexpr int PWS = CTMAP ? 2 * (PI / WAVES) : (2 * PI + WAVES - 1) / WAVES;   // slab DMA instructions per wave and pair-step
    static constexpr int PWW = WRES ? 0 : (2 * WI + WAVES - 1) / WAVES;    // weight DMA instructions per wave and pair-step
    static constexpr int ND = PWS + PWW;
    static constexpr int WBYTES = 2 * WI * 1024;                // one weight slot (two planes)
    static constexpr int WOFF = RS * PLANE;
    static constexpr int BIAS_OFF = WOFF + (WRES ? NPL * WI * 1024 : 2 * WBYTES);
    static constexpr int LDS_BYTES = BIAS_OFF + CT * 128;
    static constexpr int T = 3 * (NP + 2);
    static constexpr int AHEAD = RS / 2 - 1;                    // pair-steps the slab DMA runs ahead
    static constexpr int NW = (AHEAD - 1) * PWS;                // DMA instructions that may stay in flight at a barrier
    static_assert(RS % 2 == 0 && RS >= 4, "slab ring holds whole pairs");
    static_assert(CTMAP || (2 * PI) % WAVES == 0, "the pair's slab pieces split evenly over the waves");
    static constexpr int PWP = CTMAP ? PI / WAVES : PWS;         // loffS entries: per plane (compile-time map) or per slot
    static_assert((NP + 2) % 2 == 0, "the 6-deep B ring needs T % 6 == 0");
    static_assert(3 * CT <= NP + 2, "one A fragment per step must cover a kernel column");
};

// WV = 4: one wave per SIMD (512 registers).  WV = 8 (conv1-4 form): two waves per SIMD with 256 registers each -- the fp8
// form is bound by ONE wave's issue port (PMC: 47 % of wave cycles issuing, 35 % MFMA busy), which a second wave doubles.
// PROD = 1 (conv1-4 form, WV = 4): a FIFTH wave issues every LDS-DMA of the workgroup and nothing else.  With the MFMAs
// compiled out (S2SR_DIAG_NOMFMA) these kernels still take 73 % of their time: they run against the memory system, whose
// back-pressure stalls a DMA instruction at issue -- and, the wave being in-order, every MFMA behind it.  A wave that only
// loads can sit in that stall for free; the four compute waves then only ever wait at the step barrier for data.
// (Two waves share one SIMD: 256 registers per wave, which the conv1-4 form fits; conv5 and the fp16 kernels do not.)
template <int CT, int NP, int RS, int EPI, int WV = 4, int NPL = 0, int PROD = 0>
__global__ void __launch_bounds__((WV + PROD) * 64, PROD ? 1 : WV / 4) conv_trunk_f8(const ConvParams p) {
    using G = TG8<CT, NP, RS, WV, NPL>;
    constexpr bool kTrunk = (EPI == EPI_RDB5 || EPI == EPI_RDB5_RRDB);
    static_assert((EPI == EPI_LRELU && CT == 1) || (kTrunk && CT == 2), "conv1-4: 32 couts; conv5: 64 couts");
    static_assert(PROD == 0 || (PROD == 1 && WV == 4 && !kTrunk), "loader wave: conv1-4 form only");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pcol = lane & 31, hh = lane >> 5;

    const int nwg = gridDim.x;
    const int slot_in_round = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
    const int tpi = p.tilesX * p.tilesY;
    const int ntiles = tpi * p.N;
    const int my_tiles = (ntiles - slot_in_round + nwg - 1) / nwg;
    if (my_tiles <= 0) return;
    const int NSTEP = p.nstage >> 1;                             // pair-steps per patch (nstage is even, host)
    const int nreal = p.seg_len;                                 // real planes; the rest are phantoms
    const uint32_t sblk = (uint32_t)p.sHp * p.sWp * 32;          // bytes between planes
    const size_t oblk = (size_t)p.Hp * p.Wp * 32;

    if (tid < CT * 32) ((float*)(smem + G::BIAS_OFF))[tid] = p.bias[tid];
    int sa[CT];                                                  // E8M0 weight scale of my cout row, per cout tile
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) sa[ct] = p.wscale[ct * 32 + pcol] + (kTrunk ? 0 : p.g_exp);   // conv1-4: the accumulator comes out in the
                                                                                                  // scale of the planes it is stored in (E8M0 exponents add)

    // ---- per-lane global offsets of my DMA pieces.  Slab: instruction sl of a pair-step moves piece wave + 4*(sl % PWP) of
    // plane sl / PWP -- which plane is a compile-time property of sl, so the loop carries no selects.
    uint32_t loffS[G::PWP];
#pragma unroll
    for (int sl = 0; sl < G::PWP; ++sl) {
        int jj = wave + sl * WV;
        if (!G::CTMAP && jj >= G::PI) jj -= G::PI;               // run-time map: slot sl of this wave is piece jj of plane (wave + sl*WV) / PI
        const int i = jj * 64 + lane;
        int q = i >> 1;
        if (q >= G::SPX) q = 0;
        const int ry = q / G::SW, rx = q - ry * G::SW;
        const int h2 = (i & 1) ^ ((rx >> 3) & 1);
        loffS[sl] = (uint32_t)((ry * p.sWp + rx) * 32 + h2 * 16);
    }
    const uint32_t lane16w = (uint32_t)lane * 16 + (uint32_t)wave * 1024;   // weight piece `wave` of my lane

    // ---- two DMA cursors over my pair-steps: slabs (AHEAD steps ahead) and weights (one step ahead); both stay on the
    // last pair-step of the last patch once they get there
    struct Cur { int it, st; const char* pbase; };
    Cur cs{0, 0, nullptr}, cw{0, 0, nullptr};
    const char* sA = nullptr;   // slab sources of the pair under the slab cursor
    const char* sB = nullptr;
    const char* wS = nullptr;   // weights of the pair under the weight cursor
    bool ph_slab = false, ph_wts = false;   // the pair's second plane is a phantom (slab cursor / weight cursor)
    auto advance = [&](Cur& c) __attribute__((always_inline)) {
        if (++c.st == NSTEP) {
            if (c.it + 1 < my_tiles) { c.st = 0; ++c.it; }
            else c.st = NSTEP - 1;
        }
    };
    auto slab_next = [&]() __attribute__((always_inline)) {
        if (cs.st == 0) {
            const int tile = cs.it * nwg + slot_in_round;
            const int n = tile / tpi;
            const int trem = tile - n * tpi;
            const int ty = trem / p.tilesX, tx = trem - ty * p.tilesX;
            cs.pbase = p.src + (size_t)n * p.src_img + ((size_t)(ty * G::TH) * p.sWp + tx * G::TW) * 32;
        }
        const int pa = 2 * cs.st, pb = 2 * cs.st + 1;
        sA = cs.pbase + (size_t)pa * sblk;                       // nreal >= 1 and pairs start on even planes: plane A is always real
        sB = cs.pbase + (size_t)(pb < nreal ? pb : 0) * sblk;
        // a phantom plane (odd plane counts) is filled from ONE 16-byte piece of plane 0 (stored e4m3 bytes are never NaN:
        // the producers clamp before converting), so each of its DMA instructions is a single cache-line request instead
        // of sixteen: finite bytes against zero weights
        ph_slab = pb >= nreal;
        advance(cs);
    };
    auto wts_next = [&]() __attribute__((always_inline)) {
        wS = (const char*)p.wpack + (size_t)cw.st * G::WBYTES;
        ph_wts = 2 * cw.st + 1 >= nreal;                         // second plane's weight block is the all-zero phantom block
        advance(cw);
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
    // DMA targets of the current pair-step, set once per step: M0 values of my first piece in slab slot A / B and in the
    // weight slot (everything else about a piece is an immediate)
    uint32_t mA = 0, mB = 0, mW = 0;
    auto dma_targets = [&](uint32_t slot0, uint32_t wslot) __attribute__((always_inline)) {
        uint32_t s1 = slot0 + 1;
        if (s1 == (uint32_t)RS) s1 = 0;
        mA = lds0 + slot0 * G::PLANE + (uint32_t)wave * 1024;
        mB = lds0 + s1 * G::PLANE + (uint32_t)wave * 1024;
        mW = lds0 + G::WOFF + wslot * G::WBYTES + (uint32_t)wave * 1024;
    };
    auto dma_slab = [&](int sl) __attribute__((always_inline)) {        // sl is a compile-time constant at every call
        if (G::CTMAP) {
            const bool second = sl >= G::PWP;
            const int k = second ? sl - G::PWP : sl;
            glds16<false>(second ? sB : sA, (second && ph_slab) ? 0u : loffS[k], (second ? mB : mA) + (uint32_t)k * (WV * 1024));
        } else {
            const int i = wave + sl * WV;                                // wave-uniform
            const bool second = i >= G::PI;
            const int jj = second ? i - G::PI : i;
            glds16<false>(second ? sB : sA, (second && ph_slab) ? 0u : loffS[sl],
                          (second ? mB : mA) + (uint32_t)(jj - wave) * 1024);
        }
    };
    auto dma_wts = [&](int sl) __attribute__((always_inline)) {
        // piece i = wave + WV*sl of the pair's 2*WI KiB; past the end (last sl only): the last piece again
        const bool over = (sl * WV + WV - 1 > 2 * G::WI - 1) && (wave + sl * WV > 2 * G::WI - 1);
        const uint32_t back = over ? (uint32_t)(wave + sl * WV - (2 * G::WI - 1)) * 1024 : 0u;
        const bool zero = ph_wts && (sl * WV >= G::WI || (sl * WV + WV - 1 >= G::WI && wave + sl * WV >= G::WI));   // piece of the phantom's zero block
        glds16<false>(wS, zero ? (uint32_t)(G::WI * 1024) : lane16w + (uint32_t)sl * (WV * 1024) - back, mW + (uint32_t)sl * (WV * 1024) - back);
    };

    // ---- the loader wave (PROD): the whole workgroup's DMA schedule, one barrier per pair-step like everybody else
    if (PROD && wave == WV) {
        uint32_t loffP[G::PI];                                   // my 16 bytes of piece jj of a plane
#pragma unroll
        for (int jj = 0; jj < G::PI; ++jj) {
            const int i = jj * 64 + lane;
            int q = i >> 1;
            if (q >= G::SPX) q = 0;
            const int ry = q / G::SW, rx = q - ry * G::SW;
            const int h2 = (i & 1) ^ ((rx >> 3) & 1);
            loffP[jj] = (uint32_t)((ry * p.sWp + rx) * 32 + h2 * 16);
        }
        auto issue_slabs = [&](uint32_t slot0) __attribute__((always_inline)) {     // the pair under the slab cursor -> slots slot0, slot0 + 1
            slab_next();
            uint32_t s1 = slot0 + 1;
            if (s1 == (uint32_t)RS) s1 = 0;
            const uint32_t tA = lds0 + slot0 * G::PLANE, tB = lds0 + s1 * G::PLANE;
#pragma unroll
            for (int jj = 0; jj < G::PI; ++jj) glds16<false>(sA, loffP[jj], tA + (uint32_t)jj * 1024);
#pragma unroll
            for (int jj = 0; jj < G::PI; ++jj) glds16<false>(sB, ph_slab ? 0u : loffP[jj], tB + (uint32_t)jj * 1024);
        };
        auto issue_wts = [&](uint32_t wslot) __attribute__((always_inline)) {       // the pair under the weight cursor -> weight slot wslot
            wts_next();
            const uint32_t tW = lds0 + G::WOFF + wslot * G::WBYTES;
#pragma unroll
            for (int i = 0; i < 2 * G::WI; ++i)
                glds16<false>(wS, (ph_wts && i >= G::WI) ? (uint32_t)(G::WI * 1024) : (uint32_t)lane * 16 + (uint32_t)i * 1024, tW + (uint32_t)i * 1024);
        };
        issue_slabs(0);
        if (G::WRES) {
            const int npieces = p.nstage * G::WI;
            for (int i = 0; i < npieces; ++i)
                glds16<false>((const char*)p.wpack, (uint32_t)lane * 16 + (uint32_t)i * 1024, lds0 + G::WOFF + (uint32_t)i * 1024);
        } else {
            issue_wts(0);
            issue_wts(1);
        }
#pragma unroll
        for (int a = 1; a < G::AHEAD; ++a) issue_slabs((uint32_t)(2 * a));
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const int total_steps = my_tiles * NSTEP;
        uint32_t slot = 0, wsl = 0;
        for (int k = 0; k < total_steps; ++k) {
            uint32_t dma_slot = slot + 2 * G::AHEAD;
            while (dma_slot >= (uint32_t)RS) dma_slot -= RS;
            issue_slabs(dma_slot);                               // pair-step k + AHEAD, where pair-step k - 1 was (barrier k - 1 released it)
            // landed before barrier k: everything older than this iteration's slabs, if those run more than one step ahead
            constexpr int NPW = (G::AHEAD >= 2) ? 2 * G::PI : 0;
            static_assert(NPW < 64, "vmcnt field is 6 bits");
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NPW) : "memory");
            if (!G::WRES) issue_wts(wsl);                        // pair-step k + 2, into the weight slot barrier k released
            slot += 2;
            if (slot >= (uint32_t)RS) slot -= RS;
            wsl ^= 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---- fragment addresses: per-lane base + immediate.  b0 / b1 = logical 16-B halves 0 / 1 of pixel (row, pcol + dx)
    uint32_t b0base[3], b1base[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int c = pcol + dx, sw = (c >> 3) & 1;
        b0base[dx] = (uint32_t)((wave * NP) * G::ROWB + c * 32 + 16 * sw);
        b1base[dx] = (uint32_t)((wave * NP) * G::ROWB + c * 32 + 16 * (1 - sw));
    }
    const uint32_t abase = (uint32_t)(G::WOFF + hh * (G::WI * 1024) + pcol * 16);   // my plane's weight block inside a weight slot

    char* const trash = p.trash + (size_t)tid * 16;

    f32x16 acc[CT][NP];
    constexpr bool kBiasC = (EPI == EPI_LRELU);     // see conv_trunk_f16
    f32x16 bacc[kBiasC ? CT : 1];
    if (kBiasC) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) bacc[kBiasC ? ct : 0][4 * g + i] = __builtin_ldexpf(p.bias[ct * 32 + 8 * g + 4 * hh + i], p.g_exp);   // bias in the accumulator's scale
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) asm volatile("" : "+a"(bacc[kBiasC ? ct : 0]));
    }
    v8i acol[3][3][CT];
    v8i breg[6];
    auto ldA = [&](uint32_t woff, int tap, int ct) __attribute__((always_inline)) -> v8i {
        const char* fp = smem + woff + abase + (tap * CT + ct) * 1024;
        const v4i x0 = *(const v4i*)fp, x1 = *(const v4i*)(fp + 512);
        return __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto ldB = [&](uint32_t lane_slab, int dx, int s) __attribute__((always_inline)) -> v8i {
        const v4i x0 = *(const v4i*)(smem + lane_slab + b0base[dx] + s * G::ROWB);
        const v4i x1 = *(const v4i*)(smem + lane_slab + b1base[dx] + s * G::ROWB);
        return __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    // ---- prologue
    uint32_t cur_slot = 0;        // slab ring slot of the current pair's first plane
    uint32_t cur_w = 0;           // weight slot of the current pair
    uint32_t cur_st = 0;          // pair-step inside the patch
    if (!PROD) {
        slab_next();              // slabs of pair-step 0
        dma_targets(0, 0);
#pragma unroll
        for (int sl = 0; sl < G::PWS; ++sl) dma_slab(sl);
        if (G::WRES) {
            // every weight plane of this conv, once: piece i = wave + WV*k of nstage * WI KiB, straight copy of the packed block
            const int npieces = p.nstage * G::WI;
            for (int k = 0; k * WV < npieces; ++k) {
                int i = wave + k * WV;
                if (i > npieces - 1) i = npieces - 1;
                glds16<false>((const char*)p.wpack, (uint32_t)lane * 16 + (uint32_t)i * 1024, lds0 + G::WOFF + (uint32_t)i * 1024);
            }
        } else {
            wts_next();               // weights of pair-steps 0 and 1
#pragma unroll
            for (int sl = 0; sl < G::PWW; ++sl) dma_wts(sl);
            wts_next();
            dma_targets(0, 1);
#pragma unroll
            for (int sl = 0; sl < G::PWW; ++sl) dma_wts(sl);
        }
#pragma unroll
        for (int a = 1; a < G::AHEAD; ++a) {
            slab_next();
            dma_targets((uint32_t)(2 * a), 1);
#pragma unroll
            for (int sl = 0; sl < G::PWS; ++sl) dma_slab(sl);
        }
    }
    if (PROD) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the loader waited for the data
    else wait_release_barrier<G::NW>();
    auto lane_slab_of = [&](uint32_t slot0) __attribute__((always_inline)) -> uint32_t {
        uint32_t s1 = slot0 + 1;
        if (s1 == (uint32_t)RS) s1 = 0;
        return (hh ? s1 : slot0) * (uint32_t)G::PLANE;
    };
    {
        const uint32_t ls = lane_slab_of(0);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acol[0][dy][ct] = ldA(0, dy * 3, ct);
#pragma unroll
        for (int v = 0; v < 3; ++v) breg[v] = ldB(ls, 0, v);
    }

    // conv5: the residual operands (the fp16 trunk coming in; for rdb3 also the RRDB's input) of the patch's own pixels are
    // requested at the START of the patch's last pair-step, straight into AGPRs: they arrive under its MFMAs (that
    // step's barrier drains them with everything else issued before it) instead of stalling the epilogue for an HBM round trip
    u32x2 t_in[kTrunk ? CT : 1][kTrunk ? NP : 1][4];
    auto prefetch_residuals = [&](int it) __attribute__((always_inline)) {
        const int tile = it * nwg + slot_in_round;
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int ty = trem / p.tilesX, tx = trem - ty * p.tilesX;
        const size_t xn = (size_t)n * p.xh_img;
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            const size_t opix = (size_t)(ty * G::TH + wave * NP + np + 1) * p.Wp + (tx * G::TW + pcol + 1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const size_t off = xn + (size_t)(ct * 2 + (g >> 1)) * oblk + opix * 32 + (g & 1) * 16 + hh * 8;
                    t_in[kTrunk ? ct : 0][kTrunk ? np : 0][g] = asm_load8(p.xh_in + off);
                }
        }
    };

    // One pair-step.  FIRST: first of a patch.  `sb` = E8M0 activation scale of this pair (x planes / growth planes).
    auto step = [&](auto first_tag, bool first_patch, int sb) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        uint32_t next_slot = cur_slot + 2;
        if (next_slot >= (uint32_t)RS) next_slot -= RS;
        // DMA targets: the slabs AHEAD pair-steps ahead go where the previous pair-step's planes were
        uint32_t dma_slot = cur_slot + 2 * G::AHEAD;
        while (dma_slot >= (uint32_t)RS) dma_slot -= RS;
        const uint32_t ls = lane_slab_of(cur_slot), lsn = lane_slab_of(next_slot);
        // weight block of this pair-step and of the next one: ring slot, or (resident) the pair's place in the conv's block
        uint32_t nxt_st = cur_st + 1;
        if (nxt_st == (uint32_t)NSTEP) nxt_st = 0;
        const uint32_t wo = G::WRES ? cur_st * (uint32_t)G::WBYTES : cur_w * G::WBYTES;
        const uint32_t won = G::WRES ? nxt_st * (uint32_t)G::WBYTES : (cur_w ^ 1) * G::WBYTES;
        if (!PROD) {
            slab_next();
            dma_targets(dma_slot, cur_w);
        }
#pragma unroll
        for (int t = 0; t < G::T; ++t) {
            const int dx = t / (NP + 2), s = t % (NP + 2);
            if (t == G::T - 3) {
                // needed: the next pair-step's slabs and weights.  The weights were issued in the PREVIOUS step's tail, i.e.
                // before a patch boundary's epilogue stores; with the 6-slot ring so were the slabs, and the stores and this
                // step's slab DMA (one pair-step's worth) may stay in flight.  With the 4-slot ring the needed slabs are this
                // step's own: everything drains (vmcnt completes in issue order).
                constexpr int NEPI = (G::AHEAD >= 2) ? G::NW + CT * NP : 0;      // conv1-4 form: CT*NP plane stores per wave
                if (PROD) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // my only vector-memory traffic are stores
                else if (FIRST && !first_patch && G::AHEAD >= 2) wait_release_barrier<NEPI>();
                else wait_release_barrier<G::NW>();
                if (!G::WRES && !PROD) wts_next();            // two pair-steps ahead: into the weight slot this barrier released
            }
            {
                const int u = t + 3;
                if (u < G::T) breg[u % 6] = ldB(ls, u / (NP + 2), u % (NP + 2));
                else breg[u % 6] = ldB(lsn, 0, u - G::T);
            }
            if (dx < 2 && s < 3 * CT) {
                const int dy = s / CT, ct = s % CT;
                acol[dx + 1][dy][ct] = ldA(wo, dy * 3 + dx + 1, ct);
            }
            if (t >= G::T - 3) {
                const int dy = t - (G::T - 3);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acol[0][dy][ct] = ldA(won, dy * 3, ct);
            }
            __builtin_amdgcn_sched_barrier(0);
            // DMA: the slabs AHEAD pair-steps ahead in front of the barrier step, the weights two pair-steps ahead behind it
#pragma unroll
            for (int i = 0; i < G::PWS; ++i)
                if (!PROD && (i * (G::T - 3)) / G::PWS == t) dma_slab(i);
            if (!PROD && t >= G::T - 3) {
#pragma unroll
                for (int i = 0; i < G::PWW; ++i)
                    if (G::T - 3 + (i * 3) / (G::PWW > 0 ? G::PWW : 1) == t) dma_wts(i);
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int np = s - dy;
                if (np < 0 || np >= NP) continue;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
#if S2SR_DIAG_NOMFMA
                    // timing diagnostic only (wrong results): the fragment reads stay alive, no MFMA is issued
                    if (FIRST && dx == 0 && dy == 0) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[ct][np][i] = 0.0f;
                    }
                    asm volatile("" : "+a"(acc[ct][np]) : "v"(acol[dx][dy][ct]), "v"(breg[t % 6]));
                    continue;
#endif
                    if (FIRST && dx == 0 && dy == 0) {
                        if (kBiasC) mfma8_first_bias(acc[ct][np], acol[dx][dy][ct], breg[t % 6], sa[ct], sb, bacc[kBiasC ? ct : 0]);
                        else mfma8_first(acc[ct][np], acol[dx][dy][ct], breg[t % 6], sa[ct], sb);
                    } else mfma8_acc(acc[ct][np], acol[dx][dy][ct], breg[t % 6], sa[ct], sb);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        cur_slot = next_slot;
        cur_w ^= 1;
        cur_st = nxt_st;
    };

    // e4m3 x4 of four floats scaled by 2^e, clamped to the finite range (v_cvt_pk_fp8_f32 turns overflow into NaN)
    auto to_e4m3x4 = [&](const f32x4& v, float scale) __attribute__((always_inline)) -> uint32_t {
        float c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_fmed3f(__fmul_rn(v[i], scale), -448.0f, 448.0f);
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], w, true);
        return (uint32_t)w;
    };
    const float oscale = __builtin_ldexpf(1.0f, kTrunk ? p.x_exp : p.g_exp);   // scale of the planes this conv writes

    auto epilogue = [&](int it) __attribute__((always_inline)) {
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");       // MFMA results -> VALU reads (see conv_trunk_f16)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int np = 0; np < NP; ++np) asm_land(acc[ct][np]);   // tie the wait to the data
        const int tile = it * nwg + slot_in_round;
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int ty = trem / p.tilesX, tx = trem - ty * p.tilesX;
        const int y0 = ty * G::TH, x0 = tx * G::TW;
        const int x = x0 + pcol;
        f32x16 bv[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *(const f32x4*)(smem + G::BIAS_OFF + (ct * 32 + 8 * g + 4 * hh) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[ct][4 * g + i] = kTrunk ? v[i] * 0.2f : v[i];
            }
        const PatchLive pl = patch_live(p, y0, x0);
        bool ok[NP];
        size_t opix[NP];
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            const int y = y0 + wave * NP + np;
            ok[np] = px_live(p, pl, y0, x0, y, x);
            opix[np] = (size_t)(y + 1) * p.Wp + (x + 1);
        }
        const size_t xn = (size_t)n * p.xh_img;
        if (kTrunk) {
            // the prefetched residuals: everything issued after them in the last pair-step may still be in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::ND < 63 ? G::ND : 63) : "memory");
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int g = 0; g < 4; ++g) asm_land(t_in[kTrunk ? ct : 0][kTrunk ? np : 0][g]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // rdb3: the RRDB's input per output row (all rows at once would need 64 more AGPRs than there are); row np + 1 is
        // requested before row np is worked on
        u32x2 r_in[EPI == EPI_RDB5_RRDB ? 2 : 1][EPI == EPI_RDB5_RRDB ? CT : 1][4];
        auto load_skip = [&](int np) __attribute__((always_inline)) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    r_in[EPI == EPI_RDB5_RRDB ? np & 1 : 0][EPI == EPI_RDB5_RRDB ? ct : 0][g] =
                        asm_load8(p.xh_skip + xn + (size_t)(ct * 2 + (g >> 1)) * oblk + opix[np] * 32 + (g & 1) * 16 + hh * 8);
        };
        if (EPI == EPI_RDB5_RRDB) load_skip(0);
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            if (EPI == EPI_RDB5_RRDB) {
                // row np's skip was requested one row ago: wait for it (the stores of row np - 1 and the request of row
                // np + 1 are younger and stay in flight), then tie it down
                if (np + 1 < NP) load_skip(np + 1);
                if (np + 1 < NP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CT * 4 + (np > 0 ? CT * 3 : 0)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(np > 0 ? CT * 3 : 0) : "memory");
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int g = 0; g < 4; ++g) asm_land(r_in[EPI == EPI_RDB5_RRDB ? np & 1 : 0][EPI == EPI_RDB5_RRDB ? ct : 0][g]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                u32x2 hpk[4];
                uint32_t q8[4];     // e4m3 x4 per g: channels 8g+4hh.. of plane ct -> dword 2g+hh of the pixel's 32 bytes
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
                    if (EPI == EPI_LRELU) {
                        // bias and the plane scale 2^g_exp are already in the accumulator (C operand, scale_a): LeakyReLU and both
                        // clamps in 2.5 instructions per value: t = 0.2 a (v_pk_mul_f32), m = max(a, t) (asm: the builtin max
                        // would first canonicalise the accumulator, a second v_max per value), y = med3(m, -448, 448)
                        float y[4];
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            f32x2 a;
                            a[0] = acc[ct][np][4 * g + 2 * h2]; a[1] = acc[ct][np][4 * g + 2 * h2 + 1];
                            const f32x2 t = a * 0.2f;
                            float m0, m1;
                            asm("v_max_f32 %0, %1, %2" : "=v"(m0) : "v"(a[0]), "v"(t[0]));
                            asm("v_max_f32 %0, %1, %2" : "=v"(m1) : "v"(a[1]), "v"(t[1]));
                            y[2 * h2] = __builtin_amdgcn_fmed3f(m0, -448.0f, 448.0f);
                            y[2 * h2 + 1] = __builtin_amdgcn_fmed3f(m1, -448.0f, 448.0f);
                        }
                        int w = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], 0, false);
                        w = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], w, true);
                        q8[g] = (uint32_t)w;
                        continue;
                    } else {
                        // v = (acc + bias) * 0.2 + t [* 0.2 + skip] on pairs: acc * 0.2 + (0.2 * bias + t) is one v_pk_add_f32 and
                        // one v_pk_fma_f32 per pair (bv holds 0.2 * bias here); fp8 mode is not held to the oracle's rounding order
                        const f32x4 th = half4_to_float(t_in[kTrunk ? ct : 0][kTrunk ? np : 0][g]);
                        f32x4 rr;
                        if (EPI == EPI_RDB5_RRDB) rr = half4_to_float(r_in[EPI == EPI_RDB5_RRDB ? np & 1 : 0][EPI == EPI_RDB5_RRDB ? ct : 0][g]);
                        float y[4];
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            f32x2 a, c, r;
                            a[0] = acc[ct][np][4 * g + 2 * h2]; a[1] = acc[ct][np][4 * g + 2 * h2 + 1];
                            c[0] = th[2 * h2] + bv[ct][4 * g + 2 * h2]; c[1] = th[2 * h2 + 1] + bv[ct][4 * g + 2 * h2 + 1];
                            f32x2 w2 = __builtin_elementwise_fma(a, (f32x2){0.2f, 0.2f}, c);
                            if (EPI == EPI_RDB5_RRDB) {
                                r[0] = rr[2 * h2]; r[1] = rr[2 * h2 + 1];
                                w2 = __builtin_elementwise_fma(w2, (f32x2){0.2f, 0.2f}, r);
                            }
                            v[2 * h2] = w2[0]; v[2 * h2 + 1] = w2[1];
                            const f32x2 s2 = w2 * oscale;
                            y[2 * h2] = __builtin_amdgcn_fmed3f(s2[0], -448.0f, 448.0f);
                            y[2 * h2 + 1] = __builtin_amdgcn_fmed3f(s2[1], -448.0f, 448.0f);
                        }
                        f16x4 hv;
#pragma unroll
                        for (int i = 0; i < 4; ++i) hv[i] = (f16)v[i];
                        hpk[g] = __builtin_bit_cast(u32x2, hv);
                        int w = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], 0, false);
                        w = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], w, true);
                        q8[g] = (uint32_t)w;
                        continue;
                    }
                    q8[g] = to_e4m3x4(v, oscale);
                }
                // e4m3 plane: after the swaps lanes 0-31 hold dwords 0-3 of the pixel, lanes 32-63 dwords 4-7
                {
                    const auto r0 = __builtin_amdgcn_permlane32_swap(q8[0], q8[2], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(q8[1], q8[3], false, false);
                    u32x4 o;
                    o[0] = r0[0]; o[1] = r0[1]; o[2] = r1[0]; o[3] = r1[1];
                    *(u32x4*)(ok[np] ? p.dst + (size_t)n * p.dst_img + (size_t)ct * oblk + opix[np] * 32 + hh * 16 : trash) = o;
                }
                if (kTrunk) {
#pragma unroll
                    for (int bk = 0; bk < 2; ++bk) {
                        u32x2 lo = hpk[2 * bk], hi = hpk[2 * bk + 1];
                        const auto r0 = __builtin_amdgcn_permlane32_swap(lo[0], hi[0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(lo[1], hi[1], false, false);
                        u32x4 o;
                        o[0] = r0[0]; o[1] = r1[0]; o[2] = r0[1]; o[3] = r1[1];
                        *(u32x4*)(ok[np] ? p.xh_out + xn + (size_t)(ct * 2 + bk) * oblk + opix[np] * 32 + hh * 16 : trash) = o;
                    }
                }
            }
        }
    };

    using std::integral_constant;
    const int sbx = 127 - p.x_exp, sbg = 127 - p.g_exp;
    for (int it = 0; it < my_tiles; ++it) {
        step(integral_constant<bool, true>{}, it == 0, sbx);      // planes 0, 1 = x
        if (kTrunk) {                                             // conv5: NSTEP >= 2 (host); the last pair-step is peeled
            for (int st = 1; st < NSTEP - 1; ++st) step(integral_constant<bool, false>{}, false, sbg);
            prefetch_residuals(it);
            step(integral_constant<bool, false>{}, false, sbg);
        } else {
            for (int st = 1; st < NSTEP; ++st) step(integral_constant<bool, false>{}, false, sbg);
        }
        epilogue(it);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int CT, int NP, int RS, int EPI, int WV = 4, int NPL = 0, int PROD = 0>
hipError_t launch_trunk8_t(const ConvParams& p, hipStream_t st) {
    using G = TG8<CT, NP, RS, WV, NPL>;
    static_assert(G::LDS_BYTES <= 160 * 1024, "LDS rings do not fit");
    static_assert(G::NW >= 0 && G::NW < 64, "vmcnt field is 6 bits");
    auto kern = conv_trunk_f8<CT, NP, RS, EPI, WV, NPL, PROD>;
    if (NPL > 0 && p.nstage > NPL) return hipErrorInvalidValue;   // resident weights: the conv's planes must fit the LDS block
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
    // operand shapes the kernel's indexing assumes
    if (p.nstage < 2 || p.nstage > 6 || (p.nstage & 1) || p.seg_len < 1 || p.seg_len > p.nstage) return hipErrorInvalidValue;
    if ((EPI == EPI_RDB5 || EPI == EPI_RDB5_RRDB) && p.nstage < 4) return hipErrorInvalidValue;   // the residual prefetch rides on a later pair-step than the first
    if (p.sHp != p.Hp || p.sWp != p.Wp) return hipErrorInvalidValue;
    if (p.Hp < ((p.H + G::TH - 1) / G::TH) * G::TH + 2 || p.Wp < ((p.W + 31) / 32) * 32 + 2) return hipErrorInvalidValue;
    if (!p.src || !p.dst || !p.wpack || !p.bias || !p.trash || !p.wscale) return hipErrorInvalidValue;
    if (p.x_exp < -8 || p.x_exp > 16 || p.g_exp < -8 || p.g_exp > 16) return hipErrorInvalidValue;
    if ((EPI == EPI_RDB5 || EPI == EPI_RDB5_RRDB) && (!p.xh_in || !p.xh_out || (EPI == EPI_RDB5_RRDB && !p.xh_skip))) return hipErrorInvalidValue;
    ConvParams q = p;
    q.tilesX = (p.W + G::TW - 1) / G::TW;
    q.tilesY = (p.H + G::TH - 1) / G::TH;
    const int ntiles = q.tilesX * q.tilesY * p.N;
    int grid = ncu & ~7;
    if (ntiles < grid) grid = (ntiles + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(grid), dim3((WV + PROD) * 64), G::LDS_BYTES, st, q);
    return hipGetLastError();
}

}  // namespace

// ct = 1: conv1..4 (EPI_LRELU); ct = 2: conv5 (EPI_RDB5 / EPI_RDB5_RRDB).  Returns hipErrorNotSupported for
// anything else.
hipError_t launch_conv_trunk(const ConvParams& p, int ct, int epi, hipStream_t st, bool trace, int force_form) {
#if !S2SR_EXPERIMENTAL
    if (trace || force_form == 4 || (p.f16_form & 1)) return hipErrorNotSupported;     // stamped builds, loader-wave form: experimental library only
#endif
    if (ct == 1 && epi == EPI_LRELU) {
        if (force_form == 1) return launch_trunk_t<1, 4, 5, EPI_LRELU, false>(p, st);      // per-layer parity hook: name the patch form
        if (force_form == 2) return launch_trunk_t<1, 8, 3, EPI_LRELU, false>(p, st);
#if S2SR_EXPERIMENTAL
        if (force_form == 4) return launch_trunk_t<1, 8, 3, EPI_LRELU, false, 1>(p, st);
#endif
        if (force_form == 5) return launch_trunk_t<1, 2, 7, EPI_LRELU, false>(p, st);
        if (force_form == 6) return launch_trunk_t<1, 8, 3, EPI_LRELU, false, 0, 1>(p, st);    // whole-patch forms (invalid-value on ragged sizes / mosaics)
        if (force_form == 7) return launch_trunk_t<1, 4, 5, EPI_LRELU, false, 0, 1>(p, st);
        if (force_form == 8) return launch_trunk_t<1, 2, 7, EPI_LRELU, false, 0, 1>(p, st);
#if S2SR_EXPERIMENTAL
        if (force_form == 11) return launch_trunk_t<1, 16, 2, EPI_LRELU, false, 0, 1>(p, st);   // whole 64x32 patches
#else
        if (force_form == 11) return hipErrorNotSupported;
#endif
#if S2SR_EXPERIMENTAL
        if (force_form == 9) return launch_trunk_t<1, 8, 4, EPI_LRELU, false, 0, 1, 1>(p, st);   // whole 32x32 patches, weights from global memory (WGL)
#else
        if (force_form == 9) return hipErrorNotSupported;
#endif
        // 32x32 patches (8 rows per wave, 3-deep ring) unless that leaves most CUs without a patch (single tiles):
        // then 16x32 patches (4 rows per wave, 5-deep ring) spread the image over twice as many workgroups.  Both
        // forms accumulate in the same order, so the result does not depend on the choice.
        const long n32 = (long)((p.W + 31) / 32) * ((p.H + 31) / 32) * p.N;
        // ... and 8x32 patches (2 rows per wave, 7-deep ring) for a single tile: 256 patches for 256 CUs instead of 128 (one 256x256 tile
        // is launch-bound: 351 dependent launches; S2SR_SMALL8=0 keeps the 16x32 form)
        const bool full = !trace && p.mos_py == 0 && p.H % 32 == 0 && p.W % 32 == 0 && !(p.f16_form & 4);   // whole patches only (f16_form bit 2: diagnostic off switch)
        const bool plain = !trace && p.mos_py == 0 && !(p.f16_form & 4);                                  // ragged, but no mosaic: the extent test alone
#if S2SR_EXPERIMENTAL
        if (n32 < 96 && trace) return launch_trunk_t<1, 2, 7, EPI_LRELU, true>(p, st);               // launch anatomy of the single-tile form
#endif
        if (force_form == 10) return launch_trunk_t<1, 2, 3, EPI_LRELU, false, 0, 3, 0, S2SR_F16_LOENC, 2>(p, st);   // 8x32 patches, two planes per stage
        if (n32 < 96 && !trace && !(p.f16_form & 2)) {
#if S2SR_EXPERIMENTAL
            if (full && (p.f16_form & 8)) return launch_trunk_t<1, 2, 7, EPI_LRELU, false, 0, 1, 1>(p, st);   // r04 probe: single-tile form with the weights from global memory
#endif
            if (S2SR_SMALL_PL == 2 && full) return launch_trunk_t<1, 2, 3, EPI_LRELU, false, 0, 1, 0, S2SR_F16_LOENC, 2>(p, st);
            if (S2SR_SMALL_PL == 2 && plain) return launch_trunk_t<1, 2, 3, EPI_LRELU, false, 0, 3, 0, S2SR_F16_LOENC, 2>(p, st);
            return full ? launch_trunk_t<1, 2, 7, EPI_LRELU, false, 0, 1>(p, st)
                        : plain ? launch_trunk_t<1, 2, 7, EPI_LRELU, false, 0, 3>(p, st) : launch_trunk_t<1, 2, 7, EPI_LRELU, false>(p, st);
        }
        if (n32 < 192 && !trace)
            return full ? launch_trunk_t<1, 4, 5, EPI_LRELU, false, 0, 1>(p, st)
                        : plain ? launch_trunk_t<1, 4, 5, EPI_LRELU, false, 0, 3>(p, st) : launch_trunk_t<1, 4, 5, EPI_LRELU, false>(p, st);
#if S2SR_EXPERIMENTAL
        if (!trace && (p.f16_form & 1)) return launch_trunk_t<1, 8, 3, EPI_LRELU, false, 1>(p, st);     // S2SR_F16_LOADER=1: loader-wave form
#endif
#if S2SR_EXPERIMENTAL
        if (full && (p.f16_form & 8)) return launch_trunk_t<1, 8, 4, EPI_LRELU, false, 0, 1, 1>(p, st);   // r04 A/B: weights from global memory, 4-deep slab ring
#endif
#if S2SR_EXPERIMENTAL
        if (full && (p.f16_form & 16) && p.H % 64 == 0) return launch_trunk_t<1, 16, 2, EPI_LRELU, false, 0, 1>(p, st);   // 64x32 patches, double-buffered ring (r04 A/B: 5 % slower)
#endif
        if (full) return launch_trunk_t<1, 8, 3, EPI_LRELU, false, 0, 1>(p, st);
        if (plain && !(p.f16_form & 1)) return launch_trunk_t<1, 8, 3, EPI_LRELU, false, 0, 3>(p, st);
        if (!trace && !(p.f16_form & 4) && p.mos_py == 277 && p.mos_ry == 276 && p.mos_px == 277 && p.mos_rx == 276)
            return launch_trunk_t<1, 8, 3, EPI_LRELU, false, 0, 2>(p, st);                             // mosaics of 276-pixel windows (tile 256, pad 10)
#if S2SR_EXPERIMENTAL
        if (trace) return launch_trunk_t<1, 8, 3, EPI_LRELU, true>(p, st);
#endif
        return launch_trunk_t<1, 8, 3, EPI_LRELU, false>(p, st);
    }
    // conv5: 16x32 patches (4 rows per wave, 4-deep ring); single tiles take 8x32 patches (2 rows per wave, 5-deep ring): one
    // 256x256 tile is 128 patches of 16x32 -- half the CUs idle through twelve MFMA-bound stages (r04 kernel trace: 24 us per
    // conv5 launch, 69 of them = 37 % of one tile's latency) -- and 256 of 8x32.  Same accumulation order, same bytes.
    // force_form 1 / 5: name the patch form (per-layer parity hook).
    if (ct == 2 && (epi == EPI_RDB5 || epi == EPI_RDB5_RRDB)) {
        const long n16 = (long)((p.W + 31) / 32) * ((p.H + 15) / 16) * p.N;
        const bool small = force_form == 5 || (force_form == 0 && n16 < 192 && !(p.f16_form & 2));
#if S2SR_EXPERIMENTAL
        if (force_form == 2)      // 16x32 patches with the LONG form of the lo encoding
            return epi == EPI_RDB5 ? launch_trunk_t<2, 4, 4, EPI_RDB5, false, 0, 0, 0, 0>(p, st) : launch_trunk_t<2, 4, 4, EPI_RDB5_RRDB, false, 0, 0, 0, 0>(p, st);
#else
        if (force_form == 2) return hipErrorNotSupported;
#endif
        if (force_form == 10 || (small && force_form == 0 && !trace && S2SR_SMALL_PL == 2))      // 8x32 patches, two planes per stage, double-buffered
            return epi == EPI_RDB5 ? launch_trunk_t<2, 2, 2, EPI_RDB5, false, 0, 0, 0, S2SR_F16_LOENC, 2>(p, st)
                                   : launch_trunk_t<2, 2, 2, EPI_RDB5_RRDB, false, 0, 0, 0, S2SR_F16_LOENC, 2>(p, st);
        if (epi == EPI_RDB5) {
#if S2SR_EXPERIMENTAL
            if (trace) return small ? launch_trunk_t<2, 2, 5, EPI_RDB5, true>(p, st) : launch_trunk_t<2, 4, 4, EPI_RDB5, true>(p, st);
#endif
            return small ? launch_trunk_t<2, 2, 5, EPI_RDB5, false>(p, st) : launch_trunk_t<2, 4, 4, EPI_RDB5, false>(p, st);
        }
        return small ? launch_trunk_t<2, 2, 5, EPI_RDB5_RRDB, false>(p, st) : launch_trunk_t<2, 4, 4, EPI_RDB5_RRDB, false>(p, st);
    }
    return hipErrorNotSupported;
}

}  // namespace s2sr

namespace s2sr {

size_t conv_wpack_bytes_f8(int cin, int cout) {
    const int nreal = (cin + 31) / 32, npad = (nreal + 1) & ~1, CT = (cout + 31) / 32;
    return (size_t)npad * 9 * CT * 1024;
}

// w: OIHW fp32.  Per output channel a power-of-two scale 2^k_co puts max|w_co| into [224, 448) -- the top binade of
// e4m3, so every weight keeps its 3 mantissa bits unless it is more than 2^8 below its row's maximum -- and the MFMA
// takes 2^-k_co back out through scale_a.  wscale_out[co] = the E8M0 byte 127 - k_co.
void pack_conv_weights_f8(const float* w, int cin, int cout, void* dst_host, int32_t* wscale_out) {
    const int nreal = (cin + 31) / 32, npad = (nreal + 1) & ~1, CT = (cout + 31) / 32;
    int kco[64];
    for (int co = 0; co < 64; ++co) {
        float m = 0.f;
        if (co < cout)
            for (size_t i = 0; i < (size_t)cin * 9; ++i) m = fmaxf(m, fabsf(w[(size_t)co * cin * 9 + i]));
        int k = 0;
        if (m > 0.f) {
            k = (int)floorf(log2f(448.0f / m));
            while (ldexpf(m, k) >= 448.0f) --k;          // guard the rounding of log2f at the binade edge
            while (ldexpf(m, k + 1) < 448.0f) ++k;
        }
        if (k > 100) k = 100;
        if (k < -100) k = -100;
        kco[co] = k;
        wscale_out[co] = 127 - k;
    }
    uint8_t* d = (uint8_t*)dst_host;
    for (int pl = 0; pl < npad; ++pl)
        for (int t = 0; t < 9; ++t)
            for (int ct = 0; ct < CT; ++ct)
                for (int h16 = 0; h16 < 2; ++h16)
                    for (int row = 0; row < 32; ++row)
                        for (int j = 0; j < 16; ++j) {
                            const int co = ct * 32 + row, ci = 32 * pl + 16 * h16 + j;
                            float v = 0.f;
                            if (pl < nreal && co < cout && ci < cin) v = ldexpf(w[((size_t)co * cin + ci) * 9 + t], kco[co]);
                            *d++ = f32_to_e4m3(v);
                        }
}

hipError_t launch_conv_trunk_f8(const ConvParams& p, int ct, int epi, hipStream_t st) {
    if (ct == 1 && epi == EPI_LRELU) {
        // kernel forms, all bit-identical in their results (tests/test_gpu_net.py); p.f8_form comes from the environment at s2sr_create:
        // weights (S2SR_FP8_WSTREAM): 0 (default) conv1-3 keep theirs resident in LDS (<= 4 planes incl. a phantom, next to the 6-slot
        // slab ring), conv4 streams them (6 planes would cost two slab slots); 1 all stream; 2 all resident (conv4 on a 4-slot ring).
        // Measured on one box, conv1-4 per 5 steps: 108.7 / 109.8 / 111.9 ms -- +1 %, nothing like the -18 % a "no weight DMA"
        // diagnostic suggested (that one read zeros as weights, and an MFMA fed zeros draws less power: the chip clocked higher).
        // The loader-wave form (conv_trunk_f8 PROD) is the default: 77.5 against 79.8 us per conv1-4 launch of 32 tiles on one box,
        // A/B/A/B (+3 %; the no-MFMA floor of either form is 58 us).  S2SR_FP8_LOADER=0 selects the four-wave forms below.
#if S2SR_EXPERIMENTAL
        const bool w8 = (p.f8_form & 8) != 0, loader = (p.f8_form & 1) == 0;
        const int stream_w = (p.f8_form >> 1) & 3;
        if (w8) return launch_trunk8_t<1, 2, 6, EPI_LRELU, 8>(p, st);
        if (!loader) {
            if (stream_w == 1 || (stream_w == 0 && p.nstage > 4)) return launch_trunk8_t<1, 4, 6, EPI_LRELU>(p, st);
            return p.nstage <= 4 ? launch_trunk8_t<1, 4, 6, EPI_LRELU, 4, 4>(p, st) : launch_trunk8_t<1, 4, 4, EPI_LRELU, 4, 6>(p, st);
        }
#else
        if (p.f8_form != 0) return hipErrorNotSupported;      // the other conv1-4 forms: experimental library only
#endif
        return p.nstage <= 4 ? launch_trunk8_t<1, 4, 6, EPI_LRELU, 4, 4, 1>(p, st) : launch_trunk8_t<1, 4, 6, EPI_LRELU, 4, 0, 1>(p, st);
    }
    if (ct == 2 && epi == EPI_RDB5) return launch_trunk8_t<2, 4, 4, EPI_RDB5>(p, st);
    if (ct == 2 && epi == EPI_RDB5_RRDB) return launch_trunk8_t<2, 4, 4, EPI_RDB5_RRDB>(p, st);
    return hipErrorNotSupported;
}
}  // namespace s2sr
