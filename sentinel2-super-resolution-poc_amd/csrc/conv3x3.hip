// 3x3 / stride 1 / zero-pad 1 convolution as an im2col-free implicit GEMM on the gfx950 matrix
// cores: the kernel behind every conv of RRDBNet.forward
// (reference server/app/cnn_super_resolution.py:85-91,103-107,140-158).
//
// GEMM orientation:  D[Cout x pixels] = Wt[Cout x K] * X[K x pixels],  K = 9 taps x Cin.
//   v_mfma_f32_32x32x16_f16:  A = weights  (lane = cout + 32*(k/8)),
//                             B = activations (lane = pixel + 32*(k/8)),
//                             D: lane = pixel column, 16 registers = 16 couts
//   so a lane ends up owning 16 output channels of ONE pixel: bias, LeakyReLU, the x0.2
//   residual adds of RDB / RRDB and the stores are lane-local; one v_permlane32_swap per
//   dword pairs the two half-waves into 16-byte stores.
//
// Schedule (measured choices, see DESIGN.md "kernel history"):
//   * PERSISTENT workgroups, one per CU; each walks a list of 16x32-pixel output patches.  The
//     (patch, 16-channel input block) pairs form one stream of pipeline stages, so the loads of
//     the next patch are in flight while the current one finishes and runs its epilogue.
//   * A stage = one slab plane ((TH+2) x 34 px x 16 ch, 32 B per pixel, 16-B halves XOR-swizzled
//     with bit 3 of the pixel index: every ds_read_b128 of a B fragment is conflict free) plus
//     that block's weights in A-fragment order (9 x CT KiB).  Both arrive by LDS-DMA
//     (global_load_lds_dwordx4) into an R-deep LDS ring: R-2 stages stay in flight behind a
//     COUNTED s_waitcnt vmcnt(N) and a raw s_barrier; vmcnt(0) appears only at the tail.
//     Every wave issues the same number of DMA instructions per stage (padding slots re-load
//     the last piece) so the count is exact.  The DMA and the residual loads are inline asm:
//     any load hipcc can see inside the ring loop makes it protect register reuse with
//     vmcnt(0), which drains the ring every stage.
//   * Activations are "blocked-16" in HBM (s2sr_internal.h): a slab row is 1088 contiguous
//     bytes, each DMA instruction fetches 8 whole cache lines; the zero halo removes all bounds
//     checks; nearest-2x upsampling (cnn_super_resolution.py:146-154) is folded into the
//     loader's source offsets (>>1) and never materialised.
//   * Inside a stage the 9 A fragments stay in registers and every B fragment (slab row s,
//     column shift dx) is read ONCE and used for all kernel rows dy with 0 <= s-dy < NP.
//   * Split-operand convs of the high-precision mode (F8): fp16 main term + e4m3 correction planes
//     on the block-scaled fp8 MFMA, two planes per K=64 instruction (see the F8 / HPO notes at the
//     kernel); their up-convs run in sub-pixel form (PH): 2x2 taps on the source image.
#include <math.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "s2sr_internal.h"

#ifndef S2SR_DMA_LATE
#define S2SR_DMA_LATE 0
#endif
#ifndef S2SR_PREFETCH_D
#define S2SR_PREFETCH_D 1
#endif

namespace s2sr {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int WAVES_, int NP_, int CT_, int R_, int TAPS_ = 9>
struct Geom {
    static constexpr int WAVES = WAVES_, NP = NP_, CT = CT_, R = R_, TAPS = TAPS_;
    static constexpr int TH = WAVES * NP, TW = 32;
    static constexpr int SW = TW + 2, SH = TH + 2, SPX = SH * SW;
    static constexpr int PLANE = ((SPX * 32 + 1023) / 1024) * 1024;
    static constexpr int PI = PLANE / 1024;            // slab LDS-DMA instructions per stage
    static constexpr int WI = TAPS * CT;               // weight LDS-DMA instructions per stage
    static constexpr int NSTI = PI + WI;
    static constexpr int PW = (NSTI + WAVES - 1) / WAVES;   // DMA instructions per wave per stage
    static constexpr int STAGE_BYTES = NSTI * 1024;
    static constexpr int BIAS_OFF = R * STAGE_BYTES;
    static constexpr int LDS_BYTES = BIAS_OFF + (CT * 128 > 256 ? CT * 128 : 256);   // fp32 bias per cout tile
    static constexpr int NBSTEP = 3 * (NP + 2);        // B fragments read per stage
};

// a wave-uniform pointer made PROVABLY uniform for an "s" asm operand
__device__ __forceinline__ const char* uniform_ptr(const char* q) {
    const uint64_t v = (uint64_t)q;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}

// LDS-DMA, 16 B per lane: LDS[lds_addr + lane*16] <- global[base + voff].  M0 is written in the
// statement that uses it (hipcc keeps nothing live in M0 across statements in this kernel: no
// other LDS-DMA, GWS, sendmsg or movrel user), one wait state between the M0 write and the DMA.
// FORCE_UNIFORM re-derives base / lds_addr through v_readfirstlane: only the stamped diagnostic
// build needs it (its divergent stamp branches make hipcc keep these uniform values in VGPRs,
// which an "s" operand cannot take); s_nop 4 then covers the VALU-written SGPRs.
template <bool FORCE_UNIFORM>
__device__ __forceinline__ void glds16(const char* base, uint32_t voff, uint32_t lds_addr) {
    if (FORCE_UNIFORM) {
        base = uniform_ptr(base);
        lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr)
                     : "memory");
    } else {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr)
                     : "memory");
    }
}

// 16-B global load hidden from hipcc's waitcnt bookkeeping; valid only after an explicit
// s_waitcnt vmcnt(0) (asm) that the caller places before the first use.
// The destination is an ACCUMULATION register ("a"; gfx90a+ vector-memory instructions may target
// AGPRs): the compiler does not know the value is still in flight, so whatever it does with the
// register before the wait must be nothing.  With "=v" and more than 256 live registers (the one-wave-
// per-SIMD forms) hipcc parked freshly "loaded" VGPRs in AGPRs right behind the asm statement -- it
// copied bytes that had not arrived, handed the VGPR to someone else (an address), and the load then
// landed on top of it: the memory fault of r01's 4-wave variant.  asm_land() after the wait ties the
// value to a statement behind the wait, so every consumer (and every copy to a VGPR) comes after it.
__device__ __forceinline__ f32x4 asm_load16(const float* addr) {
    f32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(r) : "v"(addr) : "memory");
    return r;
}
template <typename T>
__device__ __forceinline__ void asm_land(T& r) { asm volatile("" : "+a"(r)); }

// LeakyReLU(0.2): max(v, 0.2v) is the same value for every finite v and one VALU op shorter
__device__ __forceinline__ u32x2 asm_load8(const char* addr) {
    u32x2 r;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=a"(r) : "v"(addr) : "memory");
    return r;
}
__device__ __forceinline__ f32x4 half4_to_float(u32x2 h) {
    const f16x4 v = __builtin_bit_cast(f16x4, h);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (float)v[i];
    return o;
}
__device__ __forceinline__ float lrelu(float v) { return fmaxf(v, __fmul_rn(v, 0.2f)); }

// Epilogue stores are unconditional (lanes outside the image write to a trash line), so their
// count per wave is a compile-time constant and the first wait after an epilogue can allow for
// them exactly: the ring keeps its R-2 stages in flight across patch boundaries.
template <int EPI, int CT, int NP, int HPO>
struct EpiStores {
    static constexpr int value = (EPI == EPI_LRELU || EPI == EPI_BODY) ? CT * (HPO == 1 ? 4 : HPO == 2 ? 3 : 2) * NP
                                 : (EPI == EPI_RDB5)                   ? CT * 4 * NP
                                 : (EPI == EPI_RDB5_RRDB)              ? CT * 8 * NP
                                 : (EPI == EPI_FIRST)                  ? CT * 12 * NP
                                                                       : -1;   // LAST / DEBUG: data-dependent, stay conservative
};

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    // counted wait for this wave's own LDS-DMA, then the workgroup barrier: past it, every
    // wave's pieces of the awaited stage are in LDS.  One statement, memory clobber: no LDS
    // read can be scheduled above it.
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// F8 schedule: for how many stage waits after an epilogue its stores may stay in flight.  vmcnt retires in order, so the
// wait for stage j may allow exactly the operations issued after stage j's DMA: the later stages' DMA and -- while the awaited
// DMA is older than the epilogue -- the epilogue's NST stores.  Both stages in flight at the epilogue qualify (2); measured
// against 1 (r03, conv_hr / conv_up): no difference, the stores are not what the next stages wait for.  1 ships.
#ifndef S2SR_HPO_SHORT
#define S2SR_HPO_SHORT 1   // short e4m3 encodings in the split-operand producers' epilogue (see there)
#endif
#ifndef S2SR_F8_STORE_SLACK
#define S2SR_F8_STORE_SLACK 1
#endif
#ifndef S2SR_DIAG_F8
#define S2SR_DIAG_F8 0   // timing diagnostics of the split-operand (F8) schedule: 1 no LDS-DMA, 2 no MFMA, 4 no epilogue, 8 epilogue without its stores
#endif
#define S2SR_STAMP(k)                                                              \
    do {                                                                           \
        if (TRACE && p.trace && !(p.dbg & 4) && tid == 0 && (k) < 20)              \
            p.trace[(size_t)blockIdx.x * 24 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
// dbg & 4: per-wave anatomy of steady-state stage 5 instead (tools/trace_waves.py): slot 0 = leaves the
// barrier before stage 5, slot 2 = has issued its last MFMA of stage 5, slot 1 = leaves the barrier before stage 6
#define S2SR_WSTAMP(slot)                                                                                 \
    do {                                                                                                  \
        if (TRACE && p.trace && (p.dbg & 4) && lane == 0)                                                 \
            p.trace[(size_t)blockIdx.x * 24 + (slot) * 8 + wave] = __builtin_amdgcn_s_memtime();          \
    } while (0)

// F8 (split-operand consumers, S2SR_PREC_F16_HP): a patch is 8 stages -- 4 fp16 blocks of x_hi
// against w_hi, then 4 fp8 (e4m3) planes of 32 channels: [x_lo*2^11 | x_hi] against
// [w_hi | w_lo*2^11], two planes per v_mfma_scale_f32_32x32x64_f8f6f4 (lanes 0-31 take their 32 K
// bytes from the first plane, lanes 32-63 from the second) with a constant 2^-11 block scale: the
// correction terms run at twice the fp16 rate on half the bytes.  An fp8 plane is 32 B per pixel,
// byte-for-byte the geometry of an fp16 block-16 plane, so loader, ring and swizzle are shared.
// HPO (their producers): besides the fp16 output, write those fp8 planes (p.T: lo8 p0, p1, hi8 p0, p1); HPO = 2: only the lo8
// planes (conv_hr when conv_last runs folded and never reads x_hi as e4m3).
// PH >= 0 (F8 kernels only): sub-pixel form of "nearest-2x upsample, then 3x3 conv" (conv_up1 / conv_up2,
// cnn_super_resolution.py:146-154).  The two upsampled rows 2y, 2y+1 are the same source row, so the
// output pixels of row parity py = PH are a 2x2-tap conv of the SOURCE image with the 3x3 taps that
// land on the same source pixel summed on the host (py = 0: rows {dy 0} {dy 1,2} on source rows y-1, y;
// py = 1: {dy 0,1} {dy 2} on y, y+1; same in x): 4 MACs per output pixel instead of 9, identical up to
// fp32 rounding of the summed weights.  One launch does both column parities, as extra "cout tiles":
// tile ct = q * CTR + rc holds couts rc*32.. of parity q, whose B fragments are the slab shifted by one
// more column -- so a lane owns the output pixels 2x and 2x+1 and a wave row writes whole cache lines
// (one parity per launch would write every other 32-B sector of each line: read-modify-write in L2/HBM).
// The launch walks patches of the source image (p.H, p.W, sHp, sWp = source; Hp, Wp = the 2x output
// tensor) and stores to (2y+py, 2x+q).
// FULL: the launch has no ragged edge and no mosaic separators (whole patches only: H % TH == 0, W % 32 == 0, mos_py == 0): the
// epilogue carries no px_live arithmetic and no trash-line selects (split-operand producers: conv_up -3.7 %, conv_hr -2.6 %).
template <int CT, int NP, int WAVES, int EPI, bool UP, int R, bool TRACE = false, int HPO = 0, int OCC = 1, bool F8 = false,
          int PH = -1, bool FULL = false>
__global__ void __launch_bounds__(WAVES * 64, OCC * WAVES / 4) conv3x3_f16(const ConvParams p) {
    using G = Geom<WAVES, NP, CT, R, (PH >= 0 ? 4 : 9)>;
    static_assert(!F8 || R == 4, "the fp8 pair schedule is written for a 4-slot ring");
    static_assert(PH < 0 || (!UP && CT % 2 == 0 && R == 4), "the sub-pixel form replaces upsample-on-load");
    constexpr int PY = PH >= 0 ? PH : 0;
    // conv_last (F8): HPO == 3 names the folded 6-stage schedule (S2SR_LAST_FOLD, default) and compiles only that; HPO == 0 only the
    // 8-stage one.  One kernel carrying both needed 1,479 SGPR-spill reloads per trip (it writes no planes: HPO is free to mean this).
    constexpr bool kFold6 = F8 && EPI == EPI_LAST && HPO == 3;
    constexpr int CTR = PH >= 0 ? CT / 2 : CT;     // real cout tiles; in the sub-pixel form tile ct = q*CTR + rc, q = column parity
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pcol = lane & 31, hh = lane >> 5;
    S2SR_STAMP(0);
    if (TRACE && p.trace && !(p.dbg & 4) && tid == 0) {
        p.trace[(size_t)blockIdx.x * 24 + 20] = __builtin_amdgcn_s_memrealtime();
        p.trace[(size_t)blockIdx.x * 24 + 22] = __builtin_amdgcn_s_memtime();
    }

    // ---- my patches.  Round `it` of the grid covers tiles [it*nwg, (it+1)*nwg); inside a round
    // the workgroups that share an XCD (same blockIdx % 8) take one contiguous run of tiles, so
    // neighbouring patches (shared halos) meet in one XCD's L2.
    const int nwg = gridDim.x;   // multiple of 8 (host)
    const int slot_in_round = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
    const int tpi = p.tilesX * p.tilesY;
    const int ntiles = tpi * p.N;
    const int my_tiles = (ntiles - slot_in_round + nwg - 1) / nwg;   // >= 0
    const int NS = p.nstage;                                         // stages per patch
    const int S = my_tiles * NS;
    const uint32_t sblk = (uint32_t)p.sHp * p.sWp * 32;              // bytes between source blocks
    const size_t oblk = (size_t)p.Hp * p.Wp * 32;                    // bytes between output blocks

    if (tid < CT * 32) ((float*)(smem + G::BIAS_OFF))[tid] = (EPI == EPI_FIRST) ? 0.f : p.bias[tid % (CTR * 32)];

    // ---- per-lane source offsets of this wave's PW DMA slots (patch independent)
    uint32_t loff[G::PW];
#pragma unroll
    for (int s = 0; s < G::PW; ++s) {
        int j = wave + s * WAVES;
        if (j > G::NSTI - 1) j = G::NSTI - 1;
        loff[s] = 0;
        if (j < G::PI) {
            const int i = j * 64 + lane;    // 16-B piece of the slab plane
            int q = i >> 1;
            if (q >= G::SPX) q = 0;         // tail pieces land in the plane's pad
            const int h2 = (i & 1) ^ ((q >> 3) & 1);
            const int ry = q / G::SW, rx = q - ry * G::SW;
            const int ly = UP ? ((ry - 1) >> 1) + 1 : ry;   // offset from the patch's source origin
            const int lx = UP ? ((rx - 1) >> 1) + 1 : rx;
            loff[s] = (uint32_t)((ly * p.sWp + lx) * 32 + h2 * 16);
        } else {
            loff[s] = (uint32_t)((j - G::PI) * 1024 + lane * 16);
        }
    }

    // ---- issue cursor: (tile iteration, stage in patch) + uniform base of its patch
    int it_i = 0, st_i = 0, seg_i = 0, blk_i = 0;
    const char* pbase = nullptr;
    const char* pbase_lo = nullptr;
    // source pointer of the stage under the issue cursor, then advance the cursor
    auto next_src = [&]() __attribute__((always_inline)) -> const char* {
        if (st_i == 0) {
            const int tile = it_i * nwg + slot_in_round;
            const int n = tile / tpi;
            const int trem = tile - n * tpi;
            const int ty = trem / p.tilesX, tx = trem - ty * p.tilesX;
            const int y0 = ty * G::TH, x0 = tx * G::TW;
            const size_t opix = UP ? (size_t)(y0 >> 1) * p.sWp + (x0 >> 1) : (size_t)y0 * p.sWp + x0;
            pbase = p.src + (size_t)n * p.src_img + opix * 32;
            pbase_lo = p.src_lo + (size_t)n * p.lo_img + opix * 32;
        }
        const char* sb = (((p.seg_lo_mask >> seg_i) & 1) ? pbase_lo : pbase) + (size_t)blk_i * sblk;
        if (++blk_i == p.seg_len) { blk_i = 0; ++seg_i; }
        if (++st_i == NS) { st_i = 0; seg_i = 0; blk_i = 0; ++it_i; }
        return sb;
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

    // ---- B-fragment addresses inside a slab plane: slab row s (relative to the wave), shift dx
    uint32_t baddr[NP + 2][3];
#pragma unroll
    for (int s = 0; s < NP + 2; ++s)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int q = (wave * NP + s) * G::SW + pcol + dx;
            baddr[s][dx] = (uint32_t)(q * 32 + 16 * (hh ^ ((q >> 3) & 1)));
        }
    const uint32_t aaddr = G::PLANE + lane * 16;

    // conv5 forms: the trunk is carried as an fp16 pair (hi = the x the convs read, lo = what fp16
    // lost), t = hi + lo.  hi of this patch's own pixels is picked out of the slab planes while
    // the first four stages (the 64 channels of x) are in LDS -- it never comes from HBM again.
    constexpr bool kTrunk = (EPI == EPI_RDB5 || EPI == EPI_RDB5_RRDB);
    u32x2 hi_cap[4][NP][2];
    uint32_t caddr[NP];
#pragma unroll
    for (int np = 0; np < NP; ++np) {
        const int q = (wave * NP + np + 1) * G::SW + pcol + 1;
        caddr[np] = (uint32_t)(q * 32 + 16 * ((q >> 3) & 1) + 8 * hh);   // half 0; half 1 is at ^16
    }

    char* const trash = p.trash + (size_t)(tid & 255) * 16;   // where out-of-image lanes park their stores

    f32x16 acc[CT][NP];
    auto init_acc = [&]() __attribute__((always_inline)) {
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

    // One stage.  `do_issue` (workgroup-uniform): also issue the DMA of the stage R-1 ahead into
    // LDS slot `sl_off`, one instruction per B step, between the MFMAs.
    // n_issue (workgroup-uniform; 0/1, the F8 schedule also 2): stages whose DMA is issued from
    // inside this stage, into LDS slots sl_off / sl_off1
    constexpr int MAXI = F8 ? 2 : 1;
    auto stage_body = [&](const char* buf, int n_issue, uint32_t sl_off, uint32_t sl_off1, int r, bool capture) __attribute__((always_inline)) {
        if (kTrunk && r < 4 && capture) {
#pragma unroll
            for (int np = 0; np < NP; ++np) {
                hi_cap[r < 4 ? r : 0][np][0] = *(const u32x2*)(buf + caddr[np]);
                hi_cap[r < 4 ? r : 0][np][1] = *(const u32x2*)(buf + (caddr[np] ^ 16));
            }
        }
        const char* sbv[MAXI] = {nullptr};
        const char* wbv[MAXI] = {nullptr};
        const uint32_t slv[2] = {sl_off, sl_off1};
#pragma unroll
        for (int w = 0; w < MAXI; ++w)
            if (w < n_issue) {
                wbv[w] = (const char*)p.wpack + (size_t)st_i * (G::WI * 1024);
                sbv[w] = next_src();
            }
        // A fragment of tap t = dy*3+dx is first needed at B step t (slab row s = dy), so the 9 taps
        // are fetched one step ahead of their first use instead of all up front: no LDS-read
        // bubble behind the barrier.
        f16x8 a[9][CT];
        constexpr int D = S2SR_PREFETCH_D;   // fragments are requested D steps ahead of their MFMAs ...
        f16x8 b[D + 1];
#pragma unroll
        for (int t = 0; t < D; ++t) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) a[t][ct] = *(const f16x8*)(buf + aaddr + (t * CT + ct) * 1024);
            b[t] = *(const f16x8*)(buf + baddr[t / 3][t % 3]);
        }
#pragma unroll
        for (int step = 0; step < G::NBSTEP; ++step) {
            const int s = step / 3, dx = step % 3;
            if (step + D < 9) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    a[step + D][ct] = *(const f16x8*)(buf + aaddr + ((step + D) * CT + ct) * 1024);
            }
            if (step + D < G::NBSTEP) b[(step + D) % (D + 1)] = *(const f16x8*)(buf + baddr[(step + D) / 3][(step + D) % 3]);
            // ... and stay there: without this fence hipcc sinks the reads back next to their use
            // (lgkmcnt(0/1) before every MFMA), which exposes the LDS latency on every step
            __builtin_amdgcn_sched_barrier(0);
            if (n_issue > 0)
#pragma unroll
            for (int sq = 0; sq < MAXI * G::PW; ++sq) {
                constexpr int LASTSTEP = G::NBSTEP - 1;
#if S2SR_DMA_LATE
                if ((LASTSTEP - sq > 0 ? LASTSTEP - sq : 0) != step) continue;   // DMA rides on the LAST steps
#else
                if ((sq < LASTSTEP ? sq : LASTSTEP) != step) continue;
#endif
                const int w = sq / G::PW, sl = sq % G::PW;
                if (w >= n_issue) continue;
                int j = wave + sl * WAVES;
                if (j > G::NSTI - 1) j = G::NSTI - 1;   // padding slot: same piece again
                const uint32_t dst = lds0 + slv[w] + (uint32_t)j * 1024;
                uint32_t vo = loff[sl];
                const char* bp = j < G::PI ? sbv[w] : wbv[w];
                if (TRACE && (((p.dbg & 1) && j >= G::PI) || ((p.dbg & 2) && j < G::PI))) { vo = lane * 16; bp = (const char*)p.wpack; }
                if (TRACE && (p.dbg & 8)) continue;   // ablation: no DMA instruction at all
                glds16<(TRACE || EPI == EPI_DEBUG)>(bp, vo, dst);
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int np = s - dy;
                if (np < 0 || np >= NP) continue;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct][np] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[dy * 3 + dx][ct], b[step % (D + 1)], acc[ct][np], 0, 0, 0);
            }
        }
        // one wave per SIMD (512 registers): the accumulators live in AGPRs.  Left alone, hipcc moves every
        // finished accumulator to VGPRs and back once per stage (~7 v_accvgpr moves per MFMA) because the
        // epilogue behind the loop wants them in VGPRs; this pins them where the MFMAs want them.
        if (WAVES == 4) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int np = 0; np < NP; ++np) asm volatile("" : "+a"(acc[ct][np]));
        }
    };

    // ---- F8 kernels: both stage kinds walk dx -> slab row -> dy, so only the three A fragments of one
    // kernel column are live (the 9-tap residency of stage_body plus the fp8 operands would spill).
    // Every B fragment is still read once and feeds the kernel rows it belongs to.
    constexpr int KT = PH >= 0 ? 2 : 3;          // kernel extent per axis (sub-pixel form: 2)
    constexpr int NBK = NP + KT - 1;             // slab rows a wave touches
    const char* sbv[2] = {nullptr, nullptr};
    const char* wbv[2] = {nullptr, nullptr};
    uint32_t slv[2] = {0, 0};
    auto plan_dma = [&](int n_issue, uint32_t sl_off, uint32_t sl_off1) __attribute__((always_inline)) {
        slv[0] = sl_off; slv[1] = sl_off1;
#pragma unroll
        for (int w = 0; w < 2; ++w)
            if (w < n_issue) {
                wbv[w] = (const char*)p.wpack + (size_t)st_i * (G::WI * 1024);
                sbv[w] = next_src();
            }
    };
    auto dma_step = [&](int step, int n_issue) __attribute__((always_inline)) {   // `step` is a compile-time constant at every call
#if S2SR_DIAG_F8 & 1
        return;   // timing diagnostic (tools/tail_anatomy.sh): the F8 schedule without its LDS-DMA
#endif
#pragma unroll
        for (int sq = 0; sq < 2 * G::PW; ++sq) {
            constexpr int LASTSTEP = 3 * NBK - 1;
            if ((sq < LASTSTEP ? sq : LASTSTEP) != step) continue;
            const int w = sq / G::PW, sl = sq % G::PW;
            if (w >= n_issue) continue;
            int j = wave + sl * WAVES;
            if (j > G::NSTI - 1) j = G::NSTI - 1;
            glds16<(TRACE || EPI == EPI_DEBUG)>(j < G::PI ? sbv[w] : wbv[w], loff[sl], lds0 + slv[w] + (uint32_t)j * 1024);
        }
    };
    // hipcc sinks the (side-effect free) MFMAs of a finished column below the next wait/barrier and keeps
    // every fragment they read alive across it; passing the accumulators through an empty volatile asm
    // pins the column's MFMAs where they were written
    auto pin_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int np = 0; np < NP; ++np) {
                if (WAVES == 4) asm volatile("" : "+a"(acc[ct][np]));   // one wave per SIMD: the accumulators live in AGPRs (see stage_body)
                else asm volatile("" : "+v"(acc[ct][np]));
            }
    };
    auto stage16_dx = [&](const char* buf, int n_issue, uint32_t sl_off, uint32_t sl_off1) __attribute__((always_inline)) {
        plan_dma(n_issue, sl_off, sl_off1);
        // columns of the slab: 3 shifts.  Plain form: shift dx = kernel column dx for every tile.  Sub-pixel
        // form: tile ct of column parity q uses kernel column bt = dx - q in {0, 1}.
        f16x8 a[2][KT][CT];   // this column's fragments and the next column's, fetched a column ahead
        auto load_a = [&](int dx) __attribute__((always_inline)) {
#pragma unroll
            for (int dy = 0; dy < KT; ++dy)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int bt = PH >= 0 ? dx - ct / CTR : dx;
                    if (bt < 0 || bt >= KT) continue;
                    a[dx & 1][dy][ct] = *(const f16x8*)(buf + aaddr + ((dy * KT + bt) * CT + ct) * 1024);
                }
        };
        constexpr bool APF16 = !(PH >= 0 && NP >= 2);
        if (APF16) load_a(0);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            if (!APF16) load_a(dx);
            f16x8 b[2];
            b[0] = *(const f16x8*)(buf + baddr[PY][dx]);
#pragma unroll
            for (int s = 0; s < NBK; ++s) {      // slab row s + PY
                if (s + 1 < NBK) b[(s + 1) & 1] = *(const f16x8*)(buf + baddr[s + 1 + PY][dx]);
                if (APF16 && s == 0 && dx + 1 < 3) load_a(dx + 1);
                __builtin_amdgcn_sched_barrier(0);
                dma_step(dx * NBK + s, n_issue);
#pragma unroll
                for (int dy = 0; dy < KT; ++dy) {
                    const int np = s - dy;
                    if (np < 0 || np >= NP) continue;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const int bt = PH >= 0 ? dx - ct / CTR : dx;
                        if (bt < 0 || bt >= KT) continue;
#if S2SR_DIAG_F8 & 2
                        asm volatile("" ::"v"(a[dx & 1][dy][ct]), "v"(b[s & 1]));   // timing diagnostic: fragment reads kept, no MFMA
#else
                        acc[ct][np] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[dx & 1][dy][ct], b[s & 1], acc[ct][np], 0, 0, 0);
#endif
                    }
                }
            }
            pin_acc();
        }
    };
    // Two fp8 planes (ring slots at LDS offsets offA / offB) as one K=64 step per tap: lanes 0-31
    // take their 32 K bytes from the first plane, lanes 32-63 from the second.  The two 16-B halves
    // of a pixel are read in physical order (one address + immediate) and put into channel order
    // with v_cndmask on the loader's swizzle bit.
    auto pair_body = [&](uint32_t offA, uint32_t offB, int n_issue, uint32_t sl_off, uint32_t sl_off1) __attribute__((always_inline)) {
        plan_dma(n_issue, sl_off, sl_off1);
        typedef int v4i __attribute__((ext_vector_type(4)));
        const char* mb = smem + (hh ? offB : offA);
        const uint32_t afrag = G::PLANE + pcol * 16;
        auto bfrag = [&](int s, int dx) __attribute__((always_inline)) -> v8i {
            const uint32_t pa = baddr[s][dx] & ~16u;                 // physical half 0 of the pixel
            const v4i x0 = *(const v4i*)(mb + pa), x1 = *(const v4i*)(mb + pa + 16);
            const bool sw = (pa >> 8) & 1;                           // bit 3 of the pixel index: halves stored swapped
            const v4i l0 = sw ? x1 : x0, l1 = sw ? x0 : x1;
            return __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        v8i a8[2][KT][CT];
        auto load_a8 = [&](int dx) __attribute__((always_inline)) {
#pragma unroll
            for (int dy = 0; dy < KT; ++dy)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int bt = PH >= 0 ? dx - ct / CTR : dx;
                    if (bt < 0 || bt >= KT) continue;
                    const char* fp = mb + afrag + (uint32_t)((dy * KT + bt) * CT + ct) * 1024;
                    const v4i x0 = *(const v4i*)(fp), x1 = *(const v4i*)(fp + 512);
                    a8[dx & 1][dy][ct] = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
        };
        constexpr bool APF = !(PH >= 0 && NP >= 2);   // fetch the next column's A fragments a column ahead (register budget permitting)
        if (APF) load_a8(0);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            if (!APF) load_a8(dx);
            v8i b8[2];
            b8[0] = bfrag(PY, dx);
#pragma unroll
            for (int s = 0; s < NBK; ++s) {
                if (s + 1 < NBK) b8[(s + 1) & 1] = bfrag(s + 1 + PY, dx);
                if (APF && s == (NBK > 2 ? 1 : 0) && dx + 1 < 3) load_a8(dx + 1);
                __builtin_amdgcn_sched_barrier(0);
                dma_step(dx * NBK + s, n_issue);
#pragma unroll
                for (int dy = 0; dy < KT; ++dy) {
                    const int np = s - dy;
                    if (np < 0 || np >= NP) continue;
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {   // e4m3 x e4m3, scale_a = 2^0, scale_b = 2^-11 (E8M0 bytes 127, 116)
                        const int bt = PH >= 0 ? dx - ct / CTR : dx;
                        if (bt < 0 || bt >= KT) continue;
#if S2SR_DIAG_F8 & 2
                        asm volatile("" ::"v"(a8[dx & 1][dy][ct]), "v"(b8[s & 1]));
#else
                        acc[ct][np] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[dx & 1][dy][ct], b8[s & 1], acc[ct][np], 0, 0, 0, 127, 0, 116);
#endif
                    }
                }
            }
            pin_acc();
        }
    };

    // ---- epilogue of the patch at tile iteration `it`
    auto epilogue = [&](int it) __attribute__((always_inline)) {
        const int tile = it * nwg + slot_in_round;
        const int n = tile / tpi;
        const int trem = tile - n * tpi;
        const int ty = trem / p.tilesX, tx = trem - ty * p.tilesX;
        const int y0 = ty * G::TH, x0 = tx * G::TW;
        const int x = x0 + pcol;
        const PatchLive pl = patch_live(p, y0, x0);
        bool ok[NP];
        size_t opix[NP];
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            const int y = y0 + wave * NP + np;
            ok[np] = FULL ? true : px_live(p, pl, y0, x0, y, x);
            opix[np] = PH >= 0 ? (size_t)(2 * y + PY + 1) * p.Wp + (2 * x + 1)    // sub-pixel form: row parity PY, column parity q added per tile
                               : (size_t)(y + 1) * p.Wp + (x + 1);
        }
        const size_t tn = (size_t)n * 8 * oblk;   // image offset inside an fp32 skip tensor (R, F), bytes
        const size_t ln = (size_t)n * 4 * oblk;   // image offset inside the fp16 lo tensor, bytes
        // residual operands: one burst of independent loads (padded tensors make every address
        // valid, so they are unconditional); asm loads + one explicit wait, see asm_load16.
        u32x2 lo_old[CT][NP][4];
        f32x4 res1[CT][NP][4];
        f32x4 res0[CT][NP][4];   // EPI_BODY only
        auto load_res = [&](int np) __attribute__((always_inline)) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    lo_old[ct][np][g] = asm_load8((const char*)p.T + ln + (size_t)(ct * 2 + (g >> 1)) * oblk + opix[np] * 32 +
                                                  (g & 1) * 16 + hh * 8);
                    if (EPI == EPI_RDB5_RRDB)
                        res1[ct][np][g] = asm_load16((const float*)((const char*)p.R + tn + (size_t)(ct * 4 + g) * oblk + opix[np] * 32 + hh * 16));
                }
        };
        auto land_res = [&](int np) __attribute__((always_inline)) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    asm_land(lo_old[ct][np][g]);
                    if (EPI == EPI_RDB5_RRDB) asm_land(res1[ct][np][g]);
                }
        };
        if (EPI == EPI_RDB5) {
#pragma unroll
            for (int np = 0; np < NP; ++np) load_res(np);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int np = 0; np < NP; ++np) land_res(np);
            __builtin_amdgcn_sched_barrier(0);
        } else if (EPI == EPI_BODY) {
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        res0[ct][np][g] = *(const f32x4*)((const char*)p.F + tn + (size_t)(ct * 4 + g) * oblk + opix[np] * 32 + hh * 16);
        }
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            const int y = y0 + wave * NP + np;
            if (EPI == EPI_RDB5_RRDB) {
                load_res(np);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                land_res(np);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int rc = ct % CTR;                                   // real cout tile
                const size_t qoff = PH >= 0 ? (size_t)(ct / CTR) * 32 : 0;   // sub-pixel form: the pixel to the right
                u32x2 hpk[4];   // fp16 x4 per g (hi / plain output)
                u32x2 lpk[4];   // fp16 x4 per g (lo), trunk forms and conv_first
                uint32_t lo8[4], hi8[4];   // e4m3 x4 per g (HPO): channels 8g+4hh .. +3 of fp8 plane ct
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (EPI == EPI_LAST && g > 0) continue;    // conv_last: couts 0..2 only (launch_t checks cout <= 3); 8.. are the folded w_lo sums
                    const int cb = ct * 32 + 8 * g + 4 * hh;   // first of this lane's 4 consecutive couts
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[ct][np][4 * g + i];
                    const size_t to = tn + (size_t)(ct * 4 + g) * oblk + opix[np] * 32 + hh * 16;
                    if (EPI == EPI_LRELU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = lrelu(v[i]);
                    } else if (kTrunk) {
                        // t = hi + lo (exact in fp32); hi was captured from LDS: block ct*2 + (g>>1), half g&1
                        const f32x4 th = half4_to_float(hi_cap[(ct * 2 + (g >> 1)) & 3][np][g & 1]);
                        const f32x4 tl = half4_to_float(lo_old[ct][np][g]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float t = __fadd_rn(th[i], tl[i]);
                            v[i] = __fadd_rn(__fmul_rn(v[i], 0.2f), t);
                            if (EPI == EPI_RDB5_RRDB) v[i] = __fadd_rn(__fmul_rn(v[i], 0.2f), res1[ct][np][g][i]);
                        }
                        if (EPI == EPI_RDB5_RRDB) *(f32x4*)(ok[np] ? (char*)p.R + to : trash) = v;
                    } else if (EPI == EPI_FIRST) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(__fmul_rn(v[i], p.in_scale), p.bias[cb + i]);
                        *(f32x4*)(ok[np] ? (char*)p.R + to : trash) = v;
                        *(f32x4*)(ok[np] ? (char*)p.F + to : trash) = v;
                    } else if (EPI == EPI_BODY) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = __fadd_rn(res0[ct][np][g][i], v[i]);
                    }
                    if (EPI == EPI_LAST || EPI == EPI_DEBUG) {
                        // where the pixel goes: image n of [N, H, W], or -- window mosaic -- window (n*ky + wy)*kx + wx of [count, ry, rx]
                        size_t oimg = (size_t)n;
                        int oy = y, ox = x, oH = p.H, oW = p.W;
                        bool live = ok[np];
                        if (!FULL && p.mos_py) {
                            const int wy = y / p.mos_py, wx = x / p.mos_px;
                            const int t = (n * p.mos_ky + wy) * p.mos_kx + wx;
                            live = live && t < p.mos_count;
                            oimg = (size_t)t; oy = y - wy * p.mos_py; ox = x - wx * p.mos_px; oH = p.mos_ry; oW = p.mos_rx;
                        }
                        if (EPI == EPI_LAST) {
                            // conv_last: couts 0..2 of the lanes hh == 0 (launch_t: cout <= 3, one cout tile; g > 0 skipped above).
                            // One pixel index, one base address per output; no per-channel 64-bit index chains.
                            float o3[3];
                            uint32_t P = 0;
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                o3[i] = (kFold6 || (!F8 && p.fold_lo)) ? __fadd_rn(v[i], acc[ct][np][4 + i]) : v[i];   // + x*w_lo (couts 8..)
                                // (out*255).clip(0,255).astype(uint8): truncation (cnn_super_resolution.py:232)
                                const float q = fminf(fmaxf(__fmul_rn(o3[i], 255.0f), 0.f), 255.f);
                                P |= (uint32_t)(int)q << (8 * i);
                            }
                            const size_t plane = (size_t)oH * oW;
                            const size_t pix = (oimg * (size_t)oH + (size_t)oy) * (size_t)oW + (size_t)ox;   // pixel index in [*, oH, oW]
                            if (p.out_f32 && live && hh == 0) {   // fp32 copy (tests, enhance_f32): planar [img, cout, oH, oW]
                                float* d = p.out_f32 + pix + oimg * (size_t)(p.cout - 1) * plane;
#pragma unroll
                                for (int i = 0; i < 3; ++i)
                                    if (i < p.cout) d[(size_t)i * plane] = o3[i];
                            }
                            if (p.out_u8) {
                                // u8 RGB rows: a wave's 32 pixels of one row are 96 contiguous bytes.  Each lane packs its pixel into
                                // 24 bits, lanes 0..23 collect one dword each with two ds_bpermute and the segment goes out as whole
                                // 32-B sectors (byte stores at stride 3 are partial-sector writes).  Taken when the 32 pixels are all
                                // live, consecutive in one output row and start on a dword (wave-uniform test); else bytes.
                                const uint32_t rowkey = (uint32_t)(oimg * (size_t)oH + (size_t)oy);
                                const uint64_t lm = __builtin_amdgcn_ballot_w64(live);
                                const int ox0 = __builtin_amdgcn_readlane(ox, 0);
                                const bool uni = p.cout == 3 && (uint32_t)lm == 0xffffffffu &&
                                                 __builtin_amdgcn_readlane((int)rowkey, 0) == __builtin_amdgcn_readlane((int)rowkey, 31) &&
                                                 __builtin_amdgcn_readlane(ox, 31) == ox0 + 31 && (ox0 & 3) == 0 && (oW & 3) == 0;
                                if (uni) {
                                    const int p0 = (4 * lane) / 3, r8 = 8 * (4 * lane - 3 * p0);
                                    const uint32_t a0 = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * p0, (int)P);
                                    const uint32_t a1 = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * p0 + 4, (int)P);
                                    const size_t pix0 = (size_t)(uint32_t)__builtin_amdgcn_readlane((int)rowkey, 0) * (size_t)oW + (size_t)ox0;
                                    if (lane < 24) *(uint32_t*)(p.out_u8 + pix0 * 3 + 4 * lane) = (a0 >> r8) | (a1 << (24 - r8));
                                } else if (live && hh == 0) {
                                    uint8_t* d = p.out_u8 + pix * 3;
#pragma unroll
                                    for (int i = 0; i < 3; ++i)
                                        if (i < p.cout) d[i] = (uint8_t)(P >> (8 * i));
                                }
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int co = cb + i;
                                if (co >= p.cout || !live) continue;
                                const float o = p.act ? lrelu(v[i]) : v[i];
                                if (p.out_f32) p.out_f32[((oimg * p.cout + co) * oH + oy) * oW + ox] = o;
                            }
                        }
                    } else {
                        f16x4 hv;
#pragma unroll
                        for (int i = 0; i < 4; ++i) hv[i] = (f16)v[i];
                        hpk[g] = __builtin_bit_cast(u32x2, hv);
                        if (kTrunk || EPI == EPI_FIRST) {
                            f16x4 lv;
#pragma unroll
                            for (int i = 0; i < 4; ++i) lv[i] = (f16)__fsub_rn(v[i], (float)hv[i]);
                            lpk[g] = __builtin_bit_cast(u32x2, lv);
                        } else if (HPO) {
                            // e4m3 copies for the consumer's correction terms: lo*2^11 and hi, clamped to the
                            // finite range (the fp8 conversions turn anything past 448 into NaN)
                            // lo, short form (S2SR_HPO_SHORT; same bytes as the long one, tools/check_hpo_forms.py): v - fp16(v) in ONE
                            // v_fma_mix_f32 that reads the packed half in place, the clamp on the unscaled value, the 2^11 inside
                            // v_cvt_scalef32_pk_fp8_f32 (it divides by the power of two of its scale operand): 2.5 instead of 5.5
                            // instructions per value.  (hi straight from the packed pair -- v_pk_min/max_f16 +
                            // v_cvt_scalef32_pk_fp8_f16 -- was tried too and does NOT give the long form's bytes; it stays long.)
                            typedef short v2s __attribute__((ext_vector_type(2)));
                            if (S2SR_HPO_SHORT) {
                                float q[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    float d;
                                    if (i & 1) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpk[g][i >> 1]), "v"(v[i]));
                                    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpk[g][i >> 1]), "v"(v[i]));
                                    q[i] = __builtin_amdgcn_fmed3f(d, -448.0f / 2048.0f, 448.0f / 2048.0f);
                                }
                                v2s w8 = {0, 0};
                                w8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w8, q[0], q[1], 1.0f / 2048.0f, false);
                                w8 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w8, q[2], q[3], 1.0f / 2048.0f, true);
                                lo8[g] = __builtin_bit_cast(uint32_t, w8);
                            } else {
                                float l4[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    l4[i] = __builtin_amdgcn_fmed3f(__fmul_rn(__fsub_rn(v[i], (float)hv[i]), 2048.0f), -448.0f, 448.0f);
                                int lw = __builtin_amdgcn_cvt_pk_fp8_f32(l4[0], l4[1], 0, false);
                                lw = __builtin_amdgcn_cvt_pk_fp8_f32(l4[2], l4[3], lw, true);
                                lo8[g] = (uint32_t)lw;
                            }
                            if (HPO == 1) {
                                float h4[4];
#pragma unroll
                                for (int i = 0; i < 4; ++i) h4[i] = __builtin_amdgcn_fmed3f((float)hv[i], -448.0f, 448.0f);
                                int hw = __builtin_amdgcn_cvt_pk_fp8_f32(h4[0], h4[1], 0, false);
                                hw = __builtin_amdgcn_cvt_pk_fp8_f32(h4[2], h4[3], hw, true);
                                hi8[g] = (uint32_t)hw;
                            }
                        }
                    }
                }
                if (EPI != EPI_LAST && EPI != EPI_DEBUG) {
                    // pair the half-waves: after the swaps lanes 0-31 hold couts 16b..16b+7 and lanes
                    // 32-63 couts 16b+8..16b+15 of their pixel -> one 16-B store per 16-channel block,
                    // 1 KiB contiguous per wave-instruction
#pragma unroll
                    for (int bk = 0; bk < 2; ++bk) {
                        u32x2 lo = hpk[2 * bk], hi = hpk[2 * bk + 1];
                        const auto r0 = __builtin_amdgcn_permlane32_swap(lo[0], hi[0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(lo[1], hi[1], false, false);
                        u32x4 o;
                        o[0] = r0[0]; o[1] = r1[0]; o[2] = r0[1]; o[3] = r1[1];
                        if ((S2SR_DIAG_F8 & 8) && F8) asm volatile("" ::"v"(o));   // timing diagnostic: the epilogue's arithmetic without its stores
                        else *(u32x4*)(ok[np] ? p.dst + (size_t)n * p.dst_img + (size_t)(rc * 2 + bk) * oblk + opix[np] * 32 + qoff + hh * 16 : trash) = o;
                        if (kTrunk || EPI == EPI_FIRST) {
                            u32x2 llo = lpk[2 * bk], lhi = lpk[2 * bk + 1];
                            const auto q0 = __builtin_amdgcn_permlane32_swap(llo[0], lhi[0], false, false);
                            const auto q1 = __builtin_amdgcn_permlane32_swap(llo[1], lhi[1], false, false);
                            u32x4 ol;
                            ol[0] = q0[0]; ol[1] = q1[0]; ol[2] = q0[1]; ol[3] = q1[1];
                            *(u32x4*)(ok[np] ? (char*)p.T + ln + (size_t)(ct * 2 + bk) * oblk + opix[np] * 32 + hh * 16 : trash) = ol;
                        }
                    }
                    if (HPO && !kTrunk && EPI != EPI_FIRST) {
                        // the lane holds dwords 2g+hh of the pixel's 32 plane bytes; after the swaps
                        // lanes 0-31 hold dwords 0-3, lanes 32-63 dwords 4-7: one 16-B store each
#pragma unroll
                        for (int w = 0; w < (HPO == 1 ? 2 : 1); ++w) {
                            const uint32_t* d = w ? hi8 : lo8;
                            const auto r0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                            const auto r1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
                            u32x4 o;
                            o[0] = r0[0]; o[1] = r0[1]; o[2] = r1[0]; o[3] = r1[1];
                            if ((S2SR_DIAG_F8 & 8) && F8) asm volatile("" ::"v"(o));
                            else *(u32x4*)(ok[np] ? (char*)p.T + ln + (size_t)(2 * w + rc) * oblk + opix[np] * 32 + qoff + hh * 16 : trash) = o;
                        }
                    }
                }
            }
        }
    };

    // ---- prologue: R-1 stages in flight
    {
        // reuse stage_body's issue path without compute: a tiny dedicated loop
#pragma unroll
        for (int r = 0; r < R - 1; ++r) {
            if (r < S) {
                const char* wb = (const char*)p.wpack + (size_t)st_i * (G::WI * 1024);
                const char* sb = next_src();
#pragma unroll
                for (int sl = 0; sl < G::PW; ++sl) {
                    int j = wave + sl * WAVES;
                    if (j > G::NSTI - 1) j = G::NSTI - 1;
                    glds16<(TRACE || EPI == EPI_DEBUG)>(j < G::PI ? sb : wb, loff[sl], lds0 + (uint32_t)(r * G::STAGE_BYTES) + (uint32_t)j * 1024);
                }
            }
        }
    }
    __syncthreads();   // bias visible in LDS
    init_acc();
    S2SR_STAMP(1);

    int k = 0, it_c = 0, st_c = 0;
    bool after_epi = false;
    int epi_age = 8;   // F8 schedule: waits since the last epilogue (saturating)
    constexpr int NST = EpiStores<EPI, CT, NP, HPO>::value;
    constexpr int NW = G::PW * (R - 2);
    if constexpr (F8) {
        // Patch = ring revolution A (4 fp16 stages, slots 0..3) + revolution B (fp8 pairs in slots
        // (0,1) and (2,3)).  A pair frees two slots at once, so the issue cursor runs 1-2-1-1 /
        // 1-2 stages per turn instead of one; `issued` counts stages whose DMA is out, the wait
        // before a turn allows exactly the stages issued after the one it needs.
        constexpr int PWn = G::PW;
        int issued = S < R - 1 ? S : R - 1;
        auto wait_pending = [&](int pend) __attribute__((always_inline)) {
            const int age = epi_age < 8 ? epi_age++ : 8;
            const bool epi = NST > 0 && age < S2SR_F8_STORE_SLACK;
            if (pend >= 2) {
                if (epi) wait_vm_barrier<(NST > 0 ? 2 * PWn + NST : 2 * PWn)>();
                else wait_vm_barrier<2 * PWn>();
            } else if (pend == 1) {
                if (epi) wait_vm_barrier<(NST > 0 ? PWn + NST : PWn)>();
                else wait_vm_barrier<PWn>();
            } else {
                wait_vm_barrier<0>();
            }
            after_epi = false;
        };
        static_assert(2 * PWn + (NST > 0 ? NST : 0) < 64, "vmcnt field is 6 bits");
        uint32_t sl0 = 0, sl1 = 0;
        auto plan_issue = [&](int upto) __attribute__((always_inline)) -> int {   // issue every stage up to `upto` (slot known free)
            const int t = upto < S - 1 ? upto : S - 1;
            int n = t - (issued - 1);
            n = n < 0 ? 0 : n;
            sl0 = (uint32_t)((issued & 3) * G::STAGE_BYTES);
            sl1 = (uint32_t)(((issued + 1) & 3) * G::STAGE_BYTES);
            issued += n;
            return n;
        };
        if constexpr (kFold6) {
            // conv_last, folded form (p.nstage == 6): w_lo rides in the idle couts 8.. of the fp16 stages, so the
            // x_hi planes never come in as e4m3 -- a patch is 4 fp16 stages + ONE pair (x_lo planes against w_hi).
            // Six stages on a four-slot ring: stage j sits in slot j & 3, so patches alternate between
            // "fp16 in slots 0..3, pair in (0,1)" and "fp16 in slots 2,3,0,1, pair in (2,3)".
            {
                while (k < S) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        if (k < S) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                wait_pending(issued - 1 - k);
                                const int n = plan_issue(k + 3);
                                stage16_dx(smem + ((r + 2 * half) & 3) * G::STAGE_BYTES, n, sl0, sl1);
                                ++k;
                            }
                            wait_pending(issued - 1 - (k + 1));
                            const int n = plan_issue(k + 3);
                            pair_body((uint32_t)((2 * half) * G::STAGE_BYTES), (uint32_t)((2 * half + 1) * G::STAGE_BYTES), n, sl0, sl1);
                            k += 2;
#if !(S2SR_DIAG_F8 & 4)
                            epilogue(it_c);
#endif
                            init_acc();
                            ++it_c;
                            after_epi = true;
                            epi_age = 0;
                        }
                    }
                }
                return;
            }
        }
        while (k < S) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wait_pending(issued - 1 - k);
                const int n = plan_issue(k + 3);
                stage16_dx(smem + r * G::STAGE_BYTES, n, sl0, sl1);
                ++k;
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                wait_pending(issued - 1 - (k + 1));
                const int n = plan_issue(k + 3);
                pair_body((uint32_t)((2 * pr) * G::STAGE_BYTES), (uint32_t)((2 * pr + 1) * G::STAGE_BYTES), n, sl0, sl1);
                k += 2;
            }
#if !(S2SR_DIAG_F8 & 4)
            epilogue(it_c);
#endif
            init_acc();
            ++it_c;
            after_epi = true;
            epi_age = 0;
        }
        return;
    }
    // one ring revolution per loop trip; every condition below is workgroup-uniform
    while (k < S) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (k < S) {
                // stage k must have landed; stages k+1 .. k+R-2 (and, right after an epilogue, its
                // stores, which are younger than all of them) may stay in flight
                if (k + (R - 2) < S) {
                    if (NST > 0 && NW + NST < 64 && after_epi) wait_vm_barrier<(NST > 0 ? NW + NST : NW)>();
                    else wait_vm_barrier<NW>();
                } else {
                    wait_vm_barrier<0>();
                }
                after_epi = false;
                S2SR_STAMP(2 + 2 * k);
                if (k == 5) S2SR_WSTAMP(0);
                if (k == 6) S2SR_WSTAMP(1);
                if constexpr (PH >= 0)
                    stage16_dx(smem + r * G::STAGE_BYTES, (k + (R - 1) < S) ? 1 : 0, (uint32_t)(((r + R - 1) % R) * G::STAGE_BYTES), 0u);
                else
                    stage_body(smem + r * G::STAGE_BYTES, (k + (R - 1) < S) ? 1 : 0, (uint32_t)(((r + R - 1) % R) * G::STAGE_BYTES), 0u, r,
                               st_c == r);   // NS % R == 0 for the trunk forms: stages 0..3 of a patch sit in slots 0..3
                S2SR_STAMP(3 + 2 * k);
                if (k == 5) S2SR_WSTAMP(2);
                if (++st_c == NS) {
                    epilogue(it_c);
                    init_acc();
                    st_c = 0;
                    ++it_c;
                    after_epi = true;
                }
                ++k;
            }
        }
    }
    if (TRACE && p.trace && !(p.dbg & 4) && tid == 0) {
        p.trace[(size_t)blockIdx.x * 24 + 21] = __builtin_amdgcn_s_memrealtime();
        p.trace[(size_t)blockIdx.x * 24 + 23] = __builtin_amdgcn_s_memtime();
    }
}

// ------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------
template <int CT, int EPI, bool UP, int WAVES, int NP, int R, bool TRACE = false, int HPO = 0, int OCC = 1, bool F8 = false,
          int PH = -1, bool FULL = false>
static hipError_t launch_t(const ConvParams& p, hipStream_t st) {
    using G = Geom<WAVES, NP, CT, R, (PH >= 0 ? 4 : 9)>;
    if (FULL && (p.mos_py != 0 || p.H % G::TH != 0 || p.W % G::TW != 0)) return hipErrorInvalidValue;
    static_assert(G::LDS_BYTES * OCC <= 160 * 1024, "LDS ring does not fit");
    static_assert(G::PW*(R - 2) < 64, "vmcnt field is 6 bits");
    auto kern = conv3x3_f16<CT, NP, WAVES, EPI, UP, R, TRACE, HPO, OCC, F8, PH, FULL>;
    if (F8 && (p.nstage != ((EPI == EPI_LAST && HPO == 3) ? 6 : 8) || ((EPI == EPI_LAST && HPO == 3) && !p.fold_lo) || p.seg_len != 4 || !p.src_lo))
        return hipErrorInvalidValue;   // 4 fp16 blocks + 4 fp8 planes (conv_last folded: + 2)
    if (PH >= 0 && !F8 && p.nstage != 4) return hipErrorInvalidValue;
    if (EPI == EPI_LAST && (p.cout > 3 || CT != 1)) return hipErrorInvalidValue;   // its epilogue writes couts 0..2 (RGB) only
    // the dynamic-LDS opt-in is per device: a process may hold handles on several GPUs, driven from
    // different threads (each handle has its own mutex, so this table needs one of its own)
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
    if ((EPI == EPI_RDB5 || EPI == EPI_RDB5_RRDB) && (p.nstage % R != 0 || p.nstage < 4))
        return hipErrorInvalidValue;   // the trunk forms pick x out of ring slots 0..3 (see hi_cap)
    ConvParams q = p;
    if (q.seg_len <= 0) { q.seg_len = q.nstage; q.seg_lo_mask = 0; }
    if (!q.src_lo) { q.src_lo = q.src; q.lo_img = q.src_img; }
    q.tilesX = (p.W + G::TW - 1) / G::TW;
    q.tilesY = (p.H + G::TH - 1) / G::TH;
    const int ntiles = q.tilesX * q.tilesY * p.N;
    int grid = (ncu * OCC) & ~7;              // OCC persistent workgroups per CU
    if (ntiles < grid) grid = (ntiles + 7) & ~7;
#if S2SR_DIAG_F8
    if (F8) {   // diagnostic builds only: run the split-operand convs on a subset of the CUs (S2SR_DIAG_GRID workgroups, multiple of 8)
        static const int dg = [] { const char* e = getenv("S2SR_DIAG_GRID"); return e ? atoi(e) & ~7 : 0; }();
        if (dg > 0 && dg < grid) grid = dg;
    }
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), G::LDS_BYTES, st, q);
    return hipGetLastError();
}

// S2SR_W4=1: the RDB convs as 4-wave workgroups, one wave per SIMD with the 512-register budget
// (8 / 4 rows per wave), instead of 8 waves x 4 / 2 rows
#if S2SR_EXPERIMENTAL
static bool use_w4() {
    static const bool v = [] { const char* e = getenv("S2SR_W4"); return e && atoi(e) != 0; }();
    return v;
}
#endif

template <int CT, int EPI, bool UP>
static hipError_t launch_w(const ConvParams& p, hipStream_t st) {
    constexpr int R = (CT == 1) ? 5 : 4;
#if !S2SR_EXPERIMENTAL
    // the RDB convs run on conv_trunk.hip; their 8-wave / 4-wave forms here (r01's trunk, S2SR_TRUNK=0 / S2SR_W4=1) and the
    // upsample-on-load up-convs (S2SR_NO_SUBPIXEL) are in the experimental library only
    if constexpr (!UP && ((CT == 1 && EPI == EPI_LRELU) || (CT == 2 && (EPI == EPI_RDB5 || EPI == EPI_RDB5_RRDB)))) return hipErrorNotSupported;
    else if constexpr (UP && EPI != EPI_DEBUG) return hipErrorNotSupported;
    else return launch_t<CT, EPI, UP, 8, 2, R>(p, st);
#else
    if constexpr (!UP && ((CT == 1 && EPI == EPI_LRELU) || (CT == 2 && (EPI == EPI_RDB5 || EPI == EPI_RDB5_RRDB)))) {
        if (use_w4()) {
            if constexpr (CT == 1) {
                const long n32 = (long)((p.W + 31) / 32) * ((p.H + 31) / 32) * p.N;
                if (n32 >= 192) return launch_t<1, EPI_LRELU, false, 4, 8, 3>(p, st);
                return launch_t<1, EPI_LRELU, false, 4, 4, 5>(p, st);
            } else {
                return launch_t<2, EPI, false, 4, 4, 4>(p, st);
            }
        }
    }
    if (CT == 1 && EPI == EPI_LRELU && !UP) {
        // the 32-cout RDB convs: 32x32 patch (4 rows per wave), 3-deep ring -- unless that leaves most
        // CUs without a patch (single tiles): then the 16x32 patch spreads the image over twice as
        // many workgroups
        const long n32 = (long)((p.W + 31) / 32) * ((p.H + 31) / 32) * p.N;
        if (n32 >= 192) return launch_t<1, EPI_LRELU, false, 8, 4, 3>(p, st);
    }
    return launch_t<CT, EPI, UP, 8, 2, R>(p, st);
#endif
}

hipError_t launch_conv(const ConvParams& p, int ct, int epi, bool up, bool lo_out, hipStream_t st, bool f8_in) {
    if (f8_in) {   // split-operand mode: fp16 main term + fp8 correction planes in; lo_out: fp8 planes out as well
#if S2SR_EXPERIMENTAL
        if (p.tail_form & 1) {   // one wave per SIMD: 4 waves x 4 rows, the same 16x32 patch and ring
            if (ct == 2 && lo_out && epi == EPI_LRELU && !up) return launch_t<2, EPI_LRELU, false, 4, 4, 4, false, 1, 1, true>(p, st);
            if (ct == 1 && !lo_out && epi == EPI_LAST && !up && p.nstage == 6 && p.fold_lo) return launch_t<1, EPI_LAST, false, 4, 4, 4, false, 3, 1, true>(p, st);
        }
#endif
        const bool full = p.mos_py == 0 && p.H % 16 == 0 && p.W % 32 == 0 && !(p.tail_form & 8);   // whole 16x32 patches (bit 3: diagnostic off switch)
        if (ct == 2 && lo_out && epi == EPI_LRELU && !up && (p.tail_form & 2))   // conv_hr in front of a folded conv_last: no hi8 planes out
            return full ? launch_t<2, EPI_LRELU, false, 8, 2, 4, false, 2, 1, true, -1, true>(p, st)
                        : launch_t<2, EPI_LRELU, false, 8, 2, 4, false, 2, 1, true>(p, st);
        if (ct == 2 && lo_out && epi == EPI_BODY && !up && full) return launch_t<2, EPI_BODY, false, 8, 2, 4, false, 1, 1, true, -1, true>(p, st);
        if (ct == 2 && lo_out) {
#if S2SR_EXPERIMENTAL
            if (epi == EPI_LRELU && up) return launch_t<2, EPI_LRELU, true, 8, 2, 4, false, 1, 1, true>(p, st);
#endif
            if (epi == EPI_LRELU && !up) return launch_t<2, EPI_LRELU, false, 8, 2, 4, false, 1, 1, true>(p, st);
            if (epi == EPI_BODY && !up) return launch_t<2, EPI_BODY, false, 8, 2, 4, false, 1, 1, true>(p, st);
        }
        if (ct == 1 && !lo_out && epi == EPI_LAST && !up) {
            if (p.nstage == 6 && p.fold_lo)
                return full ? launch_t<1, EPI_LAST, false, 8, 2, 4, false, 3, 1, true, -1, true>(p, st)
                            : launch_t<1, EPI_LAST, false, 8, 2, 4, false, 3, 1, true>(p, st);
            return launch_t<1, EPI_LAST, false, 8, 2, 4, false, 0, 1, true>(p, st);
        }
        return hipErrorInvalidValue;
    }
    if (lo_out) return hipErrorInvalidValue;
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

// one ROW parity of a split-operand up-conv in sub-pixel form (both column parities inside the launch;
// p.H, p.W = source dims, p.Hp, p.Wp = 2x tensor)
hipError_t launch_conv_phase(const ConvParams& p, int py, hipStream_t st, bool f8) {
#if S2SR_EXPERIMENTAL
    if (f8 && (p.tail_form & 1)) {   // one wave per SIMD: 4 waves x 2 source rows
        if (py == 0) return launch_t<4, EPI_LRELU, false, 4, 2, 4, false, 1, 1, true, 0>(p, st);
        if (py == 1) return launch_t<4, EPI_LRELU, false, 4, 2, 4, false, 1, 1, true, 1>(p, st);
    }
#endif
    if (f8 && p.mos_py == 0 && p.H % 8 == 0 && p.W % 32 == 0 && !(p.tail_form & 8)) {   // whole 8x32 source patches
        if (py == 0) return launch_t<4, EPI_LRELU, false, 8, 1, 4, false, 1, 1, true, 0, true>(p, st);
        if (py == 1) return launch_t<4, EPI_LRELU, false, 8, 1, 4, false, 1, 1, true, 1, true>(p, st);
    }
    if (f8) {
        if (py == 0) return launch_t<4, EPI_LRELU, false, 8, 1, 4, false, 1, 1, true, 0>(p, st);
        if (py == 1) return launch_t<4, EPI_LRELU, false, 8, 1, 4, false, 1, 1, true, 1>(p, st);
    } else {   // plain fp16 mode: the same four fp16 stages, no correction planes in or out
        if (py == 0) return launch_t<4, EPI_LRELU, false, 8, 2, 4, false, 0, 1, false, 0>(p, st);
        if (py == 1) return launch_t<4, EPI_LRELU, false, 8, 2, 4, false, 0, 1, false, 1>(p, st);
    }
    return hipErrorInvalidValue;
}

#if S2SR_EXPERIMENTAL
hipError_t launch_conv_trace(const ConvParams& p, int ct, hipStream_t st) {
    if (use_w4()) {
        if (ct == 1) return launch_t<1, EPI_LRELU, false, 4, 8, 3, true>(p, st);
        return launch_t<2, EPI_RDB5, false, 4, 4, 4, true>(p, st);
    }
    if (ct == 1) return launch_t<1, EPI_LRELU, false, 8, 4, 3, true>(p, st);
    return launch_t<2, EPI_RDB5, false, 8, 2, 4, true>(p, st);
}
#endif

// ------------------------------------------------------------------------------------------
// host-side weight repack.  Layout: [stage = cin/16][tap][ct][lane 0..63][j 0..7] fp16 with
//   value = W[cout = ct*32 + (lane&31)][cin = stage*16 + 8*(lane>>5) + j][tap/3][tap%3] * wscale
// -> each stage's weights are 9*CT contiguous KiB, each KiB is exactly what one A-fragment
// read (ds_read_b128 at lane*16) wants, so the LDS image is a straight copy of global memory.
// Missing couts / cins are zero.  Split-operand convs repeat the block per K segment:
// [w_hi][w_hi][w_lo] (or [w_hi][w_lo]), w_lo = fp16(w - w_hi).
// ------------------------------------------------------------------------------------------
size_t conv_wpack_bytes_seg(int cin, int cout, int nseg) {
    const int ns = (cin + 15) / 16, ct = (cout + 31) / 32;
    return (size_t)nseg * ns * 9 * ct * 1024;
}
size_t conv_wpack_bytes(int cin, int cout) { return conv_wpack_bytes_seg(cin, cout, 1); }

void pack_conv_weights(const float* w, int cin, int cout, int nseg, void* dst_host, bool fold) {
    const int ns = (cin + 15) / 16, CT = (cout + 31) / 32;
    f16* d = (f16*)dst_host;
    if (fold) {
        for (int seg = 0; seg < nseg; ++seg)
            for (int s = 0; s < ns; ++s)
                for (int t = 0; t < 9; ++t)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int co = l & 31, ci = s * 16 + 8 * (l >> 5) + j;
                            f16 o = (f16)0.f;
                            if (ci < cin && co < cout) o = (f16)w[((size_t)co * cin + ci) * 9 + t];
                            else if (ci < cin && co >= 8 && co - 8 < cout) {
                                const float v = w[((size_t)(co - 8) * cin + ci) * 9 + t];
                                o = (f16)(v - (float)(f16)v);
                            }
                            *d++ = o;
                        }
        return;
    }
    for (int seg = 0; seg < nseg; ++seg) {
        const bool lo = (seg == nseg - 1) && nseg > 1;   // the last segment of a split conv carries w_lo
        for (int s = 0; s < ns; ++s)
            for (int t = 0; t < 9; ++t)
                for (int ct = 0; ct < CT; ++ct)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int co = ct * 32 + (l & 31);
                            const int ci = s * 16 + 8 * (l >> 5) + j;
                            float v = 0.f;
                            if (co < cout && ci < cin) v = w[((size_t)co * cin + ci) * 9 + t];
                            const f16 hi = (f16)v;
                            *d++ = lo ? (f16)(v - (float)hi) : hi;
                        }
    }
}

uint8_t f32_to_e4m3(float f) {
    if (f != f) return 0x7f;
    const uint8_t sign = (f < 0.f || (f == 0.f && 1.f / f < 0.f)) ? 0x80 : 0;
    const float a = fabsf(f);
    if (a >= 464.f) return sign | 0x7e;          // past the midpoint to the NaN code: saturate at 448
    if (a == 0.f) return sign;
    int e;
    (void)frexpf(a, &e);                         // a = m * 2^e, m in [0.5, 1)
    int E = e - 1;
    if (E < -6) {                                // subnormal: steps of 2^-9
        int q = (int)rintf(a * 512.f);           // ties to even (default rounding mode)
        return sign | (uint8_t)(q >= 8 ? 0x08 : q);
    }
    int q = (int)rintf(ldexpf(a, 3 - E));        // 8 .. 16
    if (q == 16) { q = 8; ++E; }
    if (E > 8 || (E == 8 && q > 14)) return sign | 0x7e;
    return sign | (uint8_t)(((E + 7) << 3) | (q - 8));
}

// w: [cout][cin][taps] fp32 (taps = 9: OIHW 3x3; taps = 4: the 2x2 sub-pixel kernels below)
// fold (cout <= 8): fp16 stages carry [w_hi at couts 0.., w_lo at couts 8..] and only part 0 of the e4m3 planes follows
static void pack_f8hp_taps(const float* w, int cin, int cout, int taps, void* dst_host, bool fold = false) {
    const int CT = (cout + 31) / 32, ns = (cin + 15) / 16;
    f16* d16 = (f16*)dst_host;                                          // stages 0..ns-1: fp16 w_hi, [s][t][ct][lane][8]
    for (int s = 0; s < ns; ++s)
        for (int t = 0; t < taps; ++t)
            for (int ct = 0; ct < CT; ++ct)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int co = ct * 32 + (l & 31), ci = s * 16 + 8 * (l >> 5) + j;
                        f16 o = (f16)0.f;
                        if (co < cout && ci < cin) o = (f16)w[((size_t)co * cin + ci) * taps + t];
                        else if (fold && ci < cin && co >= 8 && co - 8 < cout) {
                            const float v = w[((size_t)(co - 8) * cin + ci) * taps + t];
                            o = (f16)(v - (float)(f16)v);
                        }
                        *d16++ = o;
                    }
    uint8_t* d = (uint8_t*)d16;
#if S2SR_EXPERIMENTAL
    static const bool diag_no_wlo = [] { const char* e = getenv("S2SR_DIAG_NO_WLO"); return e && atoi(e) != 0; }();   // numerics diagnostic
#else
    const bool diag_no_wlo = false;
#endif
    for (int part = 0; part < (fold ? 1 : 2); ++part)                   // 0: w_hi (meets x_lo), 1: w_lo * 2^11 (meets x_hi)
        for (int pl = 0; pl < 2; ++pl)
            for (int t = 0; t < taps; ++t)
                for (int ct = 0; ct < CT; ++ct)
                    for (int h16 = 0; h16 < 2; ++h16)
                        for (int row = 0; row < 32; ++row)
                            for (int j = 0; j < 16; ++j) {
                                const int co = ct * 32 + row, ci = 32 * pl + 16 * h16 + j;
                                float v = 0.f;
                                if (co < cout && ci < cin) {
                                    const float x = w[((size_t)co * cin + ci) * taps + t];
                                    const float hi = (float)(f16)x;
                                    v = part == 0 ? hi : (diag_no_wlo ? 0.0f : (x - hi) * 2048.0f);
                                }
                                *d++ = f32_to_e4m3(v);
                            }
}

void pack_conv_weights_f8hp(const float* w, int cin, int cout, void* dst_host, bool fold) { pack_f8hp_taps(w, cin, cout, 9, dst_host, fold); }

size_t conv_wpack_bytes_phase(int cin, int cout) { return (size_t)2 * ((cin + 15) / 16) * 4 * (2 * ((cout + 31) / 32)) * 1024; }

// Sub-pixel kernels of "nearest-2x, then 3x3" (see the PH template parameter): for output parity
// (py, q) the taps that read the same source pixel are summed (in double), giving a 2x2 kernel on the
// source image; tap (a, b) sits on source pixel (y + py - 1 + a, x + q - 1 + b).  One pack per row parity:
// the two column parities are stacked as cout tiles [q = 0 couts | q = 1 couts] (cout padded to 32s).
void pack_conv_weights_phase_f8hp(const float* w, int cin, int cout, int py, void* dst_host) {
    const int cpad = ((cout + 31) / 32) * 32;
    std::vector<float> w4((size_t)2 * cpad * cin * 4, 0.f);
    auto grp = [](int parity, int a, int d) {   // does original tap index d (0..2) fall on source offset a (0..1)?
        const int src = (parity + d - 1 + 2) / 2 - 1;      // floor((parity + d - 1) / 2) for values >= -1
        return src == parity - 1 + a;
    };
    for (int q = 0; q < 2; ++q)
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) {
                        double acc = 0.0;
                        for (int dy = 0; dy < 3; ++dy)
                            for (int dx = 0; dx < 3; ++dx)
                                if (grp(py, a, dy) && grp(q, b, dx)) acc += (double)w[((size_t)co * cin + ci) * 9 + dy * 3 + dx];
                        w4[((size_t)(q * cpad + co) * cin + ci) * 4 + a * 2 + b] = (float)acc;
                    }
    pack_f8hp_taps(w4.data(), cin, 2 * cpad, 4, dst_host);
}

}  // namespace s2sr
