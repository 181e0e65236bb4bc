// Internal declarations shared by the translation units of libs2sr.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <mutex>
#include <vector>

#include "../../include/s2sr.h"

// Kernel families and forms that the measurements buried (profiles/r0x_*) are compiled only with -DS2SR_EXPERIMENTAL=1
// (`make EXP=1` -> csrc/libs2sr_exp.so, selected with S2SR_LIB=...): the row-Winograd trunk form (conv_wino.hip, +-0..+1.5 % in
// the net), the fp16 loader-wave form (+-0), the one-wave-per-SIMD tail convs (10-15 % slower), the 8-wave RDB path (r01's
// kernel), the non-default fp8 conv1-4 forms, the upsample-on-load up-convs, and every stamped (TRACE) build.  The default
// library answers requests for them with hipErrorNotSupported and ignores their environment switches.
#ifndef S2SR_EXPERIMENTAL
#define S2SR_EXPERIMENTAL 0
#endif

namespace s2sr {

// ------------------------------------------------------------------------------------------
// The device gate.  While a stream captures a hipGraph, the runtime refuses device-wide operations from EVERY thread of the
// process (hipDeviceSynchronize, hipFree, legacy-stream hipMemcpy / hipMemset ...: "operation not permitted when stream is
// capturing") and voids the capture with them.  Handles are independent objects used from different threads (the x4 and the
// anime-6B engines under Starlette's pool, reference main.py:602,670-675), so one handle's workspace regrow met another's capture
// (found by tools/soak_jobs.py, r04).  Rule of this library: no legacy-stream call anywhere (copy_blocking / fill_blocking in
// engine.hip run on the handle's own stream), and every device-wide call goes through the wrappers below, which hold the same
// process-wide mutex the capture section holds.  Captures are ~1 ms of host work and device-wide calls are rare (allocation,
// regrow, teardown), so the gate costs nothing in steady state.
// ------------------------------------------------------------------------------------------
std::recursive_mutex& device_gate();
struct DeviceGate {
    std::lock_guard<std::recursive_mutex> lk;
    DeviceGate() : lk(device_gate()) {}
};
static inline hipError_t dev_sync() { DeviceGate g; return hipDeviceSynchronize(); }
template <class T> static inline hipError_t dev_malloc(T** p, size_t bytes) { DeviceGate g; return hipMalloc((void**)p, bytes); }
static inline hipError_t dev_free(void* p) { DeviceGate g; return hipFree(p); }
static inline hipError_t host_malloc(void** p, size_t bytes, unsigned flags) { DeviceGate g; return hipHostMalloc(p, bytes, flags); }
static inline hipError_t host_free(void* p) { DeviceGate g; return hipHostFree(p); }

// ------------------------------------------------------------------------------------------
// Activation tensors in HBM: "blocked-16 with a physical zero halo".
//
//   fp16 tensor, C channels (C % 16 == 0):   [N][C/16][Hp][Wp][16]   -> 32 B per (block, pixel)
//   fp32 trunk tensors, 64 channels:         [N][8]   [Hp][Wp][8]    -> 32 B per (block, pixel)
//   element (n, y, x) of block b sits at ((n*NB + b)*Hp + y+1)*Wp + x+1  (units of 32 B)
//   with Hp = roundup(H,32)+2, Wp = roundup(W,32)+2.
//
// Why: the conv kernel consumes 16 input channels per pipeline stage.  With one plane per
// 16-channel block a slab row (34 pixels) is 1088 CONTIGUOUS bytes, so every LDS-DMA
// instruction (64 lanes x 16 B) reads 8 whole 128-B lines instead of 32 partial ones (the
// NHWC layout of the first version cost ~20 % in the loader and produced multi-microsecond
// stalls), and every epilogue store instruction writes 1 KiB contiguously.
// Kernels only store to pixels with y<H, x<W, so halo and round-up slack stay zero from the
// allocation-time memset: that IS the zero padding of the reference convs
// (cnn_super_resolution.py:78-82) and it removes every bounds check from the loader.
// The channel concatenation of the dense block (torch.cat, cnn_super_resolution.py:87-90) is
// just "the first k blocks of the dense tensor": [x(4 blocks) | x1(2) | x2(2) | x3(2) | x4(2)].
// ------------------------------------------------------------------------------------------
static inline int roundup32(int v) { return (v + 31) & ~31; }
static inline int padded(int v) { return roundup32(v) + 2; }

enum Epilogue : int {
    EPI_LRELU = 0,      // y = lrelu(acc)                  -> fp16 blocks        (RDB conv1..4, up1, up2, hr)
    EPI_RDB5 = 1,       // v = acc*0.2 + (x+lo)            -> fp16 x_next, lo    (RDB conv5, rdb1/rdb2)
    EPI_RDB5_RRDB = 2,  // v = (acc*0.2+(x+lo))*0.2 + R; R=v-> fp16 x_next, lo    (RDB conv5 of rdb3)
    EPI_FIRST = 3,      // v = acc*in_scale + bias; F=R=v  -> fp16 x, lo          (conv_first)
    EPI_BODY = 4,       // v = F + acc                     -> fp16               (conv_body + trunk skip)
    EPI_LAST = 5,       // out fp32 NCHW and/or u8 NHWC (x255, clip, truncate)   (conv_last)
    EPI_DEBUG = 6,      // out fp32 NCHW, all Cout, optional lrelu               (s2sr_debug_conv)
};

struct ConvParams {
    const char* src;         // first input block of image 0 (fp16 blocked tensor)
    uint64_t src_img;        // bytes between images of src
    int32_t nstage;          // pipeline stages per patch = nseg * seg_len
    // Split-operand ("hp") convs run the K loop over segments of seg_len stages that all accumulate
    // into the same fp32 accumulator.  Bit s of seg_lo_mask says segment s reads the src_lo tensor.
    //   conv_first: (x, w_hi) (x, w_lo), both fp16 (x is an exact integer);
    //   cin-64 convs (f8_in launch): 4 fp16 stages (x_hi, w_hi), then the 4 e4m3 planes of src_lo
    //   [x_lo*2^11 p0, p1, x_hi p0, p1] against [w_hi, w_hi, w_lo*2^11, w_lo*2^11] on the fp8 MFMA.
    // Plain convs: seg_len == nstage, mask 0.
    const char* src_lo;      // second operand tensor (same padded geometry as src: fp16 lo blocks or e4m3 planes), or null
    uint64_t lo_img;         // bytes between images of src_lo
    int32_t seg_len;
    int32_t seg_lo_mask;
    const void* wpack;       // packed fp16 weights (pack_conv_weights)
    const float* bias;       // [64] fp32, zero padded
    int32_t N, H, W;         // output logical dims (images in this launch)
    int32_t Hp, Wp;          // padded dims at output resolution
    int32_t sHp, sWp;        // padded dims of the source tensor (== Hp,Wp unless upsample-on-load)
    int32_t tilesX, tilesY;  // filled by the launcher
    char* dst;               // first OUTPUT block of image 0 (fp16 blocked tensor)
    uint64_t dst_img;        // bytes between images of dst
    char* T;                 // 'lo' OUTPUT tensor: fp16 blocked-16, 4 blocks, image stride 4 blocks.  conv_first / conv5:
                             // the trunk lo (trunk = x + lo); hp convs: the 4 e4m3 planes of their 64-channel output
    float* R; float* F;      // fp32 blocked-8 skip tensors (RRDB input, global skip), 8 blocks
    float* out_f32;          // EPI_LAST / EPI_DEBUG: [N,cout,H,W] fp32 (may be null)
    uint8_t* out_u8;         // EPI_LAST: [N,H,W,3] u8 (may be null)
    int32_t cout;            // real output channels (EPI_LAST: 3; EPI_DEBUG: Cout)
    int32_t act;             // EPI_DEBUG: apply lrelu
    int32_t fold_lo;         // EPI_LAST: couts 8..8+cout-1 carry w_lo (pack_conv_weights fold): out[c] = acc[c] + acc[8+c]
    float in_scale;          // EPI_FIRST: 1/255 (inputs are fed as exact integers 0..255)
    // fp8 trunk mode (S2SR_PREC_FP8, conv_trunk.hip conv_trunk_f8): src / dst are e4m3 PLANES of 32 channels (32 B per
    // pixel, the geometry of an fp16 block-16 plane); nstage = planes per patch padded to an even count, seg_len = real
    // planes (a phantom plane re-reads plane 0 against zero weights); the trunk itself is carried in fp16:
    const char* xh_in;       // conv5: trunk x (fp16 blocked-16, 4 blocks) -- the residual operand
    const char* xh_skip;     // conv5 of rdb3: the RRDB's input x (fp16, 4 blocks).  Also used by conv_trunk_f16 (fp16 modes):
                             // there the RRDB skip is the fp16 PAIR (xh_skip, lo_skip) of the trunk at the RRDB's input, read
                             // in place of the fp32 R tensor (-256 B per pixel: no R store, two 8-B halves instead of 16 B)
    const char* lo_skip;     // ... its lo half (fp16, 4 blocks, image stride as T)
    char* xh_out;            // conv5: new trunk x (fp16, 4 blocks); may alias xh_skip (same lane reads, then writes)
    uint64_t xh_img;         // bytes between images of the three
    const int32_t* wscale;   // [64] E8M0 bytes 127 - k_co of the per-output-channel weight scales 2^k_co
    int32_t x_exp, g_exp;    // activation scales: x planes hold e4m3(x * 2^x_exp), growth planes e4m3(x_k * 2^g_exp)
    int32_t f8_form;         // conv_trunk_f8 conv1-4 kernel form (diagnostics): bit 0 no loader wave, bits 1-2 weight placement (0 default, 1 all
                             // streamed, 2 all resident), bit 3 two waves per SIMD -- from S2SR_FP8_LOADER / _WSTREAM / _W8 at s2sr_create
    int32_t f16_form;        // conv_trunk_f16 conv1-4 kernel form: bit 0 = 32x32 patches with the load-only fifth wave (S2SR_F16_LOADER), bit 1 = single
                             // tiles keep the 16x32-patch form instead of 8x32 (S2SR_SMALL8=0)
    int32_t tail_form;       // split-operand head/tail convs (conv3x3.hip F8 schedule): bit 0 = one wave per SIMD (4 waves, twice the rows per wave; S2SR_TAIL_W4), bit 1 = this conv's consumer reads no e4m3(x_hi) planes (conv_hr before a folded conv_last): do not write them, bit 3 = never the whole-patch (FULL) forms (S2SR_F16_FULL=0)
    int32_t lo_exp;          // conv_trunk_f16 conv5: the trunk's lo half is stored as e4m3(lo * 2^lo_exp) planes (xh_in, T, lo_skip)
    // Window mosaics (the AOI path, engine.hip forward_dev): equal-size windows of `_tile_process` (cnn_super_resolution.py:249-257) laid
    // out on a grid inside ONE image with a single zero row / column between neighbours -- the conv zero padding of both, at
    // every layer, because no launch ever stores to a separator pixel: at the scale of the coordinates the epilogue works in,
    // (y, x) is live iff y % mos_py < mos_ry and x % mos_px < mos_rx (period = window + 1, real extent = window).  0 = off.
    // 276-pixel windows then cost 280 x 280 of patch area instead of 288 x 288.
    int32_t mos_py, mos_ry, mos_px, mos_rx;
    uint32_t mos_my, mos_mx;             // floor(2^32 / mos_py) + 1, floor(2^32 / mos_px) + 1 (patch_live); 0: take the modulo route
    int32_t mos_kx, mos_ky, mos_count;   // EPI_LAST: grid of a mosaic and the number of windows in this launch: window t = (n * ky + wy) * kx + wx
                                         // is written as image t of [count, ry, rx]; slots past the count are not written
    char* trash;             // >= 4 KiB scratch: out-of-image lanes park their (unconditional) stores here
    unsigned long long* trace;   // diagnostic build only: s_memtime stamps, 24 per workgroup
    int32_t dbg;                 // diagnostic only (timing ablations, results wrong): 1 weights DMA from one fixed piece,
                                 // 2 slab DMA from one fixed piece, 4 per-wave stamps, 8 no DMA instructions in the loop
};

// is output pixel (y, x) of this launch one the reference writes (inside the image, and not a mosaic separator)?
__device__ __forceinline__ bool px_live(const ConvParams& p, int y, int x) {
    bool ok = (y < p.H) && (x < p.W);
    if (p.mos_py) ok = ok && (y % p.mos_py < p.mos_ry) && (x % p.mos_px < p.mos_rx);
    return ok;
}
// The same predicate for the pixels of ONE patch (first row y0, first column x0, at most 32 x 32) without a division per row
// and lane: the residues of y0 and x0 are taken once per patch (wave-uniform; multiply-high by the host's 2^32 / period, exact
// for coordinates below 65536), a pixel's residue is that plus its offset, wrapped at most once (periods of 32 and more;
// shorter periods -- toy windows in tests -- take the modulo route).
struct PatchLive { int ry0, rx0; bool fast; };
__device__ __forceinline__ PatchLive patch_live(const ConvParams& p, int y0, int x0) {
    PatchLive L{0, 0, false};
    if (p.mos_py && p.mos_py >= 32 && p.mos_px >= 32 && p.mos_my) {
        L.fast = true;
        L.ry0 = y0 - (int)__umulhi((uint32_t)y0, (uint32_t)p.mos_my) * p.mos_py;
        L.rx0 = x0 - (int)__umulhi((uint32_t)x0, (uint32_t)p.mos_mx) * p.mos_px;
    }
    return L;
}
__device__ __forceinline__ bool px_live(const ConvParams& p, const PatchLive& L, int y0, int x0, int y, int x) {
    bool ok = (y < p.H) && (x < p.W);
    if (p.mos_py) {
        if (L.fast) {
            int a = L.ry0 + (y - y0), b = L.rx0 + (x - x0);
            a = a >= p.mos_py ? a - p.mos_py : a;
            b = b >= p.mos_px ? b - p.mos_px : b;
            ok = ok && a < p.mos_ry && b < p.mos_rx;
        } else {
            ok = ok && (y % p.mos_py < p.mos_ry) && (x % p.mos_px < p.mos_rx);
        }
    }
    return ok;
}

// conv kernel (conv3x3.hip).  ct = ceil(Cout/32) in {1,2}.
hipError_t launch_conv(const ConvParams& p, int ct, int epi, bool upsample, bool lo_out, hipStream_t st, bool f8_in = false);
hipError_t launch_conv_trace(const ConvParams& p, int ct, hipStream_t st);   // stamped diagnostic build
// the RRDB trunk convs as one-wave-per-SIMD workgroups (conv_trunk.hip): ct 1 + EPI_LRELU (conv1..4), ct 2 + EPI_RDB5 /
// EPI_RDB5_RRDB (conv5).  hipErrorNotSupported = not a trunk form / launch too small: use launch_conv.
// force_form (conv1-4 only; the per-layer parity hook): 0 = by launch size, 1 = 16x32 patches / 5-deep ring, 2 = 32x32 patches / 3-deep ring,
// 4 = loader wave, 5 = 8x32 patches, 6 / 7 / 8 = the whole-patch (FULL) forms of 2 / 1 / 5
hipError_t launch_conv_trunk(const ConvParams& p, int ct, int epi, hipStream_t st, bool trace = false, int force_form = 0);
// fp16 RDB conv1..4 in the row-Winograd F(2,3) form (conv_wino.hip): weights transformed over dy (4 x 3 fragments per 16-channel
// stage instead of 3 x 3), inputs transformed over 4 consecutive slab rows in registers, 12 MFMAs per 2 output rows instead of 18
hipError_t launch_conv_trunk_wino(const ConvParams& p, hipStream_t st);
size_t conv_wpack_bytes_wino(int cin, int cout);
hipError_t launch_pack_trunk_wino(const float* d_w, int cin, int cout, void* d_out, hipStream_t st);
// the same convs on e4m3 operands, block-scaled fp8 MFMA (K = 64 = two planes per instruction)
hipError_t launch_conv_trunk_f8(const ConvParams& p, int ct, int epi, hipStream_t st);
// per-plane weight stages for conv_trunk_f8: [plane][tap][ct][16-B half][cout row 0..31][16 channel bytes] e4m3 of
// w * 2^k_co, planes padded to an even count with a zero plane; wscale_out[co] = 127 - k_co
size_t conv_wpack_bytes_f8(int cin, int cout);
void pack_conv_weights_f8(const float* w, int cin, int cout, void* dst_host, int32_t* wscale_out /*[64]*/);
// device-side repack of the RDB convs' weights (pack.hip) and gather of all biases into [nconv][64]
hipError_t launch_pack_trunk_f16(const float* d_w, int cin, int cout, void* d_out, hipStream_t st);
hipError_t launch_pack_trunk_f8(const float* d_w, int cin, int cout, void* d_out, int32_t* d_wscale /*[64]*/, hipStream_t st);
hipError_t launch_gather_bias(const float* d_blob, const uint64_t* d_off, const int32_t* d_cout, int nconv, float* d_out, hipStream_t st);
hipError_t launch_absmax_f16(const void* d, size_t n_halves, float* d_out, hipStream_t st);               // fp8 calibration
hipError_t launch_absmax_e4m3(const void* d, size_t n_bytes, int exp2, float* d_out, hipStream_t st);
hipError_t launch_xh_to_fp8(const char* xh, size_t xh_img, int N, int Hp, int Wp, int x_exp, char* out, size_t out_img, hipStream_t st);
size_t conv_wpack_bytes(int cin, int cout);
// host-side repack: OIHW fp32 -> fp16 A-fragment order [stage][tap][ct][lane][8]
// nseg 1: [w_hi]; 2: [w_hi][w_lo] (exact-integer inputs: conv_first); 3: [w_hi][w_hi][w_lo]
// fold (cout <= 8 only): every segment is [w_hi at couts 0.., w_lo at couts 8..] so a conv with idle
// output channels gets x*w_hi and x*w_lo from one pass over x (conv_last in hp mode: 2 segments, x_hi and x_lo)
void pack_conv_weights(const float* w, int cin, int cout, int nseg, void* dst_host, bool fold = false);
size_t conv_wpack_bytes_seg(int cin, int cout, int nseg);
// split-operand convs with fp8 correction terms (cin == 64): 4 fp16 stages [w_hi] + 4 e4m3 stages
// [w_hi ch 0-31][w_hi ch 32-63][w_lo*2^11 ch 0-31][w_lo*2^11 ch 32-63]; an fp8 stage fragment is
// [tap][ct][16-B half][cout row 0..31][16 channel bytes].  Same size as nseg = 2.
// fold (cout <= 8, conv_last): 4 fp16 stages [w_hi at couts 0.. | w_lo at couts 8..] + the two e4m3 w_hi planes (6 stages)
void pack_conv_weights_f8hp(const float* w, int cin, int cout, void* dst_host, bool fold = false);
// sub-pixel form of the split-operand up-convs: four 2x2-tap kernels (one per output parity), same stage layout
void pack_conv_weights_phase_f8hp(const float* w, int cin, int cout, int phase, void* dst_host);
size_t conv_wpack_bytes_phase(int cin, int cout);
hipError_t launch_conv_phase(const ConvParams& p, int row_parity, hipStream_t st, bool f8);
uint8_t f32_to_e4m3(float f);   // OCP e4m3fn, round to nearest even, saturating at +-448

// XYZ tile pyramid (tiles.hip)
hipError_t launch_warp_bilinear(const uint8_t* d_rgb, int H, int W, const float* d_grid, int gh, int gw, int step, int OH, int OW,
                                uint8_t* d_out, hipStream_t st);
hipError_t launch_tiles_base(const uint8_t* d_rgba, int W, const int32_t* d_col_lo, const int32_t* d_col_hi, const int32_t* d_row_lo,
                             const int32_t* d_row_hi, int nx, int ny, uint8_t* d_out, hipStream_t st);
hipError_t launch_tiles_overview(const uint8_t* d_child, int cnx, int cny, int ox, int oy, int pnx, int pny, uint8_t* d_out,
                                 hipStream_t st);

// PNG encoding of a tile level on the device (pngdev.hip): stats kernel -> host plan (Huffman codes, sizes) -> emit kernel.
struct PngTilePlan {
    std::vector<uint8_t> mode;            // 0 nothing to write, 1 device stream, 2 host encoder (stored blocks are smaller)
    std::vector<uint32_t> adler, deflate_bytes, eob;
    std::vector<uint64_t> eob_at, out_word;
    // what the emit kernel reads: [n][512] token table, [n][160] block header words, [n] TileMeta (pngdev.hip) -- in this order in
    // ONE block of upload_bytes: the caller's page-locked arena when it gave one (arena / arena_bytes), else `own`
    uint32_t* tb = nullptr;
    uint32_t* hdr = nullptr;
    uint8_t* meta = nullptr;
    size_t upload_bytes = 0;
    void* arena = nullptr;
    size_t arena_bytes = 0;
    std::vector<uint8_t> own;
    bool failed = false;                  // a planning thread threw (out of memory): the plan is incomplete
};
hipError_t launch_png_tile_stats(const uint8_t* d_tiles, int ntiles, uint32_t* d_hist, uint32_t* d_adler, uint32_t* d_flags, bool row_threads,
                                 hipStream_t st);
hipError_t launch_png_tile_emit(const uint8_t* d_tiles, int ntiles, const void* d_meta, const uint32_t* d_tb, const uint32_t* d_hdr,
                                uint32_t* d_out, bool row_threads, hipStream_t st);
size_t png_plan_bytes(int n);      // bytes of the upload block of n tiles
size_t png_plan_tiles(int n, const uint32_t* hist, const uint32_t* adler_rows, const uint32_t* flags, const char* const* paths,
                      bool skip_transparent, bool force_host, PngTilePlan* plan);
bool png_write_tile_file(const char* path, const uint32_t* words, uint32_t deflate_bytes, uint32_t eob, uint64_t eob_at, uint32_t adler,
                         std::vector<uint8_t>& buf);
bool png_parallel_for(int n, const std::function<void(int)>& body);   // false: a body threw (the other indices still ran)

// data-movement kernels (pack.hip)
hipError_t launch_pack_u8(const uint8_t* d_tiles, int N, int H, int W, char* blk, int Hp, int Wp, hipStream_t st);
// the same for a window mosaic: B windows of h x w into ceil(B / (kx*ky)) images of (ky*(h+1)-1) x (kx*(w+1)-1), window t at grid
// cell (t % (kx*ky)) / kx, % kx of image t / (kx*ky); separator rows / columns are never written
hipError_t launch_pack_u8_mosaic(const uint8_t* d_tiles, int B, int h, int w, int kx, int ky, char* blk, int Hp, int Wp, hipStream_t st);
hipError_t launch_pack_f32_nchw(const float* d_x, int N, int C, int H, int W, float scale, char* blk, int NB,
                                int Hp, int Wp, hipStream_t st);
hipError_t launch_trunk_to_fp8(const char* hi, size_t hi_img, const char* lo, size_t lo_img, int lo_e4m3_exp, int N, int Hp, int Wp, char* out,
                               hipStream_t st);
hipError_t launch_swap_rb_u8(const uint8_t* d_in, size_t npx, uint8_t* d_out, hipStream_t st);
hipError_t launch_gather_windows(const uint8_t* d_img, int H, int W, const int32_t* d_rects, int T, int wh, int ww,
                                 uint8_t* d_tiles, hipStream_t st);
hipError_t launch_stitch_u8(const uint8_t* d_tiles, int tilesX, int oth, int otw, const int32_t* d_rowmap,
                            const int32_t* d_colmap, int OH, int OW, uint8_t* d_out, hipStream_t st);
hipError_t launch_stitch_f32(const float* d_tiles /*[T,3,oth,otw]*/, int tilesX, int oth, int otw, const int32_t* d_rowmap,
                             const int32_t* d_colmap, int OH, int OW, float* d_out /*HWC*/, hipStream_t st);

// post-process kernels (postprocess.hip)
hipError_t launch_postprocess(const uint8_t* d_rgb, int B, int H, int W, const s2sr_pp_params& prm, uint8_t* d_out,
                              void* d_work, size_t work_bytes, hipStream_t st);
size_t postprocess_work_bytes(int B, int H, int W, const s2sr_pp_params& prm);
// ... over ONE image in row bands (an AOI's mosaic arrives band by band; CLAHE's grid is image-global): histograms as the bands
// arrive, LUTs once, then apply (R rows ahead) + sharpen band by band.  d_work = postprocess_work_bytes(1, H, W, prm) bytes; bgr:
// the image's bytes are B,G,R; swap_out: R and B exchanged in the rows written.  Same bytes as launch_postprocess.
int pp_band_radius(const s2sr_pp_params& prm);
hipError_t launch_pp_band_begin(int H, int W, const s2sr_pp_params& prm, void* d_work, hipStream_t st);
hipError_t launch_pp_band_hist(const uint8_t* d_img, int H, int W, const s2sr_pp_params& prm, int bgr, int y0, int y1, void* d_work,
                               hipStream_t st);
hipError_t launch_pp_band_lut(int H, int W, const s2sr_pp_params& prm, void* d_work, hipStream_t st);
hipError_t launch_pp_band_apply(const uint8_t* d_img, int H, int W, const s2sr_pp_params& prm, int bgr, int y0, int y1, void* d_work,
                                hipStream_t st);
hipError_t launch_pp_band_sharpen(int H, int W, const s2sr_pp_params& prm, int bgr, int swap_out, int y0, int y1, void* d_work,
                                  uint8_t* d_out, hipStream_t st);

// measured MFMA ceilings (ceiling.hip): bare / LDS-fed / LDS-DMA-fed fp16 32x32x16 loops at conv_trunk_f16's operand traffic
hipError_t launch_mfma_ceiling(int mode, char* d_src, size_t src_bytes, bool fill, float* d_sink, int grid, int stages, char* d_store,
                               size_t store_bytes, hipStream_t st);
double mfma_ceiling_flop_per_launch(int mode, int grid, int stages);
double mfma_ceiling_dma_bytes_per_launch(int mode, int grid, int stages);

// diagnostic prototype (persist.hip): an RDB-shaped loop whose workgroups stay across layers, planes handed over through flags
hipError_t launch_rdb_persistent(int variant, const char* d_wts, size_t wts_bytes, char* d_ws, uint32_t* d_flags, float* d_sink, int grid, int P,
                                 int rdbs, uint32_t* d_timeouts, hipStream_t st);
size_t rdb_persistent_ws_bytes(int variant, int grid, int P);
double rdb_persistent_flop_per_launch(int grid, int P, int rdbs);

}  // namespace s2sr
