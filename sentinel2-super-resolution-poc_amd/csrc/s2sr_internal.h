// Internal declarations shared by the translation units of libs2sr.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/s2sr.h"

namespace s2sr {

// ------------------------------------------------------------------------------------------
// Activation planes.
//
// Every feature map lives in HBM as fp16 (or fp32) "NHWC with a physical zero halo":
//   element (n, y, x, c)  ->  ((n*Hp + y+1)*Wp + x+1) * C + c
// with Hp = roundup(H,32)+2, Wp = roundup(W,32)+2.  Kernels only ever store to pixels with
// y<H, x<W, so the halo (and the round-up slack) stays zero from the allocation-time memset:
// the 3x3 zero padding of the reference convs (cnn_super_resolution.py:78-82) and ragged
// tile edges then need no bounds checks on the load side, which is what lets the loader be
// a pure LDS-DMA stream.
// ------------------------------------------------------------------------------------------
static inline int roundup32(int v) { return (v + 31) & ~31; }
static inline int padded(int v) { return roundup32(v) + 2; }

enum Epilogue : int {
    EPI_LRELU = 0,      // y = lrelu(acc)                  -> fp16 plane slice   (RDB conv1..4, up1, up2, hr)
    EPI_RDB5 = 1,       // v = acc*0.2 + T ; T=v           -> fp16 X_next        (RDB conv5, rdb1/rdb2)
    EPI_RDB5_RRDB = 2,  // v = (acc*0.2+T)*0.2 + R; T=R=v  -> fp16 X_next        (RDB conv5 of rdb3)
    EPI_FIRST = 3,      // v = acc*in_scale + bias; F=T=R=v-> fp16 X             (conv_first)
    EPI_BODY = 4,       // v = F + acc                     -> fp16               (conv_body + trunk skip)
    EPI_LAST = 5,       // out fp32 NCHW and/or u8 NHWC (x255, clip, truncate)   (conv_last)
    EPI_DEBUG = 6,      // out fp32 NCHW, all Cout, optional lrelu               (s2sr_debug_conv)
};

struct ConvParams {
    const char* src0;        // plane feeding chunks [0, split)
    const char* src1;        // plane feeding chunks [split, nchunks)
    uint32_t rec0, rec1;     // bytes per pixel record of src0 / src1
    int32_t split, nchunks;  // 32-channel chunks
    const void* wpack;       // packed fp16 weights in MFMA A-fragment order (see pack_conv_weights)
    const float* bias;       // [CT*32] fp32, zero padded
    int32_t N, H, W;         // output logical dims (images in this launch)
    int32_t Hp, Wp;          // padded dims of output-resolution planes
    int32_t sHp, sWp;        // padded dims of the source planes (== Hp,Wp unless upsample-on-load)
    int32_t tilesX, tilesY;
    char* dst;               // fp16 destination plane
    uint32_t dst_rec;        // bytes per pixel record of dst
    uint32_t dst_coff;       // byte offset of this conv's first output channel inside the record
    float* T; float* R; float* F;   // fp32 64-channel planes (trunk, RRDB skip, global skip)
    float* out_f32;          // EPI_LAST / EPI_DEBUG: [N,Cout,H,W] fp32 (may be null)
    uint8_t* out_u8;         // EPI_LAST: [N,H,W,3] u8 (may be null)
    int32_t cout;            // real output channels (EPI_LAST: 3; EPI_DEBUG: Cout)
    int32_t act;             // EPI_DEBUG: apply lrelu
    float in_scale;          // EPI_FIRST: 1/255 (inputs are fed as exact integers 0..255)
    unsigned long long* trace;   // diagnostic builds only: s_memtime stamps, 24 per workgroup
};

// conv kernel launchers (conv_mfma.hip).  ct = ceil(Cout/32) in {1,2}.
hipError_t launch_conv_f16(const ConvParams& p, int ct, int epi, bool upsample, hipStream_t st);
hipError_t launch_conv_f16_trace(const ConvParams& p, int ct, hipStream_t st);   // diagnostic build with stamps
// version 2 (conv_ring.hip): persistent workgroups + LDS ring; its own weight layout
hipError_t launch_conv_f16_ring(const ConvParams& p, int ct, int epi, bool upsample, hipStream_t st);
void pack_conv_weights_ring(const float* w, int cin, int cout, float wscale, void* dst_host);
size_t conv_wpack_bytes(int cin, int cout);
// host-side repack: OIHW fp32 -> fp16 A-fragment order, zero padded to 32-multiples
void pack_conv_weights(const float* w, int cin, int cout, float wscale, void* dst_host);

// data-movement kernels (pack.hip)
hipError_t launch_pack_u8(const uint8_t* d_tiles, int N, int H, int W, char* plane, int Hp, int Wp, hipStream_t st);
hipError_t launch_pack_f32_nchw(const float* d_x, int N, int C, int H, int W, float scale, char* plane, int Cp,
                                int Hp, int Wp, hipStream_t st);
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

}  // namespace s2sr
