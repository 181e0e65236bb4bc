// Data-movement kernels around the conv stack: HBM-bound byte work, one element per thread,
// coalesced on the side that moves the most bytes.
//   pack_u8            : [N,H,W,3] u8 -> one fp16 blocked-16 plane with zero halo; replaces
//                        `img.astype(float32)/255` + permute (cnn_super_resolution.py:220-222);
//                        values stay the exact integers 0..255, the 1/255 lives in conv_first.
//   gather_windows     : cut the _tile_process windows (cnn_super_resolution.py:249-256)
//   stitch_*           : crop + paste with the reference's overwrite order (:259-278)
#include "s2sr_internal.h"

namespace s2sr {

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));

__global__ void pack_u8_kernel(const uint8_t* __restrict__ in, int N, int H, int W, char* __restrict__ blk, int Hp,
                               int Wp) {
    const size_t total = (size_t)N * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const size_t r = i / W;
        const int y = (int)(r % H);
        const int n = (int)(r / H);
        const uint8_t* s = in + i * 3;
        f16x4 v;
        v[0] = (f16)(float)s[0];
        v[1] = (f16)(float)s[1];
        v[2] = (f16)(float)s[2];
        v[3] = (f16)0.f;
        // one 16-channel block per image; channels 4..15 stay zero from the allocation memset
        *(f16x4*)(blk + (((size_t)n * Hp + y + 1) * Wp + x + 1) * 32) = v;
    }
}

hipError_t launch_pack_u8(const uint8_t* d_tiles, int N, int H, int W, char* blk, int Hp, int Wp, hipStream_t st) {
    const size_t total = (size_t)N * H * W;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_u8_kernel, dim3(grid), dim3(256), 0, st, d_tiles, N, H, W, blk, Hp, Wp);
    return hipGetLastError();
}

__global__ void pack_u8_mosaic_kernel(const uint8_t* __restrict__ in, int B, int h, int w, int kx, int ky, char* __restrict__ blk, int Hp,
                                      int Wp) {
    const size_t total = (size_t)B * h * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lx = (int)(i % w);
        const size_t r = i / w;
        const int ly = (int)(r % h);
        const int t = (int)(r / h);
        const int n = t / (kx * ky), slot = t - n * (kx * ky);
        const int wy = slot / kx, wx = slot - wy * kx;
        const int y = wy * (h + 1) + ly, x = wx * (w + 1) + lx;
        const uint8_t* s = in + i * 3;
        f16x4 v;
        v[0] = (f16)(float)s[0];
        v[1] = (f16)(float)s[1];
        v[2] = (f16)(float)s[2];
        v[3] = (f16)0.f;
        *(f16x4*)(blk + (((size_t)n * Hp + y + 1) * Wp + x + 1) * 32) = v;
    }
}

hipError_t launch_pack_u8_mosaic(const uint8_t* d_tiles, int B, int h, int w, int kx, int ky, char* blk, int Hp, int Wp, hipStream_t st) {
    const size_t total = (size_t)B * h * w;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_u8_mosaic_kernel, dim3(grid), dim3(256), 0, st, d_tiles, B, h, w, kx, ky, blk, Hp, Wp);
    return hipGetLastError();
}

__global__ void pack_f32_nchw_kernel(const float* __restrict__ x, int N, int C, int H, int W, float scale,
                                     char* __restrict__ blk, int NB, int Hp, int Wp) {
    const size_t total = (size_t)N * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        size_t r = i / W;
        const int y = (int)(r % H);
        r /= H;
        const int c = (int)(r % C);
        const int n = (int)(r / C);
        f16* d = (f16*)(blk + ((((size_t)n * NB + (c >> 4)) * Hp + y + 1) * Wp + xx + 1) * 32) + (c & 15);
        *d = (f16)(x[i] * scale);
    }
}

hipError_t launch_pack_f32_nchw(const float* d_x, int N, int C, int H, int W, float scale, char* blk, int NB, int Hp,
                                int Wp, hipStream_t st) {
    const size_t total = (size_t)N * C * H * W;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_f32_nchw_kernel, dim3(grid), dim3(256), 0, st, d_x, N, C, H, W, scale, blk, NB, Hp, Wp);
    return hipGetLastError();
}

// conv_body's correction operands (S2SR_PREC_F16_HP): the trunk arrives as an fp16 pair (hi = dense
// blocks 0..3, lo = the trunk-lo tensor); the split-operand kernel wants e4m3 planes of 32 channels
// [lo*2^11 plane 0, plane 1, hi plane 0, plane 1] in the same padded geometry (halo pixels are zero
// in both inputs, so the whole padded tensor is converted).  One thread = one pixel of one plane.
// lo_e4m3_exp >= 0: the lo half arrives as e4m3(lo * 2^lo_e4m3_exp) planes already (the one-wave-per-SIMD trunk keeps it so);
// it is rescaled to the 2^11 the split-operand kernel expects.
__global__ void trunk_to_fp8_kernel(const char* __restrict__ hi, size_t hi_img, const char* __restrict__ lo, size_t lo_img,
                                    int lo_e4m3_exp, int N, size_t ppx, char* __restrict__ out) {
    const size_t total = (size_t)N * 4 * ppx;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i % ppx;
        const int plane = (int)((i / ppx) & 3);
        const int n = (int)(i / (4 * ppx));
        const bool is_hi = plane >= 2;
        if (!is_hi && lo_e4m3_exp >= 0) {
            const uint32_t* v = (const uint32_t*)(lo + (size_t)n * lo_img + (size_t)plane * ppx * 32 + pix * 32);
            const float rs = ldexpf(1.0f, 11 - lo_e4m3_exp);
            uint32_t o8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int w0 = (int)v[q];
                const float f0 = __builtin_amdgcn_fmed3f(__builtin_amdgcn_cvt_f32_fp8(w0, 0) * rs, -448.0f, 448.0f);
                const float f1 = __builtin_amdgcn_fmed3f(__builtin_amdgcn_cvt_f32_fp8(w0, 1) * rs, -448.0f, 448.0f);
                const float f2 = __builtin_amdgcn_fmed3f(__builtin_amdgcn_cvt_f32_fp8(w0, 2) * rs, -448.0f, 448.0f);
                const float f3 = __builtin_amdgcn_fmed3f(__builtin_amdgcn_cvt_f32_fp8(w0, 3) * rs, -448.0f, 448.0f);
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(f2, f3, w, true);
                o8[q] = (uint32_t)w;
            }
            uint4* d8 = (uint4*)(out + ((size_t)n * 4 + plane) * ppx * 32 + pix * 32);
            d8[0] = make_uint4(o8[0], o8[1], o8[2], o8[3]);
            d8[1] = make_uint4(o8[4], o8[5], o8[6], o8[7]);
            continue;
        }
        const char* src = (is_hi ? hi + (size_t)n * hi_img : lo + (size_t)n * lo_img) + (size_t)(2 * (plane & 1)) * ppx * 32 + pix * 32;
        const float scale = is_hi ? 1.0f : 2048.0f;
        uint32_t o[8];
#pragma unroll
        for (int b = 0; b < 2; ++b) {   // two fp16 blocks of 16 channels -> 32 e4m3 bytes
            const f16* v = (const f16*)(src + (size_t)b * ppx * 32);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float f[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) f[k] = __builtin_amdgcn_fmed3f((float)v[4 * q + k] * scale, -448.0f, 448.0f);
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
                o[4 * b + q] = (uint32_t)w;
            }
        }
        uint4* d = (uint4*)(out + ((size_t)n * 4 + plane) * ppx * 32 + pix * 32);
        d[0] = make_uint4(o[0], o[1], o[2], o[3]);
        d[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

hipError_t launch_trunk_to_fp8(const char* hi, size_t hi_img, const char* lo, size_t lo_img, int lo_e4m3_exp, int N, int Hp, int Wp, char* out,
                               hipStream_t st) {
    const size_t ppx = (size_t)Hp * Wp, total = (size_t)N * 4 * ppx;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(trunk_to_fp8_kernel, dim3(grid), dim3(256), 0, st, hi, hi_img, lo, lo_img, lo_e4m3_exp, N, ppx, out);
    return hipGetLastError();
}

// fp8 trunk mode: the trunk x (fp16, 4 blocks of 16 channels) -> the two e4m3 planes e4m3(x * 2^x_exp) the RDB convs
// read (whole padded tensor: halo zeros stay zeros).  One thread = one pixel of one plane.
__global__ void xh_to_fp8_kernel(const char* __restrict__ xh, size_t xh_img, int N, size_t ppx, float scale, char* __restrict__ out,
                                 size_t out_img) {
    const size_t total = (size_t)N * 2 * ppx;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = i % ppx;
        const int plane = (int)((i / ppx) & 1);
        const int n = (int)(i / (2 * ppx));
        const char* src = xh + (size_t)n * xh_img + (size_t)(2 * plane) * ppx * 32 + pix * 32;
        uint32_t o[8];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f16* v = (const f16*)(src + (size_t)b * ppx * 32);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float f[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) f[k] = __builtin_amdgcn_fmed3f((float)v[4 * q + k] * scale, -448.0f, 448.0f);
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
                o[4 * b + q] = (uint32_t)w;
            }
        }
        uint4* d = (uint4*)(out + (size_t)n * out_img + (size_t)plane * ppx * 32 + pix * 32);
        d[0] = make_uint4(o[0], o[1], o[2], o[3]);
        d[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

hipError_t launch_xh_to_fp8(const char* xh, size_t xh_img, int N, int Hp, int Wp, int x_exp, char* out, size_t out_img, hipStream_t st) {
    const size_t ppx = (size_t)Hp * Wp, total = (size_t)N * 2 * ppx;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(xh_to_fp8_kernel, dim3(grid), dim3(256), 0, st, xh, xh_img, N, ppx, ldexpf(1.0f, x_exp), out, out_img);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Device-side weight repack for the 345 RDB convs (the receive buffer of the RCCL weight broadcast goes straight into MFMA
// fragment order: no host copy of the 67 MB blob).  Same layouts, same roundings as the host packers in conv3x3.hip /
// conv_trunk.hip (pack_conv_weights nseg = 1, pack_conv_weights_f8), which stay for the six head/tail convs and the test hooks.
// ------------------------------------------------------------------------------------------
// fp16 trunk: out[stage][tap][ct][lane][8] = fp16(W[co = ct*32 + (lane&31)][ci = stage*16 + 8*(lane>>5) + j][tap])
__global__ void pack_trunk_f16_kernel(const float* __restrict__ w, int cin, int cout, int ns, int CT, f16* __restrict__ out) {
    const size_t total = (size_t)ns * 9 * CT * 64 * 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), l = (int)((i >> 3) & 63);
        size_t r = i >> 9;
        const int ct = (int)(r % CT); r /= CT;
        const int t = (int)(r % 9);
        const int st = (int)(r / 9);
        const int co = ct * 32 + (l & 31), ci = st * 16 + 8 * (l >> 5) + j;
        out[i] = (co < cout && ci < cin) ? (f16)w[((size_t)co * cin + ci) * 9 + t] : (f16)0.f;
    }
}

hipError_t launch_pack_trunk_f16(const float* d_w, int cin, int cout, void* d_out, hipStream_t st) {
    const int ns = (cin + 15) / 16, CT = (cout + 31) / 32;
    const size_t total = (size_t)ns * 9 * CT * 512;
    hipLaunchKernelGGL(pack_trunk_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_w, cin, cout, ns, CT, (f16*)d_out);
    return hipGetLastError();
}

// fp8 trunk, pass 1: per output channel the largest k with max|w_co| * 2^k < 448; wscale[co] = 127 - k (64 entries)
__global__ void f8_scale_kernel(const float* __restrict__ w, int cin, int cout, int32_t* __restrict__ wscale) {
    __shared__ float red[256];
    const int co = blockIdx.x;
    float m = 0.f;
    if (co < cout)
        for (int i = threadIdx.x; i < cin * 9; i += 256) m = fmaxf(m, fabsf(w[(size_t)co * cin * 9 + i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s2]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int k = 0;
        m = red[0];
        if (m > 0.f) {
            int e;
            const float f = frexpf(m, &e);            // m = f * 2^e, f in [0.5, 1); 448 = 0.875 * 2^9
            k = (f < 0.875f ? 9 : 8) - e;
        }
        k = k > 100 ? 100 : (k < -100 ? -100 : k);
        wscale[co] = 127 - k;
    }
}

// pass 2: out[plane][tap][ct][16-B half][cout row][16] = e4m3(W * 2^k_co), phantom plane (odd plane counts) all zero
__global__ void pack_trunk_f8_kernel(const float* __restrict__ w, int cin, int cout, int nreal, int npad, int CT,
                                     const int32_t* __restrict__ wscale, uint8_t* __restrict__ out) {
    const size_t total4 = (size_t)npad * 9 * CT * 256;          // groups of 4 bytes
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int j4 = (int)(i & 3), row = (int)((i >> 2) & 31), h16 = (int)((i >> 7) & 1);
        size_t r = i >> 8;
        const int ct = (int)(r % CT); r /= CT;
        const int t = (int)(r % 9);
        const int pl = (int)(r / 9);
        const int co = ct * 32 + row;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (pl < nreal && co < cout) {
            const float sc = ldexpf(1.0f, 127 - wscale[co]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ci = 32 * pl + 16 * h16 + 4 * j4 + q;
                if (ci < cin) v[q] = w[((size_t)co * cin + ci) * 9 + t] * sc;     // |v| < 448 by construction of k_co
            }
        }
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
        ((uint32_t*)out)[i] = (uint32_t)pk;
    }
}

hipError_t launch_pack_trunk_f8(const float* d_w, int cin, int cout, void* d_out, int32_t* d_wscale, hipStream_t st) {
    const int nreal = (cin + 31) / 32, npad = (nreal + 1) & ~1, CT = (cout + 31) / 32;
    hipLaunchKernelGGL(f8_scale_kernel, dim3(64), dim3(256), 0, st, d_w, cin, cout, d_wscale);
    const size_t total4 = (size_t)npad * 9 * CT * 256;
    hipLaunchKernelGGL(pack_trunk_f8_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, d_w, cin, cout, nreal, npad, CT, d_wscale,
                       (uint8_t*)d_out);
    return hipGetLastError();
}

// biases of all convs: blob offsets -> [nconv][64] fp32, zero padded
__global__ void gather_bias_kernel(const float* __restrict__ blob, const uint64_t* __restrict__ off, const int32_t* __restrict__ cout,
                                   int nconv, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nconv * 64) return;
    const int c = i >> 6, k = i & 63;
    out[i] = k < cout[c] ? blob[off[c] + k] : 0.f;
}

hipError_t launch_gather_bias(const float* d_blob, const uint64_t* d_off, const int32_t* d_cout, int nconv, float* d_out, hipStream_t st) {
    hipLaunchKernelGGL(gather_bias_kernel, dim3((nconv * 64 + 255) / 256), dim3(256), 0, st, d_blob, d_off, d_cout, nconv, d_out);
    return hipGetLastError();
}

// fp8 calibration: running max |v| of an fp16 blocked tensor / of an e4m3 plane tensor (whole padded extent: halos are
// zero) into one float (bit pattern compared as unsigned: valid for non-negative floats)
__global__ void absmax_f16_kernel(const f16* __restrict__ v, size_t n, float* __restrict__ out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = fabsf((float)v[i]);
        m = (a == a && a > m) ? a : m;
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned int*)out, __float_as_uint(m));
}
__global__ void absmax_e4m3_kernel(const uint8_t* __restrict__ v, size_t n, float scale_inv, float* __restrict__ out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = v[i] & 0x7f;
        if (b == 0x7f) continue;                                  // NaN code (never stored: producers clamp)
        const int e = b >> 3, mm = b & 7;
        const float a = e == 0 ? (float)mm * 0.001953125f : ldexpf((float)(8 + mm), e - 10);
        m = fmaxf(m, a);
    }
    m *= scale_inv;
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned int*)out, __float_as_uint(m));
}
hipError_t launch_absmax_f16(const void* d, size_t n_halves, float* d_out, hipStream_t st) {
    hipLaunchKernelGGL(absmax_f16_kernel, dim3(1024), dim3(256), 0, st, (const f16*)d, n_halves, d_out);
    return hipGetLastError();
}
hipError_t launch_absmax_e4m3(const void* d, size_t n_bytes, int exp2, float* d_out, hipStream_t st) {
    hipLaunchKernelGGL(absmax_e4m3_kernel, dim3(1024), dim3(256), 0, st, (const uint8_t*)d, n_bytes, ldexpf(1.0f, -exp2), d_out);
    return hipGetLastError();
}

// R <-> B of a u8 HWC image, in place or into another buffer: the reference feeds the net BGR and turns its output back
// (cv2.cvtColor RGB2BGR / BGR2RGB, wow_sr.py:85,103)
__global__ void swap_rb_kernel(const uint8_t* __restrict__ in, size_t npx, uint8_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (size_t)gridDim.x * blockDim.x) {
        const uint8_t r = in[3 * i], g = in[3 * i + 1], b = in[3 * i + 2];
        out[3 * i] = b; out[3 * i + 1] = g; out[3 * i + 2] = r;
    }
}

hipError_t launch_swap_rb_u8(const uint8_t* d_in, size_t npx, uint8_t* d_out, hipStream_t st) {
    const int grid = (int)((npx + 255) / 256 > 8192 ? 8192 : (npx + 255) / 256);
    hipLaunchKernelGGL(swap_rb_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, d_in, npx, d_out);
    return hipGetLastError();
}

__global__ void gather_windows_kernel(const uint8_t* __restrict__ img, int H, int W, const int32_t* __restrict__ rects,
                                      int T, int wh, int ww, uint8_t* __restrict__ tiles) {
    const size_t total = (size_t)T * wh * ww * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        size_t r = i / 3;
        const int x = (int)(r % ww);
        r /= ww;
        const int y = (int)(r % wh);
        const int t = (int)(r / wh);
        const int y1 = rects[t * 4 + 0], x1 = rects[t * 4 + 2];
        tiles[i] = img[((size_t)(y1 + y) * W + (x1 + x)) * 3 + c];
    }
}

hipError_t launch_gather_windows(const uint8_t* d_img, int H, int W, const int32_t* d_rects, int T, int wh, int ww,
                                 uint8_t* d_tiles, hipStream_t st) {
    const size_t total = (size_t)T * wh * ww * 3;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(gather_windows_kernel, dim3(grid), dim3(256), 0, st, d_img, H, W, d_rects, T, wh, ww, d_tiles);
    return hipGetLastError();
}

// rowmap[2*oy] = window-row index ty (or -1: not covered), rowmap[2*oy+1] = row inside that
// window's output; colmap likewise with tx.
// "Later windows overwrite" (cnn_super_resolution.py:278) == the LAST (ty, tx) in loop order
// whose paste rectangle contains the pixel; paste rectangles are row-range x column-range
// products, so that is (last covering ty, last covering tx) -- resolved on the host into
// these maps, which makes the paste race-free and order-independent.
__global__ void stitch_u8_kernel(const uint8_t* __restrict__ tiles, int oth, int otw, const int32_t* __restrict__ rowmap,
                                 const int32_t* __restrict__ colmap, int tilesX, int OH, int OW,
                                 uint8_t* __restrict__ out) {
    const size_t total = (size_t)OH * OW * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const size_t r = i / 3;
        const int ox = (int)(r % OW);
        const int oy = (int)(r / OW);
        const int ty = rowmap[2 * oy], sy = rowmap[2 * oy + 1];
        const int tx = colmap[2 * ox], sx = colmap[2 * ox + 1];
        uint8_t v = 0;   // reference output starts as zeros (cnn_super_resolution.py:242)
        if (ty >= 0 && tx >= 0) v = tiles[(((size_t)(ty * tilesX + tx) * oth + sy) * otw + sx) * 3 + c];
        out[i] = v;
    }
}

__global__ void stitch_f32_kernel(const float* __restrict__ tiles, int oth, int otw, const int32_t* __restrict__ rowmap,
                                  const int32_t* __restrict__ colmap, int tilesX, int OH, int OW,
                                  float* __restrict__ out) {
    const size_t total = (size_t)OH * OW * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const size_t r = i / 3;
        const int ox = (int)(r % OW);
        const int oy = (int)(r / OW);
        const int ty = rowmap[2 * oy], sy = rowmap[2 * oy + 1];
        const int tx = colmap[2 * ox], sx = colmap[2 * ox + 1];
        float v = 0.f;
        if (ty >= 0 && tx >= 0) v = tiles[(((size_t)(ty * tilesX + tx) * 3 + c) * oth + sy) * otw + sx];
        out[i] = v;
    }
}

hipError_t launch_stitch_u8(const uint8_t* d_tiles, int tilesX, int oth, int otw, const int32_t* d_rowmap,
                            const int32_t* d_colmap, int OH, int OW, uint8_t* d_out, hipStream_t st) {
    const size_t total = (size_t)OH * OW * 3;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(stitch_u8_kernel, dim3(grid), dim3(256), 0, st, d_tiles, oth, otw, d_rowmap, d_colmap, tilesX, OH, OW,
                       d_out);
    return hipGetLastError();
}

hipError_t launch_stitch_f32(const float* d_tiles, int tilesX, int oth, int otw, const int32_t* d_rowmap,
                             const int32_t* d_colmap, int OH, int OW, float* d_out, hipStream_t st) {
    const size_t total = (size_t)OH * OW * 3;
    const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(stitch_f32_kernel, dim3(grid), dim3(256), 0, st, d_tiles, oth, otw, d_rowmap, d_colmap, tilesX, OH,
                       OW, d_out);
    return hipGetLastError();
}

}  // namespace s2sr
