// XYZ tile pyramid kernels: the step right after the SR path (reference server/app/tiling.py:102-186
// shells out to gdalwarp -t_srs EPSG:3857 -r bilinear and gdal2tiles.py --xyz --resampling average).
// All geometry is resolved on the host (s2sr/tiles.py) into tables; these kernels are HBM-bound
// byte work:
//   warp_bilinear  : RGB u8 source -> RGBA u8 raster on the Web-Mercator grid.  Source coordinates
//                    come from a node grid (one node every `step` output pixels), interpolated
//                    linearly, then 4-tap bilinear with edge replication; alpha = inside the source.
//   tiles_base     : deepest zoom: every tile pixel = rounded mean of the valid source pixels in its
//                    footprint [col_lo..col_hi] x [row_lo..row_hi] (tables per mosaic column / row).
//   tiles_overview : parent tile pixel = rounded mean of the valid pixels of its 2x2 children group.
// Float steps are single IEEE operations in a fixed order (-ffp-contract=off): the numpy oracle
// (oracle/tiles_ref.py) reproduces them bit for bit.
#include "s2sr_internal.h"

namespace s2sr {

namespace {

__global__ void __launch_bounds__(256) warp_bilinear_kernel(const uint8_t* __restrict__ rgb, int H, int W,
                                                            const float* __restrict__ grid, int gh, int gw, int step,
                                                            int OH, int OW, uint8_t* __restrict__ out) {
    const size_t total = (size_t)OH * OW;
    const float inv = 1.0f / (float)step;   // step is a power of two: exact
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % OW), oy = (int)(i / OW);
        const int gi = oy / step, gj = ox / step;
        const int gi1 = min(gi + 1, gh - 1), gj1 = min(gj + 1, gw - 1);
        const float fi = (float)(oy - gi * step) * inv, fj = (float)(ox - gj * step) * inv;
        const float2 a = ((const float2*)grid)[(size_t)gi * gw + gj], b = ((const float2*)grid)[(size_t)gi * gw + gj1];
        const float2 c = ((const float2*)grid)[(size_t)gi1 * gw + gj], d = ((const float2*)grid)[(size_t)gi1 * gw + gj1];
        const float u0 = __fadd_rn(a.x, __fmul_rn(__fsub_rn(b.x, a.x), fj)), u1 = __fadd_rn(c.x, __fmul_rn(__fsub_rn(d.x, c.x), fj));
        const float v0 = __fadd_rn(a.y, __fmul_rn(__fsub_rn(b.y, a.y), fj)), v1 = __fadd_rn(c.y, __fmul_rn(__fsub_rn(d.y, c.y), fj));
        const float u = __fadd_rn(u0, __fmul_rn(__fsub_rn(u1, u0), fi));
        const float v = __fadd_rn(v0, __fmul_rn(__fsub_rn(v1, v0), fi));
        uchar4 o = make_uchar4(0, 0, 0, 0);
        if (u >= -0.5f && u <= (float)W - 0.5f && v >= -0.5f && v <= (float)H - 0.5f) {
            const float xf = floorf(u), yf = floorf(v);
            const float fx = __fsub_rn(u, xf), fy = __fsub_rn(v, yf);
            const int x0 = min(max((int)xf, 0), W - 1), x1 = min(max((int)xf + 1, 0), W - 1);
            const int y0 = min(max((int)yf, 0), H - 1), y1 = min(max((int)yf + 1, 0), H - 1);
            const uint8_t* p00 = rgb + ((size_t)y0 * W + x0) * 3;
            const uint8_t* p10 = rgb + ((size_t)y0 * W + x1) * 3;
            const uint8_t* p01 = rgb + ((size_t)y1 * W + x0) * 3;
            const uint8_t* p11 = rgb + ((size_t)y1 * W + x1) * 3;
            uint8_t r[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float t = __fadd_rn((float)p00[k], __fmul_rn(__fsub_rn((float)p10[k], (float)p00[k]), fx));
                const float bt = __fadd_rn((float)p01[k], __fmul_rn(__fsub_rn((float)p11[k], (float)p01[k]), fx));
                const float val = __fadd_rn(t, __fmul_rn(__fsub_rn(bt, t), fy));
                r[k] = (uint8_t)(int)__fadd_rn(val, 0.5f);
            }
            o = make_uchar4(r[0], r[1], r[2], 255);
        }
        ((uchar4*)out)[i] = o;
    }
}

// mosaic pixel (gx, gy) of the level's tile array -> out[ty][tx][py][px]
__global__ void __launch_bounds__(256) tiles_base_kernel(const uint8_t* __restrict__ rgba, int W, const int32_t* __restrict__ col_lo,
                                                         const int32_t* __restrict__ col_hi, const int32_t* __restrict__ row_lo,
                                                         const int32_t* __restrict__ row_hi, int nx, int ny,
                                                         uint8_t* __restrict__ out) {
    const int MW = nx * 256;
    const size_t total = (size_t)MW * ny * 256;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int gx = (int)(i % MW), gy = (int)(i / MW);
        const int c0 = col_lo[gx], c1 = col_hi[gx], r0 = row_lo[gy], r1 = row_hi[gy];
        uint32_t s0 = 0, s1 = 0, s2 = 0, n = 0;
        for (int r = r0; r <= r1; ++r) {
            const uchar4* row = (const uchar4*)rgba + (size_t)r * W;
            for (int c = c0; c <= c1; ++c) {
                const uchar4 p = row[c];
                if (p.w) { s0 += p.x; s1 += p.y; s2 += p.z; ++n; }
            }
        }
        uchar4 o = make_uchar4(0, 0, 0, 0);
        if (n) o = make_uchar4((uint8_t)((s0 + n / 2) / n), (uint8_t)((s1 + n / 2) / n), (uint8_t)((s2 + n / 2) / n), 255);
        const int tx = gx >> 8, ty = gy >> 8;
        ((uchar4*)out)[(((size_t)ty * nx + tx) * 256 + (gy & 255)) * 256 + (gx & 255)] = o;
    }
}

__global__ void __launch_bounds__(256) tiles_overview_kernel(const uint8_t* __restrict__ child, int cnx, int cny, int ox, int oy,
                                                             int pnx, int pny, uint8_t* __restrict__ out) {
    const int MW = pnx * 256;
    const size_t total = (size_t)MW * pny * 256;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int gx = (int)(i % MW), gy = (int)(i / MW);
        uint32_t s0 = 0, s1 = 0, s2 = 0, n = 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int cx = ox * 256 + 2 * gx + dx, cy = oy * 256 + 2 * gy + dy;   // child mosaic pixel
                if (cx < 0 || cy < 0 || cx >= cnx * 256 || cy >= cny * 256) continue;
                const uchar4 p = ((const uchar4*)child)[(((size_t)(cy >> 8) * cnx + (cx >> 8)) * 256 + (cy & 255)) * 256 + (cx & 255)];
                if (p.w) { s0 += p.x; s1 += p.y; s2 += p.z; ++n; }
            }
        uchar4 o = make_uchar4(0, 0, 0, 0);
        if (n) o = make_uchar4((uint8_t)((s0 + n / 2) / n), (uint8_t)((s1 + n / 2) / n), (uint8_t)((s2 + n / 2) / n), 255);
        ((uchar4*)out)[(((size_t)(gy >> 8) * pnx + (gx >> 8)) * 256 + (gy & 255)) * 256 + (gx & 255)] = o;
    }
}

inline int grid_for(size_t total) { return (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256); }

}  // namespace

hipError_t launch_warp_bilinear(const uint8_t* d_rgb, int H, int W, const float* d_grid, int gh, int gw, int step, int OH, int OW,
                                uint8_t* d_out, hipStream_t st) {
    if (step <= 0 || (step & (step - 1))) return hipErrorInvalidValue;
    hipLaunchKernelGGL(warp_bilinear_kernel, dim3(grid_for((size_t)OH * OW)), dim3(256), 0, st, d_rgb, H, W, d_grid, gh, gw, step, OH,
                       OW, d_out);
    return hipGetLastError();
}

hipError_t launch_tiles_base(const uint8_t* d_rgba, int W, const int32_t* d_col_lo, const int32_t* d_col_hi, const int32_t* d_row_lo,
                             const int32_t* d_row_hi, int nx, int ny, uint8_t* d_out, hipStream_t st) {
    hipLaunchKernelGGL(tiles_base_kernel, dim3(grid_for((size_t)nx * ny * 65536)), dim3(256), 0, st, d_rgba, W, d_col_lo, d_col_hi,
                       d_row_lo, d_row_hi, nx, ny, d_out);
    return hipGetLastError();
}

hipError_t launch_tiles_overview(const uint8_t* d_child, int cnx, int cny, int ox, int oy, int pnx, int pny, uint8_t* d_out,
                                 hipStream_t st) {
    hipLaunchKernelGGL(tiles_overview_kernel, dim3(grid_for((size_t)pnx * pny * 65536)), dim3(256), 0, st, d_child, cnx, cny, ox, oy,
                       pnx, pny, d_out);
    return hipGetLastError();
}

}  // namespace s2sr
