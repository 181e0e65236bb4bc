// Crop-visibility post-process on the GPU: replaces the 8 OpenCV calls of `_enhance_for_crops`
// (reference server/app/wow_sr.py:187-209) and the farm variants (server/app/farm_sr.py:61-108).
//   stage 1  RGB->Lab (8-bit fixed point), CLAHE on L, Lab->RGB        (wow_sr.py:190-193)
//   stage 2  GaussianBlur (8.8 fixed-point separable) + addWeighted     (wow_sr.py:196-197)
//   stage 3  RGB->HSV, S *= gain where lo < H < hi (f32, trunc), HSV->RGB (wow_sr.py:200-207)
// Byte/integer work, HBM-bound: four launches per batch
//   clahe_hist  : read RGB once, L on the fly, per-CLAHE-tile histograms in LDS -> global
//   clahe_lut   : clip / redistribute / prefix-sum -> 256-entry LUT per tile
//   clahe_apply : read RGB, bilinear mix of 4 LUTs on L, Lab->RGB, write
//   sharpen_veg : LDS tile + halo, separable blur, weighted add, HSV boost, write
// All float steps are the reference's float32 operations in the same order; this file is built
// with -ffp-contract=off so no multiply-add is fused.
#include <math.h>

#include <mutex>
#include <vector>

#include "s2sr_internal.h"

namespace s2sr {

namespace {

constexpr int LAB_SHIFT = 12, GAMMA_SHIFT = 3, LAB_SHIFT2 = 15;
constexpr int CBRT_TAB = 256 * 3 / 2 * (1 << GAMMA_SHIFT);   // 3072
constexpr int INV_GAMMA_TAB = 4096;
constexpr int LAB_BASE = 1 << 14;

struct PPTables {
    uint16_t srgb_gamma[256];
    uint16_t lab_cbrt[CBRT_TAB];
    uint8_t inv_gamma[INV_GAMMA_TAB];
    uint16_t lab_to_y[256];
    uint16_t lab_to_ify[256];
    int32_t sdiv[256];
    int32_t hdiv180[256];
    int32_t fwd[9];   // RGB -> XYZ/whitepoint, 12-bit
    int32_t inv[9];   // XYZ*whitepoint -> RGB, 12-bit
};

PPTables* g_d_tables[64] = {nullptr};   // per device

void build_tables(PPTables& t) {
    for (int i = 0; i < 256; ++i) {
        const double x = i / 255.0;
        const double g = x <= 0.04045 ? x / 12.92 : pow((x + 0.055) / 1.055, 2.4);
        t.srgb_gamma[i] = (uint16_t)rint(255.0 * (1 << GAMMA_SHIFT) * g);
    }
    for (int i = 0; i < CBRT_TAB; ++i) {
        const double x = i / (255.0 * (1 << GAMMA_SHIFT));
        const double f = x < 216.0 / 24389.0 ? x * (841.0 / 108.0) + 16.0 / 116.0 : cbrt(x);
        t.lab_cbrt[i] = (uint16_t)rint((1 << LAB_SHIFT2) * f);
    }
    for (int i = 0; i < INV_GAMMA_TAB; ++i) {
        const double x = (double)i / INV_GAMMA_TAB;
        const double g = x <= 0.0031308 ? x * 12.92 : 1.055 * pow(x, 1.0 / 2.4) - 0.055;
        double v = rint(255.0 * g);
        t.inv_gamma[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
    for (int i = 0; i < 256; ++i) {
        double y, ify;
        if (i <= 20) {
            y = rint((double)i * LAB_BASE * 20 * 9 / (17.0 * 29 * 29 * 29));
            ify = rint(LAB_BASE * (16.0 / 116.0 + (double)i * 5 / (3.0 * 17 * 29)));
        } else {
            const double fy = (double)i * 100 * LAB_BASE / (255.0 * 116) + 16.0 * LAB_BASE / 116.0;
            ify = rint(fy);
            y = rint(fy * fy * fy / ((double)LAB_BASE * LAB_BASE));
        }
        t.lab_to_y[i] = (uint16_t)y;
        t.lab_to_ify[i] = (uint16_t)ify;
    }
    t.sdiv[0] = t.hdiv180[0] = 0;
    for (int i = 1; i < 256; ++i) {
        t.sdiv[i] = (int32_t)rint((255 << 12) / (double)i);
        t.hdiv180[i] = (int32_t)rint((180 << 12) / (6.0 * i));
    }
    const double s2x[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
    const double x2s[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};
    const double d65[3] = {0.950456, 1.0, 1.088754};
    const double scale[3] = {(1 << LAB_SHIFT) / d65[0], (double)(1 << LAB_SHIFT), (1 << LAB_SHIFT) / d65[2]};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            t.fwd[r * 3 + c] = (int32_t)rint(scale[r] * s2x[r * 3 + c]);
            t.inv[r * 3 + c] = (int32_t)rint((1 << LAB_SHIFT) * x2s[r * 3 + c] * d65[c]);
        }
}

hipError_t get_tables(PPTables** out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    static std::mutex mu;                      // handles on different threads may get here together
    std::lock_guard<std::mutex> lk(mu);
    if (!g_d_tables[dev]) {
        PPTables* h = new PPTables();
        build_tables(*h);
        PPTables* d = nullptr;
        e = dev_malloc(&d, sizeof(PPTables));
        hipStream_t st = nullptr;                  // not the legacy stream: hipMemcpy fails (and breaks the capture) while any
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);   // other handle captures a graph
        if (e == hipSuccess) e = hipMemcpyAsync(d, h, sizeof(PPTables), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (st) hipStreamDestroy(st);
        delete h;
        if (e != hipSuccess) return e;
        g_d_tables[dev] = d;
    }
    *out = g_d_tables[dev];
    return hipSuccess;
}

// ---- device helpers -------------------------------------------------------------------------
// The colour tables are 14 KiB of random-access lookups, a dozen per pixel: every workgroup copies
// them into LDS once (global lookups cost a TA pass per distinct line per wave).
__device__ __forceinline__ const PPTables* stage_tables(const PPTables* __restrict__ t, PPTables* s_t) {
    static_assert(sizeof(PPTables) % 4 == 0, "dword copy");
    const uint32_t* src = (const uint32_t*)t;
    uint32_t* dst = (uint32_t*)s_t;
    for (int i = threadIdx.x; i < (int)(sizeof(PPTables) / 4); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    return s_t;
}

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int reflect101(int i, int n) {   // BORDER_REFLECT_101, any number of bounces
    if ((unsigned)i < (unsigned)n) return i;
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    i = (i < 0 ? -i : i) % period;
    return i >= n ? period - i : i;
}

__device__ __forceinline__ void rgb2lab(const PPTables* __restrict__ t, int r, int g, int b, int& L, int& A, int& B) {
    const int R = t->srgb_gamma[r], G = t->srgb_gamma[g], Bq = t->srgb_gamma[b];
    const int fX = t->lab_cbrt[descale(R * t->fwd[0] + G * t->fwd[1] + Bq * t->fwd[2], LAB_SHIFT)];
    const int fY = t->lab_cbrt[descale(R * t->fwd[3] + G * t->fwd[4] + Bq * t->fwd[5], LAB_SHIFT)];
    const int fZ = t->lab_cbrt[descale(R * t->fwd[6] + G * t->fwd[7] + Bq * t->fwd[8], LAB_SHIFT)];
    constexpr int Lscale = (116 * 255 + 50) / 100;
    constexpr int Lshift = -((16 * 255 * (1 << LAB_SHIFT2) + 50) / 100);
    L = clamp255(descale(Lscale * fY + Lshift, LAB_SHIFT2));
    A = clamp255(descale(500 * (fX - fY) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2));
    B = clamp255(descale(200 * (fY - fZ) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2));
}

__device__ __forceinline__ int ab_to_xz(int v) {
    // C integer division truncates toward zero, exactly what the table build does
    if (v <= 3390) return v * 108 / 841 - LAB_BASE * 16 / 116 * 108 / 841;
    return (int)((long long)(v * v / LAB_BASE) * v / LAB_BASE);
}

__device__ __forceinline__ void lab2rgb(const PPTables* __restrict__ t, int L, int A, int B, int& r, int& g, int& b) {
    const int y = t->lab_to_y[L], ify = t->lab_to_ify[L];
    const int adiv = ((5 * A * 53687 + (1 << 7)) >> 13) - 128 * LAB_BASE / 500;
    const int bdiv = ((B * 41943 + (1 << 4)) >> 9) - 128 * LAB_BASE / 200 + 1;
    const long long x = ab_to_xz(ify + adiv), z = ab_to_xz(ify - bdiv);
    constexpr int shift = LAB_SHIFT + (14 - 12);
    long long v0 = (t->inv[0] * x + (long long)t->inv[1] * y + t->inv[2] * z + (1 << (shift - 1))) >> shift;
    long long v1 = (t->inv[3] * x + (long long)t->inv[4] * y + t->inv[5] * z + (1 << (shift - 1))) >> shift;
    long long v2 = (t->inv[6] * x + (long long)t->inv[7] * y + t->inv[8] * z + (1 << (shift - 1))) >> shift;
    v0 = v0 < 0 ? 0 : (v0 > INV_GAMMA_TAB - 1 ? INV_GAMMA_TAB - 1 : v0);
    v1 = v1 < 0 ? 0 : (v1 > INV_GAMMA_TAB - 1 ? INV_GAMMA_TAB - 1 : v1);
    v2 = v2 < 0 ? 0 : (v2 > INV_GAMMA_TAB - 1 ? INV_GAMMA_TAB - 1 : v2);
    r = t->inv_gamma[v0];
    g = t->inv_gamma[v1];
    b = t->inv_gamma[v2];
}

__device__ __forceinline__ void rgb2hsv(const int32_t* __restrict__ div, int r, int g, int b, int& h, int& s, int& v) {
    v = max(max(r, g), b);     // div = sdiv[256] | hdiv180[256]
    const int vmin = min(min(r, g), b);
    const int diff = v - vmin;
    s = (diff * div[v] + (1 << 11)) >> 12;
    int hh = (v == r) ? (g - b) : ((v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff));
    hh = (hh * div[256 + diff] + (1 << 11)) >> 12;
    hh += hh < 0 ? 180 : 0;
    h = clamp255(hh);
}

__device__ __forceinline__ int sat_round(float f) {
    int v = __float2int_rn(f);
    return clamp255(v);
}

__device__ __forceinline__ void hsv2rgb(int hi, int si, int vi, int& r, int& g, int& b) {
    const float s = (float)si * (1.0f / 255.0f), v = (float)vi * (1.0f / 255.0f);
    float fb, fg, fr;
    if (si == 0) {
        fb = fg = fr = v;
    } else {
        float h = (float)hi * (6.0f / 180.0f);
        if (h >= 6.0f) h -= 6.0f;
        int sector = (int)floorf(h);
        h -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        const float t0 = v;
        const float t1 = v * (1.0f - s);
        const float t2 = v * (1.0f - s * h);
        const float t3 = v * (1.0f - s * (1.0f - h));
        // (b, g, r) <- tab index per sector: {1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}
        const int ib = (0x200311 >> (4 * sector)) & 3;   // packed nibbles, sector 0 in the low nibble
        const int ig = (0x112003 >> (4 * sector)) & 3;
        const int ir = (0x031120 >> (4 * sector)) & 3;
        fb = ib == 0 ? t0 : (ib == 1 ? t1 : (ib == 2 ? t2 : t3));
        fg = ig == 0 ? t0 : (ig == 1 ? t1 : (ig == 2 ? t2 : t3));
        fr = ir == 0 ? t0 : (ir == 1 ? t1 : (ir == 2 ? t2 : t3));
    }
    r = sat_round(fr * 255.0f);
    g = sat_round(fg * 255.0f);
    b = sat_round(fb * 255.0f);
}

struct ClaheGeom {
    int H, W;        // image
    int grid;        // tiles per side
    int th, tw;      // CLAHE tile size (on the padded image)
    int eh, ew;      // padded ("ext") image size
    int clip;        // absolute clip limit, 0 = off
    float lut_scale;
    int split;       // clahe_hist: workgroups (row bands) per CLAHE tile
    // banded runs over ONE image (launch_pp_band_*): the kernels see rows [y0, y1) of it
    int y0, y1;      // clahe_hist_rows: source rows counted; clahe_apply: y0 = image row of the first row handed over
    int bgr;         // the image's bytes are B,G,R (what RealESRGAN.enhance handles, wow_sr.py:85,94); the colour math is RGB's
};

// ---- kernel 1: per-tile L histograms --------------------------------------------------------
// grid.x = B * grid*grid * SPLIT ; every workgroup takes a horizontal band of one CLAHE tile.
__global__ void __launch_bounds__(256) clahe_hist_kernel(const uint8_t* __restrict__ rgb, ClaheGeom gm,
                                                         const PPTables* __restrict__ t, uint32_t* __restrict__ hist) {
    __shared__ uint32_t sh[256];
    __shared__ PPTables s_t;
    sh[threadIdx.x] = 0;
    t = stage_tables(t, &s_t);
    const int ntile = gm.grid * gm.grid;
    int id = blockIdx.x;
    const int part = id % gm.split; id /= gm.split;
    const int tile = id % ntile;
    const int img = id / ntile;
    const int ty = tile / gm.grid, tx = tile % gm.grid;
    const int rows_per = (gm.th + gm.split - 1) / gm.split;
    const int r0 = part * rows_per, r1 = min(gm.th, r0 + rows_per);
    const uint8_t* base = rgb + (size_t)img * gm.H * gm.W * 3;
    const int npx = (r1 - r0) * gm.tw;
    for (int i = threadIdx.x; i < npx; i += 256) {
        const int yy = ty * gm.th + r0 + i / gm.tw, xx = tx * gm.tw + i % gm.tw;
        const int sy = reflect101(yy, gm.H), sx = reflect101(xx, gm.W);   // BORDER_REFLECT_101 padding
        const uint8_t* p = base + ((size_t)sy * gm.W + sx) * 3;
        int L, A, B;
        rgb2lab(t, p[0], p[1], p[2], L, A, B);
        atomicAdd(&sh[L], 1u);
    }
    __syncthreads();
    const uint32_t v = sh[threadIdx.x];
    if (v) atomicAdd(&hist[((size_t)img * ntile + tile) * 256 + threadIdx.x], v);
}

// ---- kernel 1b: the same counts for the SOURCE rows [gm.y0, gm.y1) of one image ------------------
// An AOI's mosaic is complete band by band (engine.hip enhance_impl, s2sr/dist.py); its histograms accumulate band by band under
// the compute of the windows still to come.  A source row feeds the padded rows that map to it (itself, and its reflections in
// the BORDER_REFLECT_101 padding), so the workgroups walk their CLAHE tile's padded rows and keep those whose source row is in
// the band: integer counts, the same totals in any split.  grid.x = tile columns, grid.y = (tile rows ty_lo..ty_hi) x chunks of
// gm.split padded rows: a workgroup takes ~32k pixels, so one 1024-row band of a 16384-wide mosaic is 512 workgroups (the first form
// cut a tile into 8 parts whatever the band: 32 busy workgroups per band, 1.1 ms each, 17 ms over the 16 bands of a 4096 x 4096 AOI).
// One histogram per wave (waves take different rows; their lanes still meet on equal L values of neighbouring pixels).
__global__ void __launch_bounds__(256) clahe_hist_rows_kernel(const uint8_t* __restrict__ img, ClaheGeom gm, int ty_lo, int chunks_per_tile,
                                                              const PPTables* __restrict__ t, uint32_t* __restrict__ hist) {
    __shared__ uint32_t sh[4][256];
    __shared__ PPTables s_t;
    const int tx = blockIdx.x, ty = ty_lo + (int)blockIdx.y / chunks_per_tile, part = (int)blockIdx.y % chunks_per_tile;
    const int r0 = part * gm.split, r1 = min(gm.th, r0 + gm.split);
    // any row of this workgroup in the band?  (wave-uniform scan over at most gm.split rows; most workgroups of a band leave here)
    bool any = false;
    for (int r = r0; r < r1 && !any; ++r) {
        const int sy = reflect101(ty * gm.th + r, gm.H);
        any = sy >= gm.y0 && sy < gm.y1;
    }
    if (!any) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < 4; ++w) sh[w][threadIdx.x] = 0;
    t = stage_tables(t, &s_t);
    for (int r = r0 + wave; r < r1; r += 4) {
        const int sy = reflect101(ty * gm.th + r, gm.H);
        if (sy < gm.y0 || sy >= gm.y1) continue;
        const uint8_t* row = img + (size_t)sy * gm.W * 3;
        for (int c = lane; c < gm.tw; c += 64) {
            const int sx = reflect101(tx * gm.tw + c, gm.W);
            const uint8_t* p = row + (size_t)sx * 3;
            int L, A, B;
            rgb2lab(t, p[gm.bgr ? 2 : 0], p[1], p[gm.bgr ? 0 : 2], L, A, B);
            atomicAdd(&sh[wave][L], 1u);
        }
    }
    __syncthreads();
    const uint32_t v = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
    if (v) atomicAdd(&hist[(size_t)(ty * gm.grid + tx) * 256 + threadIdx.x], v);
}

// ---- kernel 2: clip, redistribute, CDF -> LUT -----------------------------------------------
__global__ void __launch_bounds__(256) clahe_lut_kernel(const uint32_t* __restrict__ hist, ClaheGeom gm,
                                                        uint8_t* __restrict__ lut) {
    __shared__ int sh[256];
    const int i = threadIdx.x;
    int h = (int)hist[(size_t)blockIdx.x * 256 + i];
    if (gm.clip > 0) {
        int excess = h > gm.clip ? h - gm.clip : 0;
        h = h > gm.clip ? gm.clip : h;
        sh[i] = excess;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (i < s) sh[i] += sh[i + s];
            __syncthreads();
        }
        const int clipped = sh[0];
        __syncthreads();
        const int batch = clipped / 256;
        int residual = clipped - batch * 256;
        h += batch;
        if (residual != 0) {
            const int step = max(256 / residual, 1);
            if (i % step == 0 && i / step < residual) h += 1;
        }
    }
    // inclusive prefix sum
    sh[i] = h;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = i >= off ? sh[i - off] : 0;
        __syncthreads();
        sh[i] += v;
        __syncthreads();
    }
    lut[(size_t)blockIdx.x * 256 + i] = (uint8_t)sat_round((float)sh[i] * gm.lut_scale);
}

// ---- kernel 3: apply CLAHE on L and convert back to RGB -------------------------------------
__global__ void __launch_bounds__(256) clahe_apply_kernel(const uint8_t* __restrict__ rgb, ClaheGeom gm, int B,
                                                          const PPTables* __restrict__ t,
                                                          const uint8_t* __restrict__ lut, uint8_t* __restrict__ out) {
    __shared__ PPTables s_t;
    t = stage_tables(t, &s_t);
    const size_t npx = (size_t)gm.H * gm.W, total = npx * B;
    const float inv_tw = 1.0f / (float)gm.tw, inv_th = 1.0f / (float)gm.th;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int img = (int)(i / npx);
        const size_t rem = i - (size_t)img * npx;
        const int y = (int)(rem / gm.W) + gm.y0, x = (int)(rem % gm.W);
        const uint8_t* p = rgb + i * 3;
        int L, A, Bc;
        rgb2lab(t, p[gm.bgr ? 2 : 0], p[1], p[gm.bgr ? 0 : 2], L, A, Bc);
        const float txf = (float)x * inv_tw - 0.5f, tyf = (float)y * inv_th - 0.5f;
        int tx1 = (int)floorf(txf), ty1 = (int)floorf(tyf);
        const float xa = txf - (float)tx1, ya = tyf - (float)ty1;
        const float xa1 = 1.0f - xa, ya1 = 1.0f - ya;
        const int tx2 = min(tx1 + 1, gm.grid - 1), ty2 = min(ty1 + 1, gm.grid - 1);
        tx1 = max(tx1, 0);
        ty1 = max(ty1, 0);
        const uint8_t* lb = lut + (size_t)img * gm.grid * gm.grid * 256 + L;
        const float l11 = lb[(ty1 * gm.grid + tx1) * 256], l12 = lb[(ty1 * gm.grid + tx2) * 256];
        const float l21 = lb[(ty2 * gm.grid + tx1) * 256], l22 = lb[(ty2 * gm.grid + tx2) * 256];
        const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
        int r, g, b;
        lab2rgb(t, sat_round(res), A, Bc, r, g, b);
        uint8_t* o = out + i * 3;
        o[gm.bgr ? 2 : 0] = (uint8_t)r;
        o[1] = (uint8_t)g;
        o[gm.bgr ? 0 : 2] = (uint8_t)b;
    }
}

// ---- kernel 3b: same, 4 consecutive pixels (12 bytes = 3 dwords) per thread ----------------------
// Needs 4-byte aligned tensors; lanes read and write 12 contiguous bytes each, so a wave moves 768
// contiguous bytes per instruction instead of 64 scattered single bytes.
__device__ __forceinline__ void clahe_pixel(const PPTables* __restrict__ t, const ClaheGeom& gm, const uint8_t* __restrict__ lut_img,
                                            float inv_tw, float inv_th, int y, int x, int& r, int& g, int& b) {
    int L, A, Bc;
    rgb2lab(t, r, g, b, L, A, Bc);
    const float txf = (float)x * inv_tw - 0.5f, tyf = (float)y * inv_th - 0.5f;
    int tx1 = (int)floorf(txf), ty1 = (int)floorf(tyf);
    const float xa = txf - (float)tx1, ya = tyf - (float)ty1;
    const float xa1 = 1.0f - xa, ya1 = 1.0f - ya;
    const int tx2 = min(tx1 + 1, gm.grid - 1), ty2 = min(ty1 + 1, gm.grid - 1);
    tx1 = max(tx1, 0);
    ty1 = max(ty1, 0);
    const uint8_t* lb = lut_img + L;
    const float l11 = lb[(ty1 * gm.grid + tx1) * 256], l12 = lb[(ty1 * gm.grid + tx2) * 256];
    const float l21 = lb[(ty2 * gm.grid + tx1) * 256], l22 = lb[(ty2 * gm.grid + tx2) * 256];
    const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
    lab2rgb(t, sat_round(res), A, Bc, r, g, b);
}

__global__ void __launch_bounds__(256) clahe_apply4_kernel(const uint8_t* __restrict__ rgb, ClaheGeom gm, int B,
                                                           const PPTables* __restrict__ t,
                                                           const uint8_t* __restrict__ lut, uint8_t* __restrict__ out) {
    __shared__ PPTables s_t;
    t = stage_tables(t, &s_t);
    const uint32_t npx = (uint32_t)gm.H * gm.W;
    const uint32_t total = npx * (uint32_t)B;            // launcher guarantees < 2^32
    const uint32_t ngroups = (total + 3) / 4;
    const float inv_tw = 1.0f / (float)gm.tw, inv_th = 1.0f / (float)gm.th;
    for (uint32_t gidx = blockIdx.x * blockDim.x + threadIdx.x; gidx < ngroups; gidx += gridDim.x * blockDim.x) {
        const uint32_t p0 = 4 * gidx;
        const size_t off = (size_t)p0 * 3;
        const uint32_t n = min(4u, total - p0);
        uint32_t w[3] = {0, 0, 0};
        if (n == 4) {
            const uint32_t* src = (const uint32_t*)(rgb + off);
            w[0] = src[0]; w[1] = src[1]; w[2] = src[2];
        } else {
            for (uint32_t k = 0; k < 3 * n; ++k) w[k >> 2] |= (uint32_t)rgb[off + k] << (8 * (k & 3));
        }
        uint32_t img = p0 / npx;
        const uint32_t rem = p0 - img * npx;
        uint32_t y = rem / gm.W, x = rem - y * gm.W;
        uint32_t o[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int bi = 3 * k;
            int r = (w[bi >> 2] >> (8 * (bi & 3))) & 255, g = (w[(bi + 1) >> 2] >> (8 * ((bi + 1) & 3))) & 255,
                b = (w[(bi + 2) >> 2] >> (8 * ((bi + 2) & 3))) & 255;
            if (gm.bgr) { const int s_ = r; r = b; b = s_; }
            if ((uint32_t)k < n) clahe_pixel(t, gm, lut + (size_t)img * gm.grid * gm.grid * 256, inv_tw, inv_th, (int)y + gm.y0, (int)x, r, g, b);
            if (gm.bgr) { const int s_ = r; r = b; b = s_; }
            o[bi >> 2] |= (uint32_t)r << (8 * (bi & 3));
            o[(bi + 1) >> 2] |= (uint32_t)g << (8 * ((bi + 1) & 3));
            o[(bi + 2) >> 2] |= (uint32_t)b << (8 * ((bi + 2) & 3));
            if (++x == (uint32_t)gm.W) { x = 0; if (++y == (uint32_t)gm.H) { y = 0; ++img; } }
        }
        if (n == 4) {
            uint32_t* dst = (uint32_t*)(out + off);
            dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
        } else {
            for (uint32_t k = 0; k < 3 * n; ++k) out[off + k] = (uint8_t)(o[k >> 2] >> (8 * (k & 3)));
        }
    }
}

// ---- kernel 4: unsharp mask + vegetation boost ----------------------------------------------
constexpr int ST = 32;          // output tile edge
constexpr int MAXR = 8;         // max Gaussian radius (ksize <= 17)
struct SharpParams {
    int H, W, B;
    int radius;                 // taps = 2*radius+1, 0 = no blur stage
    int taps[2 * MAXR + 1];     // 8.8 fixed point, sum 256
    float w_img, w_blur;
    int do_veg, hue_lo, hue_hi;
    float sat_gain;
    // banded runs over ONE image (launch_pp_band_sharpen): rows [y_begin, y_end) are produced, tile rows count from y_begin; the
    // halo rows come from the same image (reflected at ITS borders), so a band's bytes are those of the whole-image launch
    int y_begin, y_end;
    int bgr;         // bytes are B,G,R: blur and weighted add are per channel, only the HSV step looks at the order
    int swap_out;    // store R and B exchanged (the job's BGR2RGB, wow_sr.py:103, folded into the last pass)
};

__global__ void __launch_bounds__(256) sharpen_veg_kernel(const uint8_t* __restrict__ in, SharpParams sp,
                                                          const PPTables* __restrict__ t, uint8_t* __restrict__ out) {
    __shared__ uint8_t s_in[(ST + 2 * MAXR) * (ST + 2 * MAXR) * 3];
    __shared__ uint16_t s_h[(ST + 2 * MAXR) * ST * 3];
    __shared__ int32_t s_div[512];   // sdiv | hdiv180
    for (int i = threadIdx.x; i < 512; i += 256) s_div[i] = i < 256 ? t->sdiv[i] : t->hdiv180[i - 256];
    const int tilesX = (sp.W + ST - 1) / ST, tilesY = (sp.y_end - sp.y_begin + ST - 1) / ST;
    int id = blockIdx.x;
    const int tx = id % tilesX; id /= tilesX;
    const int ty = id % tilesY;
    const int img = id / tilesY;
    const uint8_t* src = in + (size_t)img * sp.H * sp.W * 3;
    uint8_t* dst = out + (size_t)img * sp.H * sp.W * 3;
    const int r = sp.radius, ext = ST + 2 * r;
    const int y0 = sp.y_begin + ty * ST, x0 = tx * ST;
    // stage the tile + halo (reflect-101 at the image border)
    for (int i = threadIdx.x; i < ext * ext; i += 256) {
        const int ly = i / ext, lx = i % ext;
        int sy = reflect101(min(y0 + ly - r, sp.H - 1 + r), sp.H), sx = reflect101(min(x0 + lx - r, sp.W - 1 + r), sp.W);
        sy = min(max(sy, 0), sp.H - 1);   // only reachable for images smaller than the kernel radius
        sx = min(max(sx, 0), sp.W - 1);
        const uint8_t* p = src + ((size_t)sy * sp.W + sx) * 3;
        s_in[i * 3 + 0] = p[0];
        s_in[i * 3 + 1] = p[1];
        s_in[i * 3 + 2] = p[2];
    }
    __syncthreads();
    if (r > 0) {
        // horizontal pass: (ext rows) x (ST cols) x 3, 8.8 fixed point, fits 16 bits (taps sum to 256)
        for (int i = threadIdx.x; i < ext * ST * 3; i += 256) {
            const int c = i % 3, lx = (i / 3) % ST, ly = i / (3 * ST);
            int acc = 0;
            for (int k = 0; k <= 2 * r; ++k) acc += sp.taps[k] * s_in[(ly * ext + lx + k) * 3 + c];
            s_h[i] = (uint16_t)acc;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < ST * ST; i += 256) {
        const int ly = i / ST, lx = i % ST;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= sp.y_end || x >= sp.W) continue;
        int px[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int center = s_in[((ly + r) * ext + lx + r) * 3 + c];
            if (r > 0) {
                uint32_t acc = 0;
                for (int k = 0; k <= 2 * r; ++k) acc += (uint32_t)sp.taps[k] * s_h[((ly + k) * ST + lx) * 3 + c];
                const int blur = clamp255((int)((acc + (1u << 15)) >> 16));
                px[c] = sat_round((float)center * sp.w_img + (float)blur * sp.w_blur);
            } else {
                px[c] = center;
            }
        }
        const int ir = sp.bgr ? 2 : 0, ib = 2 - ir;
        if (sp.do_veg) {
            int h, s, v;
            rgb2hsv(s_div, px[ir], px[1], px[ib], h, s, v);
            if (h > sp.hue_lo && h < sp.hue_hi) {
                // float32 S*gain, clip to [0,255], astype(uint8) == truncation (wow_sr.py:204-207)
                float fs = (float)s * sp.sat_gain;
                fs = fminf(fmaxf(fs, 0.f), 255.f);
                s = (int)fs;
            }
            hsv2rgb(h, s, v, px[ir], px[1], px[ib]);
        }
        uint8_t* o = dst + ((size_t)y * sp.W + x) * 3;
        o[0] = (uint8_t)px[sp.swap_out ? 2 : 0];
        o[1] = (uint8_t)px[1];
        o[2] = (uint8_t)px[sp.swap_out ? 0 : 2];
    }
}

// ---- kernel 4b: the same stage for the radii the reference uses (3, 4, 5), restructured ----------
// The generic kernel above spends its time in LDS byte reads (2r+1 per output and pass).  Here the
// tile is 64x32, the staged pixels are de-interleaved into channel planes so a run of consecutive
// x is a run of consecutive bytes, and both passes slide a register window: the horizontal pass
// makes 8 outputs from three 8-byte reads, the vertical pass 2x4 outputs per channel from 4+2R
// dword reads.  Interior tiles are staged with aligned dword loads (one wave per row).
constexpr int SW = 64, SH = 32, SPITCH = 96;
template <int R>
__global__ void __launch_bounds__(256) sharpen_veg_kernel_r(const uint8_t* __restrict__ in, SharpParams sp, size_t total_bytes,
                                                            const PPTables* __restrict__ t, uint8_t* __restrict__ out) {
    constexpr int EH = SH + 2 * R, EW = SW + 2 * R, NT = 2 * R + 1;
    __shared__ __attribute__((aligned(16))) uint8_t s_in[3 * EH * SPITCH];
    __shared__ __attribute__((aligned(16))) uint16_t s_h[3 * EH * SW];
    __shared__ int32_t s_div[512];   // sdiv | hdiv180
    for (int i = threadIdx.x; i < 512; i += 256) s_div[i] = i < 256 ? t->sdiv[i] : t->hdiv180[i - 256];
    const int tilesX = (sp.W + SW - 1) / SW, tilesY = (sp.y_end - sp.y_begin + SH - 1) / SH;
    int id = blockIdx.x;
    const int tx = id % tilesX; id /= tilesX;
    const int ty = id % tilesY;
    const int img = id / tilesY;
    const uint8_t* src = in + (size_t)img * sp.H * sp.W * 3;
    uint8_t* dst = out + (size_t)img * sp.H * sp.W * 3;
    const int y0 = sp.y_begin + ty * SH, x0 = tx * SW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (x0 - R >= 0 && x0 + SW - 1 + R < sp.W) {
        // interior in x: each staged row is EW*3 contiguous bytes of one image row
        for (int ly = wave; ly < EH; ly += 4) {
            int sy = reflect101(min(y0 + ly - R, sp.H - 1 + R), sp.H);
            sy = min(max(sy, 0), sp.H - 1);
            const uintptr_t a = (uintptr_t)(src + ((size_t)sy * sp.W + (x0 - R)) * 3);
            const int mis = (int)(a & 3);
            const uintptr_t al = a - mis + 4 * (uintptr_t)lane;
            if (4 * lane < mis + EW * 3) {
                uint32_t d = 0;
                if (al >= (uintptr_t)in && al + 4 <= (uintptr_t)in + total_bytes) {
                    d = *(const uint32_t*)al;
                } else {
                    for (int k = 0; k < 4; ++k)
                        if (al + k >= (uintptr_t)in && al + k < (uintptr_t)in + total_bytes) d |= (uint32_t)(*(const uint8_t*)(al + k)) << (8 * k);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int b = 4 * lane + k - mis;
                    if (b >= 0 && b < EW * 3) {
                        const int px = b / 3, c = b - 3 * px;
                        s_in[(c * EH + ly) * SPITCH + px] = (uint8_t)(d >> (8 * k));
                    }
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < EH * EW; i += 256) {
            const int ly = i / EW, lx = i % EW;
            int sy = reflect101(min(y0 + ly - R, sp.H - 1 + R), sp.H), sx = reflect101(min(x0 + lx - R, sp.W - 1 + R), sp.W);
            sy = min(max(sy, 0), sp.H - 1);   // only reachable for images smaller than the kernel radius
            sx = min(max(sx, 0), sp.W - 1);
            const uint8_t* p = src + ((size_t)sy * sp.W + sx) * 3;
            s_in[(0 * EH + ly) * SPITCH + lx] = p[0];
            s_in[(1 * EH + ly) * SPITCH + lx] = p[1];
            s_in[(2 * EH + ly) * SPITCH + lx] = p[2];
        }
    }
    __syncthreads();
    // horizontal pass: item = (channel, row, group of 8 x)
    for (int it = threadIdx.x; it < 3 * EH * (SW / 8); it += 256) {
        const int j = it % (SW / 8), cr = it / (SW / 8);   // cr = c*EH + row
        const uint8_t* rowp = s_in + cr * SPITCH + 8 * j;
        uint32_t w[6];
        const uint2 q0 = *(const uint2*)(rowp), q1 = *(const uint2*)(rowp + 8), q2 = *(const uint2*)(rowp + 16);
        w[0] = q0.x; w[1] = q0.y; w[2] = q1.x; w[3] = q1.y; w[4] = q2.x; w[5] = q2.y;
        uint32_t o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int b = i + k;
                acc += (uint32_t)sp.taps[k] * ((w[b >> 2] >> (8 * (b & 3))) & 255u);
            }
            o[i] = acc;   // <= 255*256: fits 16 bits
        }
        uint4 pk;
        pk.x = o[0] | (o[1] << 16); pk.y = o[2] | (o[3] << 16); pk.z = o[4] | (o[5] << 16); pk.w = o[6] | (o[7] << 16);
        *(uint4*)(s_h + (size_t)cr * SW + 8 * j) = pk;
    }
    __syncthreads();
    // vertical pass + weighted add + vegetation boost: thread = (x pair, 4 rows)
    const int xp = threadIdx.x & 31, yq = threadIdx.x >> 5;
    int res[4][2][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uint32_t col[4 + 2 * R];
#pragma unroll
        for (int k = 0; k < 4 + 2 * R; ++k) col[k] = *(const uint32_t*)(s_h + (size_t)(c * EH + 4 * yq + k) * SW + 2 * xp);
#pragma unroll
        for (int ry = 0; ry < 4; ++ry) {
            uint32_t a0 = 0, a1 = 0;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                a0 += (uint32_t)sp.taps[k] * (col[ry + k] & 0xffffu);
                a1 += (uint32_t)sp.taps[k] * (col[ry + k] >> 16);
            }
            const int b0 = clamp255((int)((a0 + (1u << 15)) >> 16)), b1 = clamp255((int)((a1 + (1u << 15)) >> 16));
            const uint8_t* cp = s_in + (c * EH + 4 * yq + ry + R) * SPITCH + 2 * xp + R;
            res[ry][0][c] = sat_round((float)cp[0] * sp.w_img + (float)b0 * sp.w_blur);
            res[ry][1][c] = sat_round((float)cp[1] * sp.w_img + (float)b1 * sp.w_blur);
        }
    }
#pragma unroll
    for (int ry = 0; ry < 4; ++ry) {
        const int y = y0 + 4 * yq + ry;
#pragma unroll
        for (int xx = 0; xx < 2; ++xx) {
            int pr = res[ry][xx][sp.bgr ? 2 : 0], pg = res[ry][xx][1], pb = res[ry][xx][sp.bgr ? 0 : 2];
            if (sp.do_veg) {
                int h, sa, v;
                rgb2hsv(s_div, pr, pg, pb, h, sa, v);
                if (h > sp.hue_lo && h < sp.hue_hi) {
                    float fs = (float)sa * sp.sat_gain;
                    fs = fminf(fmaxf(fs, 0.f), 255.f);
                    sa = (int)fs;
                }
                hsv2rgb(h, sa, v, pr, pg, pb);
            }
            const int x = x0 + 2 * xp + xx;
            if (y < sp.y_end && x < sp.W) {   // byte stores: measured faster than staging rows in LDS for dword stores
                uint8_t* o = dst + ((size_t)y * sp.W + x) * 3;
                const bool rfirst = (sp.bgr != 0) == (sp.swap_out != 0);   // R goes to byte 0 when the order ends up RGB
                o[0] = (uint8_t)(rfirst ? pr : pb);
                o[1] = (uint8_t)pg;
                o[2] = (uint8_t)(rfirst ? pb : pr);
            }
        }
    }
}

void gaussian_taps_q8(double sigma, int& radius, int* taps) {
    int n = ((int)rint(sigma * 6 + 1)) | 1;
    if (n > 2 * MAXR + 1) n = 2 * MAXR + 1;
    radius = n / 2;
    std::vector<double> k(n);
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        const double x = i - (n - 1) / 2.0;
        k[i] = exp(-(x * x) / (2.0 * sigma * sigma));
        sum += k[i];
    }
    double err = 0;
    int tot = 0;
    for (int i = 0; i < n / 2; ++i) {
        const double adj = k[i] / sum * 256.0 + err;
        const int v0 = (int)rint(adj);
        err = adj - v0;
        taps[i] = taps[n - 1 - i] = v0;
        tot += 2 * v0;
    }
    taps[n / 2] = 256 - tot;
}

ClaheGeom clahe_geom(int H, int W, const s2sr_pp_params& prm) {
    ClaheGeom g{};
    g.H = H; g.W = W; g.grid = prm.clahe_grid;
    if (W % g.grid == 0 && H % g.grid == 0) {
        g.eh = H; g.ew = W;
    } else {   // OpenCV pads BOTH dimensions, an evenly dividing one by a full `grid`
        g.eh = H + (g.grid - H % g.grid);
        g.ew = W + (g.grid - W % g.grid);
    }
    g.th = g.eh / g.grid; g.tw = g.ew / g.grid;
    const int area = g.th * g.tw;
    g.lut_scale = 255.0f / (float)area;
    g.clip = 0;
    if (prm.clahe_clip > 0.0f) {
        g.clip = (int)((double)prm.clahe_clip * area / 256);
        if (g.clip < 1) g.clip = 1;
    }
    return g;
}

}  // namespace

size_t postprocess_work_bytes(int B, int H, int W, const s2sr_pp_params& prm) {
    const size_t tiles = (size_t)B * prm.clahe_grid * prm.clahe_grid;
    return tiles * 256 * 4 + tiles * 256 + (size_t)B * H * W * 3 + 1024;
}

hipError_t launch_postprocess(const uint8_t* d_rgb, int B, int H, int W, const s2sr_pp_params& prm, uint8_t* d_out,
                              void* d_work, size_t work_bytes, hipStream_t st) {
    if (prm.clahe_grid <= 0 || prm.clahe_grid > 64) return hipErrorInvalidValue;
    if (work_bytes < postprocess_work_bytes(B, H, W, prm)) return hipErrorInvalidValue;
    PPTables* t = nullptr;
    hipError_t e = get_tables(&t);
    if (e != hipSuccess) return e;
    const size_t tiles = (size_t)B * prm.clahe_grid * prm.clahe_grid;
    uint32_t* d_hist = (uint32_t*)d_work;
    uint8_t* d_lut = (uint8_t*)d_work + tiles * 256 * 4;
    uint8_t* d_tmp = d_lut + ((tiles * 256 + 255) & ~(size_t)255);
    const bool s1 = prm.stages & 1, s2 = prm.stages & 2, s3 = prm.stages & 4;
    const uint8_t* cur = d_rgb;
    if (s1) {
        ClaheGeom g = clahe_geom(H, W, prm);
        if (g.th <= 0 || g.tw <= 0) return hipErrorInvalidValue;
        // enough workgroups to fill 256 CUs, but each one stages 14 KiB of tables: keep them fat
        g.split = tiles >= 2048 ? 1 : (tiles >= 1024 ? 2 : (tiles >= 512 ? 4 : 8));
        e = hipMemsetAsync(d_hist, 0, tiles * 256 * 4, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(clahe_hist_kernel, dim3((unsigned)(tiles * g.split)), dim3(256), 0, st, cur, g, t, d_hist);
        hipLaunchKernelGGL(clahe_lut_kernel, dim3((unsigned)tiles), dim3(256), 0, st, d_hist, g, d_lut);
        uint8_t* o = (s2 || s3) ? d_tmp : d_out;
        const size_t total = (size_t)B * H * W;
        const unsigned grid = (unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        const bool aligned = (((uintptr_t)cur | (uintptr_t)o) & 3) == 0 && total < (1ull << 32);
        if (aligned) {
            const size_t ng = (total + 3) / 4;
            const unsigned grid4 = (unsigned)((ng + 255) / 256 > 2048 ? 2048 : (ng + 255) / 256);
            hipLaunchKernelGGL(clahe_apply4_kernel, dim3(grid4), dim3(256), 0, st, cur, g, B, t, d_lut, o);
        } else {
            hipLaunchKernelGGL(clahe_apply_kernel, dim3(grid), dim3(256), 0, st, cur, g, B, t, d_lut, o);
        }
        cur = o;
    }
    if (s2 || s3) {
        SharpParams sp{};
        sp.H = H; sp.W = W; sp.B = B;
        sp.y_begin = 0; sp.y_end = H;
        sp.radius = 0;
        if (s2) gaussian_taps_q8(prm.blur_sigma, sp.radius, sp.taps);
        sp.w_img = prm.w_img; sp.w_blur = prm.w_blur;
        sp.do_veg = s3 ? 1 : 0; sp.hue_lo = prm.hue_lo; sp.hue_hi = prm.hue_hi; sp.sat_gain = prm.sat_gain;
        const unsigned grid = (unsigned)(((W + ST - 1) / ST) * ((H + ST - 1) / ST) * B);
        const unsigned grid_r = (unsigned)(((W + SW - 1) / SW) * ((H + SH - 1) / SH) * B);
        const size_t total_bytes = (size_t)B * H * W * 3;
        if (sp.radius == 4) hipLaunchKernelGGL(sharpen_veg_kernel_r<4>, dim3(grid_r), dim3(256), 0, st, cur, sp, total_bytes, t, d_out);
        else if (sp.radius == 5) hipLaunchKernelGGL(sharpen_veg_kernel_r<5>, dim3(grid_r), dim3(256), 0, st, cur, sp, total_bytes, t, d_out);
        else if (sp.radius == 3) hipLaunchKernelGGL(sharpen_veg_kernel_r<3>, dim3(grid_r), dim3(256), 0, st, cur, sp, total_bytes, t, d_out);
        else hipLaunchKernelGGL(sharpen_veg_kernel, dim3(grid), dim3(256), 0, st, cur, sp, t, d_out);
    } else if (!s1) {
        e = hipMemcpyAsync(d_out, d_rgb, (size_t)B * H * W * 3, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

// ---- the same post-process over ONE image in row bands ---------------------------------------------------------------------
// CLAHE's grid is image-global (wow_sr.py:191-192): no output row exists before every input row has been counted.  A mosaic that
// arrives band by band (the chunks of an AOI, engine.hip enhance_impl / s2sr/dist.py) therefore feeds the histograms as it arrives
// (launch_pp_band_hist, under the compute of the windows still to come), and once the LUTs exist the image is finished band by
// band (launch_pp_band_apply R rows ahead of launch_pp_band_sharpen), each band followed by its device-to-host copy, so what is
// exposed behind the last window is one band's kernels plus the PCIe time of the image.  Same kernels, same bytes as
// launch_postprocess on the whole image.  work: the layout of postprocess_work_bytes(1, H, W, prm) = hist | lut | CLAHE'd image.
namespace {
struct BandWork { uint32_t* hist; uint8_t* lut; uint8_t* tmp; };
BandWork band_work(void* d_work, const s2sr_pp_params& prm) {
    const size_t tiles = (size_t)prm.clahe_grid * prm.clahe_grid;
    BandWork w;
    w.hist = (uint32_t*)d_work;
    w.lut = (uint8_t*)d_work + tiles * 256 * 4;
    w.tmp = w.lut + ((tiles * 256 + 255) & ~(size_t)255);
    return w;
}
}  // namespace

int pp_band_radius(const s2sr_pp_params& prm) {
    if (!(prm.stages & 2)) return 0;
    int radius = 0, taps[2 * MAXR + 1];
    gaussian_taps_q8(prm.blur_sigma, radius, taps);
    return radius;
}

hipError_t launch_pp_band_begin(int H, int W, const s2sr_pp_params& prm, void* d_work, hipStream_t st) {
    if (prm.clahe_grid <= 0 || prm.clahe_grid > 64 || H <= 0 || W <= 0) return hipErrorInvalidValue;
    if (!(prm.stages & 1)) return hipSuccess;
    return hipMemsetAsync(d_work, 0, (size_t)prm.clahe_grid * prm.clahe_grid * 256 * 4, st);
}

hipError_t launch_pp_band_hist(const uint8_t* d_img, int H, int W, const s2sr_pp_params& prm, int bgr, int y0, int y1, void* d_work,
                               hipStream_t st) {
    if (y0 < 0 || y1 > H || y0 > y1) return hipErrorInvalidValue;
    if (!(prm.stages & 1) || y0 == y1) return hipSuccess;
    PPTables* t = nullptr;
    hipError_t e = get_tables(&t);
    if (e != hipSuccess) return e;
    ClaheGeom g = clahe_geom(H, W, prm);
    if (g.th <= 0 || g.tw <= 0) return hipErrorInvalidValue;
    g.y0 = y0; g.y1 = y1; g.bgr = bgr;
    // tile rows the band can reach: its own rows, and -- when it holds a row the bottom padding reflects -- every tile row below
    // (the kernel checks row by row; tiny images whose padding bounces more than once take every tile row)
    const int pad = g.eh - H;
    int ty_lo = y0 / g.th, ty_hi = (y1 - 1) / g.th;
    if (pad > 0 && (pad >= H || y1 > H - 1 - pad)) ty_hi = g.grid - 1;
    if (pad >= H) ty_lo = 0;
    // chunks of g.split padded rows, ~32k pixels per workgroup (each stages 14 KiB of tables), at least one row per wave
    g.split = 32768 / g.tw;
    if (g.split < 4) g.split = 4;
    if (g.split > g.th) g.split = g.th;
    const int chunks = (g.th + g.split - 1) / g.split;
    hipLaunchKernelGGL(clahe_hist_rows_kernel, dim3((unsigned)g.grid, (unsigned)((ty_hi - ty_lo + 1) * chunks)), dim3(256), 0, st, d_img, g,
                       ty_lo, chunks, t, band_work(d_work, prm).hist);
    return hipGetLastError();
}

hipError_t launch_pp_band_lut(int H, int W, const s2sr_pp_params& prm, void* d_work, hipStream_t st) {
    if (!(prm.stages & 1)) return hipSuccess;
    ClaheGeom g = clahe_geom(H, W, prm);
    if (g.th <= 0 || g.tw <= 0) return hipErrorInvalidValue;
    const BandWork w = band_work(d_work, prm);
    hipLaunchKernelGGL(clahe_lut_kernel, dim3((unsigned)(prm.clahe_grid * prm.clahe_grid)), dim3(256), 0, st, w.hist, g, w.lut);
    return hipGetLastError();
}

// rows [y0, y1) of d_img through the LUTs into the same rows of the work area's image
hipError_t launch_pp_band_apply(const uint8_t* d_img, int H, int W, const s2sr_pp_params& prm, int bgr, int y0, int y1, void* d_work,
                                hipStream_t st) {
    if (y0 < 0 || y1 > H || y0 > y1) return hipErrorInvalidValue;
    if (y0 == y1) return hipSuccess;
    const BandWork w = band_work(d_work, prm);
    const size_t off = (size_t)y0 * W * 3, nb = (size_t)(y1 - y0) * W * 3;
    if (!(prm.stages & 1)) return hipMemcpyAsync(w.tmp + off, d_img + off, nb, hipMemcpyDeviceToDevice, st);
    PPTables* t = nullptr;
    hipError_t e = get_tables(&t);
    if (e != hipSuccess) return e;
    ClaheGeom g = clahe_geom(H, W, prm);
    if (g.th <= 0 || g.tw <= 0) return hipErrorInvalidValue;
    g.H = y1 - y0;                 // the kernels index the rows they were handed ...
    g.y0 = y0; g.bgr = bgr;        // ... and place them in the grid of the whole image
    const uint8_t* src = d_img + off;
    uint8_t* dst = w.tmp + off;
    const size_t total = (size_t)(y1 - y0) * W;
    if ((((uintptr_t)src | (uintptr_t)dst) & 3) == 0 && total < (1ull << 32)) {
        const size_t ng = (total + 3) / 4;
        const unsigned grid4 = (unsigned)((ng + 255) / 256 > 2048 ? 2048 : (ng + 255) / 256);
        hipLaunchKernelGGL(clahe_apply4_kernel, dim3(grid4), dim3(256), 0, st, src, g, 1, t, w.lut, dst);
    } else {
        const unsigned grid = (unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        hipLaunchKernelGGL(clahe_apply_kernel, dim3(grid), dim3(256), 0, st, src, g, 1, t, w.lut, dst);
    }
    return hipGetLastError();
}

// rows [y0, y1) of the final image from the work area's image (whose rows [y0 - R, y1 + R) within the image must have been
// applied) into the same rows of d_out ([H, W, 3]); swap_out: R and B exchanged on the way out
hipError_t launch_pp_band_sharpen(int H, int W, const s2sr_pp_params& prm, int bgr, int swap_out, int y0, int y1, void* d_work,
                                  uint8_t* d_out, hipStream_t st) {
    if (y0 < 0 || y1 > H || y0 > y1) return hipErrorInvalidValue;
    if (y0 == y1) return hipSuccess;
    PPTables* t = nullptr;
    hipError_t e = get_tables(&t);
    if (e != hipSuccess) return e;
    const BandWork w = band_work(d_work, prm);
    SharpParams sp{};
    sp.H = H; sp.W = W; sp.B = 1;
    sp.y_begin = y0; sp.y_end = y1;
    sp.bgr = bgr; sp.swap_out = swap_out;
    sp.radius = 0;
    if (prm.stages & 2) gaussian_taps_q8(prm.blur_sigma, sp.radius, sp.taps);
    sp.w_img = prm.w_img; sp.w_blur = prm.w_blur;
    sp.do_veg = (prm.stages & 4) ? 1 : 0; sp.hue_lo = prm.hue_lo; sp.hue_hi = prm.hue_hi; sp.sat_gain = prm.sat_gain;
    const int rows = y1 - y0;
    const unsigned grid = (unsigned)(((W + ST - 1) / ST) * ((rows + ST - 1) / ST));
    const unsigned grid_r = (unsigned)(((W + SW - 1) / SW) * ((rows + SH - 1) / SH));
    const size_t total_bytes = (size_t)H * W * 3;
    if (sp.radius == 4) hipLaunchKernelGGL(sharpen_veg_kernel_r<4>, dim3(grid_r), dim3(256), 0, st, w.tmp, sp, total_bytes, t, d_out);
    else if (sp.radius == 5) hipLaunchKernelGGL(sharpen_veg_kernel_r<5>, dim3(grid_r), dim3(256), 0, st, w.tmp, sp, total_bytes, t, d_out);
    else if (sp.radius == 3) hipLaunchKernelGGL(sharpen_veg_kernel_r<3>, dim3(grid_r), dim3(256), 0, st, w.tmp, sp, total_bytes, t, d_out);
    else hipLaunchKernelGGL(sharpen_veg_kernel, dim3(grid), dim3(256), 0, st, w.tmp, sp, t, d_out);   // radius 0: veg / copy only
    return hipGetLastError();
}

}  // namespace s2sr
