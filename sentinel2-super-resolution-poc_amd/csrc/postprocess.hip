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
    if (!g_d_tables[dev]) {
        PPTables* h = new PPTables();
        build_tables(*h);
        PPTables* d = nullptr;
        e = hipMalloc((void**)&d, sizeof(PPTables));
        if (e == hipSuccess) e = hipMemcpy(d, h, sizeof(PPTables), hipMemcpyHostToDevice);
        delete h;
        if (e != hipSuccess) return e;
        g_d_tables[dev] = d;
    }
    *out = g_d_tables[dev];
    return hipSuccess;
}

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ int reflect101(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
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

__device__ __forceinline__ void rgb2hsv(const PPTables* __restrict__ t, int r, int g, int b, int& h, int& s, int& v) {
    v = max(max(r, g), b);
    const int vmin = min(min(r, g), b);
    const int diff = v - vmin;
    s = (diff * t->sdiv[v] + (1 << 11)) >> 12;
    int hh = (v == r) ? (g - b) : ((v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff));
    hh = (hh * t->hdiv180[diff] + (1 << 11)) >> 12;
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
};

// ---- kernel 1: per-tile L histograms --------------------------------------------------------
// grid.x = B * grid*grid * SPLIT ; every workgroup takes a horizontal band of one CLAHE tile.
constexpr int HIST_SPLIT = 8;
__global__ void __launch_bounds__(256) clahe_hist_kernel(const uint8_t* __restrict__ rgb, ClaheGeom gm,
                                                         const PPTables* __restrict__ t, uint32_t* __restrict__ hist) {
    __shared__ uint32_t sh[256];
    sh[threadIdx.x] = 0;
    __syncthreads();
    const int ntile = gm.grid * gm.grid;
    int id = blockIdx.x;
    const int part = id % HIST_SPLIT; id /= HIST_SPLIT;
    const int tile = id % ntile;
    const int img = id / ntile;
    const int ty = tile / gm.grid, tx = tile % gm.grid;
    const int rows_per = (gm.th + HIST_SPLIT - 1) / HIST_SPLIT;
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
    const size_t npx = (size_t)gm.H * gm.W, total = npx * B;
    const float inv_tw = 1.0f / (float)gm.tw, inv_th = 1.0f / (float)gm.th;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int img = (int)(i / npx);
        const size_t rem = i - (size_t)img * npx;
        const int y = (int)(rem / gm.W), x = (int)(rem % gm.W);
        const uint8_t* p = rgb + i * 3;
        int L, A, Bc;
        rgb2lab(t, p[0], p[1], p[2], L, A, Bc);
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
        o[0] = (uint8_t)r;
        o[1] = (uint8_t)g;
        o[2] = (uint8_t)b;
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
};

__global__ void __launch_bounds__(256) sharpen_veg_kernel(const uint8_t* __restrict__ in, SharpParams sp,
                                                          const PPTables* __restrict__ t, uint8_t* __restrict__ out) {
    __shared__ uint8_t s_in[(ST + 2 * MAXR) * (ST + 2 * MAXR) * 3];
    __shared__ uint16_t s_h[(ST + 2 * MAXR) * ST * 3];
    const int tilesX = (sp.W + ST - 1) / ST, tilesY = (sp.H + ST - 1) / ST;
    int id = blockIdx.x;
    const int tx = id % tilesX; id /= tilesX;
    const int ty = id % tilesY;
    const int img = id / tilesY;
    const uint8_t* src = in + (size_t)img * sp.H * sp.W * 3;
    uint8_t* dst = out + (size_t)img * sp.H * sp.W * 3;
    const int r = sp.radius, ext = ST + 2 * r;
    const int y0 = ty * ST, x0 = tx * ST;
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
        if (y >= sp.H || x >= sp.W) continue;
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
        if (sp.do_veg) {
            int h, s, v;
            rgb2hsv(t, px[0], px[1], px[2], h, s, v);
            if (h > sp.hue_lo && h < sp.hue_hi) {
                // float32 S*gain, clip to [0,255], astype(uint8) == truncation (wow_sr.py:204-207)
                float fs = (float)s * sp.sat_gain;
                fs = fminf(fmaxf(fs, 0.f), 255.f);
                s = (int)fs;
            }
            hsv2rgb(h, s, v, px[0], px[1], px[2]);
        }
        uint8_t* o = dst + ((size_t)y * sp.W + x) * 3;
        o[0] = (uint8_t)px[0];
        o[1] = (uint8_t)px[1];
        o[2] = (uint8_t)px[2];
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
        const ClaheGeom g = clahe_geom(H, W, prm);
        if (g.th <= 0 || g.tw <= 0) return hipErrorInvalidValue;
        e = hipMemsetAsync(d_hist, 0, tiles * 256 * 4, st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(clahe_hist_kernel, dim3((unsigned)(tiles * HIST_SPLIT)), dim3(256), 0, st, cur, g, t, d_hist);
        hipLaunchKernelGGL(clahe_lut_kernel, dim3((unsigned)tiles), dim3(256), 0, st, d_hist, g, d_lut);
        uint8_t* o = (s2 || s3) ? d_tmp : d_out;
        const size_t total = (size_t)B * H * W;
        const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
        hipLaunchKernelGGL(clahe_apply_kernel, dim3(grid), dim3(256), 0, st, cur, g, B, t, d_lut, o);
        cur = o;
    }
    if (s2 || s3) {
        SharpParams sp{};
        sp.H = H; sp.W = W; sp.B = B;
        sp.radius = 0;
        if (s2) gaussian_taps_q8(prm.blur_sigma, sp.radius, sp.taps);
        sp.w_img = prm.w_img; sp.w_blur = prm.w_blur;
        sp.do_veg = s3 ? 1 : 0; sp.hue_lo = prm.hue_lo; sp.hue_hi = prm.hue_hi; sp.sat_gain = prm.sat_gain;
        const unsigned grid = (unsigned)(((W + ST - 1) / ST) * ((H + ST - 1) / ST) * B);
        hipLaunchKernelGGL(sharpen_veg_kernel, dim3(grid), dim3(256), 0, st, cur, sp, t, d_out);
    } else if (!s1) {
        e = hipMemcpyAsync(d_out, d_rgb, (size_t)B * H * W * 3, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

}  // namespace s2sr
