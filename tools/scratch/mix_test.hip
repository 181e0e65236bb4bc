// does v_fma_mix_f32 (asm, op_sel forms used in conv_trunk.hip) agree bit-for-bit with the plain C++ expression?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool HI> __device__ float fma_f32_f32_h(float a, float b, uint32_t c16) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c16));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c16));
    return r;
}
template <bool HI> __device__ float fma_h_f32_f32(uint32_t a16, float b, float c) {
    float r;
    if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a16), "v"(b), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a16), "v"(b), "v"(c));
    return r;
}
__global__ void k(const float* x, const uint32_t* h, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[2 * i], c = x[2 * i + 1];
    const uint32_t w = h[i];
    const f16x2 hh = __builtin_bit_cast(f16x2, w);
    o[8 * i + 0] = fma_f32_f32_h<false>(a, 0.000244140625f, w);
    o[8 * i + 1] = __builtin_fmaf(a, 0.000244140625f, (float)hh[0]);
    o[8 * i + 2] = fma_f32_f32_h<true>(a, 0.000244140625f, w);
    o[8 * i + 3] = __builtin_fmaf(a, 0.000244140625f, (float)hh[1]);
    o[8 * i + 4] = fma_h_f32_f32<false>(w, -4096.0f, c);
    o[8 * i + 5] = __builtin_fmaf((float)hh[0], -4096.0f, c);
    o[8 * i + 6] = fma_h_f32_f32<true>(w, -4096.0f, c);
    o[8 * i + 7] = __builtin_fmaf((float)hh[1], -4096.0f, c);
    {   // the lo encode: old form vs new form on the same v
        f32x2 vv; vv[0] = a * 3.7f; vv[1] = c * 0.01f;
        const uint32_t hp = __builtin_bit_cast(uint32_t, __builtin_convertvector(vv, f16x2));
        const f16x2 h2 = __builtin_bit_cast(f16x2, hp);
        const f32x2 vs = vv * 4096.0f;
        const float qn0 = fma_h_f32_f32<false>(hp, -4096.0f, vs[0]), qn1 = fma_h_f32_f32<true>(hp, -4096.0f, vs[1]);
        const float qo0 = __fmul_rn(__fsub_rn(vv[0], (float)h2[0]), 4096.0f), qo1 = __fmul_rn(__fsub_rn(vv[1], (float)h2[1]), 4096.0f);
        o[8 * i + 4] = qn0; o[8 * i + 5] = qo0; o[8 * i + 6] = qn1; o[8 * i + 7] = qo1;
    }
    // cvt_pk vs single conversions
    f32x2 v; v[0] = a * 3.7f; v[1] = c * 0.01f;
    const f16x2 p = __builtin_convertvector(v, f16x2);
    const f16 s0 = (f16)v[0], s1 = (f16)v[1];
    if (__builtin_bit_cast(unsigned short, p[0]) != __builtin_bit_cast(unsigned short, s0) || __builtin_bit_cast(unsigned short, p[1]) != __builtin_bit_cast(unsigned short, s1)) o[8 * i] = __builtin_nanf("");
}
int main() {
    const int n = 1 << 16;
    float* hx = (float*)malloc(n * 2 * 4); uint32_t* hh = (uint32_t*)malloc(n * 4); float* ho = (float*)malloc(n * 8 * 4);
    srand(1);
    for (int i = 0; i < 2 * n; ++i) hx[i] = ((rand() % 20001) - 10000) * 0.0137f * ((i & 7) ? 1.f : 1e-3f);
    for (int i = 0; i < n; ++i) { f16 a = (f16)(((rand() % 2001) - 1000) * 0.013f), b = (f16)(((rand() % 2001) - 1000) * 0.0007f); unsigned short ua, ub; memcpy(&ua, &a, 2); memcpy(&ub, &b, 2); hh[i] = ua | ((uint32_t)ub << 16); }
    float *dx, *dout; uint32_t* dh;
    hipMalloc(&dx, n * 8); hipMalloc(&dh, n * 4); hipMalloc(&dout, n * 32);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice); hipMemcpy(dh, hh, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dh, dout, n);
    hipMemcpy(ho, dout, n * 32, hipMemcpyDeviceToHost);
    int bad[4] = {0, 0, 0, 0}, nan = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 4; ++j) {
            if (std::isnan(ho[8 * i + 2 * j])) { ++nan; continue; }
            if (memcmp(&ho[8 * i + 2 * j], &ho[8 * i + 2 * j + 1], 4)) ++bad[j];
        }
    printf("mismatches: f32*f32+h.lo %d, +h.hi %d, h.lo*f32+f32 %d, h.hi*f32+f32 %d, cvt_pk mismatch (nan) %d of %d\n", bad[0], bad[1], bad[2], bad[3], nan, n);
    return 0;
}
