// Is there a software-visible hazard between v_cvt_pk_f16_f32 / v_pk_mul_f32 and an immediately following v_fma_mix_f32
// that reads the fresh result (op_sel on the fp16 operand)?  Back-to-back in one asm block vs. separated by s_nop 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__global__ void k(const float* x, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v0 = x[2 * i], v1 = x[2 * i + 1];
    const float enc = 4096.0f, nenc = -4096.0f;
    float a0, a1, b0, b1, c0, c1;
    uint32_t h;
    float s0, s1;
    // (A) adjacent
    asm volatile("v_cvt_pk_f16_f32 %2, %5, %6\n\t"
                 "v_mul_f32 %3, %7, %5\n\t"
                 "v_mul_f32 %4, %7, %6\n\t"
                 "v_fma_mix_f32 %0, %2, %8, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %1, %2, %8, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                 : "=&v"(a0), "=&v"(a1), "=&v"(h), "=&v"(s0), "=&v"(s1) : "v"(v0), "v"(v1), "v"(enc), "v"(nenc));
    // (B) the mix right behind the conversion (distance 0)
    asm volatile("v_mul_f32 %3, %7, %5\n\t"
                 "v_mul_f32 %4, %7, %6\n\t"
                 "v_cvt_pk_f16_f32 %2, %5, %6\n\t"
                 "v_fma_mix_f32 %0, %2, %8, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %1, %2, %8, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                 : "=&v"(b0), "=&v"(b1), "=&v"(h), "=&v"(s0), "=&v"(s1) : "v"(v0), "v"(v1), "v"(enc), "v"(nenc));
    // (C) separated
    asm volatile("v_cvt_pk_f16_f32 %2, %5, %6\n\t"
                 "v_mul_f32 %3, %7, %5\n\t"
                 "v_mul_f32 %4, %7, %6\n\t"
                 "s_nop 7\n\t"
                 "v_fma_mix_f32 %0, %2, %8, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                 "s_nop 7\n\t"
                 "v_fma_mix_f32 %1, %2, %8, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                 : "=&v"(c0), "=&v"(c1), "=&v"(h), "=&v"(s0), "=&v"(s1) : "v"(v0), "v"(v1), "v"(enc), "v"(nenc));
    o[6 * i + 0] = a0; o[6 * i + 1] = a1; o[6 * i + 2] = b0; o[6 * i + 3] = b1; o[6 * i + 4] = c0; o[6 * i + 5] = c1;
}
int main() {
    const int n = 1 << 18;
    float* hx = (float*)malloc(n * 8); float* ho = (float*)malloc(n * 24);
    srand(3);
    for (int i = 0; i < 2 * n; ++i) hx[i] = ((rand() % 200001) - 100000) * 0.000137f * ((i & 3) ? 1.f : 37.f);
    float *dx, *dout;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dout, n * 24);
    (void)hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    (void)hipMemcpy(ho, dout, n * 24, hipMemcpyDeviceToHost);
    int badA = 0, badB = 0;
    for (int i = 0; i < n; ++i) {
        if (memcmp(&ho[6 * i], &ho[6 * i + 4], 8)) ++badA;
        if (memcmp(&ho[6 * i + 2], &ho[6 * i + 4], 8)) ++badB;
    }
    printf("vs separated: adjacent(A) mismatches %d, mix right behind cvt (B) mismatches %d of %d\n", badA, badB, n);
    return 0;
}
