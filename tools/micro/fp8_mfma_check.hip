// Checks the operand layout assumed for v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3):
// lane l holds row/col l&31 and the 32 consecutive K bytes of half l>>5; D as the fp16 32x32 form.
// Also probes v_cvt_pk_fp8_f32 rounding / saturation.  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static float dec(unsigned char b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) v = m * ldexpf(1.f, -9);
    else v = (1.f + m / 8.f) * ldexpf(1.f, e - 7);
    return s ? -v : v;
}

__global__ void k(const unsigned char* A, const unsigned char* B, float* D, int sa, int sb) {
    const int l = threadIdx.x;
    v8i a = *(const v8i*)(A + ((l & 31) * 64 + (l >> 5) * 32));
    v8i b = *(const v8i*)(B + ((l & 31) * 64 + (l >> 5) * 32));
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        D[row * 32 + col] = acc[r];
    }
}
__global__ void cvt(const float* f, unsigned char* o, int n) {
    const int i = threadIdx.x;
    if (2 * i + 1 < n) {
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2 * i], f[2 * i + 1], 0, false);
        o[2 * i] = w & 255; o[2 * i + 1] = (w >> 8) & 255;
    }
}
int main() {
    unsigned char hA[32 * 64], hB[32 * 64];
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) {
        do { hA[i] = rand() & 255; } while ((hA[i] & 0x7f) == 0x7f || ((hA[i] >> 3) & 15) > 9);
        do { hB[i] = rand() & 255; } while ((hB[i] & 0x7f) == 0x7f || ((hB[i] >> 3) & 15) > 9);
    }
    unsigned char *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int sb : {127, 116}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, 127, sb);
        float hD[32 * 32];
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                double s = 0;
                for (int kk = 0; kk < 64; ++kk) s += (double)dec(hA[r * 64 + kk]) * dec(hB[c * 64 + kk]);
                s *= ldexp(1.0, sb - 127);
                maxerr = fmax(maxerr, fabs(s - hD[r * 32 + c])); maxref = fmax(maxref, fabs(s));
            }
        printf("scale_b=%d: max |ref| %.4g  max err %.3g  (D[row=A row][col=B row])\n", sb, maxref, maxerr);
    }
    float hf[16] = {0.3f, 1.0f, 1.0625f, 1.1875f, 447.f, 448.f, 449.f, 480.f, 1000.f, -1000.f, 1e-3f, 2e-3f, 0.0009765625f, 1e30f, -0.f, 17.f};
    float* df; unsigned char* dout; unsigned char ho[16];
    hipMalloc(&df, sizeof hf); hipMalloc(&dout, 16);
    hipMemcpy(df, hf, sizeof hf, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt, dim3(1), dim3(8), 0, 0, df, dout, 16);
    hipMemcpy(ho, dout, 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("cvt %g -> 0x%02x = %g\n", hf[i], ho[i], dec(ho[i]));
    return 0;
}
