// Probe of v_cvt_scalef32_pk_fp8_f32 on gfx950 (no public semantics offline): what does the scale operand do, and does the
// conversion saturate?  Prints, per (value, scale): the byte produced, and the bytes of the plain v_cvt_pk_fp8_f32 of
// value * s and value / s for comparison.   hipcc --offload-arch=gfx950 tools/micro/cvt_scale_probe.hip -o /tmp/cvt_probe && /tmp/cvt_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef short v2s __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, const float* scale, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i >= n) return;
    v2s old = {0, 0};
    v2s r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, in[i], in[i], scale[i], false);
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(in[i] * scale[i], in[i] / scale[i], 0, false);
    out[2 * i] = (unsigned)(unsigned short)r[0];
    out[2 * i + 1] = (unsigned)p & 0xffffu;
}
static float dec(unsigned b) {
    int e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? ldexpf((float)m, -9) : ldexpf((float)(8 + m), e - 10);
    if ((b & 0x7f) == 0x7f) v = NAN;
    return (b & 0x80) ? -v : v;
}
int main() {
    const float vals[] = {1.0f, 1.0625f, 0.3f, -2.7f, 100.f, 447.f, 448.f, 460.f, 500.f, 1000.f, 1e6f, -1e6f, 0.001f, 3.0e-5f};
    const float scs[] = {1.0f, 2.0f, 0.5f, 4096.f, 1.0f / 4096.f, 3.0f, 1.5f};
    float hv[128], hs[128];
    int n = 0;
    for (float s : scs) for (float v : vals) { hv[n] = v; hs[n] = s; ++n; }
    float *dv, *ds; unsigned* dout;
    hipMalloc(&dv, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dout, n * 8);
    hipMemcpy(dv, hv, n * 4, hipMemcpyHostToDevice); hipMemcpy(ds, hs, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, dv, ds, dout, n);
    unsigned ho[256];
    hipMemcpy(ho, dout, n * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        unsigned b = ho[2 * i] & 0xff, pm = ho[2 * i + 1] & 0xff, pd = (ho[2 * i + 1] >> 8) & 0xff;
        printf("v %12g scale %10g : scalef32 byte %02x = %8g | plain(v*s) %02x = %8g | plain(v/s) %02x = %8g\n", hv[i], hs[i], b, dec(b), pm,
               dec(pm), pd, dec(pd));
    }
    return 0;
}
