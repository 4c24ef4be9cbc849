// check hi/lo f16 split helpers and the f16 MFMA accumulate precision on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ unsigned pk(float a, float b) { unsigned r; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ unsigned pk_lo(unsigned h, float p0, float p1) {
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h), "v"(p0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h), "v"(p1));
    return pk(l0, l1);
}
__global__ void k(const float* p, unsigned* hi, unsigned* lo, float* d) {
    int i = threadIdx.x;
    unsigned h = pk(p[2 * i], p[2 * i + 1]);
    hi[i] = h; lo[i] = pk_lo(h, p[2 * i], p[2 * i + 1]);
    // MFMA precision: A row = [x, x*2^-11, 0...], B col = [1, 1, 0...] -> x*(1+2^-11) ; and a sum of 32 terms of decreasing size
    union { unsigned u[4]; f16x8 v; } a, b;
    for (int j = 0; j < 4; ++j) { a.u[j] = 0; b.u[j] = 0; }
    int kg = i >> 4;
    if (kg == 0) { a.v[0] = (_Float16)8.0f; a.v[1] = (_Float16)(8.0f / 2048.f); a.v[2] = (_Float16)(1.0f / 1024.f / 1024.f); b.v[0] = (_Float16)1.0f; b.v[1] = (_Float16)1.0f; b.v[2] = (_Float16)1.0f; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c, 0, 0, 0);
    d[i] = c[0];
}
int main() {
    float hp[128]; for (int i = 0; i < 128; ++i) hp[i] = 8.0f * expf(-0.37f * i) * 1.2345f;
    float *p, *d; unsigned *hi, *lo; hipMalloc(&p, 512); hipMalloc(&hi, 256); hipMalloc(&lo, 256); hipMalloc(&d, 256);
    hipMemcpy(p, hp, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, hi, lo, d);
    unsigned hh[64], hl[64]; float hd[64];
    hipMemcpy(hh, hi, 256, hipMemcpyDeviceToHost); hipMemcpy(hl, lo, 256, hipMemcpyDeviceToHost); hipMemcpy(hd, d, 256, hipMemcpyDeviceToHost);
    auto h2f = [](unsigned short h) { union { unsigned short s; _Float16 f; } u; u.s = h; return (float)u.f; };
    double worst = 0;
    for (int i = 0; i < 64; ++i) for (int e = 0; e < 2; ++e) {
        float p0 = hp[2 * i + e]; float rec = h2f((hh[i] >> (16 * e)) & 0xffff) + h2f((hl[i] >> (16 * e)) & 0xffff);
        double rel = fabs((double)rec - p0) / p0; if (i < 4) printf("p=%g hi=%g lo=%g rel=%g\n", p0, h2f((hh[i] >> (16 * e)) & 0xffff), h2f((hl[i] >> (16 * e)) & 0xffff), rel);
        if (p0 > 1e-3 && rel > worst) worst = rel;
    }
    printf("worst rel err of hi+lo for p>1e-3: %g\n", worst);
    printf("mfma: 8 + 8/2048 + 2^-20 = %.10f (got %.10f)\n", 8.0 + 8.0 / 2048 + 1.0 / 1048576, hd[0]);
    return 0;
}
