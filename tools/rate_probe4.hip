// Per-instruction SIMD throughput (4 waves/SIMD) of the conversion ops used by the attention kernel's P split.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#define N_IT 1000
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ void k(float* out, long long* t0s, long long* t1s) {
    float a0 = threadIdx.x * 0.001f + 0.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = threadIdx.x, u1 = 3, u2 = 5, u3 = 7, u4 = 9, u5 = 11, u6 = 13, u7 = 15;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < N_IT; ++it) {
        if (MODE == 0) { asm volatile("v_cvt_pk_f16_f32 %0, %8, %9\n v_cvt_pk_f16_f32 %1, %9, %10\n v_cvt_pk_f16_f32 %2, %10, %11\n v_cvt_pk_f16_f32 %3, %11, %12\n v_cvt_pk_f16_f32 %4, %12, %13\n v_cvt_pk_f16_f32 %5, %13, %14\n v_cvt_pk_f16_f32 %6, %14, %15\n v_cvt_pk_f16_f32 %7, %15, %8" : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)); }
        if (MODE == 1) { asm volatile("v_fma_mix_f32 %0, %8, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %9, -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %10, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %11, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %12, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %13, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %14, -1.0, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %15, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(u7)); }
        if (MODE == 2) { asm volatile("v_cvt_f16_f32 %0, %8\n v_cvt_f16_f32 %1, %9\n v_cvt_f16_f32 %2, %10\n v_cvt_f16_f32 %3, %11\n v_cvt_f16_f32 %4, %12\n v_cvt_f16_f32 %5, %13\n v_cvt_f16_f32 %6, %14\n v_cvt_f16_f32 %7, %15" : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)); }
        if (MODE == 3) { asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %9\n v_sub_f32 %2, %2, %10\n v_sub_f32 %3, %3, %11\n v_sub_f32 %4, %4, %12\n v_sub_f32 %5, %5, %13\n v_sub_f32 %6, %6, %14\n v_sub_f32 %7, %7, %15" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(u7)); }
        if (MODE == 4) { asm volatile("v_cvt_pkrtz_f16_f32 %0, %8, %9\n v_cvt_pkrtz_f16_f32 %1, %9, %10\n v_cvt_pkrtz_f16_f32 %2, %10, %11\n v_cvt_pkrtz_f16_f32 %3, %11, %12\n v_cvt_pkrtz_f16_f32 %4, %12, %13\n v_cvt_pkrtz_f16_f32 %5, %13, %14\n v_cvt_pkrtz_f16_f32 %6, %14, %15\n v_cvt_pkrtz_f16_f32 %7, %15, %8" : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)); }
        if (MODE == 5) { asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        if (MODE == 6) { asm volatile("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %9\n v_cvt_f32_f16 %2, %10\n v_cvt_f32_f16 %3, %11\n v_cvt_f32_f16 %4, %12\n v_cvt_f32_f16 %5, %13\n v_cvt_f32_f16 %6, %14\n v_cvt_f32_f16 %7, %15" : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(u7)); }
        if (MODE == 7) { asm volatile("v_exp_f16 %0, %8\n v_exp_f16 %1, %9\n v_exp_f16 %2, %10\n v_exp_f16 %3, %11\n v_exp_f16 %4, %12\n v_exp_f16 %5, %13\n v_exp_f16 %6, %14\n v_exp_f16 %7, %15" : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7) : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "v"(u4), "v"(u5), "v"(u6), "v"(u7)); }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { t0s[threadIdx.x >> 6] = t0; t1s[threadIdx.x >> 6] = t1; }
}
template <int MODE> void run(const char* name) {
    float* o; long long *a, *b; hipMalloc(&o, 1 << 22); hipMalloc(&a, 128); hipMalloc(&b, 128);
    const int waves = 4;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * waves), 0, 0, o, a, b);
    hipDeviceSynchronize();
    long long h0[16], h1[16];
    hipMemcpy(h0, a, 8 * 4 * waves, hipMemcpyDeviceToHost); hipMemcpy(h1, b, 8 * 4 * waves, hipMemcpyDeviceToHost);
    long long mn = *std::min_element(h0, h0 + 4 * waves), mx = *std::max_element(h1, h1 + 4 * waves);
    printf("%-24s %.2f cycles per wave-instruction per SIMD\n", name, (double)(mx - mn) / N_IT / waves / 8);
}
int main() {
    run<0>("v_cvt_pk_f16_f32"); run<4>("v_cvt_pkrtz_f16_f32"); run<1>("v_fma_mix_f32"); run<2>("v_cvt_f16_f32"); run<6>("v_cvt_f32_f16");
    run<3>("v_sub_f32"); run<5>("v_exp_f32"); run<7>("v_exp_f16");
    return 0;
}
