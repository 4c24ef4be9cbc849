// Per-instruction SIMD throughput (4 waves/SIMD) of the bit/packed ops considered for a bf16 P split.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#define N_IT 1000
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ void k(float* out, long long* t0s, long long* t1s) {
    float a0 = threadIdx.x * 0.001f + 0.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = threadIdx.x, u1 = 3, u2 = 5, u3 = 7, u4 = 9, u5 = 11, u6 = 13, u7 = 15;
    unsigned sel = 0x07060302u;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 d0 = {a0, a1}, d1 = {a2, a3}, d2 = {a4, a5}, d3 = {a6, a7}, e0 = {1.f, 2.f}, e1 = {1.f, 0.5f}, e2 = {0.25f, 1.f}, e3 = {1.f, 1.f};
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < N_IT; ++it) {
        if (MODE == 0) { asm volatile("v_and_b32 %0, 0xffff0000, %8\n v_and_b32 %1, 0xffff0000, %9\n v_and_b32 %2, 0xffff0000, %10\n v_and_b32 %3, 0xffff0000, %11\n v_and_b32 %4, 0xffff0000, %12\n v_and_b32 %5, 0xffff0000, %13\n v_and_b32 %6, 0xffff0000, %14\n v_and_b32 %7, 0xffff0000, %15" : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)); }
        if (MODE == 1) { asm volatile("v_perm_b32 %0, %8, %9, %16\n v_perm_b32 %1, %9, %10, %16\n v_perm_b32 %2, %10, %11, %16\n v_perm_b32 %3, %11, %12, %16\n v_perm_b32 %4, %12, %13, %16\n v_perm_b32 %5, %13, %14, %16\n v_perm_b32 %6, %14, %15, %16\n v_perm_b32 %7, %15, %8, %16" : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(sel)); }
        if (MODE == 2) { asm volatile("v_pk_add_f32 %0, %0, %4 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %5 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, %2, %6 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %7 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %0, %0, %4 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %5 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %2, %2, %6 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %7 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e0), "v"(e1), "v"(e2), "v"(e3)); }
        if (MODE == 3) { asm volatile("v_cvt_pk_bf16_f32 %0, %8, %9\n v_cvt_pk_bf16_f32 %1, %9, %10\n v_cvt_pk_bf16_f32 %2, %10, %11\n v_cvt_pk_bf16_f32 %3, %11, %12\n v_cvt_pk_bf16_f32 %4, %12, %13\n v_cvt_pk_bf16_f32 %5, %13, %14\n v_cvt_pk_bf16_f32 %6, %14, %15\n v_cvt_pk_bf16_f32 %7, %15, %8" : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)); }
        if (MODE == 4) { asm volatile("v_lshlrev_b32 %0, 16, %8\n v_lshlrev_b32 %1, 16, %9\n v_lshlrev_b32 %2, 16, %10\n v_lshlrev_b32 %3, 16, %11\n v_lshlrev_b32 %4, 16, %12\n v_lshlrev_b32 %5, 16, %13\n v_lshlrev_b32 %6, 16, %14\n v_lshlrev_b32 %7, 16, %15" : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)); }
        if (MODE == 5) { asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %6\n v_pk_mul_f32 %3, %3, %7\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %6\n v_pk_mul_f32 %3, %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e0), "v"(e1), "v"(e2), "v"(e3)); }
        if (MODE == 6) { asm volatile("v_and_or_b32 %0, %8, %16, %9\n v_and_or_b32 %1, %9, %16, %10\n v_and_or_b32 %2, %10, %16, %11\n v_and_or_b32 %3, %11, %16, %12\n v_and_or_b32 %4, %12, %16, %13\n v_and_or_b32 %5, %13, %16, %14\n v_and_or_b32 %6, %14, %16, %15\n v_and_or_b32 %7, %15, %16, %8" : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(sel)); }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + d2[0] + d3[1] + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7);
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
    run<0>("v_and_b32");
    run<1>("v_perm_b32");
    run<2>("v_pk_add_f32");
    run<3>("v_cvt_pk_bf16_f32");
    run<4>("v_lshlrev_b32");
    run<5>("v_pk_mul_f32");
    run<6>("v_and_or_b32");
    return 0;
}
