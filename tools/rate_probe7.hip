// Per-instruction SIMD throughput (4 waves/SIMD) of the candidates for a cheaper P = hi + lo split in the attention kernel:
// truncating / rounding packed conversions, the packed-f32 ops, bit masks.  Units as in rate_probe6 (clock64 ticks per
// wave-instruction per SIMD; a plain f32 VOP2 reads ~2.1-2.4).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#define N_IT 1000
#define OUTS "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7)
#define INS "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(m)
#define OP2(op) asm volatile(op " %0, %8, %16\n " op " %1, %9, %16\n " op " %2, %10, %16\n " op " %3, %11, %16\n " op " %4, %12, %16\n " op " %5, %13, %16\n " op " %6, %14, %16\n " op " %7, %15, %16" : OUTS : INS)
#define OP2P(op) asm volatile(op " %0, %8, %9\n " op " %1, %9, %10\n " op " %2, %10, %11\n " op " %3, %11, %12\n " op " %4, %12, %13\n " op " %5, %13, %14\n " op " %6, %14, %15\n " op " %7, %15, %8" : OUTS : INS)
#define OP3(op) asm volatile(op " %0, %8, %16, %9\n " op " %1, %9, %16, %10\n " op " %2, %10, %16, %11\n " op " %3, %11, %16, %12\n " op " %4, %12, %16, %13\n " op " %5, %13, %16, %14\n " op " %6, %14, %16, %15\n " op " %7, %15, %16, %8" : OUTS : INS)
#define PK2(op) asm volatile(op " %0, %4, %8\n " op " %1, %5, %8\n " op " %2, %6, %8\n " op " %3, %7, %8\n " op " %0, %5, %8\n " op " %1, %6, %8\n " op " %2, %7, %8\n " op " %3, %4, %8" \
                             : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(xm))
#define PK3(op) asm volatile(op " %0, %4, %8, %5\n " op " %1, %5, %8, %6\n " op " %2, %6, %8, %7\n " op " %3, %7, %8, %4\n " op " %0, %5, %8, %6\n " op " %1, %6, %8, %7\n " op " %2, %7, %8, %4\n " op " %3, %4, %8, %5" \
                             : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(xm))
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, long long* t0s, long long* t1s) {
    float a0 = threadIdx.x * 0.001f + 0.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0, u5 = 0, u6 = 0, u7 = 0;
    unsigned m = 0xFFFFE000u;
    f2 x0 = {a0, a1}, x1 = {a2, a3}, x2 = {a4, a5}, x3 = {a6, a7}, xm = {1.5f, 2.5f};
    f2 w0 = {0, 0}, w1 = {0, 0}, w2 = {0, 0}, w3 = {0, 0};
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < N_IT; ++it) {
        if (MODE == 0) OP2P("v_cvt_pk_f16_f32");
        if (MODE == 1) OP2P("v_cvt_pkrtz_f16_f32");
        if (MODE == 2) OP2("v_and_b32");
        if (MODE == 3) OP2("v_sub_f32");
        if (MODE == 4) PK2("v_pk_add_f32");
        if (MODE == 5) PK2("v_pk_mul_f32");
        if (MODE == 6) PK3("v_pk_fma_f32");
        if (MODE == 7) OP2P("v_cvt_pk_bf16_f32");
        if (MODE == 8) OP3("v_perm_b32");
        if (MODE == 9) OP3("v_max3_f32");
        if (MODE == 10) OP2("v_pk_max_f16");
        if (MODE == 11) OP3("v_and_or_b32");
        if (MODE == 12) OP3("v_bfi_b32");
        if (MODE == 13) OP2("v_lshrrev_b32");
        if (MODE == 14) OP2("v_pk_add_f16");
        if (MODE == 15) OP3("v_fma_f32");
        if (MODE == 16) OP2("v_max_f32");
        if (MODE == 17) OP2("v_pk_max_u16");
        if (MODE == 18) asm volatile("v_exp_f32 %0, %8\n v_exp_f32 %1, %9\n v_exp_f32 %2, %10\n v_exp_f32 %3, %11\n v_exp_f32 %4, %12\n v_exp_f32 %5, %13\n v_exp_f32 %6, %14\n v_exp_f32 %7, %15" : OUTS : INS);
        if (MODE == 19) asm volatile("v_fma_mix_f32 %0, %8, -1.0, %9 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %9, -1.0, %10 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %10, -1.0, %11 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %11, -1.0, %12 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                                     "v_fma_mix_f32 %4, %12, -1.0, %13 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %13, -1.0, %14 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %14, -1.0, %15 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %15, -1.0, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : OUTS : INS);
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7) + w0[0] + w1[1] + w2[0] + w3[1];
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
    printf("%-24s %.2f\n", name, (double)(mx - mn) / N_IT / waves / 8);
}
int main() {
    printf("ticks per wave-instruction per SIMD\n");
    run<15>("v_fma_f32"); run<3>("v_sub_f32"); run<2>("v_and_b32"); run<13>("v_lshrrev_b32");
    run<0>("v_cvt_pk_f16_f32"); run<1>("v_cvt_pkrtz_f16_f32"); run<7>("v_cvt_pk_bf16_f32"); run<19>("v_fma_mix_f32");
    run<4>("v_pk_add_f32"); run<5>("v_pk_mul_f32"); run<6>("v_pk_fma_f32");
    run<8>("v_perm_b32"); run<11>("v_and_or_b32"); run<12>("v_bfi_b32");
    run<16>("v_max_f32"); run<9>("v_max3_f32"); run<10>("v_pk_max_f16"); run<17>("v_pk_max_u16"); run<14>("v_pk_add_f16");
    run<18>("v_exp_f32");
    return 0;
}
