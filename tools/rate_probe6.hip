// Per-instruction SIMD throughput (4 waves/SIMD) of the integer-multiply / fp64 / transcendental ops of the
// posterior step kernel (Philox rounds, fp64 log-softmax sum, expf / logf bodies).  Units as in rate_probe5
// (clock64 ticks; a plain f32 VOP2 reads ~2.1).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#define N_IT 1000
#define OUTS "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7)
#define INS "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(m)
#define OP2(op) asm volatile(op " %0, %8, %16\n " op " %1, %9, %16\n " op " %2, %10, %16\n " op " %3, %11, %16\n " op " %4, %12, %16\n " op " %5, %13, %16\n " op " %6, %14, %16\n " op " %7, %15, %16" : OUTS : INS)
#define OP1(op) asm volatile(op " %0, %8\n " op " %1, %9\n " op " %2, %10\n " op " %3, %11\n " op " %4, %12\n " op " %5, %13\n " op " %6, %14\n " op " %7, %15" : OUTS : INS)
#define OP3(op) asm volatile(op " %0, %8, %16, %9\n " op " %1, %9, %16, %10\n " op " %2, %10, %16, %11\n " op " %3, %11, %16, %12\n " op " %4, %12, %16, %13\n " op " %5, %13, %16, %14\n " op " %6, %14, %16, %15\n " op " %7, %15, %16, %8" : OUTS : INS)
template <int MODE>
__global__ void k(float* out, long long* t0s, long long* t1s) {
    unsigned a0 = threadIdx.x * 2654435761u + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15;
    unsigned u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0, u5 = 0, u6 = 0, u7 = 0;
    unsigned m = 0xD2511F53u;
    unsigned long long w0 = a0, w1 = a1, w2 = a2, w3 = a3;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < N_IT; ++it) {
        if (MODE == 0) OP2("v_mul_lo_u32");
        if (MODE == 1) OP2("v_mul_hi_u32");
        if (MODE == 2) {  // 64-bit result, carry-out to vcc
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %8, 0\n v_mad_u64_u32 %1, vcc, %5, %8, 0\n v_mad_u64_u32 %2, vcc, %6, %8, 0\n v_mad_u64_u32 %3, vcc, %7, %8, 0\n"
                         "v_mad_u64_u32 %0, vcc, %5, %8, 0\n v_mad_u64_u32 %1, vcc, %6, %8, 0\n v_mad_u64_u32 %2, vcc, %7, %8, 0\n v_mad_u64_u32 %3, vcc, %4, %8, 0"
                         : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(m) : "vcc");
        }
        if (MODE == 3) OP2("v_mul_u32_u24");
        if (MODE == 4) OP2("v_mul_hi_u32_u24");
        if (MODE == 5) OP3("v_mad_u32_u24");
        if (MODE == 6) OP2("v_xor_b32");
        if (MODE == 7) OP1("v_log_f32");
        if (MODE == 8) OP1("v_rndne_f32");
        if (MODE == 9) OP2("v_ldexp_f32");
        if (MODE == 10) OP1("v_cvt_i32_f32");
        if (MODE == 11) {
            asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                         "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(1.0));
        }
        if (MODE == 12) {
            asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
                         "v_cvt_f64_f32 %0, %5\n v_cvt_f64_f32 %1, %6\n v_cvt_f64_f32 %2, %7\n v_cvt_f64_f32 %3, %4"
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
        }
        if (MODE == 13) OP2("v_cndmask_b32");   // vcc implicit
        if (MODE == 14) OP3("v_fma_f32");
        if (MODE == 15) OP2("v_mul_f32");
        if (MODE == 16) OP1("v_exp_f32");
        if (MODE == 17) OP3("v_xad_u32");
        if (MODE == 19) asm volatile("v_cndmask_b32_e64 %0, %8, %16, s[10:11]\n v_cndmask_b32_e64 %1, %9, %16, s[10:11]\n v_cndmask_b32_e64 %2, %10, %16, s[10:11]\n v_cndmask_b32_e64 %3, %11, %16, s[10:11]\n v_cndmask_b32_e64 %4, %12, %16, s[10:11]\n v_cndmask_b32_e64 %5, %13, %16, s[10:11]\n v_cndmask_b32_e64 %6, %14, %16, s[10:11]\n v_cndmask_b32_e64 %7, %15, %16, s[10:11]" : OUTS : INS : "s10", "s11");
        if (MODE == 20) asm volatile("v_cmp_gt_f32 vcc, %8, %16\n v_cndmask_b32 %0, %8, %16, vcc\n v_cmp_gt_f32 vcc, %9, %16\n v_cndmask_b32 %1, %9, %16, vcc\n v_cmp_gt_f32 vcc, %10, %16\n v_cndmask_b32 %2, %10, %16, vcc\n v_cmp_gt_f32 vcc, %11, %16\n v_cndmask_b32 %3, %11, %16, vcc\n"
                                     "v_cmp_gt_f32 vcc, %12, %16\n v_cndmask_b32 %4, %12, %16, vcc\n v_cmp_gt_f32 vcc, %13, %16\n v_cndmask_b32 %5, %13, %16, vcc\n v_cmp_gt_f32 vcc, %14, %16\n v_cndmask_b32 %6, %14, %16, vcc\n v_cmp_gt_f32 vcc, %15, %16\n v_cndmask_b32 %7, %15, %16, vcc" : OUTS : INS : "vcc");
        if (MODE == 21) OP2("v_max_f32");
        if (MODE == 22) OP3("v_med3_f32");
        if (MODE == 23) asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8" : : INS : "vcc");
        if (MODE == 24) asm volatile("v_cmp_gt_f32 s[10:11], %8, %16\n v_cmp_gt_f32 s[12:13], %9, %16\n v_cmp_gt_f32 s[14:15], %10, %16\n v_cmp_gt_f32 s[16:17], %11, %16\n v_cndmask_b32_e64 %0, %8, %16, s[10:11]\n v_cndmask_b32_e64 %1, %9, %16, s[12:13]\n v_cndmask_b32_e64 %2, %10, %16, s[14:15]\n v_cndmask_b32_e64 %3, %11, %16, s[16:17]\n"
                                     "v_cmp_gt_f32 s[10:11], %12, %16\n v_cmp_gt_f32 s[12:13], %13, %16\n v_cmp_gt_f32 s[14:15], %14, %16\n v_cmp_gt_f32 s[16:17], %15, %16\n v_cndmask_b32_e64 %4, %12, %16, s[10:11]\n v_cndmask_b32_e64 %5, %13, %16, s[12:13]\n v_cndmask_b32_e64 %6, %14, %16, s[14:15]\n v_cndmask_b32_e64 %7, %15, %16, s[16:17]" : OUTS : INS : "s10","s11","s12","s13","s14","s15","s16","s17");
        if (MODE == 18) OP3("v_add3_u32");
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7) + (float)(w0 + w1 + w2 + w3) + (float)(d0 + d1 + d2 + d3);
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
    printf("ticks per wave-instruction per SIMD (plain f32 VOP2 ~2.1)\n");
    run<15>("v_mul_f32");
    run<14>("v_fma_f32");
    run<0>("v_mul_lo_u32");
    run<1>("v_mul_hi_u32");
    run<2>("v_mad_u64_u32");
    run<3>("v_mul_u32_u24");
    run<4>("v_mul_hi_u32_u24");
    run<5>("v_mad_u32_u24");
    run<6>("v_xor_b32");
    run<17>("v_xad_u32");
    run<18>("v_add3_u32");
    run<13>("v_cndmask_b32 (vcc)");
    run<19>("v_cndmask_b32_e64 sgpr");
    run<23>("v_cmp_gt_f32 vcc");
    run<20>("cmp+cndmask pair /2");
    run<24>("cmp+cndmask sgpr grouped /2");
    run<21>("v_max_f32");
    run<22>("v_med3_f32");
    run<16>("v_exp_f32");
    run<7>("v_log_f32");
    run<8>("v_rndne_f32");
    run<9>("v_ldexp_f32");
    run<10>("v_cvt_i32_f32");
    run<11>("v_add_f64");
    run<12>("v_cvt_f64_f32");
    return 0;
}
