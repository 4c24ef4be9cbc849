// Instruction-rate probe on gfx950: cycles per wave-instruction for the building blocks of the d=4 attention loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define N_IT 2000

template <int MODE>
__global__ void k(float* out, long long* cyc) {
    float a = threadIdx.x * 0.001f + 0.5f, b = 1.0001f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f32x16 d0, d1;
    for (int i = 0; i < 16; ++i) { d0[i] = 0; d1[i] = 0; }
    float e0 = a, e1 = a + 1, e2 = a + 2, e3 = a + 3, e4 = a + 4, e5 = a + 5, e6 = a + 6, e7 = a + 7;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < N_IT; ++it) {
        if (MODE == 0) {  // 4 independent 16x16x4
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
        } else if (MODE == 1) {  // 4 independent 4x4x1
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
        } else if (MODE == 2) {  // 4 dependent 4x4x1
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        } else if (MODE == 3) {  // 8 independent v_exp
            e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1); e2 = __builtin_amdgcn_exp2f(e2); e3 = __builtin_amdgcn_exp2f(e3);
            e4 = __builtin_amdgcn_exp2f(e4); e5 = __builtin_amdgcn_exp2f(e5); e6 = __builtin_amdgcn_exp2f(e6); e7 = __builtin_amdgcn_exp2f(e7);
        } else if (MODE == 4) {  // 8 independent v_fma
            e0 = __builtin_fmaf(e0, b, a); e1 = __builtin_fmaf(e1, b, a); e2 = __builtin_fmaf(e2, b, a); e3 = __builtin_fmaf(e3, b, a);
            e4 = __builtin_fmaf(e4, b, a); e5 = __builtin_fmaf(e5, b, a); e6 = __builtin_fmaf(e6, b, a); e7 = __builtin_fmaf(e7, b, a);
        } else if (MODE == 5) {  // 2 independent 32x32x2
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d1, 0, 0, 0);
        } else if (MODE == 6) {  // 4 independent 16x16x4 + 8 exp (co-issue)
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            e2 = __builtin_amdgcn_exp2f(e2); e3 = __builtin_amdgcn_exp2f(e3);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            e4 = __builtin_amdgcn_exp2f(e4); e5 = __builtin_amdgcn_exp2f(e5);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
            e6 = __builtin_amdgcn_exp2f(e6); e7 = __builtin_amdgcn_exp2f(e7);
        } else if (MODE == 7) {  // 8 pk_fma (2 fma per lane each)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 x0 = {e0, e1}, x1 = {e2, e3}, x2 = {e4, e5}, x3 = {e6, e7}, bb = {b, b}, aa = {a, a};
            x0 = __builtin_elementwise_fma(x0, bb, aa); x1 = __builtin_elementwise_fma(x1, bb, aa);
            x2 = __builtin_elementwise_fma(x2, bb, aa); x3 = __builtin_elementwise_fma(x3, bb, aa);
            e0 = x0[0]; e1 = x0[1]; e2 = x1[0]; e3 = x1[1]; e4 = x2[0]; e5 = x2[1]; e6 = x3[0]; e7 = x3[1];
        }
    }
    long long t1 = clock64();
    float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5] + e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter) {
    float* o; long long* c; hipMalloc(&o, 1 << 22); hipMalloc(&c, 8);
    for (int waves : {1, 2, 4, 8}) {   // waves per SIMD: block = 256*waves threads? use blocks of 256 threads, waves blocks per CU
        int threads = 256 * (waves > 4 ? 4 : waves);
        int blocks = 256 * (waves > 4 ? waves / 4 : 1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, o, c);
        hipDeviceSynchronize();
        long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%-34s waves/SIMD=%d: %.2f cyc/iter  (%.2f cyc per instr per wave; %.2f per SIMD-instr)\n", name, waves, (double)h / N_IT,
               (double)h / N_IT / per_iter, (double)h / N_IT / per_iter / waves);
    }
}
int main() {
    run<0>("mfma_16x16x4 x4 indep", 4);
    run<1>("mfma_4x4x1 x4 indep", 4);
    run<2>("mfma_4x4x1 x4 dependent", 4);
    run<5>("mfma_32x32x2 x2 indep", 2);
    run<3>("v_exp x8", 8);
    run<4>("v_fma x8", 8);
    run<7>("v_pk_fma x4 (8 fma)", 4);
    run<6>("4 mfma16 + 8 exp interleaved", 12);
    return 0;
}
