// Multi-wave throughput probe (gfx950): span = max(end) - min(start) over all waves of one 256*W-thread block
// (one block per CU).  Inputs are made opaque each iteration (empty asm) so nothing is hoisted.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define N_IT 1000
#define OPAQUE(x) asm volatile("" : "+v"(x))

template <int MODE>
__global__ void k(float* out, long long* t0s, long long* t1s) {
    float a = threadIdx.x * 0.001f + 0.5f, b = 1.0001f;
    f32x4 nm = {-1.f, -1.f, -1.f, -1.f};
    f32x4 acc[4];
    float ls[4] = {0, 0, 0, 0};
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0, 0, 0, 0};
    float vb[4] = {a, a + 1, a + 2, a + 3};
    f32x4 s[4];
    for (int j = 0; j < 4; ++j) s[j] = f32x4{-1.f - j, -2.f, -3.f, -4.f};
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < N_IT; ++it) {
        OPAQUE(a);
        if (MODE == 0 || MODE >= 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b + j, nm, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { OPAQUE(s[j][0]); OPAQUE(s[j][1]); OPAQUE(s[j][2]); OPAQUE(s[j][3]); }
        }
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ls[j] += s[j][0];
        }
        float p[4][4];
        if (MODE == 1 || MODE >= 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) p[j][r] = __builtin_amdgcn_exp2f(s[j][r]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) p[j][r] = s[j][r];
        }
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ls[j] += p[j][0] + p[j][1] + p[j][2] + p[j][3];
        }
        if (MODE == 2 || MODE == 4) {   // PV on 4x4x1 MFMA, r-outer so consecutive MFMAs hit different accumulators
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(p[j][r], vb[r], acc[j], 0, 0, 0);
        }
        if (MODE == 3 || MODE == 5) {   // PV on VALU: 64 fma
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][e] = __builtin_fmaf(p[j][r], vb[(r + e) & 3], acc[j][e]);
        }
        if (MODE >= 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ls[j] += (p[j][0] + p[j][1]) + (p[j][2] + p[j][3]);
        }
    }
    long long t1 = clock64();
    float r = ls[0] + ls[1] + ls[2] + ls[3];
    for (int j = 0; j < 4; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { t0s[threadIdx.x >> 6] = t0; t1s[threadIdx.x >> 6] = t1; }
}

template <int MODE>
void run(const char* name) {
    float* o; long long *a, *b; hipMalloc(&o, 1 << 22); hipMalloc(&a, 8 * 16); hipMalloc(&b, 8 * 16);
    for (int waves : {1, 2, 4}) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * waves), 0, 0, o, a, b);
        hipDeviceSynchronize();
        long long h0[16], h1[16];
        hipMemcpy(h0, a, 8 * 4 * waves, hipMemcpyDeviceToHost); hipMemcpy(h1, b, 8 * 4 * waves, hipMemcpyDeviceToHost);
        long long mn = *std::min_element(h0, h0 + 4 * waves), mx = *std::max_element(h1, h1 + 4 * waves);
        printf("%-44s waves/SIMD=%d: %.1f cycles per iteration per SIMD\n", name, waves, (double)(mx - mn) / N_IT / waves);
    }
}
int main() {
    run<0>("T0 4x mfma16x16x4");
    run<1>("T1 16x v_exp (+12 add)");
    run<2>("T2 16x mfma4x4x1 (4 accumulators, r-outer)");
    run<3>("T3 64x v_fma");
    run<4>("T4 full tile, PV on mfma4x4x1");
    run<5>("T5 full tile, PV on VALU");
    return 0;
}
