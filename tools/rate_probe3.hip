// Throughput probe for the v3 attention inner loop (gfx950): bf16 matrix-pipe QK^T beside the f32 datapath.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define N_IT 1000
#define OPAQUE(x) asm volatile("" : "+v"(x))

template <int MODE>
__global__ void k(float* out, long long* t0s, long long* t1s) {
    float a = threadIdx.x * 0.001f + 0.5f;
    union { uint4 u; bf16x8 v; } ka, qb[4];
    ka.u = make_uint4(0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    for (int j = 0; j < 4; ++j) qb[j].u = make_uint4(0x3c003c00u + j, 0x3c003c00u, 0xbc003c00u, 0x3c00bc00u);
    f32x4 acc[4];
    float ls[4] = {0, 0, 0, 0};
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0, 0, 0, 0};
    float vb[4] = {a, a + 1, a + 2, a + 3};
    const f32x4 zero = {0, 0, 0, 0};
    f32x4 s[4];
    for (int j = 0; j < 4; ++j) s[j] = f32x4{-1.f - j, -2.f, -3.f, -4.f};
    float guard = 0.f;
    float pp[4][4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) pp[j][r] = 0.5f + j + r;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < N_IT; ++it) {
        OPAQUE(ka.u.x);
        if (MODE != 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka.v, qb[j].v, zero, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { OPAQUE(s[j][0]); OPAQUE(s[j][1]); OPAQUE(s[j][2]); OPAQUE(s[j][3]); }
        }
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ls[j] += s[j][0];
        }
        if (MODE == 3 || MODE == 5) {
            float mx = s[0][0];
#pragma unroll
            for (int j = 0; j < 4; ++j) { mx = fmaxf(fmaxf(mx, s[j][0]), s[j][1]); mx = fmaxf(fmaxf(mx, s[j][2]), s[j][3]); }
            if (__any(mx > 40.f)) guard += 1.f;
        }
        if (MODE == 8 || MODE == 9) {
            float p[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) p[j][r] = __builtin_amdgcn_exp2f(s[j][r]);
            if (MODE == 8) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j][e] = __builtin_fmaf(p[j][r], vb[(r + e) & 3], acc[j][e]);
#pragma unroll
                for (int j = 0; j < 4; ++j) ls[j] += (p[j][0] + p[j][1]) + (p[j][2] + p[j][3]);
            } else {   // MODE 9: exp only + sums (no PV, no QK dependence)
#pragma unroll
                for (int j = 0; j < 4; ++j) ls[j] += (p[j][0] + p[j][1]) + (p[j][2] + p[j][3]);
            }
        } else if (MODE == 6 || MODE == 7) {
            // software pipelined: PV MFMAs consume the previous tile's p while this tile's exps are issued
            float p[4][4];
            if (MODE == 6) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        p[j][r] = __builtin_amdgcn_exp2f(s[j][r]);
                        acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(pp[j][r], vb[r], acc[j], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) p[j][r] = __builtin_amdgcn_exp2f(s[j][r]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(pp[j][r], vb[r], acc[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) ls[j] += (p[j][0] + p[j][1]) + (p[j][2] + p[j][3]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) pp[j][r] = p[j][r];
        } else if (MODE >= 1) {
            float p[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) p[j][r] = __builtin_amdgcn_exp2f(s[j][r]);
            if (MODE != 4 && MODE != 5) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(p[j][r], vb[r], acc[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) ls[j] += (p[j][0] + p[j][1]) + (p[j][2] + p[j][3]);
        }
    }
    long long t1 = clock64();
    float r = ls[0] + ls[1] + ls[2] + ls[3] + guard;
    for (int j = 0; j < 4; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3] + pp[j][0] + pp[j][3];
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
        printf("%-52s waves/SIMD=%d: %.1f cycles per iteration per SIMD\n", name, waves, (double)(mx - mn) / N_IT / waves);
    }
}
int main() {
    run<0>("M0 4x mfma_bf16_16x16x32 only");
    run<1>("M1 16 exp + 16 mfma4x4x1 + sums (no QK)");
    run<2>("M2 4 bf16 mfma + 16 exp + 16 mfma4x4x1 + sums");
    run<3>("M3 = M2 + max check");
    run<4>("M4 4 bf16 mfma + 16 exp + sums (no PV)");
    run<5>("M5 = M4 + max check");
    run<8>("M8 4 bf16 mfma + 16 exp + 64 v_fma + sums");
    run<9>("M9 4 bf16 mfma + 16 exp + sums");
    run<6>("M6 M2 software-pipelined, exp/mfma interleaved");
    run<7>("M7 M2 software-pipelined, 16 exp then 16 mfma");
    return 0;
}
