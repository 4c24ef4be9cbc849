// Wall-clock ceiling of v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 on gfx950: pure register loops, whole chip, with 1 or 2
// waves per SIMD (run on the GPU box: ./tools/mfma_peak_probe).  Tells what "100 %" means under the sustained clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = a0 * 2.f + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = a0 * 2.f + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static void run(const char* name, F launch, double flop_per_launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flop_per_launch / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 4096;
    for (int blocks : {256, 512, 1024}) {
        char nm[128];
        snprintf(nm, sizeof nm, "32x32x2 f32, 4 acc, %d blocks x 4 waves", blocks);
        run(nm, [&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); }, (double)blocks * 4 * iters * 16 * 4096.0);
        snprintf(nm, sizeof nm, "32x32x2 f32, 1 acc (dependent), %d blocks", blocks);
        run(nm, [&] { hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); }, (double)blocks * 4 * iters * 4 * 4096.0);
        snprintf(nm, sizeof nm, "16x16x4 f32, 4 acc, %d blocks", blocks);
        run(nm, [&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); }, (double)blocks * 4 * iters * 16 * 2048.0);
        snprintf(nm, sizeof nm, "16x16x4 f32, 1 acc (dependent), %d blocks", blocks);
        run(nm, [&] { hipLaunchKernelGGL(k16<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); }, (double)blocks * 4 * iters * 4 * 2048.0);
    }
    return 0;
}
