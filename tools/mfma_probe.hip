// Probe the lane/register maps of v_mfma_f32_4x4x1_16b_f32 on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(int* out) {
    const int lane = threadIdx.x;
    for (int a = 0; a < 64; ++a)
        for (int b = 0; b < 64; ++b) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            const float av = lane == a ? 1.f : 0.f, bv = lane == b ? 1.f : 0.f;
            c = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, c, 0, 0, 0);
            for (int r = 0; r < 4; ++r)
                if (c[r] != 0.f) out[a * 64 + b] = lane * 4 + r;
        }
}
int main() {
    int* d; hipMalloc(&d, 64 * 64 * 4); hipMemset(d, 0xff, 64 * 64 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    static int h[64 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // print for A lane a and B lane b which (lane,reg) receives the product
    for (int a = 0; a < 8; ++a) { for (int b = 0; b < 8; ++b) { int v = h[a * 64 + b]; if (v < 0) printf("   .   "); else printf(" l%02dr%d ", v / 4, v % 4); } printf("\n"); }
    int cnt = 0; for (int i = 0; i < 4096; ++i) cnt += h[i] >= 0; printf("nonzero pairs: %d\n", cnt);
    for (int a = 60; a < 64; ++a) { for (int b = 60; b < 64; ++b) { int v = h[a * 64 + b]; if (v < 0) printf("   .   "); else printf(" l%02dr%d ", v / 4, v % 4); } printf("\n"); }
    return 0;
}
