// Dense self-attention for head dim 4 (the D3PM denoiser: n_embd 64 / 16 heads), exact fp32.
//
// Replaces FullAttention.forward (transformer_utils.py:46-62): softmax(q k^T / sqrt(4)) v without
// materialising the (B,16,L,L) score tensor (1 GiB / sample / layer in the reference at L = 4096).
//
// Structure (per wave64): 64 queries = 4 sub-tiles of 16.  Per 16-key tile
//   S^T[key][query] = v_mfma_f32_16x16x4_f32(A = K tile, B = Q^T sub-tile, C = -m)      (contraction = head dim 4)
// so a lane owns 4 keys x 1 query per sub-tile and keeps its OWN running (m, l, acc[4]) over the
// keys it sees; the 4 key-groups (lane>>4) are merged once at the end.  Softmax runs in the log2
// domain (q pre-scaled by log2(e)/sqrt(4)), p = v_exp_f32(S) needs no subtraction because -m rides
// in the MFMA accumulator input.  The running max is only raised when a score exceeds it by 2^40
// (fp32 has the headroom), a wave-uniform rare branch.
// K/V tiles are staged through LDS in chunks of 256 keys, double-buffered.
#include "common.hpp"

namespace gsdd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 256;            // keys per LDS chunk
constexpr int KPAD = KC + 16;      // dim-major pitch: dims 0/1 of a 32-lane read group land on disjoint banks
constexpr float RESCALE_THR = 40.f;

struct AttnSmem {
    float k[2][4][KPAD];           // [buf][dim][key]
    float v[2][KC][4];             // [buf][key][dim]
};

__global__ __launch_bounds__(256) void d3pm_attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, int B, int L, int H,
                                                             float* __restrict__ out) {
    __shared__ AttnSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const int64_t M = (int64_t)B * L;
    const int64_t base = ((int64_t)h * M + (int64_t)b * L) * 4;      // first row of this (b,h)
    const float* qh = q + base;
    const float* kh = k + base;
    const float* vh = v + base;
    const int li = lane & 15, lg = lane >> 4;
    const int q0 = blockIdx.x * 256 + wave * 64;

    // ---- Q^T operand: lane holds Q[q0 + 16j + li][dim = lg], pre-scaled into the log2 domain
    const float qscale = 0.5f * 1.4426950408889634f;
    float qf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int qi = q0 + 16 * j + li;
        qi = qi < L ? qi : L - 1;
        qf[j] = qh[(int64_t)qi * 4 + lg] * qscale;
    }

    const int nchunks = (L + KC - 1) / KC;
    float4 rk, rv;
    auto load_chunk = [&](int c) {
        const int key = c * KC + tid;
        rk = make_float4(0.f, 0.f, 0.f, 0.f);
        rv = rk;
        if (key < L) {
            rk = *reinterpret_cast<const float4*>(kh + (int64_t)key * 4);
            rv = *reinterpret_cast<const float4*>(vh + (int64_t)key * 4);
        }
    };
    auto store_chunk = [&](int buf) {
        sm.k[buf][0][tid] = rk.x; sm.k[buf][1][tid] = rk.y; sm.k[buf][2][tid] = rk.z; sm.k[buf][3][tid] = rk.w;
        *reinterpret_cast<float4*>(&sm.v[buf][tid][0]) = rv;
    };

    f32x4 negm[4];
    float lsum[4];
    float acc[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lsum[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    }

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    {   // running max initialised from the first key tile (per lane: its own 4 keys)
        const float kf = sm.k[0][lg][li];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf, qf[j], zero, 0, 0, 0);
            const float m0 = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            negm[j][0] = -m0; negm[j][1] = -m0; negm[j][2] = -m0; negm[j][3] = -m0;
        }
    }

    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk(c + 1);
        const int ntiles = min(KC, L - c * KC) >> 4;
        for (int t = 0; t < ntiles; ++t) {
            const float kf = sm.k[buf][lg][t * 16 + li];
            f32x4 s[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf, qf[j], negm[j], 0, 0, 0);
            float4 vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) vv[r] = *reinterpret_cast<const float4*>(&sm.v[buf][t * 16 + lg * 4 + r][0]);
            float mx = s[0][0];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[j][r]);
            if (__any(mx > RESCALE_THR)) {   // rare: raise the running max of the lanes that need it
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float tm = fmaxf(fmaxf(s[j][0], s[j][1]), fmaxf(s[j][2], s[j][3]));
                    const float delta = tm > RESCALE_THR ? tm : 0.f;
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    lsum[j] *= alpha;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[j][e] *= alpha; negm[j][e] -= delta; s[j][e] -= delta; }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[j][r]);
                    lsum[j] += p;
                    acc[j][0] = fmaf(p, vv[r].x, acc[j][0]);
                    acc[j][1] = fmaf(p, vv[r].y, acc[j][1]);
                    acc[j][2] = fmaf(p, vv[r].z, acc[j][2]);
                    acc[j][3] = fmaf(p, vv[r].w, acc[j][3]);
                }
            }
        }
        if (c + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- merge the 4 key groups of each query, normalise, store rows [M][H*4]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float m = -negm[j][0];
        float mm = fmaxf(m, __shfl_xor(m, 16));
        mm = fmaxf(mm, __shfl_xor(mm, 32));
        const float sc = __builtin_amdgcn_exp2f(m - mm);
        float l = lsum[j] * sc;
        float a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = acc[j][e] * sc;
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            l += __shfl_xor(l, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += __shfl_xor(a[e], o);
        }
        const int qi = q0 + 16 * j + li;
        if (lg == 0 && qi < L) {
            const float inv = 1.0f / l;
            float4 o4 = make_float4(a[0] * inv, a[1] * inv, a[2] * inv, a[3] * inv);
            *reinterpret_cast<float4*>(out + ((int64_t)b * L + qi) * (H * 4) + h * 4) = o4;
        }
    }
}

// General cross-attention with Te condition tokens (tiny: Te <= 77), one thread per (row, head).
__global__ void d3pm_cross_attention_kernel(const float* q, const float* kc, const float* vc, int B, int L, int Te, int H,
                                            float* out) {
    const int64_t M = (int64_t)B * L;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * H) return;
    const int h = (int)(i / M);
    const int64_t m = i % M;
    const int b = (int)(m / L);
    const float4 qv = *reinterpret_cast<const float4*>(q + ((int64_t)h * M + m) * 4);
    float mx = -INFINITY;
    for (int e = 0; e < Te; ++e) {
        const float4 kv = *reinterpret_cast<const float4*>(kc + ((int64_t)b * Te + e) * (H * 4) + h * 4);
        const float s = (qv.x * kv.x + qv.y * kv.y + qv.z * kv.z + qv.w * kv.w) * 0.5f;
        mx = fmaxf(mx, s);
    }
    float l = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int e = 0; e < Te; ++e) {
        const float4 kv = *reinterpret_cast<const float4*>(kc + ((int64_t)b * Te + e) * (H * 4) + h * 4);
        const float4 vv = *reinterpret_cast<const float4*>(vc + ((int64_t)b * Te + e) * (H * 4) + h * 4);
        const float s = (qv.x * kv.x + qv.y * kv.y + qv.z * kv.z + qv.w * kv.w) * 0.5f;
        const float p = expf(s - mx);
        l += p;
        a0 += p * vv.x; a1 += p * vv.y; a2 += p * vv.z; a3 += p * vv.w;
    }
    const float inv = 1.0f / l;
    *reinterpret_cast<float4*>(out + m * (H * 4) + h * 4) = make_float4(a0 * inv, a1 * inv, a2 * inv, a3 * inv);
}

}  // namespace gsdd

using namespace gsdd;

extern "C" int gsdd_d3pm_attention(const float* q, const float* k, const float* v, int B, int L, int H, float* out,
                                   void* stream) {
    GSDD_CHECK_ARG(q && k && v && out, "null pointer");
    GSDD_CHECK_ARG(B > 0 && H > 0 && L >= 16 && L % 16 == 0, "L must be a positive multiple of 16");
    GSDD_CHECK_ARG(B <= 65535 && H <= 65535, "grid too large");
    const dim3 grid((L + 255) / 256, H, B);
    hipLaunchKernelGGL(d3pm_attention_kernel, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, B, L, H, out);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_d3pm_cross_attention(const float* q, const float* kc, const float* vc, int B, int L, int Te, int H,
                                         float* out, void* stream) {
    GSDD_CHECK_ARG(q && kc && vc && out, "null pointer");
    GSDD_CHECK_ARG(B > 0 && L > 0 && Te > 0 && H > 0, "bad sizes");
    const int64_t n = (int64_t)B * L * H;
    hipLaunchKernelGGL(d3pm_cross_attention_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       q, kc, vc, B, L, Te, H, out);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}
