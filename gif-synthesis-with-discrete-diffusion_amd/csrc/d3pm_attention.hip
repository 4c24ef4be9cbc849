// Dense self-attention for head dim 4 (the D3PM denoiser: n_embd 64 / 16 heads), fp32-accurate.
//
// Replaces FullAttention.forward (transformer_utils.py:46-62): softmax(q k^T / sqrt(4)) v without
// materialising the (B,16,L,L) score tensor (1 GiB / sample / layer in the reference at L = 4096).
//
// Measured on gfx950 (tools/rate_probe2.hip): f32 MFMA and f32 VALU share one datapath (times add up),
// v_exp_f32 costs 8.4 cycles per wave-instruction.  Per 16x64 score tile the f32 datapath would pay
// QK^T 128 + exp 134 + PV 128 cycles.  So QK^T is moved to the *bf16 matrix pipe*, which runs beside the
// f32 datapath, with an error-free split: q = q1+q2+q3, k = k1+k2+k3 (bf16 pieces, 24 bits) and the six
// significant cross products (k1q1 k2q1 k1q2 k2q2 k3q1 k1q3) laid along the K=32 contraction of ONE
// v_mfma_f32_16x16x32_bf16 (products of bf16 pairs are exact in the f32 accumulator; dropped terms < 2^-24).
// Three more contraction slots carry 1 * (-m) so the running max is subtracted for free (C = 0).
//
// Per wave64: 64 queries = 4 sub-tiles of 16.  Per 16-key tile
//   S^T[key][query] = mfma_16x16x32_bf16(A = split K tile, B = split Q^T sub-tile)        (matrix pipe)
//   p = v_exp_f32(S)                                                                        (log2 domain)
//   out += P V via v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products, exact f32)
// The running max m is an integer (exactly representable, rescale factors are powers of two) and is raised
// only when a score exceeds it by 2^40 (fp32 headroom): a rare wave-uniform branch.
// K/V tiles are staged through LDS in chunks of 256 keys, double-buffered.
#include <type_traits>

#include "common.hpp"

namespace gsdd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KC = 256;            // keys per LDS chunk
constexpr float RESCALE_THR = 40.f;

struct AttnSmem {
    uint4 k[2][KC][2];             // [buf][key][slot]: pieces A=[k1|k2], B=[k3|k1]; slots swapped for (key&15)>=8 (banks)
    float v[2][KC / 4][4][4];      // [buf][key/4][dim][key%4]   (B operands of PV: one ds_read_b128 per tile)
    uint4 ones[16];                // [1,1,1,0,0,0,0,0] bf16 at dword offsets 0 and 4 (mod 64) for the -m slots
};

__device__ __forceinline__ uint32_t bf16_rn(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// x = a + b + c with a, b, c bf16 (error-free up to the last piece's rounding, < 2^-24 |x|)
__device__ __forceinline__ void split3(float x, uint32_t& a, uint32_t& b, uint32_t& c) {
    a = bf16_rn(x);
    const float r = x - __uint_as_float(a << 16);
    b = bf16_rn(r);
    const float r2 = r - __uint_as_float(b << 16);
    c = bf16_rn(r2);
}
__device__ __forceinline__ uint4 pack8(const uint32_t (&lo)[4], const uint32_t (&hi)[4]) {
    return make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
}
__device__ __forceinline__ bf16x8 as_frag(uint4 u) {
    union { uint4 u; bf16x8 v; } c;
    c.u = u;
    return c.v;
}
// -m as three bf16 pieces in contraction slots 24..26 (lane group 3's fragment)
__device__ __forceinline__ uint4 negm_frag(float m) {
    uint32_t a, b, c;
    split3(-m, a, b, c);
    return make_uint4(a | (b << 16), c, 0u, 0u);
}

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
    const int li = lane & 15, lg = lane >> 4, lj = lane & 3;
    const int q0 = blockIdx.x * 256 + wave * 64;
    const int quad = lane & ~3;

    if (tid < 16) sm.ones[tid] = (tid == 0 || tid == 1) ? make_uint4(0x3F803F80u, 0x00003F80u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);

    // ---- Q^T operand fragments: lane (li, lg) holds the pieces of query q0+16j+li that face lane group lg's K slots
    //      lg0: [q1|q1]  lg1: [q2|q2]  lg2: [q1|q3]  lg3: [-m1,-m2,-m3,0|0]   (q pre-scaled into the log2 domain)
    const float qscale = 0.5f * 1.4426950408889634f;
    uint4 qfrag[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int qi = q0 + 16 * j + li;
        qi = qi < L ? qi : L - 1;
        const float4 qv = *reinterpret_cast<const float4*>(qh + (int64_t)qi * 4);
        const float qs[4] = {qv.x * qscale, qv.y * qscale, qv.z * qscale, qv.w * qscale};
        uint32_t q1[4], q2[4], q3[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) split3(qs[d], q1[d], q2[d], q3[d]);
        const uint4 f0 = pack8(q1, q1), f1 = pack8(q2, q2), f2 = pack8(q1, q3);
        qfrag[j] = lg == 0 ? f0 : (lg == 1 ? f1 : (lg == 2 ? f2 : make_uint4(0u, 0u, 0u, 0u)));
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
        const float ks[4] = {rk.x, rk.y, rk.z, rk.w};
        uint32_t k1[4], k2[4], k3[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) split3(ks[d], k1[d], k2[d], k3[d]);
        const int sw = (tid >> 3) & 1;                        // (key & 15) >= 8: swap the two 16-B slots
        sm.k[buf][tid][sw] = pack8(k1, k2);                   // piece A: faces [q1|q1] and [q2|q2]
        sm.k[buf][tid][sw ^ 1] = pack8(k3, k1);               // piece B: faces [q1|q3]
        float* pv = &sm.v[buf][tid >> 2][0][tid & 3];
        pv[0] = rv.x; pv[4] = rv.y; pv[8] = rv.z; pv[12] = rv.w;
    };

    f32x4 acc[4];
    float lsum[4], mq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lsum[j] = 0.f; mq[j] = 0.f;
        acc[j][0] = 0.f; acc[j][1] = 0.f; acc[j][2] = 0.f; acc[j][3] = 0.f;
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    // per-lane K fragment address inside a chunk buffer: lanes lg 0/1 read piece A, lg 2 reads piece B, lg 3 reads the
    // constant "ones" piece (two copies on different banks so every ds_read_b128 lane group stays conflict-free)
    const int slot = ((lg >> 1) ^ (li >> 3)) & 1;
    const uint4* kbase0 = (lg == 3) ? &sm.ones[(li >= 4 && li < 12) ? 0 : 1] : &sm.k[0][li][slot];
    const int kstep = (lg == 3) ? 0 : 32;                     // uint4 per 16-key tile
    const int kbuf = (lg == 3) ? 0 : KC * 2;                  // uint4 per buffer
    {   // running max (an integer, shared by the 4 key groups of a query) from the first key tile
        const bf16x8 kf = as_frag(kbase0[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, as_frag(qfrag[j]), zero, 0, 0, 0);
            float m0 = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
            m0 = fmaxf(m0, __shfl_xor(m0, 16));
            m0 = fmaxf(m0, __shfl_xor(m0, 32));
            mq[j] = ceilf(m0);
            if (lg == 3) qfrag[j] = negm_frag(mq[j]);
        }
    }

    // One sweep over all keys.  CHECK=false is the fast path: no per-tile max maintenance at all (fp32 range lets p
    // reach 2^100 before anything is lost); CHECK=true raises the running max whenever a score exceeds it by 2^40.
    auto sweep = [&](auto check_tag) {
        constexpr bool CHECK = decltype(check_tag)::value;
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunks) load_chunk(c + 1);
            const int ntiles = min(KC, L - c * KC) >> 4;
            const uint4* kb = kbase0 + buf * kbuf;
#pragma unroll 2
            for (int t = 0; t < ntiles; ++t) {
                const bf16x8 kf = as_frag(kb[t * kstep]);
                const float4 vb = *reinterpret_cast<const float4*>(&sm.v[buf][t * 4 + lg][lj][0]);
                f32x4 s[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, as_frag(qfrag[j]), zero, 0, 0, 0);
                if (CHECK) {
                    float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        mx = fmaxf(fmaxf(mx, s[j][0]), s[j][1]);
                        mx = fmaxf(fmaxf(mx, s[j][2]), s[j][3]);
                    }
                    if (__any(mx > RESCALE_THR)) {   // rare: raise the running max of the queries that need it
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float tm = fmaxf(fmaxf(s[j][0], s[j][1]), fmaxf(s[j][2], s[j][3]));
                            tm = fmaxf(tm, __shfl_xor(tm, 16));
                            tm = fmaxf(tm, __shfl_xor(tm, 32));
                            const float delta = tm > RESCALE_THR ? ceilf(tm) : 0.f;   // integer: alpha is a power of two
                            const float alpha = __builtin_amdgcn_exp2f(-delta);
                            mq[j] += delta;
                            lsum[j] *= alpha;
                            if (lg == 3) qfrag[j] = negm_frag(mq[j]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                s[j][e] -= delta;
                                acc[j][e] *= __shfl(alpha, quad + e);   // register e belongs to the query of lane quad+e
                            }
                        }
                    }
                }
                float p[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) p[j][r] = __builtin_amdgcn_exp2f(s[j][r]);
#pragma unroll
                for (int j = 0; j < 4; ++j) lsum[j] += (p[j][0] + p[j][1]) + (p[j][2] + p[j][3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(p[j][0], vb.x, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(p[j][1], vb.y, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(p[j][2], vb.z, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(p[j][3], vb.w, acc[j], 0, 0, 0);
            }
            if (c + 1 < nchunks) store_chunk(buf ^ 1);
            __syncthreads();
        }
    };

    sweep(std::false_type{});
    {   // overflow screen: a row sum beyond 2^100 (or inf/nan) means some score ran > 100 octaves above the first
        // tile's maximum.  Never seen on real weights; then the whole block redoes the sweep with max maintenance.
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) bad = bad || !(lsum[j] < 1.2676506e30f);
        if (__syncthreads_or(bad ? 1 : 0)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lsum[j] = 0.f;
                acc[j][0] = 0.f; acc[j][1] = 0.f; acc[j][2] = 0.f; acc[j][3] = 0.f;
            }
            load_chunk(0);
            store_chunk(0);
            __syncthreads();
            sweep(std::true_type{});
        }
    }

    // ---- merge the 4 key groups of each query (they share one running max), normalise, store rows [M][H*4]
    // lane (blk, dim) register i holds out[query 4*(blk&3)+i][dim]; the row sum of that query lives in lane quad+i
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float l = __shfl(lsum[j], quad + i);
            float a = acc[j][i];
            l += __shfl_xor(l, 16); a += __shfl_xor(a, 16);
            l += __shfl_xor(l, 32); a += __shfl_xor(a, 32);
            o[i] = a / l;
        }
        if (lg == 0) {      // lanes 0..15: lane = 4*qq + dim
            const int qq = lane >> 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int qi = q0 + 16 * j + 4 * qq + i;
                if (qi < L) out[((int64_t)b * L + qi) * (H * 4) + h * 4 + lj] = o[i];
            }
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
