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
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace gsdd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KC = 256;            // keys per LDS chunk (exact-f32 kernel)
// keys per LDS chunk of the matrix-pipe kernel = template parameter KC4 (384: 3 workgroups x 48 KB fill the CU's LDS; every
// chunk boundary — barrier skew, overflow screen, pipeline refill — costs ~0.6 % of the run; GSDD_ATTN_KC=256 for A/B)
constexpr float RESCALE_THR = 40.f;

struct AttnSmem {
    uint4 k[2][KC][2];             // [buf][key][slot]: pieces A=[k1|k2], B=[k3|k1]; slots swapped for (key&15)>=8 (banks)
    float v[2][KC / 4][4][4];      // [buf][key/4][dim][key%4]   (B operands of PV: one ds_read_b128 per tile)
    uint4 ones[16];                // [1,1,1,0,0,0,0,0] bf16 at dword offsets 0 and 4 (mod 64) for the -m slots
};

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
    // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs (private L2s), so the query blocks of
    // one (b,h) -- which all stream the same K/V -- are renumbered to share an XCD (bijective remap; speed only).
    const int nqb = (L + 255) / 256;
    const unsigned nwg = gridDim.x;
    unsigned wg = blockIdx.x;
    {
        const unsigned q8 = nwg / 8, r8 = nwg % 8, xcd = wg % 8, idx = wg / 8;
        wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int qblk = wg % nqb;
    const int h = (wg / nqb) % H, b = wg / (nqb * H);
    const int64_t M = (int64_t)B * L;
    const int64_t base = ((int64_t)h * M + (int64_t)b * L) * 4;      // first row of this (b,h)
    const float* qh = q + base;
    const float* kh = k + base;
    const float* vh = v + base;
    const int li = lane & 15, lg = lane >> 4, lj = lane & 3;
    const int q0 = qblk * 256 + wave * 64;
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

// =====================================================================================================
// v4: P.V and the softmax row sum on the f16 MATRIX pipe as well.
//
// What the f32 datapath still has to do per score is then only: v_exp_f32 + (v_cvt_pk_f16 + v_fma_mix + v_cvt_pk_f16)/score
// = 1 transcendental + 2 plain VALU ops, instead of 1 + 4 FMA + 1 add.
//   P = hi + lo           hi = f16(p) (RN), lo = f16(p - hi)       (v_fma_mix_f32 subtracts the f16 half directly)
//   V = v1 + v2/2^11 + v3/2^22   (f16 pieces, scaled so no piece is subnormal)
//   D[query][col] = sum_key (hi + lo)[query][key] * [v1 | v2 | v3 | 1][key][col]     2 x v_mfma_f32_16x16x32_f16 per 32 keys
// f16 x f16 products are exact in the f32 accumulator; the six P x V cross terms keep >= 22 bits of P and all of V.
// The "1" column yields the row sum for free.
//
// f16 range: p' = 2^(s - m) must stay below 2^16, and the largest p' of a row at or above 2^3 (so that the summed
// absolute rounding of subnormal-range p' stays below 2^-22 of the row sum).  m starts at (max over the first 64 keys)
// - 3; when an accumulator overflows during a 256-key chunk, that chunk is redone for the affected queries with m += 12
// (accumulators restored from the chunk-start copy and scaled by the exact 2^-12).  So max(p') stays in [2^3, 2^16).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

template <int KC4>
struct AttnSmem4 {
    uint4 k[2][KC4][2];            // as AttnSmem.k
    uint4 v[2][KC4 / 32][4][16];    // [buf][32-key pair-tile][key group g][col j] -> 8 f16: keys (tile0: 4g+r, tile1: 4g+r)
    uint4 ones[16];
};

__device__ __forceinline__ f16x8 as_h8(uint4 u) {
    union { uint4 u; f16x8 v; } c;
    c.u = u;
    return c.v;
}
// Split 8 probabilities into packed f16 hi / lo fragments (one MFMA A operand each):
//   hi = f16(p) round-to-nearest (>= 65520 -> inf, which the overflow screen looks for; v_cvt_pkrtz would saturate)
//   lo = f16(p - hi), the subtraction done by v_fma_mix_f32 reading the f16 half directly (exact in f32).  (v_fma_mixlo_f16 /
//   v_fma_mixhi_f16 would write the rounded difference straight into one half of the packed result -- 8 instructions instead of 12,
//   same bits -- but the partial-register writes are slower: 1.40-1.58 ms against 1.30 for the hi + lo kernel, round 3.)
// One asm block; it ends with s_nop 1 because hipcc does not pad the VALU-write -> MFMA-read hazard for registers
// written inside inline asm (cdna_hip_programming.md section 5.7 item 2): without it the MFMA may read stale operands.
__device__ __forceinline__ void split_p8(const float (&p)[8], uint4& hi, uint4& lo) {
    float t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "v_cvt_pk_f16_f32 %0, %16, %17\n\t"
        "v_cvt_pk_f16_f32 %1, %18, %19\n\t"
        "v_cvt_pk_f16_f32 %2, %20, %21\n\t"
        "v_cvt_pk_f16_f32 %3, %22, %23\n\t"
        "v_fma_mix_f32 %8, %0, -1.0, %16 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %9, %0, -1.0, %17 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %10, %1, -1.0, %18 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %11, %1, -1.0, %19 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %12, %2, -1.0, %20 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %13, %2, -1.0, %21 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %14, %3, -1.0, %22 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %15, %3, -1.0, %23 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_f16_f32 %4, %8, %9\n\t"
        "v_cvt_pk_f16_f32 %5, %10, %11\n\t"
        "v_cvt_pk_f16_f32 %6, %12, %13\n\t"
        "v_cvt_pk_f16_f32 %7, %14, %15\n\t"
        "s_nop 1"
        : "=&v"(hi.x), "=&v"(hi.y), "=&v"(hi.z), "=&v"(hi.w), "=&v"(lo.x), "=&v"(lo.y), "=&v"(lo.z), "=&v"(lo.w),
          "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]));
}

// hi part only (the "P11" variant: 11 significant bits of every probability instead of 22; the row sum comes from the SAME
// rounded values through the ones column, so the weights still sum to one exactly).
__device__ __forceinline__ void round_p8(const float (&p)[8], uint4& hi) {
    asm volatile(
        "v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
        "v_cvt_pk_f16_f32 %2, %8, %9\n\t"
        "v_cvt_pk_f16_f32 %3, %10, %11\n\t"
        "s_nop 1"
        : "=&v"(hi.x), "=&v"(hi.y), "=&v"(hi.z), "=&v"(hi.w)
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]));
}

// lo = f16(p - hi) for an hi made by round_p8 (the second half of split_p8)
__device__ __forceinline__ void resid_p8(const float (&p)[8], const uint4& hi, uint4& lo) {
    float t0, t1, t2, t3, t4, t5, t6, t7;
    asm volatile(
        "v_fma_mix_f32 %4, %20, -1.0, %12 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %5, %20, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %6, %21, -1.0, %14 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %7, %21, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %8, %22, -1.0, %16 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %9, %22, -1.0, %17 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %10, %23, -1.0, %18 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %11, %23, -1.0, %19 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_cvt_pk_f16_f32 %1, %6, %7\n\t"
        "v_cvt_pk_f16_f32 %2, %8, %9\n\t"
        "v_cvt_pk_f16_f32 %3, %10, %11\n\t"
        "s_nop 1"
        : "=&v"(lo.x), "=&v"(lo.y), "=&v"(lo.z), "=&v"(lo.w),
          "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]),
          "v"(hi.x), "v"(hi.y), "v"(hi.z), "v"(hi.w));
}
// true if any of the 8 non-negative f16 in hi exceeds the f16 replicated in both halves of thr2: two v_pk_maximum3_f16 (gfx950)
// fold the eight values and the threshold into one packed maximum, which differs from the threshold iff something exceeded it.
// (hi may hold +inf after an overflow: then either this is true and the lo half is computed, or the threshold itself is +inf; in
// both cases the inf reaches the accumulator and the overflow screen sees it.)
__device__ __forceinline__ bool any_gt_h8(const uint4& hi, uint32_t thr2) {
    uint32_t m, t;
    asm("v_pk_maximum3_f16 %0, %2, %3, %4\n\t"
        "v_pk_maximum3_f16 %1, %0, %5, %6"
        : "=&v"(m), "=&v"(t)
        : "v"(hi.x), "v"(hi.y), "v"(hi.z), "v"(hi.w), "v"(thr2));
    return t != thr2;
}

// Pre-split K and V once per (b,h) into the exact LDS images the attention workgroups consume (16 query blocks
// share them), so staging inside the attention kernel is a plain copy:
//   Kp[row][2] uint4 : pieces A=[k1|k2], B=[k3|k1] (bf16), slots swapped for (key&15)>=8
//   Vp[row/32][4][16] uint4 : per 32-key pair-tile, [key group g][col j] -> 8 f16 (tile0 keys 4g+r, tile1 keys 4g+r),
//                             cols = [v1 | v2*2^11 | v3*2^22 | 1 | 0 0 0]
__global__ __launch_bounds__(256) void d3pm_attn_prep_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                             int64_t rows, uint4* __restrict__ kp, uint4* __restrict__ vp,
                                                             float* __restrict__ knorm, float4* __restrict__ ksum) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;       // row = h*M + b*L + key (L % 32 == 0)
    if (row >= rows) return;                                          // rows % 32 == 0: whole 32-lane halves leave together
    const float4 rk = *reinterpret_cast<const float4*>(k + row * 4);
    const float4 rv = *reinterpret_cast<const float4*>(v + row * 4);
    const float ks[4] = {rk.x, rk.y, rk.z, rk.w}, vs[4] = {rv.x, rv.y, rv.z, rv.w};
    kv_image_store_k(ks, row, kp);
    kv_image_store_v(vs, row, vp);
    const float nb = half32_norm_bound((rk.x * rk.x + rk.y * rk.y) + (rk.z * rk.z + rk.w * rk.w));
    const float4 sk = make_float4(half32_sum(rk.x), half32_sum(rk.y), half32_sum(rk.z), half32_sum(rk.w));
    if ((threadIdx.x & 31) == 16) {                                   // (the reductions are valid in the upper 16 lanes of each half)
        knorm[row >> 5] = nb;
        ksum[row >> 5] = sk;
    }
}

// One pass over the pair-tiles of a staged chunk for the wave's 4 x 16 queries.  MODE 1: P = hi + lo; 0: hi only; 2: hi, and lo only
// where a tile holds a probability above the lane's threshold thr2 (see the kernel's note) -- returns the number of (pair-tile,
// query sub-tile) pairs that skipped the lo half.  (Deciding once per pair-tile for the four sub-tiles together, with the rare path
// recomputing the scores, measured 3 % faster on flat rows and 7 % slower on trained-like ones: not kept.)
//
// MODE 2 asks the *bound* first: bit u of skipmask[j] (scalar registers) says that no probability of pair-tile u can exceed 2^-PM of
// any row sum of query sub-tile j -- proven from ||q|| ||k|| (kernel note), so the tile takes the hi half only and nothing is
// measured.  Where the bound does not decide, `measure` (wave-uniform) selects the measured test above or hi + lo outright.
// MASK / MEASURE are compile-time: each scalar branch inside the tile loop costs a pipeline bubble per (pair-tile, sub-tile), so the
// loop only carries the tests its chunk needs (measured: 1.43 -> 1.3x ms on rows where nothing is cleared and little is skipped).
template <int KC4, int MODE, bool MASK = false, bool MEASURE = true>
__device__ __forceinline__ int attn_tiles(const AttnSmem4<KC4>& sm, int buf, int npairs, const uint4* kb, int kstep, int lg, int li,
                                          const uint4 (&qfrag)[4], f32x4 (&acc)[4], const uint32_t (&thr2)[4],
                                          const uint32_t (&skipmask)[4]) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int skipped = 0;
    for (int u = 0; u < npairs; ++u) {
        const bf16x8 kf0 = as_frag(kb[(2 * u) * kstep]);
        const bf16x8 kf1 = as_frag(kb[(2 * u + 1) * kstep]);
        const f16x8 vb = as_h8(sm.v[buf][u][lg][li]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, as_frag(qfrag[j]), zero, 0, 0, 0);
            const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, as_frag(qfrag[j]), zero, 0, 0, 0);
            float p[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { p[r] = __builtin_amdgcn_exp2f(s0[r]); p[4 + r] = __builtin_amdgcn_exp2f(s1[r]); }
            uint4 hi, lo;
            if (MODE == 1) {
                split_p8(p, hi, lo);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(hi), vb, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(lo), vb, acc[j], 0, 0, 0);
            } else {
                round_p8(p, hi);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(hi), vb, acc[j], 0, 0, 0);
                if (MODE == 2) {
                    if (!MASK || !((skipmask[j] >> u) & 1u)) {            // scalar test: the bound did not clear this tile
                        if (!MEASURE || __any(any_gt_h8(hi, thr2[j]))) { // wave-uniform: some probability of this tile matters
                            resid_p8(p, hi, lo);
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(lo), vb, acc[j], 0, 0, 0);
                        } else {
                            ++skipped;
                        }
                    }
                }
            }
        }
    }
    return skipped;
}

// Diagnostic (`redo`, optional caller-owned device counter): how often the rare "accumulator overflowed, redo this chunk with a larger
// exponent offset" branch ran (one count per wave and redo attempt); bench.py reports the count per pass.

// PM selects how the probabilities reach the P.V product:
//   PM = 1   P = hi + lo (22 bits) everywhere: the parity default.
//   PM = 0   P = hi (11 bits): 12 of the 16 split instructions and one of the two P.V MFMAs per 512 scores go away (GSDD_ATTN_P=11).
//   PM >= 2  adaptive: the lo half is only added for a (16-query, 32-key) tile in which some probability exceeds 2^-PM of that
//            query's row sum so far.  A probability below that carries a rounding error below 2^-(12+PM) of the row sum; the
//            errors of the skipped tiles add up (randomly signed) to at most 2^-(PM/2) * 2^-12 / sqrt(3) of |v - o|.  With flat
//            rows everything after the first few hundred keys is "small"; with peaky rows only the tiles holding a peak pay.
//            Whether a tile can hold such a probability is first asked of a BOUND, not of the scores: every score of query q
//            against pair-tile u is at most ||q'|| * knorm[u] (Cauchy-Schwarz; q' = q scaled into the log2 domain, knorm = the
//            tile's largest ||k||, written next to the K image by whoever made it), so all its probabilities 2^(s - m) stay below
//            2^-PM * (row sum at the start of the chunk) whenever
//                knorm[u] < (log2(row sum) - PM + m - slack) / ||q'||          (the right side: one number per query and chunk).
//            The minimum of that number over the 16 queries of a sub-tile is compared with the chunk's 12 tile norms in one
//            v_cmp; the resulting mask lives in a scalar register and the tile loop tests a bit of it -- no per-score vector work.
//            A chunk whose tiles are all cleared runs the hi-only loop (flat rows: every chunk after the first); tiles the bound
//            cannot clear fall back to the measured test (two v_pk_maximum3_f16 on the rounded probabilities) or to hi + lo.
//            The bound only ever clears tiles the measured test would have cleared too, so the error budget above is unchanged.
// The error each mode costs is measured by tools/attn_error.py and tabled in DESIGN.md.
template <int KC4, int PM = 1>
__global__ __launch_bounds__(256) void d3pm_attention_v4_kernel(const float* __restrict__ q, const uint4* __restrict__ kp,
                                                                const uint4* __restrict__ vp, const float* __restrict__ knorm,
                                                                const float4* __restrict__ ksum, int B, int L, int H,
                                                                float* __restrict__ out,
                                                                float* __restrict__ lse, unsigned long long* __restrict__ redo) {
    __shared__ AttnSmem4<KC4> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nqb = (L + 255) / 256;
    const unsigned nwg = gridDim.x;
    unsigned wg = blockIdx.x;
    {   // XCD-aware renumbering (see v3)
        const unsigned q8 = nwg / 8, r8 = nwg % 8, xcd = wg % 8, idx = wg / 8;
        wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int qblk = wg % nqb;
    const int h = (wg / nqb) % H, b = wg / (nqb * H);
    const int64_t M = (int64_t)B * L;
    const int64_t base = ((int64_t)h * M + (int64_t)b * L) * 4;
    const float* qh = q + base;
    const uint4* kph = kp + ((int64_t)h * M + (int64_t)b * L) * 2;        // 2 uint4 per key
    const uint4* vph = vp + (((int64_t)h * M + (int64_t)b * L) >> 5) * 64; // 64 uint4 per 32-key pair-tile
    const float* knh = knorm + (((int64_t)h * M + (int64_t)b * L) >> 5);   // one float per 32-key pair-tile
    const float4* ksh = ksum + (((int64_t)h * M + (int64_t)b * L) >> 5);  // one float4 per 32-key pair-tile
    const int li = lane & 15, lg = lane >> 4;
    const int q0 = qblk * 256 + wave * 64;

    // PM >= 2: the tile norms and key sums of this (b, h), requested first so that their latency hides behind the q splits, the first
    // chunk's staging and the first 64 keys' score maximum (they are consumed just before the chunk loop)
    float knm = 0.f;
    float4 ks = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PM >= 2) {
        for (int i = lane; i < (L >> 5); i += 64) {
            knm = fmaxf(knm, knh[i]);
            const float4 t = ksh[i];
            ks.x += t.x; ks.y += t.y; ks.z += t.z; ks.w += t.w;
        }
    }

    if (tid < 16) sm.ones[tid] = (tid == 0 || tid == 1) ? make_uint4(0x3F803F80u, 0x00003F80u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);

    const float qscale = 0.5f * 1.4426950408889634f;
    uint4 qfrag[4];
    float rqn[4];                               // PM >= 2: 1 / (||q'|| of query li, rounded up): the score bound's slope
    float4 qsv[4];                              // PM >= 2: q' of query li (log2 domain), for the row-sum lower bound
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
        // 1.0001: the split products, the norms' own rounding and v_rcp/v_log/v_sqrt (1 ulp each) are all below 2^-20 relative
        rqn[j] = 1.0f / (sqrtf((qs[0] * qs[0] + qs[1] * qs[1]) + (qs[2] * qs[2] + qs[3] * qs[3])) * 1.0001f + 1e-30f);
        qsv[j] = make_float4(qs[0], qs[1], qs[2], qs[3]);
    }

    const int nchunks = (L + KC4 - 1) / KC4;
    // chunk staging = plain copy of the pre-split images: K 768 uint4 + V 768 uint4 per 384-key chunk
    uint4 rk0, rk1, rk2, rv0, rv1, rv2;
    float rkn = 0.f, kn_cur = 0.f;              // PM >= 2: lane u < KC4 / 32 holds the norm bound of pair-tile u of the chunk
    auto load_chunk = [&](int c) {
        // Unconditional loads from a clamped index: a load under "cond ? p[i] : zero" becomes a *flat* load from either the
        // image or a scratch copy of the zero, and flat loads also count on lgkmcnt, so the first LDS fragment wait of the chunk
        // would wait for global memory.  Rows past a short last chunk are staged but never read (npairs bounds the tile loop).
        const int last = 2 * min(KC4, L - c * KC4) - 1;           // keys: a multiple of 32
        const uint4* ksrc = kph + (int64_t)c * KC4 * 2;
        const uint4* vsrc = vph + (int64_t)c * (KC4 / 32) * 64;
        rk0 = ksrc[min(tid, last)]; rk1 = ksrc[min(tid + 256, last)]; rk2 = ksrc[min(tid + 512, last)];
        rv0 = vsrc[min(tid, last)]; rv1 = vsrc[min(tid + 256, last)]; rv2 = vsrc[min(tid + 512, last)];
        if (PM >= 2) rkn = knh[min(c * (KC4 / 32) + min(lane, KC4 / 32 - 1), (L >> 5) - 1)];
    };
    auto store_chunk = [&](int buf, int) {
        uint4* kd = &sm.k[buf][0][0];
        uint4* vd = &sm.v[buf][0][0][0];
        kd[tid] = rk0; kd[tid + 256] = rk1; kd[tid + 512] = rk2;
        vd[tid] = rv0; vd[tid + 256] = rv1; vd[tid + 512] = rv2;
        kn_cur = rkn;
    };

    f32x4 acc[4], sav[4];
    float mq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mq[j] = 0.f;
        acc[j][0] = 0.f; acc[j][1] = 0.f; acc[j][2] = 0.f; acc[j][3] = 0.f;
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    const int slot = ((lg >> 1) ^ (li >> 3)) & 1;
    const uint4* kbase0 = (lg == 3) ? &sm.ones[(li >= 4 && li < 12) ? 0 : 1] : &sm.k[0][li][slot];
    const int kstep = (lg == 3) ? 0 : 32;
    const int kbuf = (lg == 3) ? 0 : KC4 * 2;
    {   // m = ceil(max over the first 64 keys) - 3, shared by the 4 key groups of a query
        const int nt0 = min(4, L >> 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float m0 = -INFINITY;
            for (int t = 0; t < nt0; ++t) {
                const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(kbase0[t * kstep]), as_frag(qfrag[j]), zero, 0, 0, 0);
                m0 = fmaxf(m0, fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
            }
            m0 = fmaxf(m0, __shfl_xor(m0, 16));
            m0 = fmaxf(m0, __shfl_xor(m0, 32));
            mq[j] = ceilf(m0) - 3.f;
            if (lg == 3) qfrag[j] = negm_frag(mq[j]);
        }
    }

    bool prescan = false;                       // wave-uniform
    // PM >= 2: whether this wave tests tiles at all is decided chunk by chunk.  The first chunk has no row sum to compare with; after
    // that a chunk runs the adaptive loop, and if fewer than half of its tiles could skip the lo half the wave goes back to the plain
    // hi + lo loop (no per-tile test, one basic block per pair-tile) and probes again later, each time twice as much later (a row
    // that was not flat after 384 keys seldom becomes flat): chunks 1, 6, 15, 32, ...
    bool adapt = PM >= 2;                       // (the row-sum lower bound below gives the first chunk a threshold too)
    int probe = 1, backoff = 4;
    float jb[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};      // log2 of a lower bound of query li's FINAL row sum (absolute scale)
    float kbj[4] = {0.f, 0.f, 0.f, 0.f};        // the bound's per-sub-tile numbers (wave-uniform values); 0 = clears nothing
    int kb_next = 1, kb_gap = 1;                // chunk of the next recomputation, and the gap after it
    if (PM >= 2) {
        // A priori form of the bound, before any key has been seen: the FINAL row sum of query q is at least L * 2^(-||q'|| KNmax - m)
        // (every score is at least -||q'|| ||k||; KNmax = the largest tile norm of this (b, h)), and the error budget is relative to the
        // final row sum, so a tile whose probabilities stay below 2^-PM of that lower bound may take the hi half only:
        //     ||q'|| knorm[u] - m < log2(L) - PM - ||q'|| KNmax - m   <=>   knorm[u] < (log2(L) - PM - slack) / ||q'|| - KNmax.
        // With near-flat rows (||q'|| ||k|| << log2(L) - PM = 4 at L = 4096) this clears every tile of every chunk, the first included,
        // and the running-sum form below never has to be computed.
        knm = wave_max(knm) * 1.0001f;
        const float budget = log2f((float)L) - (float)PM - 0.02f;
        // Sharper, and per query: Jensen -- log2 sum_j 2^(q'.k_j) >= log2(L) + q'.kmean, kmean from the per-tile key sums the K image's
        // producer left next to the tile norms (fixed summation order: the same bits in every workgroup and every run).  It is within
        // sigma^2 / 2 nats of the true log row sum for scores of spread sigma, where the bound above is off by ||q'|| (KNmax + ||kmean||).
        // It replaces the a priori row sum in the tile bound, and it gives the measured test a threshold relative to the FINAL row sum from
        // the first chunk on (the running sum after one chunk of eleven is 3.5 bits short of it, which made early chunks of rows that are
        // nowhere near peaked take the lo half: trained-like weights).
        ks.x = wave_sum(ks.x); ks.y = wave_sum(ks.y); ks.z = wave_sum(ks.z); ks.w = wave_sum(ks.w);
        const float invL = 1.0f / (float)L;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dx = qsv[j].x * ks.x, dy = qsv[j].y * ks.y, dz = qsv[j].z * ks.z, dw = qsv[j].w * ks.w;
            const float mean = ((dx + dy) + (dz + dw)) * invL, mag = ((fabsf(dx) + fabsf(dy)) + (fabsf(dz) + fabsf(dw))) * invL;
            jb[j] = log2f((float)L) + mean - 1e-5f * mag - 0.02f;       // slack: the float sums behind kmean and the dot product
            kbj[j] = row16_min(fmaxf(fmaxf(budget * rqn[j] - knm, (jb[j] - (float)PM) * rqn[j]), 0.f));
        }
    }
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk(c + 1);
        const int npairs = min(KC4, L - c * KC4) >> 5;
        const uint4* kb = kbase0 + buf * kbuf;
#pragma unroll
        for (int j = 0; j < 4; ++j) sav[j] = acc[j];
        // attempt -1 = "pre-scan": when the previous chunk of this wave had to move an exponent offset, the scores of this chunk are
        // scanned for their maximum BEFORE it is processed and the offsets are moved ahead of the overflow (an S-only pass, ~1/4 of a
        // chunk) instead of after it (the pass plus the whole chunk again).  Rows whose maximum keeps growing along the sequence
        // (peaky softmax rows) then cost ~1.3x a flat row instead of ~2.5x; flat rows never pre-scan.
        bool moved = false;
        for (int attempt = prescan ? -1 : 0; attempt < 16; ++attempt) {
          if (attempt >= 0) {
            uint32_t thr2[4] = {0u, 0u, 0u, 0u};
            uint32_t skipmask[4] = {0u, 0u, 0u, 0u};
            const uint32_t allmask = (1u << npairs) - 1u;
            if (PM >= 2) {
                // The bound's per-sub-tile numbers kbj (kernel note) only grow along a row -- the row sum grows, and log2(row sum) + m
                // does not change when m moves -- so a stale kbj stays a valid (only more cautious) bound: the masks of a chunk are four
                // compares of its tile norms with the kbj in hand.  The expensive part (row sums through the LDS permute path, a
                // logarithm, a row minimum: a serial latency chain of ~600 cycles per chunk that cost 8 % when it ran every chunk) is
                // redone only when the stale numbers leave tiles uncleared, and then at doubling intervals (chunks 1, 2, 4, 8, ...): flat
                // rows compute it once, rows it cannot help four times.
#pragma unroll
                for (int j = 0; j < 4; ++j) skipmask[j] = (uint32_t)__ballot(kn_cur < kbj[j]);
                const bool stale_clears = (skipmask[0] & skipmask[1] & skipmask[2] & skipmask[3] & allmask) == allmask;
                const bool refresh = !stale_clears && c >= kb_next;         // (kb_next >= 1: the first chunk has no row sum yet)
                if (refresh) { kb_next = c + kb_gap; kb_gap *= 2; }
                if (refresh || (adapt && !stale_clears)) {       // (a chunk the bound clears whole never looks at thresholds)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // row sum of query li at the start of this chunk: accumulator column 12, row li = lane 16 (li >> 2) + 12, register li & 3
                        const int src = ((li >> 2) << 4) + 12;
                        const float r0 = __shfl(sav[j][0], src), r1 = __shfl(sav[j][1], src), r2 = __shfl(sav[j][2], src),
                                    r3 = __shfl(sav[j][3], src);
                        const int rr = li & 3;
                        const float rs = rr == 0 ? r0 : (rr == 1 ? r1 : (rr == 2 ? r2 : r3));
                        if (refresh) {
                            // per query the largest tile norm that still keeps every probability below 2^-PM of the row sum; its minimum
                            // over the sub-tile's 16 queries (a DPP row; the four rows of a wave hold the same queries) against the tile
                            // norms held by lanes 0..11.  0.02 of slack in the exponent: v_log_f32 is good to 1 ulp of a value below 2^5
                            // here.  A zero row sum gives -inf, a NaN one NaN: fmaxf turns both into 0, "never cleared".
                            float kbq = (__builtin_amdgcn_logf(rs) - (float)PM + mq[j] - 0.02f) * rqn[j];
                            kbq = fmaxf(kbq, 0.f);
                            kbj[j] = fmaxf(kbj[j], row16_min(kbq));             // never below the a priori number

                            skipmask[j] = (uint32_t)__ballot(kn_cur < kbj[j]);
                        }
                        if (adapt && !stale_clears) {
                            // measured test: threshold of query li = 2^-PM * that row sum, as an f16 replicated in both halves.  A zero
                            // row sum makes every tile take the lo half; an inf threshold (row sum beyond the f16 range) none.
                            // (the row sum so far or the lower bound of the final one, whichever is larger; both relative to 2^m)
                            const float rlb = fmaxf(rs, __builtin_amdgcn_exp2f(jb[j] - mq[j]));
                            const _Float16 th = (_Float16)(rlb * (1.0f / (float)(1 << PM)));
                            const uint32_t hb = (uint32_t)__builtin_bit_cast(unsigned short, th);
                            thr2[j] = hb | (hb << 16);
                        }
                    }
                }
            }
            const uint32_t clr_all = skipmask[0] & skipmask[1] & skipmask[2] & skipmask[3] & allmask;
            int ncleared = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) ncleared += __builtin_popcount(skipmask[j] & allmask);
            int skipped = 0;
            const bool cleared = PM >= 2 && clr_all == allmask;          // every tile of the chunk, for all 64 queries
            if (PM == 0 || cleared) attn_tiles<KC4, 0>(sm, buf, npairs, kb, kstep, lg, li, qfrag, acc, thr2, skipmask);
            else if (PM == 1 || (!adapt && 2 * ncleared < 4 * npairs))
                // (the tile loop with its per-tile branches only pays when at least half of the tiles skip the lo half: measured
                // 1.45 vs 1.27 ms on rows where the bound clears a few tiles per chunk)
                attn_tiles<KC4, 1>(sm, buf, npairs, kb, kstep, lg, li, qfrag, acc, thr2, skipmask);
            else if (!adapt) skipped = ncleared + attn_tiles<KC4, 2, true, false>(sm, buf, npairs, kb, kstep, lg, li, qfrag, acc, thr2, skipmask);
            else if (ncleared == 0) skipped = attn_tiles<KC4, 2, false, true>(sm, buf, npairs, kb, kstep, lg, li, qfrag, acc, thr2, skipmask);
            else skipped = ncleared + attn_tiles<KC4, 2, true, true>(sm, buf, npairs, kb, kstep, lg, li, qfrag, acc, thr2, skipmask);
            // overflow screen (f16 hi part saturated to inf somewhere in this chunk): rare
            // (inf and NaN survive additions, and full-rate adds are cheaper than sixteen half-rate compares)
            float chk = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) chk += (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
            const bool bad = !(fabsf(chk) < 3.0e38f);
            if (!__any(bad)) {
                if (PM >= 2 && !cleared) {                               // (a cleared chunk says nothing about the measured test)
                    if (adapt) {
                        // (a tile that fails the test pays for the test, a taken branch and the lo half in a loop that cannot overlap
                        // them: measured, a chunk with 30 % such tiles costs 1.13 plain hi + lo chunks; the break-even is between 15 and 30 %)
                        if (10 * skipped < 7 * 4 * npairs) { adapt = false; probe = backoff; backoff *= 2; }
                    } else if (--probe <= 0) {
                        adapt = true;
                    }
                }
                break;
            }
            moved = true;
            if (lane == 0 && redo != nullptr) atomicAdd(redo, 1ull);
          }
            // exact maximum of this chunk's scores per query (relative to the current m), then move m so that the chunk maximum lands
            // in (2^2, 2^3]; accumulators restart from the chunk-start copy scaled by 2^-delta.  After an overflow only the queries
            // that can overflow move (maximum >= 2^15); the pre-scan moves earlier (>= 2^12) to keep headroom for the next chunks.
            const float trigger = attempt < 0 ? 12.f : 15.f;
            float cmax[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (int t = 0; t < 2 * npairs; ++t) {
                const bf16x8 kf = as_frag(kb[t * kstep]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 sx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, as_frag(qfrag[j]), zero, 0, 0, 0);
                    cmax[j] = fmaxf(cmax[j], fmaxf(fmaxf(sx[0], sx[1]), fmaxf(sx[2], sx[3])));
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float cm = fmaxf(cmax[j], __shfl_xor(cmax[j], 16));        // lane (query li): max over the 4 key groups
                cm = fmaxf(cm, __shfl_xor(cm, 32));
                const float delta = cm >= trigger ? ceilf(cm) - 3.f : 0.f;   // integer
                if (attempt < 0 && __any(delta != 0.f)) moved = true;
                mq[j] += delta;
                if (lg == 3) qfrag[j] = negm_frag(mq[j]);
                // accumulator rows of this lane are queries 4*lg + r: fetch their delta from the lane holding that query
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dr = __shfl(delta, (lane & 48) + 4 * lg + r);
                    acc[j][r] = sav[j][r] * __builtin_amdgcn_exp2f(-dr);
                }
                sav[j] = acc[j];
            }
        }
        prescan = moved;
        if (c + 1 < nchunks) store_chunk(buf ^ 1, c + 1);
        __syncthreads();
    }

    // ---- epilogue: D[query][col], col on the lane (l&15): out_d = D[d] + D[4+d]/2^11 + D[8+d]/2^22, row sum = D[12]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = acc[j][r];
            const int rowbase = lane & 48;
            const float a1 = __shfl(a, rowbase + (li & 3) + 4);
            const float a2 = __shfl(a, rowbase + (li & 3) + 8);
            const float l = __shfl(a, rowbase + 12);
            const float mrow = __shfl(mq[j], rowbase + 4 * lg + r);       // this row's exponent offset lives on lane (query li)
            const int qi = q0 + 16 * j + 4 * lg + r;
            if (li < 4 && qi < L) {
                const float o = (a + a1 * 0.00048828125f + a2 * 2.384185791015625e-07f) / l;
                out[((int64_t)b * L + qi) * (H * 4) + h * 4 + li] = o;
                if (lse != nullptr && li == 0) lse[(int64_t)h * M + (int64_t)b * L + qi] = mrow + log2f(l);   // log2 domain
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

// The sampler's attention arithmetic (the kernel's PM parameter).  Default: adaptive with threshold 2^-8 of the row sum ("a8") for
// L >= 2048, hi + lo everywhere below that: the rounding errors of the skipped lo halves average out over the keys of a row, as
// 1.4e-4 * sqrt(L - 256) / L for flat rows -- 2.1e-6 rms at L = 4096, but 3.8e-6 at L = 1024 (measured maximum 2.5e-5, above the
// 2e-5 this kernel is held to), and short sequences have too few chunks to gain anything.
// The caller picks per call (`mode`, include/gsdd.h): GSDD_ATTN_P22 -> 1 (hi + lo everywhere: the most exact variant), GSDD_ATTN_P11 -> 0
// (hi only), GSDD_ATTN_A8 / _A12 -> adaptive with threshold 2^-8 / 2^-12 at any L.  Errors and times: DESIGN.md.
static int attn_p_mode(int mode, int L) {
    switch (mode) {
        case GSDD_ATTN_P22: return 1;
        case GSDD_ATTN_P11: return 0;
        case GSDD_ATTN_A8: return 8;
        case GSDD_ATTN_A12: return 12;
        default: return L >= 2048 ? 8 : 1;
    }
}

extern "C" int64_t gsdd_d3pm_attention_workspace_bytes(int B, int L, int H) {
    const int64_t rows = (int64_t)B * L * H;
    return rows * 64 + ((rows + 31) / 32) * 20;  // 32 B (K pieces) + 32 B (V image) per key and head; a key sum (16 B) and a norm bound (4 B) per 32 keys
}

int gsdd_attention_valu(const float* q, const float* k, const float* v, int B, int L, int H, float* out, float* lse,
                        void* stream);                                                              // d3pm_bwd.hip

extern "C" int gsdd_d3pm_attention(const float* q, const float* k, const float* v, int B, int L, int H, float* out,
                                   void* workspace, int64_t workspace_bytes, uint64_t* redo_events, int mode, void* stream) {
    GSDD_CHECK_ARG(q && out && ((k == nullptr) == (v == nullptr)), "null pointer");
    GSDD_CHECK_ARG(mode >= GSDD_ATTN_AUTO && mode <= GSDD_ATTN_KC256, "mode: one of GSDD_ATTN_*");
    GSDD_CHECK_ARG(B > 0 && H > 0 && L > 0, "bad sizes");
    GSDD_CHECK_ARG((int64_t)B * H * ((L + 255) / 256) < (1ll << 31), "grid too large");
    const bool premade = k == nullptr;        // the workspace already holds the K / V images (gsdd_d3pm_layer wrote them)
    const bool force_v3 = mode == GSDD_ATTN_F32PV;     // exact-f32 P.V (mfma 4x4x1) kernel, no workspace use
    GSDD_CHECK_ARG(!premade || (L % 32 == 0 && workspace != nullptr && !force_v3), "k = v = NULL needs the matrix-pipe kernel's images in the workspace");
    if (L % 16 != 0) return gsdd_attention_valu(q, k, v, B, L, H, out, nullptr, stream);     // ragged lengths: VALU kernel
    const dim3 grid((unsigned)(B * H * ((L + 255) / 256)));
    hipStream_t st = (hipStream_t)stream;
    if (L % 32 == 0 && !force_v3 && workspace != nullptr) {
        GSDD_CHECK_ARG(workspace_bytes >= gsdd_d3pm_attention_workspace_bytes(B, L, H), "workspace too small");
        const int64_t rows = (int64_t)B * L * H;
        uint4* kp = reinterpret_cast<uint4*>(workspace);
        uint4* vp = kp + rows * 2;
        float* kn = kv_image_knorm(workspace, rows);
        float4* ksm = kv_image_ksum(workspace, rows);
        unsigned long long* redo = reinterpret_cast<unsigned long long*>(redo_events);
        if (!premade) {
            hipLaunchKernelGGL(d3pm_attn_prep_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, k, v, rows, kp, vp, kn, ksm);
            GSDD_CHECK_LAUNCH();
        }
        const bool kc256 = mode == GSDD_ATTN_KC256;          // development variant: 256-key chunks, hi + lo everywhere
        const int pmode = attn_p_mode(mode, L);
        float* nolse = nullptr;
        if (kc256) hipLaunchKernelGGL(d3pm_attention_v4_kernel<256>, grid, dim3(256), 0, st, q, kp, vp, kn, ksm, B, L, H, out, nolse, redo);
        else if (pmode == 0) hipLaunchKernelGGL((d3pm_attention_v4_kernel<384, 0>), grid, dim3(256), 0, st, q, kp, vp, kn, ksm, B, L, H, out, nolse, redo);
        else if (pmode == 8) hipLaunchKernelGGL((d3pm_attention_v4_kernel<384, 8>), grid, dim3(256), 0, st, q, kp, vp, kn, ksm, B, L, H, out, nolse, redo);
        else if (pmode == 12) hipLaunchKernelGGL((d3pm_attention_v4_kernel<384, 12>), grid, dim3(256), 0, st, q, kp, vp, kn, ksm, B, L, H, out, nolse, redo);
        else hipLaunchKernelGGL(d3pm_attention_v4_kernel<384>, grid, dim3(256), 0, st, q, kp, vp, kn, ksm, B, L, H, out, nolse, redo);
    } else {
        hipLaunchKernelGGL(d3pm_attention_kernel, grid, dim3(256), 0, st, q, k, v, B, L, H, out);      // (never redoes a chunk)
    }
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

// Training forward on the same matrix-pipe kernel: also writes the log2-domain log-sum-exp the backward kernels consume.
// Returns GSDD_OK and sets *done = 1 when the fast path applies (L % 32 == 0 and a workspace was given).
int gsdd_attention_v4_with_lse(const float* q, const float* k, const float* v, int B, int L, int H, float* out, float* lse,
                               void* workspace, int64_t workspace_bytes, int mode, void* stream, int* done) {
    *done = 0;
    GSDD_CHECK_ARG(mode == GSDD_ATTN_AUTO || mode == GSDD_ATTN_P22 || mode == GSDD_ATTN_A8,
                   "training forward: mode is GSDD_ATTN_AUTO, GSDD_ATTN_P22 or GSDD_ATTN_A8");
    if (L % 32 != 0 || workspace == nullptr) return GSDD_OK;
    GSDD_CHECK_ARG(workspace_bytes >= gsdd_d3pm_attention_workspace_bytes(B, L, H), "workspace too small");
    GSDD_CHECK_ARG((int64_t)B * H * ((L + 255) / 256) < (1ll << 31), "grid too large");
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)B * L * H;
    uint4* kp = reinterpret_cast<uint4*>(workspace);
    uint4* vp = kp + rows * 2;
    float* kn = kv_image_knorm(workspace, rows);
    float4* ksm = kv_image_ksum(workspace, rows);
    unsigned long long* noredo = nullptr;
    hipLaunchKernelGGL(d3pm_attn_prep_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, k, v, rows, kp, vp, kn, ksm);
    GSDD_CHECK_LAUNCH();
    // The training forward uses the sampler's adaptive lo half at L >= 2048 (the norm-bound form: on flat rows every tile is cleared a
    // priori): its output error <= 2e-5 of the row scale is below what the gradient parity tests resolve (full-size gradient parity and
    // the 2e-5 attention-backward bar pass unchanged) and the step is 3.2 ms shorter (64.6 -> 61.4 ms).  mode GSDD_ATTN_P22: hi + lo
    // everywhere, the round-2 behaviour; GSDD_ATTN_A8: adaptive at any L.
    if (mode == GSDD_ATTN_A8 || (mode == GSDD_ATTN_AUTO && L >= 2048))
        hipLaunchKernelGGL((d3pm_attention_v4_kernel<384, 8>), dim3((unsigned)(B * H * ((L + 255) / 256))), dim3(256), 0, st, q, kp, vp, kn, ksm,
                           B, L, H, out, lse, noredo);
    else
        hipLaunchKernelGGL(d3pm_attention_v4_kernel<384>, dim3((unsigned)(B * H * ((L + 255) / 256))), dim3(256), 0, st, q, kp, vp, kn, ksm,
                           B, L, H, out, lse, noredo);
    GSDD_CHECK_LAUNCH();
    *done = 1;
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
