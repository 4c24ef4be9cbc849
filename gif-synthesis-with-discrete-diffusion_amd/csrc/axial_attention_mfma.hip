// Axial attention (model_utils.py:318-337, :586-600) for lines of exactly 16 positions and head width 64 or 128 — the shape of
// every axis of the 16x16x16 latent grid at 256 channels / 2 heads — on the f32 matrix pipe, entirely in registers.
//
// One wave per (line, head).  All products are 16x16 tiles of v_mfma_f32_16x16x4_f32 (exact f32 multiply-add):
//   A operand: lane (row = l & 15, slot = l >> 4) gives A[row][k = slot];  B operand: lane gives B[k = slot][col = l & 15];
//   result: lane holds D[4 (l >> 4) + r][l & 15], r = 0..3.
// The order in which the contraction index is walked is free as long as both operands agree, which is what makes the global
// loads coalesce:
//   "row style"  X[row = l & 15][e = 16 (s >> 2) + 4 (l >> 4) + (s & 3)], s = 0 .. D/4-1   (contraction over the head width:
//                a float4 load covers 64 contiguous bytes of each of 16 rows)
//   "tile style" X[row = 4 (l >> 4) + r][e = 64 (t >> 2) + 4 (l & 15) + (t & 3)], t = 0 .. D/16-1   (contraction over the 16
//                positions: a float4 load covers 256 contiguous bytes of a row); results come out in the same (row, e) pattern.
// A 16x16 score tile computed as K Q^T has the query on the lane and four keys in registers, which is exactly the A operand of
// the product with V (tile style): no shuffles, no LDS.  The backward pass needs the scores in both orientations (dQ contracts
// over keys, dK / dV over queries); the second orientation is the same MFMA with the operands swapped, so Q, K, V and dO are
// loaded once in row style and the softmax statistics travel between the orientations with three lane shuffles.
#include "common.hpp"

namespace gsdd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AxialLine {
    int64_t basepos, stride;
};
__device__ __forceinline__ AxialLine axial_line(int64_t line, int axis, int T, int H, int W) {
    AxialLine a;
    if (axis == 0) { a.stride = 1; a.basepos = line * W; }                                              // line = (n, t, h)
    else if (axis == 1) { a.stride = W; a.basepos = (line / W) * H * W + line % W; }                    // line = (n, t, w)
    else { a.stride = (int64_t)H * W; a.basepos = (line / a.stride) * T * a.stride + line % a.stride; } // line = (n, h, w)
    return a;
}

template <int D>
__device__ __forceinline__ void load_row_style(const float* p, float (&x)[D / 4]) {   // p -> X[row][4 (l >> 4)]
#pragma unroll
    for (int i = 0; i < D / 16; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(p + 16 * i);
        x[4 * i + 0] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
}
// rows 4 (l >> 4) + r of a matrix whose row r' starts at p + r' * pitch; p already points at column 4 (l & 15)
template <int D>
__device__ __forceinline__ void load_tile_style(const float* p, int64_t pitch, int g, float (&x)[4][D / 16]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int h = 0; h < D / 64; ++h) {
            const float4 v = *reinterpret_cast<const float4*>(p + (4 * g + r) * pitch + 64 * h);
            x[r][4 * h + 0] = v.x; x[r][4 * h + 1] = v.y; x[r][4 * h + 2] = v.z; x[r][4 * h + 3] = v.w;
        }
}
template <int D>
__device__ __forceinline__ void store_tile_style(float* p, int64_t pitch, int g, const f32x4 (&acc)[D / 16]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int h = 0; h < D / 64; ++h)
            *reinterpret_cast<float4*>(p + (4 * g + r) * pitch + 64 * h) =
                make_float4(acc[4 * h + 0][r], acc[4 * h + 1][r], acc[4 * h + 2][r], acc[4 * h + 3][r]);
}
// D[4g + r][l & 15] = sum_e A[4g + r][e] B[l & 15][e] with both operands in row style
template <int D>
__device__ __forceinline__ f32x4 dot_rows(const float (&a)[D / 4], const float (&b)[D / 4]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < D / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
    return acc;
}
// out tile t: D[4g + r'][e(l & 15, t)] = sum_r W[l & 15][4 (l>>4) + r] X[4 (l>>4) + r][e(l & 15, t)]  (w = the 16x16 weights with
// the output row on the lane, x in tile style)
template <int D>
__device__ __forceinline__ void mix_rows(const f32x4& w, const float (&x)[4][D / 16], f32x4 (&acc)[D / 16]) {
#pragma unroll
    for (int t = 0; t < D / 16; ++t) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], x[r][t], a, 0, 0, 0);
        acc[t] = a;
    }
}
// reductions over the four lane groups (same l & 15)
__device__ __forceinline__ float groups_max(float v) { v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }
__device__ __forceinline__ float groups_sum(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }

template <int D>
__global__ __launch_bounds__(256) void axial_attention_mfma_kernel(const float* __restrict__ qkv, int T, int H, int W, int C,
                                                                   int n_head, int axis, int64_t nwork, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int64_t work = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);        // (line, head), head fastest
    if (work >= nwork) return;
    const int head = (int)(work % n_head);
    const AxialLine ln = axial_line(work / n_head, axis, T, H, W);
    const int64_t pitch_in = 9 * (int64_t)C * ln.stride, pitch_out = 3 * (int64_t)C * ln.stride;
    const float* base = qkv + ln.basepos * (9 * (int64_t)C) + (int64_t)axis * 3 * C + head * D;
    float q[D / 4], k[D / 4], v[4][D / 16];
    load_row_style<D>(base + li * pitch_in + 4 * g, q);
    load_row_style<D>(base + li * pitch_in + C + 4 * g, k);
    load_tile_style<D>(base + 2 * C + 4 * li, pitch_in, g, v);
    // scores with the query on the lane: s[r] = <q[li], k[4g + r]> / sqrt(D)
    f32x4 s = dot_rows<D>(k, q);
    const float scale = 1.0f / sqrtf((float)D);
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[r] *= scale; m = fmaxf(m, s[r]); }
    m = groups_max(m);
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[r] = expf(s[r] - m); l += s[r]; }
    const float inv = 1.f / groups_sum(l);
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] *= inv;
    f32x4 o[D / 16];
    mix_rows<D>(s, v, o);
    store_tile_style<D>(out + ln.basepos * (3 * (int64_t)C) + (int64_t)axis * C + head * D + 4 * li, pitch_out, g, o);
}

template <int D>
__global__ __launch_bounds__(256) void axial_attention_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ datt,
                                                                       int T, int H, int W, int C, int n_head, int axis,
                                                                       int64_t nwork, float* __restrict__ dqkv) {
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int64_t work = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (work >= nwork) return;
    const int head = (int)(work % n_head);
    const AxialLine ln = axial_line(work / n_head, axis, T, H, W);
    const int64_t pitch_in = 9 * (int64_t)C * ln.stride, pitch_out = 3 * (int64_t)C * ln.stride;
    const int64_t off_in = ln.basepos * (9 * (int64_t)C) + (int64_t)axis * 3 * C + head * D;
    const float* base = qkv + off_in;
    const float* gbase = datt + ln.basepos * (3 * (int64_t)C) + (int64_t)axis * C + head * D;
    float* obase = dqkv + off_in;
    const float scale = 1.0f / sqrtf((float)D);

    f32x4 pt, dst, pn, dsn;       // probabilities / score gradients: "t" query on the lane (keys 4g + r), "n" key on the lane
    {
        float q[D / 4], k[D / 4], v[D / 4], go[D / 4];
        load_row_style<D>(base + li * pitch_in + 4 * g, q);
        load_row_style<D>(base + li * pitch_in + C + 4 * g, k);
        load_row_style<D>(base + li * pitch_in + 2 * C + 4 * g, v);
        load_row_style<D>(gbase + li * pitch_out + 4 * g, go);
        pt = dot_rows<D>(k, q);                 // S[query li][key 4g + r]
        pn = dot_rows<D>(q, k);                 // S[query 4g + r][key li]
        dst = dot_rows<D>(v, go);               // dP[query li][key 4g + r]
        dsn = dot_rows<D>(go, v);               // dP[query 4g + r][key li]
    }
    // softmax statistics of query li
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) { pt[r] *= scale; m = fmaxf(m, pt[r]); }
    m = groups_max(m);
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { pt[r] = expf(pt[r] - m); l += pt[r]; }
    const float inv = 1.f / groups_sum(l);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { pt[r] *= inv; rs += pt[r] * dst[r]; }
    rs = groups_sum(rs);
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = pt[r] * (dst[r] - rs) * scale;
    // the other orientation: statistics of query 4g + r live on lane 4g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float mr = __shfl(m, 4 * g + r), ir = __shfl(inv, 4 * g + r), rr = __shfl(rs, 4 * g + r);
        pn[r] = expf(pn[r] * scale - mr) * ir;
        dsn[r] = pn[r] * (dsn[r] - rr) * scale;
    }
    f32x4 acc[D / 16];
    float x[4][D / 16];
    load_tile_style<D>(base + C + 4 * li, pitch_in, g, x);         // K
    mix_rows<D>(dst, x, acc);                                      // dQ = dS K
    store_tile_style<D>(obase + 4 * li, pitch_in, g, acc);
    load_tile_style<D>(base + 4 * li, pitch_in, g, x);             // Q
    mix_rows<D>(dsn, x, acc);                                      // dK = dS^T Q
    store_tile_style<D>(obase + C + 4 * li, pitch_in, g, acc);
    load_tile_style<D>(gbase + 4 * li, pitch_out, g, x);           // dO
    mix_rows<D>(pn, x, acc);                                       // dV = P^T dO
    store_tile_style<D>(obase + 2 * C + 4 * li, pitch_in, g, acc);
}

// Launch helpers for the C entry points (small_ops.hip / vqvae_bwd.hip).  Return false when the shape is not theirs.
bool axial_attention_mfma_launch(const float* qkv, int N, int T, int H, int W, int C, int n_head, int axis, float* out,
                                 hipStream_t st) {
    const int S = axis == 0 ? W : (axis == 1 ? H : T), d = C / n_head;
    if (S != 16 || (d != 64 && d != 128) || C % 4 != 0) return false;
    const int64_t nwork = (int64_t)N * T * H * W / S * n_head;
    const dim3 grid((unsigned)((nwork + 3) / 4)), block(256);
    if (d == 128) hipLaunchKernelGGL(axial_attention_mfma_kernel<128>, grid, block, 0, st, qkv, T, H, W, C, n_head, axis, nwork, out);
    else hipLaunchKernelGGL(axial_attention_mfma_kernel<64>, grid, block, 0, st, qkv, T, H, W, C, n_head, axis, nwork, out);
    return true;
}
bool axial_attention_bwd_mfma_launch(const float* qkv, const float* datt, int N, int T, int H, int W, int C, int n_head, int axis,
                                     float* dqkv, hipStream_t st) {
    const int S = axis == 0 ? W : (axis == 1 ? H : T), d = C / n_head;
    if (S != 16 || (d != 64 && d != 128) || C % 4 != 0) return false;
    const int64_t nwork = (int64_t)N * T * H * W / S * n_head;
    const dim3 grid((unsigned)((nwork + 3) / 4)), block(256);
    if (d == 128) hipLaunchKernelGGL(axial_attention_bwd_mfma_kernel<128>, grid, block, 0, st, qkv, datt, T, H, W, C, n_head, axis, nwork, dqkv);
    else hipLaunchKernelGGL(axial_attention_bwd_mfma_kernel<64>, grid, block, 0, st, qkv, datt, T, H, W, C, n_head, axis, nwork, dqkv);
    return true;
}

}  // namespace gsdd
