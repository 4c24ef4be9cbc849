// Backward kernels of the D3PM denoiser training step (correctness-first versions; the sampling path does not use them).
//   elementwise GELU2, LayerNorm backward (affine / AdaLN with per-batch table gradients), weight-gradient GEMM with the
//   contraction over rows, column sums (bias gradients), head-dim-4 attention forward-with-LSE and its two backward
//   kernels (dQ ; dK,dV), embedding scatter, tiny per-batch linears, Adam.
// Reference semantics: autograd of transformer_utils.py:24-62,138-159,258-282,353-356 and dalle_mask_image_embedding.py:59-79.
#include <stdlib.h>

#include "common.hpp"

namespace gsdd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------ GELU2 (x * sigmoid(1.702 x))
__global__ void gelu2_fwd_kernel(const float* a, float* u, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 v = *reinterpret_cast<const float4*>(a + i);
    float4 o;
    o.x = v.x / (1.f + expf(-1.702f * v.x)); o.y = v.y / (1.f + expf(-1.702f * v.y));
    o.z = v.z / (1.f + expf(-1.702f * v.z)); o.w = v.w / (1.f + expf(-1.702f * v.w));
    *reinterpret_cast<float4*>(u + i) = o;
}
__device__ __forceinline__ float gelu2_grad(float x) {
    const float s = 1.f / (1.f + expf(-1.702f * x));
    return s + 1.702f * x * s * (1.f - s);
}
__global__ void gelu2_bwd_kernel(const float* du, const float* a, float* da, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 g = *reinterpret_cast<const float4*>(du + i);
    const float4 v = *reinterpret_cast<const float4*>(a + i);
    *reinterpret_cast<float4*>(da + i) = make_float4(g.x * gelu2_grad(v.x), g.y * gelu2_grad(v.y), g.z * gelu2_grad(v.z),
                                                     g.w * gelu2_grad(v.w));
}

// ------------------------------------------------------------------ LayerNorm forward for the training step, C = 64, 16 lanes per row:
//   stats[row] = (mean, rstd) (two-pass, biased variance, like nn.LayerNorm) and y = (x - mean) * rstd * gamma[sel] + beta[sel]
//   in one pass over x (the training step keeps y: it is the weight-gradient operand of the GEMM that consumes it)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int64_t M, float eps, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const int64_t* __restrict__ sel, int gstride,
                                                     int rows_per_batch, float* __restrict__ stats, float* __restrict__ y) {
    const int lane16 = threadIdx.x & 15;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool ok = row < M;
    const int64_t rc = ok ? row : 0;
    const int c = lane16 * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + rc * 64 + c);
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.f / 64.f);
    const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
    float q = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q * (1.f / 64.f) + eps);
    const int b = (int)((uint32_t)rc / (uint32_t)rows_per_batch);
    const int64_t sl = sel != nullptr ? sel[b] : 0;
    const float4 g = *reinterpret_cast<const float4*>(gamma + sl * gstride + c);
    const float4 bt = *reinterpret_cast<const float4*>(beta + sl * gstride + c);
    if (ok) {
        if (lane16 == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
        *reinterpret_cast<float4*>(y + row * 64 + c) =
            make_float4(a0 * rstd * g.x + bt.x, a1 * rstd * g.y + bt.y, a2 * rstd * g.z + bt.z, a3 * rstd * g.w + bt.w);
    }
}

// ------------------------------------------------------------------ LayerNorm backward, C = 64, 16 lanes per row
//   h = xhat * gamma[sel] + beta[sel],  xhat = (x - mean) * rstd
//   dx_out = dx_in + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dh * gamma
//   dgamma[sel][c] += sum_rows dh*xhat ; dbeta[sel][c] += sum_rows dh      (float atomics after a block reduction)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* dh, const float* x, const float* stats, const float* gamma,
                                                     const int64_t* sel, int gstride, int rows_per_batch, int64_t M,
                                                     const float* dx_in, float* dx_out, float* dgamma, float* dbeta,
                                                     int gacc_stride, int acc_by_batch, int rit) {
    // a block covers `rit` groups of 16 rows; the gamma / beta gradient contributions of its rows are summed in registers and LDS
    // first, so that the atomics onto the (few) gradient columns are one per column and block: with one 16-row group per block the
    // 64 K atomics per call onto 128 addresses were most of the kernel's time
    __shared__ float sg[16][64], sb[16][64];
    const int tid = threadIdx.x, lane16 = tid & 15, rloc = tid >> 4;
    const int c = lane16 * 4;
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < rit; ++it) {
        const int64_t row = ((int64_t)blockIdx.x * rit + it) * 16 + rloc;
        const bool ok = row < M;
        const int64_t rc = ok ? row : 0;
        const int b = (int)((uint32_t)rc / (uint32_t)rows_per_batch);
        const int64_t s = sel != nullptr ? sel[b] : 0;
        const float4 d = *reinterpret_cast<const float4*>(dh + rc * 64 + c);
        const float4 xv = *reinterpret_cast<const float4*>(x + rc * 64 + c);
        const float4 gm = *reinterpret_cast<const float4*>(gamma + s * gstride + c);
        const float mean = stats[2 * rc], rstd = stats[2 * rc + 1];
        const float xh[4] = {(xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd};
        const float dv[4] = {d.x, d.y, d.z, d.w};
        const float g[4] = {d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w};
        float s1 = (g[0] + g[1]) + (g[2] + g[3]);
        float s2 = (g[0] * xh[0] + g[1] * xh[1]) + (g[2] * xh[2] + g[3] * xh[3]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        const float m1 = s1 * (1.f / 64.f), m2 = s2 * (1.f / 64.f);
        if (ok) {
            float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dx_in != nullptr) o4 = *reinterpret_cast<const float4*>(dx_in + rc * 64 + c);
            o4.x += rstd * (g[0] - m1 - xh[0] * m2); o4.y += rstd * (g[1] - m1 - xh[1] * m2);
            o4.z += rstd * (g[2] - m1 - xh[2] * m2); o4.w += rstd * (g[3] - m1 - xh[3] * m2);
            *reinterpret_cast<float4*>(dx_out + rc * 64 + c) = o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) { ag[e] += dv[e] * xh[e]; ab[e] += dv[e]; }
        }
    }
    if (dgamma != nullptr) {
        // the rows of a block belong to one batch element when rows_per_batch % (16 * rit) == 0 (checked on the host)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sg[rloc][c + e] = ag[e]; sb[rloc][c + e] = ab[e]; }
        __syncthreads();
        if (tid < 64) {
            float a = 0.f, bb = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a += sg[r][tid]; bb += sb[r][tid]; }
            const int64_t row0 = (int64_t)blockIdx.x * rit * 16;
            const int b0 = (int)((uint32_t)(row0 < M ? row0 : 0) / (uint32_t)rows_per_batch);
            const int64_t slot = acc_by_batch ? b0 : 0;
            atomicAdd(dgamma + slot * gacc_stride + tid, a);
            atomicAdd(dbeta + slot * gacc_stride + tid, bb);
        }
    }
}

// ------------------------------------------------------------------ weight gradient: dW[n][k] += sum_m dY[m][n] * X[m][k]
// grid (row slabs, N/64, K/64); each workgroup accumulates `slabs` x 128 rows in registers, then one atomic pass.
constexpr int WG_ROWS = 128;
// db != NULL: the blocks of the first K tile also add the column sums of their dY rows (the bias gradient) -- every block stages
// those rows anyway; one atomic per column and block instead of a separate pass over dY.
__global__ __launch_bounds__(256) void wgrad_kernel(const float* dY, int ldy, const float* X, int ldx, int64_t M, int N, int K,
                                                    float* dW, int slabs, float* db) {
    __shared__ float sy[WG_ROWS][64], sx[WG_ROWS][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.y * 64, k0 = blockIdx.z * 64;
    const int wn = wave >> 1, wk = wave & 1;                   // 32x32 quadrant of the 64x64 output tile
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const bool do_bias = db != nullptr && blockIdx.z == 0;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);              // column sums of this thread's 4 columns ((tid & 15) * 4) over its rows
    // A slab's 2 x 8 float4 per thread are requested together and unconditionally (clamped addresses; what lies outside the matrix is
    // zeroed when it is staged), and the next slab's are in flight during this slab's MFMAs.  (With the loads under their bounds tests
    // inside the staging loop the kernel paid eight memory round trips per slab, one after the other: 65 us where the bytes need 28.)
    const int c4 = (tid & 15) * 4, rb = tid >> 4;              // this thread's column run and first row of a slab (then every 16th)
    const bool y_ok = n0 + c4 < N, x_ok = k0 + c4 < K;
    const int cy = y_ok ? n0 + c4 : 0, cx = x_ok ? k0 + c4 : 0;
    float4 vy[8], vx[8];
    auto load_slab = [&](int64_t r0) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int64_t rr = r0 + rb + 16 * it;
            const int64_t rc = rr < M ? rr : M - 1;
            vy[it] = *reinterpret_cast<const float4*>(dY + rc * ldy + cy);
            vx[it] = *reinterpret_cast<const float4*>(X + rc * ldx + cx);
        }
    };
    const int64_t first = (int64_t)blockIdx.x * slabs * WG_ROWS;
    if (first < M) load_slab(first);
    for (int sl = 0; sl < slabs; ++sl) {
        const int64_t r0 = ((int64_t)blockIdx.x * slabs + sl) * WG_ROWS;
        if (r0 >= M) break;
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = rb + 16 * it;
            const bool row_ok = r0 + r < M;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 y4 = (row_ok && y_ok) ? vy[it] : z, x4 = (row_ok && x_ok) ? vx[it] : z;
            *reinterpret_cast<float4*>(&sy[r][c4]) = y4;
            *reinterpret_cast<float4*>(&sx[r][c4]) = x4;
            cs.x += y4.x; cs.y += y4.y; cs.z += y4.z; cs.w += y4.w;
        }
        __syncthreads();
        if (sl + 1 < slabs && r0 + WG_ROWS < M) load_slab(r0 + WG_ROWS);
#pragma unroll 8
        for (int s = 0; s < WG_ROWS / 2; ++s) {
            const float a = sy[2 * s + lh][wn * 32 + li];          // A[i = n][k = m]
            const float b = sx[2 * s + lh][wk * 32 + li];          // B[k = m][j = k]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int k = k0 + wk * 32 + li;
        if (n < N && k < K) atomicAdd(dW + (int64_t)n * K + k, acc[r]);
    }
    if (do_bias) {                                             // block-uniform
        __syncthreads();
        *reinterpret_cast<float4*>(&sy[tid >> 4][(tid & 15) * 4]) = cs;
        __syncthreads();
        if (tid < 64 && n0 + tid < N) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += sy[g][tid];
            atomicAdd(db + n0 + tid, t);
        }
    }
}

// column sums: out[n] += sum_m Y[m][n]
// out[n] += sum_r Y[r][n].  256 threads = cpb columns x (256 / cpb) row groups; CS_ROWS rows per block, LDS reduction over the row
// groups, one atomic per column and block.
constexpr int CS_ROWS = 64;    // rows per block for the per-batch sums; the column sums pick 64 (narrow and short: latency-bound, wants blocks)
                               // or 256 (wide or long: fewer atomics per column) per launch
__global__ __launch_bounds__(256) void colsum_kernel(const float* Y, int ld, int64_t M, int N, float* out, int cpb, int rows) {
    __shared__ float red[256];
    const int c = threadIdx.x % cpb, rg = threadIdx.x / cpb, nrg = 256 / cpb;
    const int n = blockIdx.y * cpb + c;
    const int64_t r0 = (int64_t)blockIdx.x * rows;
    const int64_t r1 = r0 + rows < M ? r0 + rows : M;
    float s = 0.f;
    if (n < N)
        for (int64_t r = r0 + rg; r < r1; r += nrg) s += Y[r * ld + n];
    red[threadIdx.x] = s;
    __syncthreads();
    if (rg == 0 && n < N) {
        for (int g = 1; g < nrg; ++g) s += red[g * cpb + c];
        atomicAdd(out + n, s);
    }
}
static inline int colsum_cpb(int N) {
    int cpb = 256;
    while (cpb > 1 && cpb / 2 >= N) cpb /= 2;
    return cpb;
}

// per-batch row sums: out[b][c] = sum_{l} Y[b*L + l][c]   (gradient of the broadcast cross-attention vector); out is zeroed
// by the caller, blocks of 256 rows accumulate with atomics
__global__ __launch_bounds__(256) void batch_rowsum_kernel(const float* Y, int L, int C, float* out, int cpb) {
    __shared__ float red[256];
    const int c = threadIdx.x % cpb, rg = threadIdx.x / cpb, nrg = 256 / cpb;
    const int b = blockIdx.z, n = blockIdx.y * cpb + c;
    const int l0 = blockIdx.x * CS_ROWS;
    const int l1 = l0 + CS_ROWS < L ? l0 + CS_ROWS : L;
    float s = 0.f;
    if (n < C)
        for (int l = l0 + rg; l < l1; l += nrg) s += Y[((int64_t)b * L + l) * C + n];
    red[threadIdx.x] = s;
    __syncthreads();
    if (rg == 0 && n < C) {
        for (int g = 1; g < nrg; ++g) s += red[g * cpb + c];
        atomicAdd(out + (int64_t)b * C + n, s);
    }
}

// ------------------------------------------------------------------ attention, head dim 4, training versions (VALU)
// forward with log-sum-exp: one lane per query, keys streamed through LDS (broadcast reads)
constexpr int AT_KC = 256;
__global__ __launch_bounds__(256) void attn_train_fwd_kernel(const float* q, const float* k, const float* v, int B, int L, int H,
                                                             float* out, float* lse) {
    __shared__ float4 sk[AT_KC], sv[AT_KC];
    const int h = blockIdx.y, b = blockIdx.z;
    const int64_t M = (int64_t)B * L, base = (int64_t)h * M + (int64_t)b * L;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const bool ok = qi < L;
    const float c = 0.5f * 1.4426950408889634f;
    float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) qv = *reinterpret_cast<const float4*>(q + (base + qi) * 4);
    qv.x *= c; qv.y *= c; qv.z *= c; qv.w *= c;
    float m = -INFINITY, l = 0.f, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
    for (int c0 = 0; c0 < L; c0 += AT_KC) {
        __syncthreads();
        const int key = c0 + threadIdx.x;
        sk[threadIdx.x] = key < L ? *reinterpret_cast<const float4*>(k + (base + key) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        sv[threadIdx.x] = key < L ? *reinterpret_cast<const float4*>(v + (base + key) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        const int n = min(AT_KC, L - c0);
        for (int j = 0; j < n; ++j) {
            const float4 kk = sk[j], vv = sv[j];
            const float s = fmaf(qv.x, kk.x, fmaf(qv.y, kk.y, fmaf(qv.z, kk.z, qv.w * kk.w)));
            if (s > m) {
                const float a = __builtin_amdgcn_exp2f(m - s);
                l *= a; o0 *= a; o1 *= a; o2 *= a; o3 *= a;
                m = s;
            }
            const float p = __builtin_amdgcn_exp2f(s - m);
            l += p;
            o0 = fmaf(p, vv.x, o0); o1 = fmaf(p, vv.y, o1); o2 = fmaf(p, vv.z, o2); o3 = fmaf(p, vv.w, o3);
        }
    }
    if (ok) {
        const float inv = 1.f / l;
        *reinterpret_cast<float4*>(out + ((int64_t)b * L + qi) * (H * 4) + h * 4) = make_float4(o0 * inv, o1 * inv, o2 * inv, o3 * inv);
        if (lse != nullptr) lse[base + qi] = m + log2f(l);    // log2 domain
    }
}

// dQ: lane per query.  dqkv rows [M][3*H*4] = (dq | dk | dv) per row; also writes D = dO . O per (head, row)
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const float* q, const float* k, const float* v, const float* o,
                                                          const float* dO, const float* lse, int B, int L, int H, float* dqkv,
                                                          float* Dout) {
    __shared__ float4 sk[AT_KC], sv[AT_KC];
    const int h = blockIdx.y, b = blockIdx.z;
    const int64_t M = (int64_t)B * L, base = (int64_t)h * M + (int64_t)b * L;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const bool ok = qi < L;
    const int qc = ok ? qi : L - 1;
    const float c = 0.5f * 1.4426950408889634f;
    float4 qv = *reinterpret_cast<const float4*>(q + (base + qc) * 4);
    qv.x *= c; qv.y *= c; qv.z *= c; qv.w *= c;
    const int64_t row = (int64_t)b * L + qc;
    const float4 g = *reinterpret_cast<const float4*>(dO + row * (H * 4) + h * 4);
    const float4 ov = *reinterpret_cast<const float4*>(o + row * (H * 4) + h * 4);
    const float Dq = (g.x * ov.x + g.y * ov.y) + (g.z * ov.z + g.w * ov.w);
    const float ls = lse[base + qc];
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    for (int c0 = 0; c0 < L; c0 += AT_KC) {
        __syncthreads();
        const int key = c0 + threadIdx.x;
        sk[threadIdx.x] = key < L ? *reinterpret_cast<const float4*>(k + (base + key) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        sv[threadIdx.x] = key < L ? *reinterpret_cast<const float4*>(v + (base + key) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        const int n = min(AT_KC, L - c0);
        for (int j = 0; j < n; ++j) {
            const float4 kk = sk[j], vv = sv[j];
            const float s = fmaf(qv.x, kk.x, fmaf(qv.y, kk.y, fmaf(qv.z, kk.z, qv.w * kk.w)));
            const float p = __builtin_amdgcn_exp2f(s - ls);
            const float dp = fmaf(g.x, vv.x, fmaf(g.y, vv.y, fmaf(g.z, vv.z, g.w * vv.w)));
            const float ds = p * (dp - Dq);
            d0 = fmaf(ds, kk.x, d0); d1 = fmaf(ds, kk.y, d1); d2 = fmaf(ds, kk.z, d2); d3 = fmaf(ds, kk.w, d3);
        }
    }
    if (ok) {
        *reinterpret_cast<float4*>(dqkv + row * (3 * H * 4) + h * 4) = make_float4(0.5f * d0, 0.5f * d1, 0.5f * d2, 0.5f * d3);
        Dout[base + qi] = Dq;
    }
}

// dK, dV: lane per key, queries streamed through LDS
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const float* q, const float* k, const float* v, const float* dO,
                                                           const float* lse, const float* Dq, int B, int L, int H, float* dqkv) {
    __shared__ float4 sq[AT_KC], sg[AT_KC];
    __shared__ float sl[AT_KC], sd[AT_KC];
    const int h = blockIdx.y, b = blockIdx.z;
    const int64_t M = (int64_t)B * L, base = (int64_t)h * M + (int64_t)b * L;
    const int kj = blockIdx.x * 256 + threadIdx.x;
    const bool ok = kj < L;
    const int kc = ok ? kj : L - 1;
    const float c = 0.5f * 1.4426950408889634f;
    const float4 kk = *reinterpret_cast<const float4*>(k + (base + kc) * 4);
    const float4 vv = *reinterpret_cast<const float4*>(v + (base + kc) * 4);
    float dk0 = 0.f, dk1 = 0.f, dk2 = 0.f, dk3 = 0.f, dv0 = 0.f, dv1 = 0.f, dv2 = 0.f, dv3 = 0.f;
    for (int c0 = 0; c0 < L; c0 += AT_KC) {
        __syncthreads();
        const int qi = c0 + threadIdx.x;
        if (qi < L) {
            sq[threadIdx.x] = *reinterpret_cast<const float4*>(q + (base + qi) * 4);
            sg[threadIdx.x] = *reinterpret_cast<const float4*>(dO + ((int64_t)b * L + qi) * (H * 4) + h * 4);
            sl[threadIdx.x] = lse[base + qi];
            sd[threadIdx.x] = Dq[base + qi];
        }
        __syncthreads();
        const int n = min(AT_KC, L - c0);
        for (int i = 0; i < n; ++i) {
            const float4 qq = sq[i], g = sg[i];
            const float s = c * fmaf(qq.x, kk.x, fmaf(qq.y, kk.y, fmaf(qq.z, kk.z, qq.w * kk.w)));
            const float p = __builtin_amdgcn_exp2f(s - sl[i]);
            dv0 = fmaf(p, g.x, dv0); dv1 = fmaf(p, g.y, dv1); dv2 = fmaf(p, g.z, dv2); dv3 = fmaf(p, g.w, dv3);
            const float dp = fmaf(g.x, vv.x, fmaf(g.y, vv.y, fmaf(g.z, vv.z, g.w * vv.w)));
            const float ds = p * (dp - sd[i]);
            dk0 = fmaf(ds, qq.x, dk0); dk1 = fmaf(ds, qq.y, dk1); dk2 = fmaf(ds, qq.z, dk2); dk3 = fmaf(ds, qq.w, dk3);
        }
    }
    if (ok) {
        const int64_t row = (int64_t)b * L + kj;
        float* dst = dqkv + row * (3 * H * 4);
        *reinterpret_cast<float4*>(dst + H * 4 + h * 4) = make_float4(0.5f * dk0, 0.5f * dk1, 0.5f * dk2, 0.5f * dk3);
        *reinterpret_cast<float4*>(dst + 2 * H * 4 + h * 4) = make_float4(dv0, dv1, dv2, dv3);
    }
}

// ------------------------------------------------------------------ embedding backward: demb[tok] += dx ; dpos[l] += dx
// Half of x_t is the [MASK] token (id n_embed - 1) on average, so its table row would take tens of thousands of atomics on the
// same 64 addresses; each block sums its [MASK] rows locally (registers, then LDS over the 16 row lanes) and issues one atomic
// per column.  Block = 256 rows: thread (row lane tid >> 4, column quad tid & 15 [+16 ...]).
constexpr int EB_ROWS = 256;
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* dx, const int64_t* tok, int64_t rows, int L, int D, int n_embed,
                                                        float* demb, float* dpos) {
    __shared__ float4 red[16][16];
    const int rl = threadIdx.x >> 4, cq = threadIdx.x & 15;
    const int64_t r0 = (int64_t)blockIdx.x * EB_ROWS;
    const int hot = n_embed - 1;
    for (int c = 4 * cq; c < D; c += 64) {
        float4 hs = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t row = r0 + rl; row < r0 + EB_ROWS && row < rows; row += 16) {
            int64_t t = tok[row];
            t = t < 0 ? 0 : (t >= n_embed ? n_embed - 1 : t);
            const float4 g = *reinterpret_cast<const float4*>(dx + row * D + c);
            float* p = dpos + (int64_t)(row % L) * D + c;
            atomicAdd(p + 0, g.x); atomicAdd(p + 1, g.y); atomicAdd(p + 2, g.z); atomicAdd(p + 3, g.w);
            if (t == hot) {
                hs.x += g.x; hs.y += g.y; hs.z += g.z; hs.w += g.w;
            } else {
                float* e = demb + t * D + c;
                atomicAdd(e + 0, g.x); atomicAdd(e + 1, g.y); atomicAdd(e + 2, g.z); atomicAdd(e + 3, g.w);
            }
        }
        red[rl][cq] = hs;
        __syncthreads();
        if (rl == 0) {
            float4 a = red[0][cq];
            for (int i = 1; i < 16; ++i) { const float4 b = red[i][cq]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            float* e = demb + (int64_t)hot * D + c;
            atomicAdd(e + 0, a.x); atomicAdd(e + 1, a.y); atomicAdd(e + 2, a.z); atomicAdd(e + 3, a.w);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ tiny linears over a handful of rows (one thread per output)
// y[r][j] = W[j][:] . x[r][:] + b[j] backward:  dx[r][k] = sum_j dy[r][j] W[j][k] ; dW[j][k] += sum_r dy[r][j] x[r][k] ; db[j] += sum_r dy
__global__ void small_linear_bwd_kernel(const float* dy, const float* x, const float* w, int R, int Cin, int Cout, float* dx,
                                        float* dw, float* db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (dx != nullptr && i < R * Cin) {
        const int r = i / Cin, k = i % Cin;
        float s = 0.f;
#pragma unroll 16
        for (int j = 0; j < Cout; ++j) s = fmaf(dy[r * Cout + j], w[(int64_t)j * Cin + k], s);      // (unrolled: loads in flight together)
        dx[i] = s;
    }
    if (i < Cout * Cin) {
        const int j = i / Cin, k = i % Cin;
        float s = 0.f;
#pragma unroll 8
        for (int r = 0; r < R; ++r) s = fmaf(dy[r * Cout + j], x[(int64_t)r * Cin + k], s);
        dw[i] += s;
    }
    if (db != nullptr && i < Cout) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += dy[r * Cout + i];
        db[i] += s;
    }
}

// AdaLayerNorm table backward: table[t] = (1 + W silu(e_t) + b | ...) ; per-batch dtable rows (B x 2D) at timesteps t[b]
//   dW[j][k] += sum_b dtab[b][j] silu(e[t_b][k]) ; db[j] += sum_b dtab[b][j] ; de[t_b][k] += silu'(e) * sum_j dtab[b][j] W[j][k]
// SMEM: the B x D values silu(e[t_b][k]) and silu'(e[t_b][k]) are made once per block in LDS (2 B D floats); without it every thread
// walks B dependent (t[b] -> emb row) loads with an expf each, which made this 58 us of pure latency per layer.
template <bool SMEM>
__global__ __launch_bounds__(256) void adaln_bwd_kernel(const float* dtab, const int64_t* t, int B, int D, const float* emb, const float* w,
                                                        float* demb, float* dw, float* db) {
    extern __shared__ float adaln_lds[];
    float* sl = adaln_lds;                 // silu(e)   [B][D]
    float* dsl = adaln_lds + B * D;        // silu'(e)  [B][D]
    if (SMEM) {
        for (int u = threadIdx.x; u < B * D; u += 256) {
            const float e = emb[t[u / D] * D + u % D];
            const float sg = 1.f / (1.f + expf(-e));
            sl[u] = e * sg;
            dsl[u] = sg * (1.f + e * (1.f - sg));
        }
        __syncthreads();
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int D2 = 2 * D;
    if (i < D2 * D) {
        const int j = i / D, k = i % D;
        float s = 0.f;
#pragma unroll 8
        for (int b = 0; b < B; ++b) {
            float si;
            if (SMEM) si = sl[b * D + k];
            else { const float e = emb[t[b] * D + k]; si = e / (1.f + expf(-e)); }
            s = fmaf(dtab[b * D2 + j], si, s);
        }
        dw[i] += s;
    }
    if (i < D2) {
        float s = 0.f;
#pragma unroll 8
        for (int b = 0; b < B; ++b) s += dtab[b * D2 + i];
        db[i] += s;
    }
    if (i < B * D) {
        const int b = i / D, k = i % D;
        float s = 0.f;
#pragma unroll 16
        for (int j = 0; j < D2; ++j) s = fmaf(dtab[b * D2 + j], w[(int64_t)j * D + k], s);      // (unrolled: 16 loads in flight, not one)
        float ds;
        if (SMEM) ds = dsl[i];
        else { const float e = emb[t[b] * D + k]; const float sg = 1.f / (1.f + expf(-e)); ds = sg * (1.f + e * (1.f - sg)); }
        atomicAdd(demb + t[b] * D + k, s * ds);
    }
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam semantics, no weight decay / amsgrad)
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                            float bc1, float bc2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

// Adam over many tensors in one launch: table[b] = {p, g, m, v, n} for block b (up to ADAM_CHUNK elements of one tensor)
__global__ __launch_bounds__(256) void adam_multi_kernel(const int64_t* __restrict__ table, float lr, float b1, float b2, float eps,
                                                         float bc1, float bc2) {
    const int64_t* e = table + (int64_t)blockIdx.x * 5;
    float* p = reinterpret_cast<float*>(e[0]);
    const float* g = reinterpret_cast<const float*>(e[1]);
    float* m = reinterpret_cast<float*>(e[2]);
    float* v = reinterpret_cast<float*>(e[3]);
    const int n = (int)e[4];
    for (int i = threadIdx.x; i < n; i += 256) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

// the same with the step count in device memory (a captured training step is replayed: nothing of the update may be baked into the
// launch): the bias corrections 1 - beta^step are evaluated per workgroup, in double (two pow calls per 4096 elements)
__global__ __launch_bounds__(256) void adam_multi_dev_kernel(const int64_t* __restrict__ table, float lr, float b1, float b2, float eps,
                                                             const int64_t* __restrict__ step_dev) {
    const double st = (double)step_dev[0];
    const float bc1 = (float)(1.0 - pow((double)b1, st)), bc2 = (float)(1.0 - pow((double)b2, st));
    const int64_t* e = table + (int64_t)blockIdx.x * 5;
    float* p = reinterpret_cast<float*>(e[0]);
    const float* g = reinterpret_cast<const float*>(e[1]);
    float* m = reinterpret_cast<float*>(e[2]);
    float* v = reinterpret_cast<float*>(e[3]);
    const int n = (int)e[4];
    for (int i = threadIdx.x; i < n; i += 256) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

}  // namespace gsdd

using namespace gsdd;

extern "C" int gsdd_adam_multi_dev(const int64_t* table, int n_blocks, float lr, float beta1, float beta2, float eps,
                                   const int64_t* step_dev, void* stream) {
    GSDD_CHECK_ARG(table != nullptr && n_blocks > 0 && step_dev != nullptr, "bad args");
    hipLaunchKernelGGL(adam_multi_dev_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, table, lr, beta1, beta2, eps,
                       step_dev);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_adam_multi(const int64_t* table, int n_blocks, float lr, float beta1, float beta2, float eps, int step,
                               void* stream) {
    GSDD_CHECK_ARG(table != nullptr && n_blocks > 0 && step >= 1, "bad args");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, table, lr, beta1, beta2, eps,
                       bc1, bc2);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_gelu2(const float* a, const float* du, float* out, int64_t n, int backward, void* stream) {
    GSDD_CHECK_ARG(a && out && n > 0 && n % 4 == 0 && (!backward || du), "bad args");
    const unsigned grid = (unsigned)((n / 4 + 255) / 256);
    if (backward) hipLaunchKernelGGL(gelu2_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, du, a, out, n);
    else hipLaunchKernelGGL(gelu2_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, out, n);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_ln_fwd(const float* x, int64_t M, int C, float eps, const float* gamma, const float* beta, const int64_t* sel,
                           int gstride, int rows_per_batch, float* stats, float* y, void* stream) {
    GSDD_CHECK_ARG(x && gamma && beta && stats && y, "null pointer");
    GSDD_CHECK_ARG(C == 64 && M > 0 && rows_per_batch > 0, "C must be 64");
    GSDD_CHECK_ARG(sel != nullptr || gstride == 0, "gstride needs a row selector");
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((M * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, M, eps, gamma,
                       beta, sel, gstride, rows_per_batch, stats, y);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_ln_bwd(const float* dh, const float* x, const float* stats, const float* gamma, const int64_t* sel,
                           int gstride, int rows_per_batch, int64_t M, int C, const float* dx_in, float* dx_out, float* dgamma,
                           float* dbeta, int gacc_stride, int acc_by_batch, void* stream) {
    GSDD_CHECK_ARG(dh && x && stats && gamma && dx_out, "null pointer");
    GSDD_CHECK_ARG(C == 64 && M > 0 && rows_per_batch > 0, "kernel is specialised for 64 features");
    GSDD_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "dgamma/dbeta come together");
    GSDD_CHECK_ARG(!acc_by_batch || rows_per_batch % 16 == 0, "per-batch accumulation needs rows_per_batch % 16 == 0");
    int rit = 4;                                               // 16-row groups per block: fewer, fatter atomics
    while (rit > 1 && ((acc_by_batch && rows_per_batch % (16 * rit) != 0) || M < (int64_t)16 * rit * 512)) rit >>= 1;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)((M + 16 * rit - 1) / (16 * rit))), dim3(256), 0, (hipStream_t)stream, dh, x, stats,
                       gamma, sel, gstride, rows_per_batch, M, dx_in, dx_out, dgamma, dbeta, gacc_stride, acc_by_batch, rit);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_wgrad(const float* dY, int ldy, const float* X, int ldx, int64_t M, int N, int K, float* dW, float* db,
                          void* stream) {
    GSDD_CHECK_ARG(dY && X && dW && M > 0 && N > 0 && K > 0, "bad args");
    GSDD_CHECK_ARG(N % 4 == 0 && K % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0, "N, K and pitches must be multiples of 4");
    // 8 slabs of 128 rows per workgroup amortise the atomic pass, but a 64 x 64 weight over 65536 rows is then 64 workgroups on 256
    // CUs: fewer slabs until the grid reaches ~2 workgroups per CU
    int slabs = 8;
    const int64_t tiles = (int64_t)((N + 63) / 64) * ((K + 63) / 64);
    while (slabs > 1 && ((M + (int64_t)WG_ROWS * slabs - 1) / ((int64_t)WG_ROWS * slabs)) * tiles < 512) slabs >>= 1;
    const dim3 grid((unsigned)((M + (int64_t)WG_ROWS * slabs - 1) / ((int64_t)WG_ROWS * slabs)), (N + 63) / 64, (K + 63) / 64);
    hipLaunchKernelGGL(wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, dY, ldy, X, ldx, M, N, K, dW, slabs, db);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_colsum(const float* Y, int ld, int64_t M, int N, float* out, void* stream) {
    GSDD_CHECK_ARG(Y && out && M > 0 && N > 0, "bad args");
    const int cpb = colsum_cpb(N), rows = (N <= 256 && M < 262144) ? 64 : 256;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((M + rows - 1) / rows), (N + cpb - 1) / cpb), dim3(256), 0,
                       (hipStream_t)stream, Y, ld, M, N, out, cpb, rows);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_batch_rowsum(const float* Y, int B, int L, int C, float* out, void* stream) {
    GSDD_CHECK_ARG(Y && out && B > 0 && L > 0 && C > 0, "bad args");
    GSDD_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)B * C * sizeof(float), (hipStream_t)stream));
    const int cpb = colsum_cpb(C);
    hipLaunchKernelGGL(batch_rowsum_kernel, dim3((L + CS_ROWS - 1) / CS_ROWS, (C + cpb - 1) / cpb, B), dim3(256), 0, (hipStream_t)stream,
                       Y, L, C, out, cpb);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

int gsdd_attention_v4_with_lse(const float* q, const float* k, const float* v, int B, int L, int H, float* out, float* lse,
                               void* workspace, int64_t workspace_bytes, int mode, void* stream, int* done);      // d3pm_attention.hip

// any sequence length (one lane per query, keys broadcast from LDS): the sampler's fallback for L % 16 != 0
int gsdd_attention_valu(const float* q, const float* k, const float* v, int B, int L, int H, float* out, float* lse, void* stream) {
    hipLaunchKernelGGL(attn_train_fwd_kernel, dim3((L + 255) / 256, H, B), dim3(256), 0, (hipStream_t)stream, q, k, v, B, L, H,
                       out, lse);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_d3pm_attention_train(const float* q, const float* k, const float* v, int B, int L, int H, float* out,
                                         float* lse, void* workspace, int64_t workspace_bytes, int mode, void* stream) {
    GSDD_CHECK_ARG(q && k && v && out && lse && B > 0 && L > 0 && H > 0, "bad args");
    int done = 0;
    const int rc = gsdd_attention_v4_with_lse(q, k, v, B, L, H, out, lse, workspace, workspace_bytes, mode, stream, &done);
    if (rc != GSDD_OK || done) return rc;
    hipLaunchKernelGGL(attn_train_fwd_kernel, dim3((L + 255) / 256, H, B), dim3(256), 0, (hipStream_t)stream, q, k, v, B, L, H,
                       out, lse);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

int gsdd_attention_bwd_mfma(const float* q, const float* k, const float* v, const float* o, const float* dO, const float* lse, int B,
                            int L, int H, float* dqkv, void* workspace, int64_t workspace_bytes, int variant, void* stream,
                            int* done);                                                      // d3pm_attention_bwd.hip

extern "C" int gsdd_d3pm_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* dO,
                                       const float* lse, int B, int L, int H, float* dqkv, float* scratch, void* workspace,
                                       int64_t workspace_bytes, int variant, void* stream) {
    GSDD_CHECK_ARG(q && k && v && o && dO && lse && dqkv && B > 0 && L > 0 && H > 0, "bad args");
    GSDD_CHECK_ARG(variant >= GSDD_ATTN_BWD_AUTO && variant <= GSDD_ATTN_BWD_DEV_LAST, "variant: one of GSDD_ATTN_BWD_*");
    const bool force_valu = variant == GSDD_ATTN_BWD_VALU;
    if (!force_valu) {
        int done = 0;
        const int rc = gsdd_attention_bwd_mfma(q, k, v, o, dO, lse, B, L, H, dqkv, workspace, workspace_bytes, variant, stream, &done);
        if (rc != GSDD_OK || done) return rc;
    }
    GSDD_CHECK_ARG(scratch != nullptr, "the VALU backward needs the float[H*M] scratch");
    const dim3 grid((L + 255) / 256, H, B);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, o, dO, lse, B, L, H, dqkv, scratch);
    GSDD_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, dO, lse, scratch, B, L, H, dqkv);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_d3pm_embed_bwd(const float* dx, const int64_t* tok, int B, int L, int D, int n_embed, float* demb,
                                   float* dpos, void* stream) {
    GSDD_CHECK_ARG(dx && tok && demb && dpos && B > 0 && L > 0 && D % 4 == 0, "bad args");
    const int64_t rows = (int64_t)B * L;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)((rows + EB_ROWS - 1) / EB_ROWS)), dim3(256), 0, (hipStream_t)stream, dx, tok,
                       rows, L, D, n_embed, demb, dpos);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_small_linear_bwd(const float* dy, const float* x, const float* w, int R, int Cin, int Cout, float* dx,
                                     float* dw, float* db, void* stream) {
    GSDD_CHECK_ARG(dy && x && w && dw && R > 0 && Cin > 0 && Cout > 0, "bad args");
    int n = Cout * Cin;
    if (R * Cin > n) n = R * Cin;
    hipLaunchKernelGGL(small_linear_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dy, x, w, R, Cin, Cout,
                       dx, dw, db);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_adaln_bwd(const float* dtab, const int64_t* t, int B, int D, const float* emb, const float* w, float* demb,
                              float* dw, float* db, void* stream) {
    GSDD_CHECK_ARG(dtab && t && emb && w && demb && dw && db && B > 0 && D > 0, "bad args");
    int n = 2 * D * D;
    if (B * D > n) n = B * D;
    const size_t lds = (size_t)2 * B * D * sizeof(float);
    if (lds <= 48 * 1024)
        hipLaunchKernelGGL(adaln_bwd_kernel<true>, dim3((n + 255) / 256), dim3(256), lds, (hipStream_t)stream, dtab, t, B, D, emb, w, demb,
                           dw, db);
    else
        hipLaunchKernelGGL(adaln_bwd_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dtab, t, B, D, emb, w, demb,
                           dw, db);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                         int step, void* stream) {
    GSDD_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "bad args");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, bc1, bc2);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}
