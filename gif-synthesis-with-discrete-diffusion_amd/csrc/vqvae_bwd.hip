// Backward kernels of the VQ-VAE training step (correctness-first): implicit-GEMM weight gradient with tap tables and the
// BN+ReLU prologue, BatchNorm(train)+ReLU backward, ReLU masks, axial-attention backward, small elementwise helpers.
// Data gradients of (transposed) convolutions reuse gsdd_gemm with transposed weights and mirrored tap tables.
// Reference semantics: autograd of videogpt_vq_vae.py:102-138, 228-332 and model_utils.py:211-289, 318-337, 586-600.
#include <stdlib.h>

#include "common.hpp"

namespace gsdd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------ conv weight gradient
//   dW[tap][n][c] += sum_m dY[orow(m)][n] * pro(X[src(m,tap)][c])
// grid: (row slabs, n-tiles * c-tiles, taps); 128 rows per slab, 64x64 output tile, 4 waves (32x32 quadrants) contracting over
// the slab's rows on v_mfma_f32_32x32x2_f32.  Per slab: 128 threads decode one row each (source row of this tap or -1, output
// row) into LDS, every thread then issues its 16 branch-free float4 loads of slab s+1 before the 64 MFMAs of slab s and stages
// them into the (single) LDS tile afterwards, so global latency hides under the MFMAs.
constexpr int CW_ROWS = 128;
struct CwSmem {
    float y[CW_ROWS][64];
    float x[CW_ROWS][64];
    int src[2][CW_ROWS];
    int dst[2][CW_ROWS];
};
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const gsdd_gemm_desc d, const float* dY, int dy_pitch, float* dW,
                                                            const int64_t M, int slabs, int ctiles) {
    extern __shared__ __attribute__((aligned(16))) char cw_raw[];
    CwSmem& sm = *reinterpret_cast<CwSmem*>(cw_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = (blockIdx.y / ctiles) * 64, c0 = (blockIdx.y % ctiles) * 64;
    const int tap = blockIdx.z;
    const int wn = wave >> 1, wk = wave & 1;
    int dt = 0, dh = 0, dw = 0;
    if (d.taps != nullptr) { dt = d.taps[3 * tap]; dh = d.taps[3 * tap + 1]; dw = d.taps[3 * tap + 2]; }
    const bool linear_rows = (d.oD == d.Do && d.oH == d.Ho && d.oW == d.Wo && d.osd == 1 && d.osh == 1 && d.osw == 1 &&
                              d.ood == 0 && d.ooh == 0 && d.oow == 0);
    const int64_t slab0 = (int64_t)blockIdx.x * slabs;
    int nsl = (int)((M - slab0 * CW_ROWS + CW_ROWS - 1) / CW_ROWS);
    nsl = nsl < slabs ? nsl : slabs;
    if (nsl <= 0) return;

    const int c4 = (tid & 15) * 4, rr = tid >> 4;
    const bool cy_ok = n0 + c4 < d.Cout, cx_ok = c0 + c4 < d.Cin;
    const int ycol = cy_ok ? n0 + c4 : 0, xcol = cx_ok ? c0 + c4 : 0;
    const bool has_pro = d.pro_scale != nullptr;
    float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_pro) { ps = *reinterpret_cast<const float4*>(d.pro_scale + xcol); pb = *reinterpret_cast<const float4*>(d.pro_shift + xcol); }

    auto decode = [&](int sl) {                       // one row per thread (threads 0..127)
        if (tid < CW_ROWS) {
            const int64_t m = (slab0 + sl) * CW_ROWS + tid;
            int src = -1, dst = -1;
            if (m < M) {
                uint32_t q = (uint32_t)m;
                const uint32_t wo = q % (uint32_t)d.Wo; q /= (uint32_t)d.Wo;
                const uint32_t ho = q % (uint32_t)d.Ho; q /= (uint32_t)d.Ho;
                const uint32_t to = q % (uint32_t)d.Do; q /= (uint32_t)d.Do;
                dst = (int)m;
                if (!linear_rows)
                    dst = (((int)q * d.oD + ((int)to * d.osd + d.ood)) * d.oH + ((int)ho * d.osh + d.ooh)) * d.oW + ((int)wo * d.osw + d.oow);
                const int ti = (int)to * d.sd + dt, hi = (int)ho * d.sh + dh, wi = (int)wo * d.sw + dw;
                if ((unsigned)ti < (unsigned)d.Di && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi)
                    src = (((int)q * d.Di + ti) * d.Hi + hi) * d.Wi + wi;
            }
            sm.src[sl & 1][tid] = src;
            sm.dst[sl & 1][tid] = dst;
        }
    };
    float4 vy[8], vx[8];
    unsigned okbits = 0;
    auto issue = [&](int sl) {
        okbits = 0;
        const float* py[8];
        const float* px[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int src = sm.src[sl & 1][rr + 16 * k], dst = sm.dst[sl & 1][rr + 16 * k];
            const bool oky = (dst >= 0) & cy_ok, okx = (src >= 0) & cx_ok;
            py[k] = dY + ((int64_t)(oky ? dst : 0) * dy_pitch + ycol);
            px[k] = d.in + ((int64_t)(okx ? src : 0) * d.in_pitch + xcol);
            okbits |= ((oky ? 1u : 0u) << k) | ((okx ? 1u : 0u) << (8 + k));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) vy[k] = *reinterpret_cast<const float4*>(py[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) vx[k] = *reinterpret_cast<const float4*>(px[k]);
    };
    auto stage = [&]() {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float4 y = vy[k], x = vx[k];
            if (has_pro) {
                x.x = fmaxf(fmaf(x.x, ps.x, pb.x), 0.f); x.y = fmaxf(fmaf(x.y, ps.y, pb.y), 0.f);
                x.z = fmaxf(fmaf(x.z, ps.z, pb.z), 0.f); x.w = fmaxf(fmaf(x.w, ps.w, pb.w), 0.f);
            }
            if (!((okbits >> k) & 1u)) y = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!((okbits >> (8 + k)) & 1u)) x = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&sm.y[rr + 16 * k][c4]) = y;
            *reinterpret_cast<float4*>(&sm.x[rr + 16 * k][c4]) = x;
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    decode(0);
    __syncthreads();
    issue(0);
    stage();
    for (int sl = 0; sl < nsl; ++sl) {
        const bool more = sl + 1 < nsl;
        if (more) decode(sl + 1);
        __syncthreads();                               // tile of slab sl staged, row table of slab sl+1 visible
        if (more) issue(sl + 1);
        {                                              // 64 k-steps, operands fetched one batch of 8 ahead of the MFMAs
            float yb[2][8], xb[2][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { yb[0][u] = sm.y[2 * u + lh][wn * 32 + li]; xb[0][u] = sm.x[2 * u + lh][wk * 32 + li]; }
#pragma unroll
            for (int b8 = 0; b8 < 8; ++b8) {
                if (b8 + 1 < 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        yb[(b8 + 1) & 1][u] = sm.y[2 * (8 * (b8 + 1) + u) + lh][wn * 32 + li];
                        xb[(b8 + 1) & 1][u] = sm.x[2 * (8 * (b8 + 1) + u) + lh][wk * 32 + li];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yb[b8 & 1][u], xb[b8 & 1][u], acc, 0, 0, 0);
            }
        }
        __syncthreads();                               // every wave is done with the tile
        if (more) stage();
    }
    float* out = dW + (int64_t)tap * d.Cout * d.Cin;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int c = c0 + wk * 32 + li;
        if (n < d.Cout && c < d.Cin) atomicAdd(out + (int64_t)n * d.Cin + c, acc[r]);
    }
}

// ------------------------------------------------------------------ conv weight gradient on the bf16 matrix pipe
// Same contraction with both operands split error-free into three bf16 pieces (as gemm_kernel<.., true>): per 64-row slab a
// 32x32 quadrant is 4 k-steps x 6 cross products of v_mfma_f32_32x32x16_bf16.  The MFMA wants 8 consecutive slab rows per
// lane, so a thread stages an 8-row x 4-column patch: splitting pairs of rows packs them directly, and each (column, piece) is
// one 16-byte LDS write into the transposed tile [piece][column][row] (pitch 72 bf16: conflict-free fragment reads; the
// thread -> patch map keeps every 16-lane group of the writes on distinct banks too).
constexpr int CX_ROWS = 64, CX_PITCH = 72;
typedef __bf16 cw_bf16x8 __attribute__((ext_vector_type(8)));
template <int T>                             // T x T output tile (64 or 128)
struct CwSmemX3 {
    uint16_t t[2][3][T][CX_PITCH];           // [operand: dY | x][piece][column][slab row]
    int src[2][CX_ROWS];
    int dst[2][CX_ROWS];
};
__device__ __forceinline__ uint32_t cw_cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// T = 64: 256 threads, a wave owns one 32x32 quadrant.  T = 128 (both channel counts >= 128): 512 threads, a wave owns two
// quadrants; per MAC half the L2 traffic and half the split work of the 64x64 tile.
template <int T>
__global__ __launch_bounds__(4 * T, 128 / T) void conv_wgrad_x3_kernel(const gsdd_gemm_desc d, const float* dY, int dy_pitch,
                                                                         float* dW, const int64_t M, int slabs, int ctiles) {
    constexpr int NTH = 4 * T, HALF = NTH / 2, QC = T / 64;       // threads, threads per operand, c-quadrants per wave
    extern __shared__ __attribute__((aligned(16))) char cwx_raw[];
    CwSmemX3<T>& sm = *reinterpret_cast<CwSmemX3<T>*>(cwx_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = (blockIdx.y / ctiles) * T, c0 = (blockIdx.y % ctiles) * T;
    const int tap = blockIdx.z;
    const int wn = wave >> 1, wk = wave & 1;                      // n quadrant, first c quadrant = QC * wk
    int dt = 0, dh = 0, dw = 0;
    if (d.taps != nullptr) { dt = d.taps[3 * tap]; dh = d.taps[3 * tap + 1]; dw = d.taps[3 * tap + 2]; }
    const bool linear_rows = (d.oD == d.Do && d.oH == d.Ho && d.oW == d.Wo && d.osd == 1 && d.osh == 1 && d.osw == 1 &&
                              d.ood == 0 && d.ooh == 0 && d.oow == 0);
    const int64_t slab0 = (int64_t)blockIdx.x * slabs;
    int nsl = (int)((M - slab0 * CX_ROWS + CX_ROWS - 1) / CX_ROWS);
    nsl = nsl < slabs ? nsl : slabs;
    if (nsl <= 0) return;

    // staging patch of this thread: operand op, columns 4*cgrp..+3, slab rows 8*rgrp..+7 (bits: cgrp_lo 2, rgrp_lo 2, cgrp_hi, rgrp_hi)
    const int op = tid / HALF, th = tid % HALF;
    const int cgrp = 4 * ((th >> 4) % (T / 16)) + (th & 3), rgrp = 4 * (th / T) + ((th >> 2) & 3);
    const int c4 = 4 * cgrp;
    const bool col_ok = op == 0 ? (n0 + c4 < d.Cout) : (c0 + c4 < d.Cin);
    const int gcol = col_ok ? (op == 0 ? n0 + c4 : c0 + c4) : 0;
    const float* gbase = op == 0 ? dY : d.in;
    const int gpitch = op == 0 ? dy_pitch : d.in_pitch;
    const bool has_pro = op == 1 && d.pro_scale != nullptr;
    float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (has_pro) { ps = *reinterpret_cast<const float4*>(d.pro_scale + gcol); pb = *reinterpret_cast<const float4*>(d.pro_shift + gcol); }

    auto decode = [&](int sl) {                       // one row per thread (threads 0..63)
        if (tid < CX_ROWS) {
            const int64_t m = (slab0 + sl) * CX_ROWS + tid;
            int src = -1, dst = -1;
            if (m < M) {
                uint32_t q = (uint32_t)m;
                const uint32_t wo = q % (uint32_t)d.Wo; q /= (uint32_t)d.Wo;
                const uint32_t ho = q % (uint32_t)d.Ho; q /= (uint32_t)d.Ho;
                const uint32_t to = q % (uint32_t)d.Do; q /= (uint32_t)d.Do;
                dst = (int)m;
                if (!linear_rows)
                    dst = (((int)q * d.oD + ((int)to * d.osd + d.ood)) * d.oH + ((int)ho * d.osh + d.ooh)) * d.oW + ((int)wo * d.osw + d.oow);
                const int ti = (int)to * d.sd + dt, hi = (int)ho * d.sh + dh, wi = (int)wo * d.sw + dw;
                if ((unsigned)ti < (unsigned)d.Di && (unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi)
                    src = (((int)q * d.Di + ti) * d.Hi + hi) * d.Wi + wi;
            }
            sm.src[sl & 1][tid] = src;
            sm.dst[sl & 1][tid] = dst;
        }
    };
    float4 v0, v1, v2, v3, v4, v5, v6, v7;            // named scalars: an array captured by the lambdas would live in scratch
    unsigned okbits = 0;
    auto issue = [&](int sl) {
        const int* tab = op == 0 ? sm.dst[sl & 1] : sm.src[sl & 1];
        okbits = 0;
        const float* pp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = tab[8 * rgrp + i];
            const bool ok = (row >= 0) & col_ok;
            pp[i] = gbase + ((int64_t)(ok ? row : 0) * gpitch + gcol);
            okbits |= (ok ? 1u : 0u) << i;
        }
        v0 = *reinterpret_cast<const float4*>(pp[0]); v1 = *reinterpret_cast<const float4*>(pp[1]);
        v2 = *reinterpret_cast<const float4*>(pp[2]); v3 = *reinterpret_cast<const float4*>(pp[3]);
        v4 = *reinterpret_cast<const float4*>(pp[4]); v5 = *reinterpret_cast<const float4*>(pp[5]);
        v6 = *reinterpret_cast<const float4*>(pp[6]); v7 = *reinterpret_cast<const float4*>(pp[7]);
    };
    auto stage = [&]() {
        float x[8][4];
        const float4 vv[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float4 v = vv[i];
            if (has_pro) {
                v.x = fmaxf(fmaf(v.x, ps.x, pb.x), 0.f); v.y = fmaxf(fmaf(v.y, ps.y, pb.y), 0.f);
                v.z = fmaxf(fmaf(v.z, ps.z, pb.z), 0.f); v.w = fmaxf(fmaf(v.w, ps.w, pb.w), 0.f);
            }
            if (!((okbits >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            x[i][0] = v.x; x[i][1] = v.y; x[i][2] = v.z; x[i][3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                   // column 4*cgrp + j: 8 consecutive slab rows -> one uint4 per piece
            float r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = x[i][j];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                uint32_t w[4];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    w[i2] = cw_cvt_pk_bf16(r[2 * i2], r[2 * i2 + 1]);
                    if (pc < 2) {
                        r[2 * i2] -= __uint_as_float(w[i2] << 16);
                        r[2 * i2 + 1] -= __uint_as_float(w[i2] & 0xffff0000u);
                    }
                }
                *reinterpret_cast<uint4*>(&sm.t[op][pc][c4 + j][8 * rgrp]) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    };

    f32x16 acc[QC];
#pragma unroll
    for (int qc = 0; qc < QC; ++qc)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[qc][i] = 0.f;
    decode(0);
    __syncthreads();
    issue(0);
    stage();
    for (int sl = 0; sl < nsl; ++sl) {
        const bool more = sl + 1 < nsl;
        if (more) decode(sl + 1);
        __syncthreads();                               // tile of slab sl staged, row table of slab sl+1 visible
        if (more) issue(sl + 1);
#pragma unroll
        for (int ks = 0; ks < CX_ROWS / 16; ++ks) {
            cw_bf16x8 yf[3], xf[QC][3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                union { uint4 u; cw_bf16x8 v; } a;
                a.u = *reinterpret_cast<const uint4*>(&sm.t[0][pc][wn * 32 + li][16 * ks + 8 * lh]);
                yf[pc] = a.v;
#pragma unroll
                for (int qc = 0; qc < QC; ++qc) {
                    union { uint4 u; cw_bf16x8 v; } b;
                    b.u = *reinterpret_cast<const uint4*>(&sm.t[1][pc][(QC * wk + qc) * 32 + li][16 * ks + 8 * lh]);
                    xf[qc][pc] = b.v;
                }
            }
#pragma unroll
            for (int qc = 0; qc < QC; ++qc) {
                acc[qc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[2], xf[qc][0], acc[qc], 0, 0, 0);
                acc[qc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[1], xf[qc][1], acc[qc], 0, 0, 0);
                acc[qc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[0], xf[qc][2], acc[qc], 0, 0, 0);
                acc[qc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[1], xf[qc][0], acc[qc], 0, 0, 0);
                acc[qc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[0], xf[qc][1], acc[qc], 0, 0, 0);
                acc[qc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[0], xf[qc][0], acc[qc], 0, 0, 0);
            }
        }
        __syncthreads();                               // every wave is done with the tile
        if (more) stage();
    }
    float* out = dW + (int64_t)tap * d.Cout * d.Cin;
#pragma unroll
    for (int qc = 0; qc < QC; ++qc)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int c = c0 + (QC * wk + qc) * 32 + li;
            if (n < d.Cout && c < d.Cin) atomicAdd(out + (int64_t)n * d.Cin + c, acc[qc][r]);
        }
}

// ------------------------------------------------------------------ BatchNorm(train) + ReLU backward on rows x[M][C]
//   a = relu(y), y = gamma*xhat + beta, xhat = (x - mean)*rstd ; given da:
//   dy = da*[y>0] ; dbeta = sum dy ; dgamma = sum dy*xhat ; dx = gamma*rstd*(dy - mean(dy) - xhat*mean(dy*xhat))
constexpr int BB_ROWS = 64;        // >= 1024 stage-1 blocks at the training shapes
__global__ __launch_bounds__(256) void bn_relu_bwd_partial_kernel(const float* da, const float* x, const float* mean_rstd,
                                                                  const float* gamma, const float* beta, int64_t M, int C,
                                                                  double* part) {
    const int64_t r0 = (int64_t)blockIdx.x * BB_ROWS;
    const int64_t r1 = r0 + BB_ROWS < M ? r0 + BB_ROWS : M;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float mu = mean_rstd[2 * c], rs = mean_rstd[2 * c + 1], g = gamma[c], bt = beta[c];
        double s1 = 0.0, s2 = 0.0;
        for (int64_t r = r0; r < r1; ++r) {
            const float xh = (x[r * C + c] - mu) * rs;
            const float y = g * xh + bt;
            const float dy = y > 0.f ? da[r * C + c] : 0.f;
            s1 += (double)dy;
            s2 += (double)(dy * xh);
        }
        part[((int64_t)blockIdx.x * C + c) * 2 + 0] = s1;
        part[((int64_t)blockIdx.x * C + c) * 2 + 1] = s2;
    }
}
// one 256-thread block per channel
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const double* part, int nblk, int C, float* dgamma, float* dbeta,
                                                                 float* sums) {
    __shared__ double red[2][4];
    const int c = blockIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) { s1 += part[((int64_t)b * C + c) * 2]; s2 += part[((int64_t)b * C + c) * 2 + 1]; }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    dbeta[c] += (float)s1;
    dgamma[c] += (float)s2;
    sums[2 * c] = (float)s1;
    sums[2 * c + 1] = (float)s2;
}
__global__ void bn_relu_bwd_apply_kernel(const float* da, const float* x, const float* mean_rstd, const float* gamma,
                                         const float* beta, const float* sums, int64_t M, int C, const float* dx_in, float* dx) {
    const int q4 = C >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * q4) return;
    const int64_t r = i / q4;
    const int c = (int)(i % q4) * 4;
    const float4 xv = *reinterpret_cast<const float4*>(x + r * C + c);
    const float4 dv = *reinterpret_cast<const float4*>(da + r * C + c);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dx_in != nullptr) o = *reinterpret_cast<const float4*>(dx_in + r * C + c);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
    float res[4];
    const float invM = 1.f / (float)M;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float mu = mean_rstd[2 * (c + e)], rs = mean_rstd[2 * (c + e) + 1], g = gamma[c + e], bt = beta[c + e];
        const float xh = (xs[e] - mu) * rs;
        const float y = g * xh + bt;
        const float dy = y > 0.f ? ds[e] : 0.f;
        res[e] = g * rs * (dy - sums[2 * (c + e)] * invM - xh * sums[2 * (c + e) + 1] * invM);
    }
    o.x += res[0]; o.y += res[1]; o.z += res[2]; o.w += res[3];
    *reinterpret_cast<float4*>(dx + r * C + c) = o;
}

// dpre = dout * [out > 0]   (ReLU backward through a saved post-activation)
__global__ void relu_mask_kernel(const float* dout, const float* out, float* dpre, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 g = *reinterpret_cast<const float4*>(dout + i);
    const float4 o = *reinterpret_cast<const float4*>(out + i);
    *reinterpret_cast<float4*>(dpre + i) = make_float4(o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f,
                                                       o.w > 0.f ? g.w : 0.f);
}

// out = (a ? a : 0) + alpha * (b - c)
__global__ void lincomb_kernel(const float* a, const float* b, const float* c, float alpha, float* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = (a != nullptr ? a[i] : 0.f) + alpha * (b[i] - c[i]);
}

// ------------------------------------------------------------------ axial attention backward: one wave per (line, head, axis)
__global__ __launch_bounds__(64) void axial_attention_bwd_kernel(const float* qkv, const float* datt, int N, int T, int H, int W,
                                                                 int C, int n_head, int axis, float* dqkv) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int head = blockIdx.y;
    const int d = C / n_head;
    const int S = axis == 0 ? W : (axis == 1 ? H : T);
    const int64_t line = blockIdx.x;
    int64_t stride, basepos;
    if (axis == 0) { stride = 1; basepos = line * W; }
    else if (axis == 1) { stride = W; const int64_t w = line % W; const int64_t nt = line / W; basepos = nt * H * W + w; }
    else { stride = (int64_t)H * W; const int64_t hw = line % ((int64_t)H * W); const int64_t n = line / ((int64_t)H * W);
           basepos = n * T * H * W + hw; }
    const int P = d + 1;
    float* sq = smf; float* sk = sq + S * P; float* sv = sk + S * P; float* sg = sv + S * P;      // [S][d+1] each
    float* sp = sg + S * P;                                                                        // [S][S] probabilities
    float* sd = sp + S * S;                                                                        // [S][S] dS
    const int lane = threadIdx.x;
    const int64_t rowpitch = 9 * (int64_t)C;
    const int64_t colbase = (int64_t)axis * 3 * C + head * d;
    for (int i = lane; i < S * d; i += 64) {
        const int s = i / d, e = i % d;
        const float* row = qkv + (basepos + s * stride) * rowpitch + colbase + e;
        sq[s * P + e] = row[0]; sk[s * P + e] = row[C]; sv[s * P + e] = row[2 * C];
        sg[s * P + e] = datt[(basepos + s * stride) * (3 * (int64_t)C) + (int64_t)axis * C + head * d + e];
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)d);
    for (int i = lane; i < S * S; i += 64) {
        const int a = i / S, b = i % S;
        float s = 0.f, dp = 0.f;
        for (int e = 0; e < d; ++e) { s = fmaf(sq[a * P + e], sk[b * P + e], s); dp = fmaf(sg[a * P + e], sv[b * P + e], dp); }
        sp[i] = s * scale;
        sd[i] = dp;
    }
    __syncthreads();
    for (int a = lane; a < S; a += 64) {
        float mx = -INFINITY;
        for (int j = 0; j < S; ++j) mx = fmaxf(mx, sp[a * S + j]);
        float l = 0.f;
        for (int j = 0; j < S; ++j) { const float p = expf(sp[a * S + j] - mx); sp[a * S + j] = p; l += p; }
        const float inv = 1.f / l;
        float rs = 0.f;
        for (int j = 0; j < S; ++j) { sp[a * S + j] *= inv; rs += sp[a * S + j] * sd[a * S + j]; }
        for (int j = 0; j < S; ++j) sd[a * S + j] = sp[a * S + j] * (sd[a * S + j] - rs) * scale;
    }
    __syncthreads();
    for (int i = lane; i < S * d; i += 64) {
        const int a = i / d, e = i % d;
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int j = 0; j < S; ++j) {
            dq = fmaf(sd[a * S + j], sk[j * P + e], dq);
            dk = fmaf(sd[j * S + a], sq[j * P + e], dk);
            dv = fmaf(sp[j * S + a], sg[j * P + e], dv);
        }
        float* row = dqkv + (basepos + a * stride) * rowpitch + colbase + e;
        row[0] = dq; row[C] = dk; row[2 * C] = dv;
    }
}

}  // namespace gsdd

using namespace gsdd;

extern "C" int gsdd_conv_wgrad(const gsdd_gemm_desc* d, const float* dY, int dy_pitch, float* dW, void* stream) {
    GSDD_CHECK_ARG(d != nullptr && dY != nullptr && dW != nullptr && d->in != nullptr, "null pointer");
    GSDD_CHECK_ARG(d->N > 0 && d->Do > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0 && d->Cin > 0, "bad sizes");
    GSDD_CHECK_ARG(d->Cin % 4 == 0 && d->in_pitch % 4 == 0 && d->Cout % 4 == 0 && dy_pitch % 4 == 0, "channel counts / pitches must be multiples of 4");
    GSDD_CHECK_ARG(d->ntaps >= 1 && (d->ntaps == 1 || d->taps != nullptr) && d->gather == nullptr && d->ln_stats == nullptr,
                   "unsupported descriptor");
    const int64_t M = (int64_t)d->N * d->Do * d->Ho * d->Wo;
    GSDD_CHECK_ARG(M < (1ll << 31), "more than 2^31 rows");
    GSDD_CHECK_ARG((int64_t)d->N * d->Di * d->Hi * d->Wi < (1ll << 31), "more than 2^31 input rows");
    GSDD_CHECK_ARG(d->oD > 0 && d->oH > 0 && d->oW > 0 && (int64_t)d->N * d->oD * d->oH * d->oW < (1ll << 31), "bad output dims");
    GSDD_CHECK_ARG((d->flags & ~GSDD_GEMM_EXACT_F32) == 0, "unknown flags");
    const bool force_f32 = (d->flags & GSDD_GEMM_EXACT_F32) != 0;
    const bool big = !force_f32 && d->Cout >= 128 && d->Cin >= 128;           // 128 x 128 output tiles on 512 threads
    const int T = big ? 128 : 64;
    const int ntiles = (d->Cout + T - 1) / T, ctiles = (d->Cin + T - 1) / T;
    const int rows_per_slab = force_f32 ? CW_ROWS : CX_ROWS;
    // rows per block: enough blocks to fill the chip (>= ~2048, half that for the big tile), as few atomics per dW element as
    // that allows
    const int64_t nslabs = (M + rows_per_slab - 1) / rows_per_slab;
    const int64_t per_x = (int64_t)ntiles * ctiles * d->ntaps;
    int64_t gx = ((big ? 1024 : 2048) + per_x - 1) / per_x;
    gx = gx < 1 ? 1 : (gx > nslabs ? nslabs : gx);
    int slabs = (int)((nslabs + gx - 1) / gx);
    slabs = slabs < 8 ? (nslabs < 8 ? (int)nslabs : 8) : (slabs > 128 ? 128 : slabs);
    const dim3 grid((unsigned)((nslabs + slabs - 1) / slabs), ntiles * ctiles, d->ntaps);
    if (force_f32) {
        hipLaunchKernelGGL(conv_wgrad_kernel, grid, dim3(256), sizeof(CwSmem), (hipStream_t)stream, *d, dY, dy_pitch, dW, M, slabs, ctiles);
    } else if (big) {
        GSDD_ONCE_PER_DEVICE(attr_done,
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)conv_wgrad_x3_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)sizeof(CwSmemX3<128>)));
        );
        hipLaunchKernelGGL(conv_wgrad_x3_kernel<128>, grid, dim3(512), sizeof(CwSmemX3<128>), (hipStream_t)stream, *d, dY, dy_pitch, dW, M,
                           slabs, ctiles);
    } else {
        hipLaunchKernelGGL(conv_wgrad_x3_kernel<64>, grid, dim3(256), sizeof(CwSmemX3<64>), (hipStream_t)stream, *d, dY, dy_pitch, dW, M,
                           slabs, ctiles);
    }
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int64_t gsdd_bn_relu_bwd_workspace_bytes(int64_t M, int C) {
    return (((M + BB_ROWS - 1) / BB_ROWS) * (int64_t)C * 2) * (int64_t)sizeof(double) + (int64_t)C * 2 * (int64_t)sizeof(float);
}

extern "C" int gsdd_bn_relu_bwd(const float* da, const float* x, int64_t M, int C, const float* mean_rstd, const float* gamma,
                                const float* beta, const float* dx_in, float* dx, float* dgamma, float* dbeta, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    GSDD_CHECK_ARG(da && x && mean_rstd && gamma && beta && dx && dgamma && dbeta && workspace, "null pointer");
    GSDD_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "bad sizes");
    GSDD_CHECK_ARG(workspace_bytes >= gsdd_bn_relu_bwd_workspace_bytes(M, C), "workspace too small");
    const int nblk = (int)((M + BB_ROWS - 1) / BB_ROWS);
    double* part = (double*)workspace;
    float* sums = (float*)(part + (int64_t)nblk * C * 2);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_relu_bwd_partial_kernel, dim3(nblk), dim3(256), 0, st, da, x, mean_rstd, gamma, beta, M, C, part);
    GSDD_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(C), dim3(256), 0, st, part, nblk, C, dgamma, dbeta, sums);
    GSDD_CHECK_LAUNCH();
    const int64_t n = M * (C / 4);
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, da, x, mean_rstd, gamma, beta,
                       sums, M, C, dx_in, dx);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_relu_mask(const float* dout, const float* out, float* dpre, int64_t n, void* stream) {
    GSDD_CHECK_ARG(dout && out && dpre && n > 0 && n % 4 == 0, "bad args");
    hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, out, dpre, n);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_lincomb(const float* a, const float* b, const float* c, float alpha, float* out, int64_t n, void* stream) {
    GSDD_CHECK_ARG(b && c && out && n > 0, "bad args");
    hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, alpha, out, n);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_axial_attention_bwd(const float* qkv, const float* datt, int N, int T, int H, int W, int C, int n_head,
                                        float* dqkv, int variant, void* stream) {
    GSDD_CHECK_ARG(qkv && datt && dqkv, "null pointer");
    GSDD_CHECK_ARG(variant == GSDD_AXIAL_AUTO || variant == GSDD_AXIAL_VALU, "variant: GSDD_AXIAL_AUTO or GSDD_AXIAL_VALU");
    GSDD_CHECK_ARG(N > 0 && T > 0 && H > 0 && W > 0 && C > 0 && n_head > 0 && C % n_head == 0, "bad sizes");
    GSDD_CHECK_ARG(T <= 64 && H <= 64 && W <= 64, "axis length > 64 unsupported");
    const int d = C / n_head;
    const int axes_len[3] = {W, H, T};
    const int64_t pos = (int64_t)N * T * H * W;
    const bool force_valu = variant == GSDD_AXIAL_VALU;
    for (int axis = 0; axis < 3; ++axis) {
        if (!force_valu && axial_attention_bwd_mfma_launch(qkv, datt, N, T, H, W, C, n_head, axis, dqkv, (hipStream_t)stream)) {
            GSDD_CHECK_LAUNCH();
            continue;
        }
        const int S = axes_len[axis];
        const size_t lds = (size_t)(4 * S * (d + 1) + 2 * S * S) * sizeof(float);
        GSDD_CHECK_ARG(lds <= 64 * 1024, "line does not fit LDS");
        hipLaunchKernelGGL(axial_attention_bwd_kernel, dim3((unsigned)(pos / S), n_head, 1), dim3(64), lds, (hipStream_t)stream, qkv,
                           datt, N, T, H, W, C, n_head, axis, dqkv);
        GSDD_CHECK_LAUNCH();
    }
    return GSDD_OK;
}
