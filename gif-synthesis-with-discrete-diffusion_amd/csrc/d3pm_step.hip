// Fused D3PM reverse step: log_softmax x2 -> classifier-free mix -> q_posterior -> Gumbel arg-max.
//
// One wave64 owns one token position; the K(+1)-wide categorical lives in registers
// (lane owns quads k = 4*lane + 256*j), every reduction over classes is a wave reduction.
// HBM-bound by design: algorithmic traffic = the two logit rows (2*K*4 B) per position.
//
// Mirrors the reference op-for-op in fp32 (diffusion_transformer.py:220-283, :354-359); the
// reference's fp64 log_softmax (:231) is matched by accumulating the exp-sum and taking its log in fp64.
#include "common.hpp"

namespace gsdd {

constexpr float LOG_ZERO = -69.07755278982137f;  // log(1e-30)

__device__ __forceinline__ float lae(float a, float b) {  // reference log_add_exp (:32-34)
    const float m = fmaxf(a, b);
    return m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float clamp70(float v) { return fminf(fmaxf(v, -70.f), 0.f); }

struct StepSched {
    float la, lb, lc, l1mc;          // per-step at t
    float lca, lcb, lcc, l1mcc;      // cumulative at t
    float pca, pcb, pcc, p1mcc;      // cumulative at t-1 (wrapped)
};

__device__ __forceinline__ StepSched load_sched(const float* const* s, int64_t t, int T) {
    StepSched r;
    r.la = s[0][t]; r.lb = s[1][t]; r.lc = s[2][t]; r.l1mc = s[3][t];
    r.lca = s[4][t]; r.lcb = s[5][t]; r.lcc = s[6][t]; r.l1mcc = s[7][t];
    const int64_t tp = (t - 1 + (T + 1)) % (T + 1);
    r.pca = s[4][tp]; r.pcb = s[5][tp]; r.pcc = s[6][tp]; r.p1mcc = s[7][tp];
    return r;
}

struct SchedPtrs { const float* p[8]; };

// log_softmax over the wave's row (fp64 sum/log), clamp to [-70,0]   (predict_start, :231-236)
template <int J>
__device__ __forceinline__ void log_softmax_clamp(float (&x)[J][4]) {
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, x[j][e]);
    mx = wave_max(mx);
    double se = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) se += (double)expf(x[j][e] - mx);
    se = wave_sum(se);
    const double lse = (double)mx + log(se);
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) x[j][e] = clamp70((float)((double)x[j][e] - lse));
}

template <int J>
__device__ __forceinline__ float wave_logsumexp(const float (&x)[J][4], float extra, bool has_extra) {
    float mx = has_extra ? extra : -INFINITY;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, x[j][e]);
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) se += expf(x[j][e] - mx);
    se = wave_sum(se);
    if (has_extra) se += expf(extra - mx);
    return mx + logf(se);
}

__device__ __forceinline__ float gumbel(float u) {  // log_sample_categorical (:355-356)
    return -logf(-logf(u + 1e-30f) + 1e-30f);
}

// arg-max of (val, idx) over the wave, first index wins ties (torch.argmax)
__device__ __forceinline__ int wave_argmax(float v, int idx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(idx, o);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    return idx;
}

template <int J>
__global__ __launch_bounds__(256) void d3pm_step_kernel(gsdd_step_desc d, SchedPtrs sp) {
    const int lane = threadIdx.x & 63;
    const int64_t pos = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pos >= (int64_t)d.B * d.L) return;
    const int b = (int)(pos / d.L), l = (int)(pos % d.L);
    const int K = d.K;
    const float NEG = -INFINITY;

    float x0[J][4];
    {   // ---- predict_start on the conditional logits
        const float* row = d.logits_c + pos * (int64_t)K;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int k = 4 * lane + 256 * j;
            if (k < K) {
                const float4 v = *reinterpret_cast<const float4*>(row + k);
                x0[j][0] = v.x; x0[j][1] = v.y; x0[j][2] = v.z; x0[j][3] = v.w;
            } else {
                x0[j][0] = x0[j][1] = x0[j][2] = x0[j][3] = NEG;
            }
        }
        log_softmax_clamp<J>(x0);
    }
    if (d.logits_u != nullptr) {  // ---- cf_predict_start (:240-249)
        float xu[J][4];
        const float* row = d.logits_u + pos * (int64_t)K;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int k = 4 * lane + 256 * j;
            if (k < K) {
                const float4 v = *reinterpret_cast<const float4*>(row + k);
                xu[j][0] = v.x; xu[j][1] = v.y; xu[j][2] = v.z; xu[j][3] = v.w;
            } else {
                xu[j][0] = xu[j][1] = xu[j][2] = xu[j][3] = NEG;
            }
        }
        log_softmax_clamp<J>(xu);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const bool valid = (4 * lane + 256 * j) < K;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float diff = x0[j][e] - xu[j][e];
                const float sc = d.guidance * diff;
                x0[j][e] = valid ? (xu[j][e] + sc) : NEG;
            }
        }
        const float lse = wave_logsumexp<J>(x0, 0.f, false);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const bool valid = (4 * lane + 256 * j) < K;
#pragma unroll
            for (int e = 0; e < 4; ++e) x0[j][e] = valid ? clamp70(x0[j][e] - lse) : NEG;
        }
    }
    if (d.x0_dbg != nullptr) {
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * lane + 256 * j + e;
                if (k < K) d.x0_dbg[((int64_t)b * (K + 1) + k) * d.L + l] = x0[j][e];
            }
        if (lane == 0) d.x0_dbg[((int64_t)b * (K + 1) + K) * d.L + l] = -70.f;
    }

    // ---- q_posterior (:251-283)
    const int64_t t = d.t_dev[b];
    const StepSched s = load_sched(sp.p, t, d.T);
    const int64_t xt = d.tok_in[pos];
    const bool masked = (xt == K);
    const float qt_hit = lae(0.f + s.lca, s.lcb), qt_miss = lae(LOG_ZERO + s.lca, s.lcb);
    const float q1_hit = lae(0.f + s.la, s.lb), q1_miss = lae(LOG_ZERO + s.la, s.lb);
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = 4 * lane + 256 * j + e;
            const float log_qt = masked ? s.lcc : (k == xt ? qt_hit : qt_miss);
            x0[j][e] = (k < K) ? (x0[j][e] - log_qt) : NEG;
        }
    const float S = wave_logsumexp<J>(x0, LOG_ZERO, true);
    float best = NEG;
    int best_k = 0;
    const uint32_t kp4 = (uint32_t)((K + 1 + 3) / 4);
    const uint32_t stream_id = (uint32_t)d.stream_dev[0];
    const uint64_t grow = (uint64_t)(d.row0 + pos);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k0 = 4 * lane + 256 * j;
        if (k0 < K) {
            const float4 u4 = philox_uniform4(d.seed, stream_id, grow, kp4, (uint32_t)(k0 >> 2));
            const float u[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + e;
                const float qn = x0[j][e] - S;
                const float log_q1 = masked ? s.lc : (k == xt ? q1_hit : q1_miss);
                const float o = clamp70(lae(qn + s.pca, s.pcb) + log_q1 + S);
                if (d.post_dbg != nullptr) d.post_dbg[((int64_t)b * (K + 1) + k) * d.L + l] = o;
                const float v = gumbel(u[e]) + o;
                if (v > best) { best = v; best_k = k; }
            }
        }
    }
    if (lane == ((K >> 2) & 63)) {  // the [MASK] class k = K (K % 4 == 0 -> word 0 of quad K/4)
        const float4 u4 = philox_uniform4(d.seed, stream_id, grow, kp4, (uint32_t)(K >> 2));
        const float qn = LOG_ZERO - S;
        const float log_q1 = masked ? 0.f : LOG_ZERO;
        const float o = clamp70(lae(qn + s.p1mcc, s.pcc) + log_q1 + S);
        if (d.post_dbg != nullptr) d.post_dbg[((int64_t)b * (K + 1) + K) * d.L + l] = o;
        const float v = gumbel(u4.x) + o;
        if (v > best) { best = v; best_k = K; }
    }
    const int win = wave_argmax(best, best_k);
    if (lane == 0) d.tok_out[pos] = win;
}

// q_sample (:361-366): x_t ~ Gumbel-argmax(q_pred(onehot(x0), t))
__global__ __launch_bounds__(256) void d3pm_q_sample_kernel(const int64_t* x0, int64_t* xt, int B, int L, int K, int T,
                                                            SchedPtrs sp, const int64_t* t_dev, uint64_t seed,
                                                            const int64_t* stream_dev, int64_t row0) {
    const int lane = threadIdx.x & 63;
    const int64_t pos = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pos >= (int64_t)B * L) return;
    const int b = (int)(pos / L);
    const int64_t t = (t_dev[b] + (T + 1)) % (T + 1);
    const float lca = sp.p[4][t], lcb = sp.p[5][t], lcc = sp.p[6][t], l1mcc = sp.p[7][t];
    const int64_t tok = x0[pos];
    const float hit = lae(0.f + lca, lcb), miss = lae(LOG_ZERO + lca, lcb);
    const float mval = lae((tok == K ? 0.f : LOG_ZERO) + l1mcc, lcc);
    const uint32_t kp4 = (uint32_t)((K + 1 + 3) / 4);
    const uint32_t stream_id = (uint32_t)stream_dev[0];
    const uint64_t grow = (uint64_t)(row0 + pos);
    float best = -INFINITY;
    int best_k = 0;
    for (int k0 = 4 * lane; k0 <= K; k0 += 256) {
        const float4 u4 = philox_uniform4(seed, stream_id, grow, kp4, (uint32_t)(k0 >> 2));
        const float u[4] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + e;
            if (k <= K) {
                const float lp = (k == K) ? mval : (k == tok ? hit : miss);
                const float v = gumbel(u[e]) + lp;
                if (v > best) { best = v; best_k = k; }
            }
        }
    }
    const int win = wave_argmax(best, best_k);
    if (lane == 0) xt[pos] = win;
}

__global__ void advance_kernel(int64_t* t_dev, int B, int64_t dt, int64_t* stream_dev, int64_t ds) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (t_dev != nullptr && i < B) t_dev[i] += dt;
    if (stream_dev != nullptr && i == 0) stream_dev[0] += ds;
}

__global__ void philox_uniform_kernel(uint64_t seed, uint32_t stream_id, int64_t row0, int64_t n_rows, int n_cols,
                                      float* out) {
    const uint32_t kp4 = (uint32_t)((n_cols + 3) / 4);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * (int64_t)kp4) return;
    const int64_t r = i / kp4;
    const uint32_t c4 = (uint32_t)(i % kp4);
    const float4 u4 = philox_uniform4(seed, stream_id, (uint64_t)(row0 + r), kp4, c4);
    const float u[4] = {u4.x, u4.y, u4.z, u4.w};
    for (int e = 0; e < 4; ++e) {
        const int c = (int)c4 * 4 + e;
        if (c < n_cols) out[r * n_cols + c] = u[e];
    }
}

}  // namespace gsdd

using namespace gsdd;

extern "C" int gsdd_d3pm_step(const gsdd_step_desc* d, void* stream) {
    GSDD_CHECK_ARG(d != nullptr, "null descriptor");
    GSDD_CHECK_ARG(d->logits_c && d->tok_in && d->tok_out && d->t_dev && d->stream_dev, "null pointer");
    GSDD_CHECK_ARG(d->B > 0 && d->L > 0 && d->T > 0, "bad sizes");
    GSDD_CHECK_ARG(d->K >= 4 && d->K % 4 == 0 && d->K <= 8192, "K must be a multiple of 4 in [4, 8192]");
    for (int i = 0; i < 8; ++i) GSDD_CHECK_ARG(d->sched[i] != nullptr, "null schedule buffer");
    SchedPtrs sp;
    for (int i = 0; i < 8; ++i) sp.p[i] = d->sched[i];
    const int64_t npos = (int64_t)d->B * d->L;
    const dim3 grid((unsigned)((npos + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int J = (d->K + 255) / 256;
    if (J <= 1) hipLaunchKernelGGL(d3pm_step_kernel<1>, grid, block, 0, st, *d, sp);
    else if (J <= 2) hipLaunchKernelGGL(d3pm_step_kernel<2>, grid, block, 0, st, *d, sp);
    else if (J <= 4) hipLaunchKernelGGL(d3pm_step_kernel<4>, grid, block, 0, st, *d, sp);
    else if (J <= 8) hipLaunchKernelGGL(d3pm_step_kernel<8>, grid, block, 0, st, *d, sp);
    else if (J <= 16) hipLaunchKernelGGL(d3pm_step_kernel<16>, grid, block, 0, st, *d, sp);
    else hipLaunchKernelGGL(d3pm_step_kernel<32>, grid, block, 0, st, *d, sp);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_d3pm_q_sample(const int64_t* x0, int64_t* xt, int B, int L, int K, int T,
                                  const float* const* sched, const int64_t* t_dev, uint64_t seed,
                                  const int64_t* stream_dev, int64_t row0, void* stream) {
    GSDD_CHECK_ARG(x0 && xt && sched && t_dev && stream_dev, "null pointer");
    GSDD_CHECK_ARG(B > 0 && L > 0 && T > 0 && K >= 4 && K % 4 == 0, "bad sizes");
    SchedPtrs sp;
    for (int i = 0; i < 8; ++i) {
        GSDD_CHECK_ARG(sched[i] != nullptr, "null schedule buffer");
        sp.p[i] = sched[i];
    }
    const int64_t npos = (int64_t)B * L;
    hipLaunchKernelGGL(d3pm_q_sample_kernel, dim3((unsigned)((npos + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x0,
                       xt, B, L, K, T, sp, t_dev, seed, stream_dev, row0);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_advance(int64_t* t_dev, int B, int64_t dt, int64_t* stream_dev, int64_t ds, void* stream) {
    GSDD_CHECK_ARG(B >= 0 && B <= 65536, "bad B");
    const int n = B > 0 ? B : 1;
    hipLaunchKernelGGL(advance_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t_dev, B, dt,
                       stream_dev, ds);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_philox_uniform(uint64_t seed, int64_t stream_id, int64_t row0, int64_t n_rows, int n_cols,
                                   float* out, void* stream) {
    GSDD_CHECK_ARG(out != nullptr && n_rows > 0 && n_cols > 0, "bad args");
    const int64_t n = n_rows * (int64_t)((n_cols + 3) / 4);
    hipLaunchKernelGGL(philox_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       seed, (uint32_t)stream_id, row0, n_rows, n_cols, out);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}
