// Fused D3PM reverse step: log_softmax x2 -> classifier-free mix -> q_posterior -> Gumbel arg-max.
//
// One wave64 owns one token position; the K(+1)-wide categorical lives in registers
// (lane owns quads k = 4*lane + 256*j), every reduction over classes is a wave reduction.
// HBM-bound by design: algorithmic traffic = the two logit rows (2*K*4 B) per position.
//
// Mirrors the reference op-for-op in fp32 (diffusion_transformer.py:220-283, :354-359); the
// reference's fp64 log_softmax (:231) is matched by accumulating the exp-sum and taking its log in fp64.
#include <stdlib.h>

#include "common.hpp"

namespace gsdd {

constexpr float LOG_ZERO = -69.07755278982137f;  // log(1e-30)

// expf / logf restricted to the argument ranges this file feeds them.  Both run the device library's own
// arithmetic (two-constant log2(e) / ln 2 products around v_exp_f32 / v_log_f32, same constants, same order), so
// results are bit-identical to expf / logf on those ranges; what is dropped is the library's range plumbing
// (two compare+select pairs per expf, denormal pre-scaling and the inf/nan select per logf), which costs more issue
// slots than the arithmetic itself (tools/rate_probe6.hip: compares and selects issue at half the f32 add/mul rate).
//   exp_le0(x): x <= 0 (or -inf).  The lower clamp stands in for the library's "x < -103.28 -> 0" select:
//               exp(-104) < 2^-150 rounds to 0 in v_ldexp_f32, and it keeps a -inf / absurdly negative
//               argument away from the hi/lo product.
//   log_norm(x): x finite and >= 2^-126 (here always >= 1e-30).
//               CLAMP = false where the argument is known to be finite and of moderate size (differences of clamped
//               log-probabilities and schedule constants): the clamp is one more half-rate instruction.
template <bool CLAMP = true>
__device__ __forceinline__ float exp_le0(float x) {
    const float L2E_HI = __builtin_bit_cast(float, 0x3fb8aa3bu), L2E_LO = __builtin_bit_cast(float, 0x32a5705fu);
    if (CLAMP) x = fmaxf(x, -104.f);
    const float ph = x * L2E_HI;
    float pl = fmaf(x, L2E_HI, -ph);
    const float e = rintf(ph);
    pl = fmaf(x, L2E_LO, pl);
    const float a = (ph - e) + pl;
    return ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
}
// exp(d), d <= 0 (or -inf), for the TERMS OF A SUM over classes: one product and v_exp_f32 (relative error ~ |d| * 1.7e-7, i.e. small
// exactly where the term is not).  A log-sum-exp enters every class of its row as the same additive constant, which the arg-max at
// the end of the step does not see; what its accuracy decides is how often x - lse rounds to the neighbouring float.  With these
// terms the fp64 sum over a row is accurate to ~1e-9 relative (the largest term is exp(0) = 1 exactly), the same as with the
// 12-instruction exp_le0, which is kept for the per-class values (log_add_exp), where every class has its own error.
#ifdef GSDD_DEV_EXACT_SUM_EXP      // development A/B only (tools/neartie_diff.py): the sum terms on the library-exact exponential, as before 2f17ed5
__device__ __forceinline__ float exp_term(float d) { return exp_le0(d); }
#else
__device__ __forceinline__ float exp_term(float d) { return __builtin_amdgcn_exp2f(d * 1.44269504088896340736f); }
#endif
__device__ __forceinline__ float log_norm(float x) {
    const float LN2_HI = __builtin_bit_cast(float, 0x3f317217u), LN2_LO = __builtin_bit_cast(float, 0x3377d1cfu);
    const float r = __builtin_amdgcn_logf(x);
    const float ph = r * LN2_HI;
    float pl = fmaf(r, LN2_HI, -ph);
    pl = fmaf(r, LN2_LO, pl);
    return ph + pl;
}

// reference log_add_exp (:32-34): m + log(exp(a - m) + exp(b - m)), m = max(a, b).  One of the two exponentials is
// exp(0) = 1 exactly and the other one's argument is -|a - b| exactly, so a single exp gives the same bits.
__device__ __forceinline__ float lae(float a, float b) {
    const float m = fmaxf(a, b);
    return m + log_norm(1.f + exp_le0(-fabsf(a - b)));   // clamped: a schedule constant may be log(0) = -inf (t - 1 wrap)
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
        for (int e = 0; e < 4; ++e) se += (double)exp_term(x[j][e] - mx);
    se = wave_sum(se);
    const double lse = (double)mx + log(se);
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) x[j][e] = clamp70((float)((double)x[j][e] - lse));
}

// CLAMP = false: every slot holds a finite, bounded value (FULL rows of clamped log-probabilities)
template <int J, bool CLAMP = true>
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
        for (int e = 0; e < 4; ++e) se += exp_term(x[j][e] - mx);
    se = wave_sum(se);
    if (has_extra) se += exp_le0(extra - mx);
    return mx + logf(se);
}

__device__ __forceinline__ float gumbel(float u) {  // log_sample_categorical (:355-356); u in [0, 1)
    return -log_norm(-log_norm(u + 1e-30f) + 1e-30f);
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

// FULL: K == 256 * J, every register slot holds a class (no validity selects).  DBG: the posterior / x0 test hooks
// are compiled in; the production instantiation has none, so the unrolled class loops are single basic blocks.
// Everything that depends only on the position (token, timestep, schedule row, which register slot holds class x_t)
// is wave-uniform: the wave index goes through readfirstlane so that the compiler keeps it in SGPRs, and the
// "k == x_t" special case becomes a scalar branch on j plus one lane compare instead of a compare+select per class.
// OCC = waves per SIMD the register budget is sized for: 3 -> 168 VGPRs, 2 -> 256.  At K = 4096 (J = 16) the two 64-register
// rows plus the temporaries of the unrolled class loops need ~180: with OCC = 3 the compiler parks 8 of them in scratch memory
// (2 KB written per position, 131 MB per launch at B*L = 65536 -- the WRITE_SIZE of profiles/r1_pmc_traffic.csv); OCC = 2 has none.
template <int J, bool FULL, bool DBG, int OCC = 3>
__global__ __launch_bounds__(256, OCC) void d3pm_step_kernel(gsdd_step_desc d, SchedPtrs sp) {
    const int lane = threadIdx.x & 63;
    const int64_t pos = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
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
            if (FULL || k < K) {
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
            if (FULL || k < K) {
                const float4 v = *reinterpret_cast<const float4*>(row + k);
                xu[j][0] = v.x; xu[j][1] = v.y; xu[j][2] = v.z; xu[j][3] = v.w;
            } else {
                xu[j][0] = xu[j][1] = xu[j][2] = xu[j][3] = NEG;
            }
        }
        log_softmax_clamp<J>(xu);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const bool valid = FULL || (4 * lane + 256 * j) < K;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float diff = x0[j][e] - xu[j][e];
                const float sc = d.guidance * diff;
                x0[j][e] = valid ? (xu[j][e] + sc) : NEG;
            }
        }
        const float lse = wave_logsumexp<J, !FULL>(x0, 0.f, false);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const bool valid = FULL || (4 * lane + 256 * j) < K;
#pragma unroll
            for (int e = 0; e < 4; ++e) x0[j][e] = valid ? clamp70(x0[j][e] - lse) : NEG;
        }
    }
    if (DBG && d.x0_dbg != nullptr) {
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
    const int xt = (int)d.tok_in[pos];
    const bool masked = (xt == K);
    // class x_t sits in register slot [xj][xe] of lane xl (for x_t = [MASK] that slot does not exist or is invalid,
    // and the "hit" constants equal the "miss" ones, so no case distinction is needed below)
    const int xj = xt >> 8, xl = (xt >> 2) & 63, xe = xt & 3;
    const bool mine = (lane == xl);
    const float qt_miss = masked ? s.lcc : lae(LOG_ZERO + s.lca, s.lcb);
    const float qt_hit = masked ? s.lcc : lae(0.f + s.lca, s.lcb);
    const float q1_miss = masked ? s.lc : lae(LOG_ZERO + s.la, s.lb);
    const float q1_hit = masked ? s.lc : lae(0.f + s.la, s.lb);
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const bool valid = FULL || (4 * lane + 256 * j) < K;
        if (j == xj) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x0[j][e] = valid ? (x0[j][e] - ((mine && e == xe) ? qt_hit : qt_miss)) : NEG;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x0[j][e] = valid ? (x0[j][e] - qt_miss) : NEG;
        }
    }
    const float S = wave_logsumexp<J, !FULL>(x0, LOG_ZERO, true);
    float best = NEG;
    int best_k = 0;
    const uint32_t kp4 = (uint32_t)((K + 1 + 3) / 4);
    const uint32_t stream_id = (uint32_t)d.stream_dev[0];
    const uint64_t grow = (uint64_t)(d.row0 + pos);
    auto draw = [&](int j, int e, float log_q1, float u) {
        const int k = 4 * lane + 256 * j + e;
        const float qn = x0[j][e] - S;
        const float o = clamp70(lae(qn + s.pca, s.pcb) + log_q1 + S);
        if (DBG && d.post_dbg != nullptr) d.post_dbg[((int64_t)b * (K + 1) + k) * d.L + l] = o;
        const float v = gumbel(u) + o;
        if (v > best) { best = v; best_k = k; }
    };
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k0 = 4 * lane + 256 * j;
        if (FULL || k0 < K) {
            const float4 u4 = philox_uniform4(d.seed, stream_id, grow, kp4, (uint32_t)(k0 >> 2));
            const float u[4] = {u4.x, u4.y, u4.z, u4.w};
            if (j == xj) {
#pragma unroll
                for (int e = 0; e < 4; ++e) draw(j, e, (mine && e == xe) ? q1_hit : q1_miss, u[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) draw(j, e, q1_miss, u[e]);
            }
        }
    }
    if (lane == ((K >> 2) & 63)) {  // the [MASK] class k = K (K % 4 == 0 -> word 0 of quad K/4)
        const float4 u4 = philox_uniform4(d.seed, stream_id, grow, kp4, (uint32_t)(K >> 2));
        const float qn = LOG_ZERO - S;
        const float log_q1 = masked ? 0.f : LOG_ZERO;
        const float o = clamp70(lae(qn + s.p1mcc, s.pcc) + log_q1 + S);
        if (DBG && d.post_dbg != nullptr) d.post_dbg[((int64_t)b * (K + 1) + K) * d.L + l] = o;
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


// ------------------------------------------------------------------------------------------------------------
// Training objective, forward value (diffusion_transformer.py:391-457): per position
//   log_x0_recon = predict_start(logits)                      (:231-236)
//   log_model    = q_posterior(log_x0_recon, onehot(x_t), t)   (:251-283)
//   log_true     = q_posterior(onehot(x_0),  onehot(x_t), t)
//   kl = sum_k exp(log_true)(log_true - log_model) ; nll = -sum_k exp(onehot(x_0)) log_model ;
//   kl_aux = sum_{k<K} exp(onehot(x_0)) (onehot(x_0) - log_x0_recon)
// written per position; x0_recon / x_{t-1}-recon arg-max tokens for pred_data and the acc/keep statistics.
struct TrainArgs {
    const float* logits;      // [B*L][K]
    const int64_t* x0;
    const int64_t* xt;
    const int64_t* t_dev;
    int B, L, K, T;
    float mw_mask, mw_other;  // mask_weight
    float* kl; float* nll; float* aux;       // [B*L]
    int64_t* x0_recon; int64_t* xt1_recon;   // [B*L]
    float* probs;             // optional [B][K+1][L] = exp(log_model)
};

// (two waves per SIMD: left to itself the compiler interleaves so many classes of the unrolled loops that it needs 256 VGPRs plus
// AGPR spill space, i.e. one wave per SIMD with nothing to cover its transcendentals' latency)
// FULL: K == 256 J, every register slot holds a class: no validity predicates (each `k < K` otherwise becomes an exec-mask region
// around its class: 245 s_and_saveexec + 291 branches per position in the K = 4096 instantiation).
template <int J, bool FULL = false>
__global__ __launch_bounds__(256, 2) void d3pm_train_loss_kernel(TrainArgs d, SchedPtrs sp) {
    const int lane = threadIdx.x & 63;
    const int64_t pos = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (pos >= (int64_t)d.B * d.L) return;
    const int b = (int)(pos / d.L), l = (int)(pos % d.L);
    const int K = d.K;
    const float NEG = -INFINITY;
    float xr[J][4];
    const float* row = d.logits + pos * (int64_t)K;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = 4 * lane + 256 * j;
        if (FULL || k < K) {
            const float4 v = *reinterpret_cast<const float4*>(row + k);
            xr[j][0] = v.x; xr[j][1] = v.y; xr[j][2] = v.z; xr[j][3] = v.w;
        } else {
            xr[j][0] = xr[j][1] = xr[j][2] = xr[j][3] = NEG;
        }
    }
    log_softmax_clamp<J>(xr);                       // log_x0_recon rows k < K; row K is -70
    const int64_t t = d.t_dev[b];
    const StepSched s = load_sched(sp.p, t, d.T);
    const int xt = (int)d.xt[pos], x0 = (int)d.x0[pos];       // (32-bit: the per-class compares are half-rate 64-bit ones otherwise)
    const bool masked = (xt == K);
    const float qt_hit = lae(0.f + s.lca, s.lcb), qt_miss = lae(LOG_ZERO + s.lca, s.lcb);
    const float q1_hit = lae(0.f + s.la, s.lb), q1_miss = lae(LOG_ZERO + s.la, s.lb);
    // As in d3pm_train_bwd_kernel: the position is wave-uniform, the two special classes (k = x_t, k = x_0) sit in register quads
    // xj / x0j, every other quad runs on scalar constants behind a scalar branch; no second 64-register array (q = x - log_qt is
    // recomputed: one subtraction).
    const int xj = masked ? -1 : (xt >> 8), x0j = x0 >> 8;
    const float qt_c = masked ? s.lcc : qt_miss, q1_c = masked ? s.lc : q1_miss;
    const float E30 = expf(LOG_ZERO);               // exp(log-onehot "zero") as the reference computes it
    const float s_pca = s.pca, s_pcb = s.pcb, s_p1mcc = s.p1mcc, s_pcc = s.pcc;
    auto for_classes = [&](auto f) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (j == xj || j == x0j) {                                                  // scalar branch
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * lane + 256 * j + e;
                    if (FULL || k < K) {
                        const bool hit = !masked && k == xt, tru = k == x0;
                        f(j, e, k, hit ? qt_hit : qt_c, hit ? q1_hit : q1_c, tru ? 0.f : LOG_ZERO, tru ? 1.f : E30);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * lane + 256 * j + e;
                    if (FULL || k < K) f(j, e, k, qt_c, q1_c, LOG_ZERO, E30);
                }
            }
        }
    };
    // pass 1: the two normalisers S_m (model) and S_t (true); arg-max of log_x0_recon (row K = -70 competes)
    float mxm = LOG_ZERO, mxt = LOG_ZERO;            // include the [MASK] row (= LOG_ZERO) in both maxima
    float best0 = NEG; int best0_k = 0;
    for_classes([&](int j, int e, int k, float log_qt, float, float lx0, float) {
        if (xr[j][e] > best0) { best0 = xr[j][e]; best0_k = k; }
        mxm = fmaxf(mxm, xr[j][e] - log_qt);
        mxt = fmaxf(mxt, lx0 - log_qt);
    });
    if (lane == 0 && -70.f > best0) { best0 = -70.f; best0_k = K; }     // first-max rule across the wave: wave_argmax
    const int x0rec = wave_argmax(best0, best0_k);
    mxm = wave_max(mxm); mxt = wave_max(mxt);
    float sem = 0.f, set = 0.f;
    for_classes([&](int j, int e, int, float log_qt, float, float lx0, float) {
        sem += exp_term((xr[j][e] - log_qt) - mxm);                                      // terms of sums: see exp_term
        set += exp_term((lx0 - log_qt) - mxt);
    });
    sem = wave_sum(sem) + expf(LOG_ZERO - mxm);
    set = wave_sum(set) + expf(LOG_ZERO - mxt);
    const float Sm = mxm + logf(sem), St = mxt + logf(set);
    // pass 2: posteriors, KL / NLL / aux partial sums, arg-max of log_model
    float kl = 0.f, nll = 0.f, aux = 0.f;
    float bestm = NEG; int bestm_k = 0;
    for_classes([&](int j, int e, int k, float log_qt, float log_q1, float lx0, float w0) {
        const float lm = clamp70(lae(((xr[j][e] - log_qt) - Sm) + s_pca, s_pcb) + log_q1 + Sm);
        const float ltr = clamp70(lae(((lx0 - log_qt) - St) + s_pca, s_pcb) + log_q1 + St);
        kl += exp_le0(ltr) * (ltr - lm);
        nll += w0 * lm;
        aux += w0 * (lx0 - xr[j][e]);
        if (lm > bestm) { bestm = lm; bestm_k = k; }
        if (d.probs != nullptr) d.probs[((int64_t)b * (K + 1) + k) * d.L + l] = exp_le0(lm);
    });
    kl = wave_sum(kl); nll = wave_sum(nll); aux = wave_sum(aux);
    {   // the [MASK] class row
        const float log_q1K = masked ? 0.f : LOG_ZERO;
        const float lmK = clamp70(lae((LOG_ZERO - Sm) + s_p1mcc, s_pcc) + log_q1K + Sm);
        const float ltK = clamp70(lae((LOG_ZERO - St) + s_p1mcc, s_pcc) + log_q1K + St);
        kl += expf(ltK) * (ltK - lmK);
        nll += (x0 == K ? 1.f : E30) * lmK;
        if (lane == 0) {
            if (lmK > bestm) { bestm = lmK; bestm_k = K; }
            if (d.probs != nullptr) d.probs[((int64_t)b * (K + 1) + K) * d.L + l] = expf(lmK);
        }
    }
    const int xt1 = wave_argmax(bestm, bestm_k);
    if (lane == 0) {
        const float mw = masked ? d.mw_mask : d.mw_other;
        d.kl[pos] = kl * mw;
        d.nll[pos] = -nll;
        d.aux[pos] = aux * mw;
        d.x0_recon[pos] = x0rec;
        d.xt1_recon[pos] = xt1;
    }
}

// per-sample reductions + the scalar tail of _train_loss / forward (:425-457, :548): one workgroup
struct TrainFinArgs {
    const float* kl; const float* nll; const float* aux;
    const int64_t* x0; const int64_t* xt; const int64_t* x0_recon; const int64_t* xt1_recon;
    const int64_t* t_dev; const float* pt;
    int B, L, T;
    float aux_weight; int adaptive;
    float* Lt_history; float* Lt_count;
    float* loss;              // [1]
    float* per_sample;        // [B][4]: kl_loss, vb_loss, acc rate, keep rate
};

__global__ __launch_bounds__(256) void d3pm_train_finalize_kernel(TrainFinArgs a) {
    __shared__ float red[4][256];
    __shared__ float vb_all[1024];
    const int tid = threadIdx.x;
    for (int b = 0; b < a.B; ++b) {
        float kl = 0.f, nll = 0.f, aux = 0.f, same0 = 0.f, same1 = 0.f;
        for (int l = tid; l < a.L; l += 256) {
            const int64_t p = (int64_t)b * a.L + l;
            kl += a.kl[p]; nll += a.nll[p]; aux += a.aux[p];
            same0 += (a.x0_recon[p] == a.x0[p]) ? 1.f : 0.f;
            same1 += (a.xt1_recon[p] == a.xt[p]) ? 1.f : 0.f;
        }
        // block sums: wave shuffles, then the four wave totals through LDS (two barriers per sample instead of forty-five)
        float vals[5] = {wave_sum(kl), wave_sum(nll), wave_sum(aux), wave_sum(same0), wave_sum(same1)};
        float tot[5];
        if ((tid & 63) == 0)
            for (int q = 0; q < 5; ++q) red[0][(tid >> 6) * 8 + q] = vals[q];
        __syncthreads();
        for (int q = 0; q < 5; ++q) tot[q] = (red[0][q] + red[0][8 + q]) + (red[0][16 + q] + red[0][24 + q]);
        __syncthreads();
        if (tid == 0) {
            const int64_t t = a.t_dev[b];
            const float m0 = (t == 0) ? 1.f : 0.f;
            const float kl_loss = m0 * tot[1] + (1.f - m0) * tot[0];
            float vb = kl_loss / a.pt[b];
            if (a.aux_weight != 0.f) {
                const float kl_aux_loss = m0 * tot[1] + (1.f - m0) * tot[2];
                const float w = a.adaptive ? ((1.f - (float)t / (float)a.T) + 1.0f) : 1.0f;
                vb += w * a.aux_weight * kl_aux_loss / a.pt[b];
            }
            // Lt_history.scatter_(t, 0.1*Lt2 + 0.9*prev) ; Lt_count.scatter_add_(t, 1)   (:432-436; sample order)
            const float lt2 = kl_loss * kl_loss;
            a.Lt_history[t] = 0.1f * lt2 + 0.9f * a.Lt_history[t];
            a.Lt_count[t] += 1.f;
            a.per_sample[4 * b + 0] = kl_loss;
            a.per_sample[4 * b + 1] = vb;
            a.per_sample[4 * b + 2] = tot[3] / (float)a.L;
            a.per_sample[4 * b + 3] = tot[4] / (float)a.L;
            vb_all[b] = vb;
        }
        __syncthreads();
    }
    if (tid == 0) {
        float sum = 0.f;
        for (int b = 0; b < a.B; ++b) sum += vb_all[b];
        a.loss[0] = sum / ((float)a.B * (float)a.L);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Gradient of the training objective w.r.t. the denoiser logits (backward of predict_start -> q_posterior -> KL / NLL /
// aux KL, diffusion_transformer.py:391-457).  Per position, with G_c = dL/d log_model_c = -(g_kl exp(log_true_c) + g_nll w0_c):
//   lm = clamp(e + lq1 + S), e = lae(qn + alpha, beta), qn = q - S, S = lse(q), q_k = r_k - lqt_k, r = clamp(log_softmax(x))
// g_* are the per-sample weights of the three sums in loss = sum_b vb_b / (B L).
struct TrainBwdArgs {
    const float* logits; const int64_t* x0; const int64_t* xt; const int64_t* t_dev; const float* pt;
    int B, L, K, T;
    float mw_mask, mw_other, aux_weight; int adaptive;
    float* dlogits;
    float* kl; float* nll; float* aux; int64_t* x0_recon; int64_t* xt1_recon;      // LOSS = true: the forward kernel's outputs too
};

// LOSS = true (gsdd_d3pm_train_loss_grad): the same pass also leaves what d3pm_train_loss_kernel computes -- the per-position KL /
// NLL / auxiliary-KL sums and the two arg-max tokens, by the same operations in the same order (bit-identical values) -- so the
// training step reads the (B L, K) logits once instead of twice and evaluates the posteriors once (1.7 ms of a 67 ms step at bs 16).
template <int J, bool LOSS = false, bool FULL = false>
__global__ __launch_bounds__(256, 2) void d3pm_train_bwd_kernel(TrainBwdArgs d, SchedPtrs sp) {
    const int lane = threadIdx.x & 63;
    const int64_t pos = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (pos >= (int64_t)d.B * d.L) return;
    const int b = (int)(pos / d.L);
    const int K = d.K;
    const float NEG = -INFINITY;
    float a[J][4];                                  // unclamped log_softmax
    const float* row = d.logits + pos * (int64_t)K;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = 4 * lane + 256 * j;
        if (FULL || k < K) {
            const float4 v = *reinterpret_cast<const float4*>(row + k);
            a[j][0] = v.x; a[j][1] = v.y; a[j][2] = v.z; a[j][3] = v.w;
        } else {
            a[j][0] = a[j][1] = a[j][2] = a[j][3] = NEG;
        }
    }
    {   // log_softmax (fp64 sum) without the clamp
        float mx = NEG;
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, a[j][e]);
        mx = wave_max(mx);
        double se = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) se += (double)exp_term(a[j][e] - mx);
        se = wave_sum(se);
        const double lse = (double)mx + log(se);
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[j][e] = (float)((double)a[j][e] - lse);
    }
    int x0rec = 0;
    if (LOSS) {     // arg-max of log_x0_recon = clamp(log_softmax) over k < K, row K = -70 competing (first maximum wins)
        float best0 = NEG; int best0_k = 0;
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * lane + 256 * j + e;
                if (FULL || k < K) {
                    const float xr = clamp70(a[j][e]);
                    if (xr > best0) { best0 = xr; best0_k = k; }
                }
            }
        if (lane == 0 && -70.f > best0) { best0 = -70.f; best0_k = K; }
        x0rec = wave_argmax(best0, best0_k);
    }
    const int64_t t = d.t_dev[b];
    const StepSched s = load_sched(sp.p, t, d.T);
    const int xt = (int)d.xt[pos], x0 = (int)d.x0[pos];
    const bool masked = (xt == K);
    const float qt_hit = lae(0.f + s.lca, s.lcb), qt_miss = lae(LOG_ZERO + s.lca, s.lcb);
    const float q1_hit = lae(0.f + s.la, s.lb), q1_miss = lae(LOG_ZERO + s.la, s.lb);
    // Everything that depends only on the position is wave-uniform (pos comes through readfirstlane, so x_t, x_0, t and the schedule
    // row are scalar loads).  The two classes that are special -- k = x_t (the "hit" transition constants) and k = x_0 (the one-hot of
    // the true posterior) -- sit in register quads xj = x_t / 256 and x0j = x_0 / 256: every other quad runs with scalar constants, and
    // only those (at most two) quads evaluate per-class selects.  With a select per class in every pass the compiler kept 64-entry
    // vectors of them alive across the passes: 256 VGPRs + AGPR / scratch spills and one wave per SIMD.
    // (named scalars, not the struct: a struct captured by reference in the class-walking lambdas is demoted to scratch memory)
    const float s_pca = s.pca, s_pcb = s.pcb, s_p1mcc = s.p1mcc, s_pcc = s.pcc, s_lcc = s.lcc, s_lc = s.lc;
    const int xj = masked ? -1 : (xt >> 8), x0j = x0 >> 8;
    const float qt_c = masked ? s_lcc : qt_miss, q1_c = masked ? s_lc : q1_miss;       // the constants of a plain class
    const float E30 = expf(LOG_ZERO);
    // per-sample weights
    const float m0 = (t == 0) ? 1.f : 0.f;
    const float mw = masked ? d.mw_mask : d.mw_other;
    const float inv = 1.f / (d.pt[b] * (float)d.B * (float)d.L);
    const float w = d.adaptive ? ((1.f - (float)t / (float)d.T) + 1.0f) : 1.0f;
    const float g_kl = (1.f - m0) * mw * inv;
    const float g_nll = m0 * (1.f + w * d.aux_weight) * inv;
    const float g_aux = (1.f - m0) * w * d.aux_weight * mw * inv;
    // walks the classes of the row: f(j, e, k, log_qt, log_q1, lx0, w0) with the constants of class k
    auto for_classes = [&](auto f) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            if (j == xj || j == x0j) {                                                  // scalar branch
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * lane + 256 * j + e;
                    if (FULL || k < K) {
                        const bool hit = !masked && k == xt, tru = k == x0;
                        f(j, e, k, hit ? qt_hit : qt_c, hit ? q1_hit : q1_c, tru ? 0.f : LOG_ZERO, tru ? 1.f : E30);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * lane + 256 * j + e;
                    if (FULL || k < K) f(j, e, k, qt_c, q1_c, LOG_ZERO, E30);
                }
            }
        }
    };
    // normalisers of the model / true posteriors
    float mxm = LOG_ZERO, mxt = LOG_ZERO;
    for_classes([&](int j, int e, int, float log_qt, float, float lx0, float) {
        mxm = fmaxf(mxm, clamp70(a[j][e]) - log_qt);
        mxt = fmaxf(mxt, lx0 - log_qt);
    });
    mxm = wave_max(mxm); mxt = wave_max(mxt);
    float sem = 0.f, set = 0.f;
    for_classes([&](int j, int e, int, float log_qt, float, float lx0, float) {
        sem += exp_term((clamp70(a[j][e]) - log_qt) - mxm);
        set += exp_term((lx0 - log_qt) - mxt);
    });
    sem = wave_sum(sem) + expf(LOG_ZERO - mxm);
    set = wave_sum(set) + expf(LOG_ZERO - mxt);
    const float Sm = mxm + logf(sem), St = mxt + logf(set);
    // pass A: Gqn_c and the gradient reaching S.  The per-class gradient under construction lives in LDS (one 16-byte slot per lane
    // and register quad, lane-contiguous: conflict-free b128 accesses), not in a second 64-register array.
    extern __shared__ __attribute__((aligned(16))) float gq_lds[];
    float* const gqs = gq_lds + (((threadIdx.x >> 6) * J) * 64 + lane) * 4;            // class (j, e) at gqs[256 j + e]
    float sumGe = 0.f, sumGqn = 0.f;
    float kl = 0.f, nll = 0.f, aux = 0.f;           // LOSS: the forward kernel's three sums and the arg-max of log_model
    float bestm = NEG; int bestm_k = 0;
    for_classes([&](int j, int e, int k, float log_qt, float log_q1, float lx0, float w0) {
        const float xr = clamp70(a[j][e]);
        const float qn = (xr - log_qt) - Sm;
        const float ee = lae(qn + s_pca, s_pcb);
        const float pre = ee + log_q1 + Sm;
        const float ltr = clamp70(lae(((lx0 - log_qt) - St) + s_pca, s_pcb) + log_q1 + St);
        const float eltr = exp_le0(ltr);
        const float G = -(g_kl * eltr + g_nll * w0);
        const float Ge = (pre >= -70.f && pre <= 0.f) ? G : 0.f;
        const float Gqn = Ge * exp_le0((qn + s_pca) - ee);
        gqs[256 * j + e] = Gqn;
        sumGe += Ge; sumGqn += Gqn;
        if (LOSS) {
            const float lm = clamp70(pre);
            kl += eltr * (ltr - lm);
            nll += w0 * lm;
            aux += w0 * (lx0 - xr);
            if (lm > bestm) { bestm = lm; bestm_k = k; }
        }
    });
    sumGe = wave_sum(sumGe); sumGqn = wave_sum(sumGqn);
    if (LOSS) { kl = wave_sum(kl); nll = wave_sum(nll); aux = wave_sum(aux); }
    {   // the [MASK] class: q_K is a constant, it only feeds S
        const float log_q1K = masked ? 0.f : LOG_ZERO;
        const float qnK = LOG_ZERO - Sm;
        const float eK = lae(qnK + s_p1mcc, s_pcc);
        const float preK = eK + log_q1K + Sm;
        const float ltK = clamp70(lae((LOG_ZERO - St) + s_p1mcc, s_pcc) + log_q1K + St);
        const float eltK = expf(ltK);
        const float GK = -(g_kl * eltK + g_nll * (x0 == K ? 1.f : E30));
        const float GeK = (preK >= -70.f && preK <= 0.f) ? GK : 0.f;
        sumGe += GeK;
        sumGqn += GeK * expf((qnK + s_p1mcc) - eK);
        if (LOSS) {
            const float lmK = clamp70(preK);
            kl += eltK * (ltK - lmK);
            nll += (x0 == K ? 1.f : E30) * lmK;
            if (lane == 0 && lmK > bestm) { bestm = lmK; bestm_k = K; }
        }
    }
    if (LOSS) {
        const int xt1 = wave_argmax(bestm, bestm_k);
        if (lane == 0) {
            d.kl[pos] = kl * mw;
            d.nll[pos] = -nll;
            d.aux[pos] = aux * mw;
            d.x0_recon[pos] = x0rec;
            d.xt1_recon[pos] = xt1;
        }
    }
    const float GS = sumGe - sumGqn;
    // pass B: through q -> r -> clamp -> log_softmax
    float sumGa = 0.f;
    for_classes([&](int j, int e, int, float log_qt, float, float, float w0) {
        const float r = clamp70(a[j][e]);
        const float pi = exp_le0((r - log_qt) - Sm);
        float Gr = gqs[256 * j + e] + GS * pi;
        Gr -= g_aux * w0;
        const float Ga = (a[j][e] >= -70.f && a[j][e] <= 0.f) ? Gr : 0.f;
        gqs[256 * j + e] = Ga;
        sumGa += Ga;
    });
    sumGa = wave_sum(sumGa);
    float* drow = d.dlogits + pos * (int64_t)K;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int k = 4 * lane + 256 * j;
        if (FULL || k < K) {
            float4 o;
            const float4 gv = *reinterpret_cast<const float4*>(gqs + 256 * j);
            // softmax probabilities of the gradient (relative error of exp_term: |a| 1.7e-7, at most 1.2e-5 and only where p < e^-69)
            o.x = gv.x - exp_term(a[j][0]) * sumGa; o.y = gv.y - exp_term(a[j][1]) * sumGa;
            o.z = gv.z - exp_term(a[j][2]) * sumGa; o.w = gv.w - exp_term(a[j][3]) * sumGa;
            *reinterpret_cast<float4*>(drow + k) = o;
        }
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
    const bool dbg = d->post_dbg != nullptr || d->x0_dbg != nullptr;
    GSDD_CHECK_ARG(d->occupancy == 0 || d->occupancy == 2 || d->occupancy == 3, "occupancy: 0 (auto), 2 or 3 waves per SIMD");
    if (J > 8 && J <= 16 && d->K == 4096 && !dbg && d->occupancy != 3) {       // the production shape: no scratch (see the kernel's note)
        hipLaunchKernelGGL((d3pm_step_kernel<16, true, false, 2>), grid, block, 0, st, *d, sp);
        GSDD_CHECK_LAUNCH();
        return GSDD_OK;
    }
#define GSDD_STEP_LAUNCH(JJ)                                                                                         \
    do {                                                                                                             \
        const bool full = d->K == 256 * (JJ);                                                                        \
        if (full && !dbg) hipLaunchKernelGGL((d3pm_step_kernel<JJ, true, false>), grid, block, 0, st, *d, sp);      \
        else if (full) hipLaunchKernelGGL((d3pm_step_kernel<JJ, true, true>), grid, block, 0, st, *d, sp);          \
        else if (!dbg) hipLaunchKernelGGL((d3pm_step_kernel<JJ, false, false>), grid, block, 0, st, *d, sp);        \
        else hipLaunchKernelGGL((d3pm_step_kernel<JJ, false, true>), grid, block, 0, st, *d, sp);                   \
    } while (0)
    if (J <= 1) GSDD_STEP_LAUNCH(1);
    else if (J <= 2) GSDD_STEP_LAUNCH(2);
    else if (J <= 4) GSDD_STEP_LAUNCH(4);
    else if (J <= 8) GSDD_STEP_LAUNCH(8);
    else if (J <= 16) GSDD_STEP_LAUNCH(16);
    else GSDD_STEP_LAUNCH(32);
#undef GSDD_STEP_LAUNCH
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

extern "C" int gsdd_d3pm_train_loss(const gsdd_train_desc* d, void* stream) {
    GSDD_CHECK_ARG(d != nullptr, "null descriptor");
    GSDD_CHECK_ARG(d->logits && d->x0 && d->xt && d->t_dev && d->pt && d->kl && d->nll && d->aux && d->x0_recon &&
                   d->xt1_recon && d->Lt_history && d->Lt_count && d->loss && d->per_sample, "null pointer");
    GSDD_CHECK_ARG(d->B > 0 && d->B <= 1024 && d->L > 0 && d->T > 0, "bad sizes (B <= 1024)");
    GSDD_CHECK_ARG(d->K >= 4 && d->K % 4 == 0 && d->K <= 8192, "K must be a multiple of 4 in [4, 8192]");
    SchedPtrs sp;
    for (int i = 0; i < 8; ++i) {
        GSDD_CHECK_ARG(d->sched[i] != nullptr, "null schedule buffer");
        sp.p[i] = d->sched[i];
    }
    TrainArgs a;
    a.logits = d->logits; a.x0 = d->x0; a.xt = d->xt; a.t_dev = d->t_dev; a.B = d->B; a.L = d->L; a.K = d->K; a.T = d->T;
    a.mw_mask = d->mask_weight[0]; a.mw_other = d->mask_weight[1];
    a.kl = d->kl; a.nll = d->nll; a.aux = d->aux; a.x0_recon = d->x0_recon; a.xt1_recon = d->xt1_recon; a.probs = d->probs;
    const int64_t npos = (int64_t)d->B * d->L;
    const dim3 grid((unsigned)((npos + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int J = (d->K + 255) / 256;
    if (d->K == 4096) hipLaunchKernelGGL((d3pm_train_loss_kernel<16, true>), grid, block, 0, st, a, sp);
    else if (J <= 1) hipLaunchKernelGGL(d3pm_train_loss_kernel<1>, grid, block, 0, st, a, sp);
    else if (J <= 2) hipLaunchKernelGGL(d3pm_train_loss_kernel<2>, grid, block, 0, st, a, sp);
    else if (J <= 4) hipLaunchKernelGGL(d3pm_train_loss_kernel<4>, grid, block, 0, st, a, sp);
    else if (J <= 8) hipLaunchKernelGGL(d3pm_train_loss_kernel<8>, grid, block, 0, st, a, sp);
    else if (J <= 16) hipLaunchKernelGGL(d3pm_train_loss_kernel<16>, grid, block, 0, st, a, sp);
    else hipLaunchKernelGGL(d3pm_train_loss_kernel<32>, grid, block, 0, st, a, sp);
    GSDD_CHECK_LAUNCH();
    TrainFinArgs f;
    f.kl = d->kl; f.nll = d->nll; f.aux = d->aux; f.x0 = d->x0; f.xt = d->xt; f.x0_recon = d->x0_recon;
    f.xt1_recon = d->xt1_recon; f.t_dev = d->t_dev; f.pt = d->pt; f.B = d->B; f.L = d->L; f.T = d->T;
    f.aux_weight = d->aux_weight; f.adaptive = d->adaptive_aux; f.Lt_history = d->Lt_history; f.Lt_count = d->Lt_count;
    f.loss = d->loss; f.per_sample = d->per_sample;
    hipLaunchKernelGGL(d3pm_train_finalize_kernel, dim3(1), dim3(256), 0, st, f);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

static int train_bwd_launch(const gsdd_train_desc* d, float* dlogits, bool with_loss, void* stream);

extern "C" int gsdd_d3pm_train_loss_bwd(const gsdd_train_desc* d, float* dlogits, void* stream) {
    return train_bwd_launch(d, dlogits, false, stream);
}

extern "C" int gsdd_d3pm_train_loss_grad(const gsdd_train_desc* d, float* dlogits, void* stream) {
    GSDD_CHECK_ARG(d != nullptr && dlogits != nullptr, "null pointer");
    GSDD_CHECK_ARG(d->kl && d->nll && d->aux && d->x0_recon && d->xt1_recon && d->Lt_history && d->Lt_count && d->loss && d->per_sample,
                   "null pointer");
    GSDD_CHECK_ARG(d->probs == nullptr, "the fused pass does not write probs: call gsdd_d3pm_train_loss + gsdd_d3pm_train_loss_bwd");
    GSDD_CHECK_ARG(d->B <= 1024, "bad sizes (B <= 1024)");
    const int rc = train_bwd_launch(d, dlogits, true, stream);
    if (rc != GSDD_OK) return rc;
    TrainFinArgs f;
    f.kl = d->kl; f.nll = d->nll; f.aux = d->aux; f.x0 = d->x0; f.xt = d->xt; f.x0_recon = d->x0_recon;
    f.xt1_recon = d->xt1_recon; f.t_dev = d->t_dev; f.pt = d->pt; f.B = d->B; f.L = d->L; f.T = d->T;
    f.aux_weight = d->aux_weight; f.adaptive = d->adaptive_aux; f.Lt_history = d->Lt_history; f.Lt_count = d->Lt_count;
    f.loss = d->loss; f.per_sample = d->per_sample;
    hipLaunchKernelGGL(d3pm_train_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, f);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

static int train_bwd_launch(const gsdd_train_desc* d, float* dlogits, bool with_loss, void* stream) {
    GSDD_CHECK_ARG(d != nullptr && dlogits != nullptr, "null pointer");
    GSDD_CHECK_ARG(d->logits && d->x0 && d->xt && d->t_dev && d->pt, "null pointer");
    GSDD_CHECK_ARG(d->B > 0 && d->L > 0 && d->T > 0 && d->K >= 4 && d->K % 4 == 0 && d->K <= 8192, "bad sizes");
    SchedPtrs sp;
    for (int i = 0; i < 8; ++i) {
        GSDD_CHECK_ARG(d->sched[i] != nullptr, "null schedule buffer");
        sp.p[i] = d->sched[i];
    }
    TrainBwdArgs a;
    a.logits = d->logits; a.x0 = d->x0; a.xt = d->xt; a.t_dev = d->t_dev; a.pt = d->pt;
    a.B = d->B; a.L = d->L; a.K = d->K; a.T = d->T;
    a.mw_mask = d->mask_weight[0]; a.mw_other = d->mask_weight[1]; a.aux_weight = d->aux_weight; a.adaptive = d->adaptive_aux;
    a.dlogits = dlogits;
    a.kl = d->kl; a.nll = d->nll; a.aux = d->aux; a.x0_recon = d->x0_recon; a.xt1_recon = d->xt1_recon;
    const int64_t npos = (int64_t)d->B * d->L;
    const dim3 grid((unsigned)((npos + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int J = (d->K + 255) / 256;
    // dynamic LDS: one float4 per (wave, register quad, lane) = 4 KB * J per workgroup (64 KB at K = 4096: two workgroups per CU)
#define GSDD_BWD_LAUNCH(JJ)                                                                                                   \
    do {                                                                                                                      \
        if (with_loss) hipLaunchKernelGGL((d3pm_train_bwd_kernel<JJ, true>), grid, block, (size_t)4096 * JJ, st, a, sp);      \
        else hipLaunchKernelGGL((d3pm_train_bwd_kernel<JJ, false>), grid, block, (size_t)4096 * JJ, st, a, sp);               \
    } while (0)
    if (J > 16) {                                  // J = 32 (4096 < K <= 8192) needs 128 KB of dynamic LDS: above the 64 KB default
        int dev = 0, lds_max = 0;
        GSDD_CHECK_HIP(hipGetDevice(&dev));
        GSDD_CHECK_HIP(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
        GSDD_CHECK_ARG(lds_max >= 4096 * 32, "K > 4096 needs 128 KB of LDS per workgroup (gfx950 has 160 KB)");
        GSDD_ONCE_PER_DEVICE(attr_done,
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)d3pm_train_bwd_kernel<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 32));
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)d3pm_train_bwd_kernel<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 32));
        );
    }
    if (d->K == 4096) {                            // the workload's class count: every register slot holds a class
        if (with_loss) hipLaunchKernelGGL((d3pm_train_bwd_kernel<16, true, true>), grid, block, (size_t)4096 * 16, st, a, sp);
        else hipLaunchKernelGGL((d3pm_train_bwd_kernel<16, false, true>), grid, block, (size_t)4096 * 16, st, a, sp);
    } else if (J <= 1) GSDD_BWD_LAUNCH(1);
    else if (J <= 2) GSDD_BWD_LAUNCH(2);
    else if (J <= 4) GSDD_BWD_LAUNCH(4);
    else if (J <= 8) GSDD_BWD_LAUNCH(8);
    else if (J <= 16) GSDD_BWD_LAUNCH(16);
    else GSDD_BWD_LAUNCH(32);
#undef GSDD_BWD_LAUNCH
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}
