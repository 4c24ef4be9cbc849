// Train-mode pieces of the VQ-VAE forward: BatchNorm batch statistics (+ running-stat update and folding to
// scale/shift), codebook EMA statistics / update / restart, and the mean-squared-error reductions.
// Reference: nn.BatchNorm3d in train mode (videogpt_vq_vae.py:125-133, 242-247), Codebook.forward :190-219, VQVAE.forward :64.
#include "common.hpp"

namespace gsdd {

constexpr int ST_ROWS = 64;       // rows per stage-1 block: >= 1024 blocks at the training shapes (one thread per channel, serial over rows)

// stage 1: per (row slab, channel) partial sum and sum of squares
__global__ __launch_bounds__(256) void channel_stats_partial_kernel(const float* x, int64_t M, int C, double* part) {
    const int64_t r0 = (int64_t)blockIdx.x * ST_ROWS;
    const int64_t r1 = r0 + ST_ROWS < M ? r0 + ST_ROWS : M;
    for (int c = threadIdx.x; c < C; c += 256) {
        double s = 0.0, q = 0.0;
        for (int64_t r = r0; r < r1; ++r) {
            const double v = (double)x[r * C + c];
            s += v;
            q += v * v;
        }
        part[((int64_t)blockIdx.x * C + c) * 2 + 0] = s;
        part[((int64_t)blockIdx.x * C + c) * 2 + 1] = q;
    }
}

// stage 2 + BatchNorm bookkeeping: batch mean / biased variance -> folded (scale, shift); running stats updated with
// the unbiased variance and `momentum` exactly like torch (running = (1-m) running + m stat)
// block sum of a pair of doubles over 256 threads (result valid in thread 0)
__device__ __forceinline__ void block_sum2(double& s, double& q) {
    __shared__ double red[2][4];
    s = wave_sum(s); q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
}

// stage 2 + BatchNorm bookkeeping, one 256-thread block per channel: batch mean / biased variance -> folded (scale, shift);
// running stats updated with the unbiased variance and `momentum` exactly like torch (running = (1-m) running + m stat)
__global__ __launch_bounds__(256) void bn_train_finalize_kernel(const double* part, int nblk, int64_t M, int C, const float* weight,
                                                                const float* bias, float eps, float momentum, float* running_mean,
                                                                float* running_var, float* scale, float* shift, float* mean_rstd) {
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        s += part[((int64_t)b * C + c) * 2 + 0];
        q += part[((int64_t)b * C + c) * 2 + 1];
    }
    block_sum2(s, q);
    if (threadIdx.x != 0) return;
    const double mean = s / (double)M;
    double var = q / (double)M - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float meanf = (float)mean, varf = (float)var;
    const float sc = weight[c] / sqrtf(varf + eps);
    scale[c] = sc;
    shift[c] = bias[c] - meanf * sc;
    if (mean_rstd != nullptr) { mean_rstd[2 * c] = meanf; mean_rstd[2 * c + 1] = 1.0f / sqrtf(varf + eps); }
    if (running_mean != nullptr) {
        const float unbiased = (float)(var * ((double)M / (double)(M > 1 ? M - 1 : 1)));
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// n_total[k] = #rows with idx == k ; encode_sum[k][:] = sum of those rows (float atomics: order-dependent in the last bits)
__global__ __launch_bounds__(256) void code_stats_kernel(const float* z, const int64_t* idx, int64_t M, int E, float* n_total,
                                                         float* encode_sum) {
    const int q4 = E >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * q4) return;
    const int64_t row = i / q4;
    const int c = (int)(i % q4) * 4;
    const int64_t k = idx[row];
    const float4 v = *reinterpret_cast<const float4*>(z + row * E + c);
    float* dst = encode_sum + k * E + c;
    atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
    if (c == 0) atomicAdd(n_total + k, 1.0f);
}

// N <- 0.99 N + 0.01 n_total ; n = sum(N) ; perplexity from n_total     (one workgroup)
__global__ __launch_bounds__(1024) void codebook_ema_n_kernel(float* N, const float* n_total, int K, int64_t M, float decay,
                                                              float* n_sum, float* perplexity) {
    __shared__ double red[1024];
    double s = 0.0, h = 0.0;
    for (int k = threadIdx.x; k < K; k += 1024) {
        const float nv = N[k] * decay + (1.f - decay) * n_total[k];
        N[k] = nv;
        s += (double)nv;
        const float p = n_total[k] / (float)M;
        h += (double)(p * logf(p + 1e-10f));
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const double tot = red[0];
    __syncthreads();
    red[threadIdx.x] = h;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { n_sum[0] = (float)tot; perplexity[0] = expf(-(float)red[0]); }
}

// perplexity of the LOCAL one-hot mean (videogpt_vq_vae.py:218-219: avg_probs = mean(encode_onehot) over this rank's latents)
__global__ __launch_bounds__(1024) void code_perplexity_kernel(const float* n_local, int K, int64_t M, float* out) {
    __shared__ double red[1024];
    double h = 0.0;
    for (int k = threadIdx.x; k < K; k += 1024) {
        const float p = n_local[k] / (float)M;
        h += (double)(p * logf(p + 1e-10f));
    }
    red[threadIdx.x] = h;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = expf(-(float)red[0]);
}

// z_avg <- 0.99 z_avg + 0.01 encode_sum ; emb = z_avg / weights, dead codes (N < 1) restart from k_rand
__global__ void codebook_ema_emb_kernel(const float* N, float* z_avg, float* emb, const float* encode_sum, const float* z,
                                        const int64_t* perm, int K, int E, float decay, const float* n_sum) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * E) return;
    const int k = i / E, e = i % E;
    const float za = z_avg[i] * decay + (1.f - decay) * encode_sum[i];
    z_avg[i] = za;
    const float n = n_sum[0];
    const float w = (N[k] + 1e-7f) / (n + (float)K * 1e-7f) * n;
    const float usage = N[k] >= 1.f ? 1.f : 0.f;
    const float kr = z[perm[k] * E + e];
    emb[i] = (za / w) * usage + kr * (1.f - usage);
}

// sum of squared differences, deterministic two-stage reduction in fp64
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const float* a, const float* b, int64_t n, double* part) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = a[i] - b[i];
        s += (double)(d * d);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void sqdiff_final_kernel(const double* part, int nblk, double scale, float* out) {
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += part[i];
    out[0] = (float)(s * scale);
}

}  // namespace gsdd

using namespace gsdd;

extern "C" int64_t gsdd_bn_train_workspace_bytes(int64_t M, int C) {
    return ((M + ST_ROWS - 1) / ST_ROWS) * (int64_t)C * 2 * (int64_t)sizeof(double);
}

extern "C" int gsdd_bn_train(const float* x, int64_t M, int C, const float* weight, const float* bias, float eps, float momentum,
                             float* running_mean, float* running_var, float* scale, float* shift, float* mean_rstd,
                             void* workspace, int64_t workspace_bytes, void* stream) {
    GSDD_CHECK_ARG(x && weight && bias && scale && shift && workspace, "null pointer");
    GSDD_CHECK_ARG(M > 0 && C > 0, "bad sizes");
    GSDD_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running stats come together");
    GSDD_CHECK_ARG(workspace_bytes >= gsdd_bn_train_workspace_bytes(M, C), "workspace too small");
    const int nblk = (int)((M + ST_ROWS - 1) / ST_ROWS);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(channel_stats_partial_kernel, dim3(nblk), dim3(256), 0, st, x, M, C, (double*)workspace);
    GSDD_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_train_finalize_kernel, dim3(C), dim3(256), 0, st, (const double*)workspace, nblk, M, C,
                       weight, bias, eps, momentum, running_mean, running_var, scale, shift, mean_rstd);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_codebook_ema(const float* z, const int64_t* idx, int64_t M, int E, int K, float decay, const int64_t* perm,
                                 float* N, float* z_avg, float* embeddings, float* n_total, float* encode_sum,
                                 float* scalars, int phase, void* stream) {
    GSDD_CHECK_ARG(z && idx && N && z_avg && embeddings && n_total && encode_sum && scalars, "null pointer");
    GSDD_CHECK_ARG(M > 0 && E > 0 && E % 4 == 0 && K > 0, "bad sizes");
    GSDD_CHECK_ARG(phase == 0 || phase == 1, "phase 0 = statistics, 1 = update");
    hipStream_t st = (hipStream_t)stream;
    if (phase == 0) {      // local statistics; the caller all-reduces n_total / encode_sum across ranks before phase 1
        GSDD_CHECK_HIP(hipMemsetAsync(n_total, 0, (size_t)K * sizeof(float), st));
        GSDD_CHECK_HIP(hipMemsetAsync(encode_sum, 0, (size_t)K * E * sizeof(float), st));
        const int64_t n = M * (E / 4);
        hipLaunchKernelGGL(code_stats_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z, idx, M, E, n_total,
                           encode_sum);
    } else {
        GSDD_CHECK_ARG(perm != nullptr, "restart permutation required");
        hipLaunchKernelGGL(codebook_ema_n_kernel, dim3(1), dim3(1024), 0, st, N, n_total, K, M, decay, scalars, scalars + 1);
        GSDD_CHECK_LAUNCH();
        hipLaunchKernelGGL(codebook_ema_emb_kernel, dim3((K * E + 255) / 256), dim3(256), 0, st, N, z_avg, embeddings,
                           encode_sum, z, perm, K, E, decay, scalars);
    }
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_code_perplexity(const float* n_local, int K, int64_t M, float* out, void* stream) {
    GSDD_CHECK_ARG(n_local && out && K > 0 && M > 0, "bad args");
    hipLaunchKernelGGL(code_perplexity_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_local, K, M, out);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_mse(const float* a, const float* b, int64_t n, float scale, float* out, void* workspace,
                        int64_t workspace_bytes, void* stream) {
    GSDD_CHECK_ARG(a && b && out && workspace && n > 0, "bad args");
    const int nblk = (int)std::min<int64_t>((n + 255) / 256, 1024);
    GSDD_CHECK_ARG(workspace_bytes >= (int64_t)nblk * (int64_t)sizeof(double), "workspace too small (8 KiB)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(nblk), dim3(256), 0, st, a, b, n, (double*)workspace);
    GSDD_CHECK_LAUNCH();
    hipLaunchKernelGGL(sqdiff_final_kernel, dim3(1), dim3(1), 0, st, (const double*)workspace, nblk, (double)scale / (double)n, out);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}
