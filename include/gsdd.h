/*
 * gsdd.h — C ABI of the MI355X (gfx950) video-token generation hot path.
 *
 * Drop-in boundary for the VQ-VAE encode/quantise/decode + D3PM reverse-diffusion path of
 * Developer-Zer0/GIF-synthesis-with-Discrete-Diffusion.  The reference has NO native FFI for this
 * path: its seam is Hydra `_target_` instantiation of torch nn.Modules.  Each entry
 * point below therefore cites the reference *op sequence* (file:line under /root/reference) that
 * it replaces; the Python shells in gif-synthesis-with-discrete-diffusion_amd/ bind them with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'd / torch-ROCm storage); the library never
 *     allocates, frees or synchronises; all work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - fp32 everywhere; token / code indices are int64 (torch.argmin/argmax dtype).
 *   - return 0 on success, <0 on error (GSDD_E_*); gsdd_last_error() gives a message (thread-local).
 *   - activations are channels-last: a "row" is one position (n,t,h,w) with C contiguous floats.
 *   - arithmetic mode / kernel variant are ARGUMENTS (a `mode` / `variant` / `flags` value, 0 = the documented default), chosen per
 *     call and therefore per stream; the library reads no environment variable and keeps no mode latched in static state.  (The
 *     Python shells map their GSDD_* debugging environment variables onto these arguments: gif-synthesis-with-discrete-diffusion_amd/ops.py.)
 */
#ifndef GSDD_H
#define GSDD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSDD_OK 0
#define GSDD_E_ARG (-1)     /* bad argument (null pointer, unsupported shape)  */
#define GSDD_E_HIP (-2)     /* a HIP runtime call failed                        */
#define GSDD_E_STATE (-3)   /* graph capture misuse                             */

const char* gsdd_last_error(void);
int gsdd_version(void);
/* sizeof of a descriptor struct as this build of the library sees it (which: 0 gsdd_gemm_desc, 1 gsdd_layer_desc, 2 gsdd_step_desc,
 * 3 gsdd_train_desc; anything else: -1).  A binding checks its own struct sizes against these when it loads the library, so that a
 * library built from another revision of this header fails at load time instead of misreading a descriptor. */
int64_t gsdd_abi_sizeof(int which);

/* ------------------------------------------------------------------ generic implicit GEMM
 * out[orow(m)][n] = epi( sum_{tap,c} pro(in[src(m,tap)][c]) * w[tap][n][c] )
 * One kernel serves every dense contraction on the VQ-VAE path and the denoiser's linears:
 *   SamePadConv3d / nn.Conv3d            videogpt_vq_vae.py:289-309   (taps = kt*kh*kw)
 *   SamePadConvTranspose3d (per phase)   videogpt_vq_vae.py:312-332   (sub-pixel phases)
 *   nn.Linear                            model_utils.py:223-233, transformer_utils.py:36-43,258-263,353-356
 * Activation modes (act):  0 none, 1 ReLU, 2 x*sigmoid(1.702x) (GELU2, transformer_utils.py:115-119)
 */
typedef struct {
    /* input tensor: N x Di x Hi x Wi positions, row pitch `in_pitch` floats, Cin used channels */
    const float* in;
    int N, Di, Hi, Wi, Cin, in_pitch;
    /* output grid (positions enumerated (n,to,ho,wo)); rows M = N*Do*Ho*Wo */
    int Do, Ho, Wo;
    int sd, sh, sw;                 /* input stride per output step                          */
    int ntaps;
    const int* taps;                /* device int[ntaps*3]: (dt,dh,dw) added to o*stride     */
    const int64_t* gather;          /* optional: in row = gather[m] (ntaps must be 1)        */
    /* weights: [ntaps][Cout][Cin] floats (Cin contiguous) */
    const float* w;
    int Cout;
    /* prologue on the input operand */
    const float* pro_scale;         /* per input channel a = relu(x*ps[c]+pb[c]) (BN+ReLU)   */
    const float* pro_shift;
    const float* ln_stats;          /* per input row (mean, rstd) pairs -> LayerNorm         */
    const float* ln_gamma;          /* [sel][Cin]   a = (x-mu)*rs*gamma + beta               */
    const float* ln_beta;
    const int64_t* ln_sel;          /* optional per-batch selector (e.g. timestep), else 0   */
    int ln_stride;                  /* floats between selector rows                          */
    int rows_per_batch;             /* rows per batch element (for ln_sel / bvec)            */
    /* epilogue */
    const float* epi_scale;         /* per output channel, optional                          */
    const float* epi_shift;         /* per output channel (bias / folded BN), optional       */
    const float* bvec;              /* optional [batch][Cout] vector added per batch element */
    int act;
    const float* residual;          /* optional, same addressing as out                      */
    /* output addressing: row = ((n*oD + to*osd+ood)*oH + ho*osh+ooh)*oW + wo*osw+oow         */
    float* out;
    int oD, oH, oW, osd, osh, osw, ood, ooh, oow, out_pitch;
    int out_mode;                   /* 0 channels-last rows; 1 NCDHW (out[n][c][d][h][w]);
                                       2 head-major [n/4][M][4] (denoiser q/k/v)             */
    int flags;                      /* 0, or GSDD_GEMM_EXACT_F32                             */
} gsdd_gemm_desc;
/* flags: contractions of 64 or more run as error-free 3-way bf16 splits on the matrix pipe by default (six cross products, dropped
 * terms < 2^-24 relative); GSDD_GEMM_EXACT_F32 runs them on v_mfma_f32_32x32x2_f32 (an fmaf chain in f32) instead -- also honoured by
 * gsdd_conv_wgrad. */
#define GSDD_GEMM_EXACT_F32 1
int gsdd_gemm(const gsdd_gemm_desc* d, void* stream);

/* per-row LayerNorm statistics (mean, rstd) of x[M][C] (eps inside rsqrt).
 * Replaces the statistics half of nn.LayerNorm: transformer_utils.py:147,157,217,354 */
int gsdd_row_stats(const float* x, int64_t M, int C, float eps, float* stats, void* stream);

/* ------------------------------------------------------------------ layout at the NCDHW boundary
 * videogpt_vq_vae.py:58-60 hands (B,3,T,H,W); kernels are channels-last. */
int gsdd_ncdhw_to_rows(const float* x, int N, int C, int D, int H, int W, int Cpad, int padw,
                       float* out, void* stream);

/* ------------------------------------------------------------------ VQ-VAE specific
 * Axial attention over one axis of a (N,T,H,W) grid of fused q|k|v rows.
 * qkv row layout: [9*C] = (axis a: q,k,v) for a in (w,h,t); out row layout [3*C] = (a_w|a_h|a_t).
 * Replaces AxialAttention + scaled_dot_product_attention: model_utils.py:318-337, :586-600. */
#define GSDD_AXIAL_AUTO 0           /* register-resident MFMA kernel where the line length and head dim allow (16-position lines) */
#define GSDD_AXIAL_VALU 1           /* the LDS / vector kernel for every axis (any line length <= 64): the cross-check variant    */
int gsdd_axial_attention(const float* qkv, int N, int T, int H, int W, int C, int n_head,
                         float* out, int variant, void* stream);

/* Nearest codebook entry: idx[m] = argmin_k (|z_m|^2 - 2 z_m.e_k + |e_k|^2), first minimum wins.
 * Replaces Codebook.forward distance+argmin: videogpt_vq_vae.py:178-183.
 * z: [M][E] rows, cb: [K][E]; optional zq[M][E] = cb[idx] (the F.embedding at :186).
 * workspace: gsdd_nearest_code_workspace_bytes(K) bytes (the |e_k|^2 vector).  With it, E == 128 and K % 32 == 0 the distances
 * are a GEMM on the f32 matrix cores with the arg-min as its epilogue (never materialised); otherwise, or with workspace ==
 * NULL, the register-tiled vector kernel runs (any E % 4 == 0, any K). */
int64_t gsdd_nearest_code_workspace_bytes(int K);
int gsdd_nearest_code(const float* z, int64_t M, int E, const float* cb, int K,
                      int64_t* idx, float* zq, void* workspace, int64_t workspace_bytes, void* stream);

/* 3-D max / mean pooling on channels-last rows, for the FVD evaluator's I3D feature extractor (MaxPool3dSamePadding and the
 * final AvgPool3d: src/models/motionencoder/pytorch_i3d.py:7-34, :296).  in: N x Di x Hi x Wi positions of `in_pitch` floats, C of
 * them pooled; window kernel[3] / stride[3] starting at o * stride - pad_front; out grid Do x Ho x Wo with row pitch out_pitch
 * (a channel slice of a wider concatenated row is addressed by offsetting `out`).  mode 0: maximum, positions outside the input
 * count as 0 (the reference zero-pads, then pools unpadded); mode 1: mean over the full window. */
int gsdd_pool3d(const float* in, int N, int Di, int Hi, int Wi, int C, int in_pitch, const int* kernel, const int* stride,
                const int* pad_front, int Do, int Ho, int Wo, int mode, float* out, int out_pitch, void* stream);

/* ------------------------------------------------------------------ VQ-VAE train-mode forward pieces
 * nn.BatchNorm3d in train mode on rows x[M][C] (videogpt_vq_vae.py:125-133, 242-247): batch mean / biased variance ->
 * folded scale = w/sqrt(var+eps), shift = b - mean*scale (consumed by gsdd_gemm's pro_scale/pro_shift); running stats
 * (may be NULL) updated with momentum and the unbiased variance like torch. */
int64_t gsdd_bn_train_workspace_bytes(int64_t M, int C);
int gsdd_bn_train(const float* x, int64_t M, int C, const float* weight, const float* bias, float eps, float momentum,
                  float* running_mean, float* running_var, float* scale, float* shift, float* mean_rstd /* optional [C][2] */,
                  void* workspace, int64_t workspace_bytes, void* stream);

/* Codebook EMA (Codebook.forward, videogpt_vq_vae.py:193-214) in two phases so the caller can all-reduce between them
 * (:196-198): phase 0: n_total[K], encode_sum[K][E] from (z rows, idx); phase 1: N, z_avg EMA with `decay`, Laplace-
 * smoothed embeddings, dead codes (N < 1) restarted from z[perm[k]] (:205-214).  scalars[0] = sum(N), scalars[1] =
 * exp(-H(n_total / M)) (the perplexity when n_total is this rank's own count over M latents; otherwise use
 * gsdd_code_perplexity on the local counts). */
int gsdd_codebook_ema(const float* z, const int64_t* idx, int64_t M, int E, int K, float decay, const int64_t* perm,
                      float* N, float* z_avg, float* embeddings, float* n_total, float* encode_sum, float* scalars,
                      int phase, void* stream);

/* out[0] = exp(-sum_k p_k log(p_k + 1e-10)), p_k = n_local[k] / M: the codebook perplexity of THIS rank's M latents
 * (videogpt_vq_vae.py:218-219 take the mean of the local one-hot, before any all-reduce and before _tile). */
int gsdd_code_perplexity(const float* n_local, int K, int64_t M, float* out, void* stream);

/* out[0] = scale * mean((a-b)^2), deterministic fp64 two-stage reduction (F.mse_loss at videogpt_vq_vae.py:64, :190);
 * workspace >= 8 KiB. */
int gsdd_mse(const float* a, const float* b, int64_t n, float scale, float* out, void* workspace, int64_t workspace_bytes,
             void* stream);

/* ------------------------------------------------------------------ VQ-VAE training step: backward building blocks
 * (autograd of videogpt_vq_vae.py:102-138, 228-332; data gradients of (transposed) convolutions are gsdd_gemm calls with
 * transposed weights and mirrored tap tables) */
/* dW[tap][n][c] += sum_m dY[orow(m)][n] * pro(in[src(m,tap)][c]); geometry / taps / BN+ReLU prologue / dY row addressing
 * (the output-addressing fields) taken from the forward descriptor; Cout and dy_pitch multiples of 4. */
int gsdd_conv_wgrad(const gsdd_gemm_desc* d, const float* dY, int dy_pitch, float* dW, void* stream);
/* a = relu(BatchNorm_train(x)) backward: given da -> dx (= dx_in + ...), dgamma += , dbeta += */
int64_t gsdd_bn_relu_bwd_workspace_bytes(int64_t M, int C);
int gsdd_bn_relu_bwd(const float* da, const float* x, int64_t M, int C, const float* mean_rstd, const float* gamma,
                     const float* beta, const float* dx_in, float* dx, float* dgamma, float* dbeta, void* workspace,
                     int64_t workspace_bytes, void* stream);
/* dpre = dout * [out > 0] */
int gsdd_relu_mask(const float* dout, const float* out, float* dpre, int64_t n, void* stream);
/* out = (a ? a : 0) + alpha * (b - c) */
int gsdd_lincomb(const float* a, const float* b, const float* c, float alpha, float* out, int64_t n, void* stream);
/* backward of gsdd_axial_attention: datt rows [M][3C] -> dqkv rows [M][9C] */
int gsdd_axial_attention_bwd(const float* qkv, const float* datt, int N, int T, int H, int W, int C, int n_head, float* dqkv,
                             int variant /* GSDD_AXIAL_* */, void* stream);

/* ------------------------------------------------------------------ D3PM denoiser pieces
 * x[b][l][:] = emb[tok[b][l]] + pos[l]   (DalleMaskImageEmbedding.forward, dalle_mask_image_embedding.py:59-79;
 * pos = height_emb[l/W]+width_emb[l%W] precomputed once). rep: x is written for `rep` stacked copies. */
int gsdd_d3pm_embed(const int64_t* tok, int B, int L, int D, const float* emb, int n_embed,
                    const float* pos, int rep, float* x, void* stream);

/* AdaLayerNorm modulation table: out[t][0:D] = 1+scale, out[t][D:2D] = shift with
 * [scale|shift] = Linear(SiLU(emb[t]))  (AdaLayerNorm, transformer_utils.py:138-159). */
int gsdd_adaln_table(const float* emb, int T, int D, const float* lin_w, const float* lin_b,
                     float* out, void* stream);

/* y[r][:] = W x[r] + b for a handful of rows (cross-attention value/proj of the condition token,
 * transformer_utils.py:95-113 with T_E == 1: softmax over one key == 1). */
int gsdd_small_linear(const float* x, int R, int Cin, const float* w, const float* b, int Cout,
                      float* y, void* stream);

/* Self-attention for head dim 4: q,k,v head-major [H][M][4] (M = B*L rows), out rows [M][H*4].
 * softmax(q k^T / sqrt(4)) v, scores never leave registers.
 * Replaces FullAttention.forward: transformer_utils.py:46-62 (head-mean att is dropped: unused). */
/* workspace: gsdd_d3pm_attention_workspace_bytes(B,L,H) bytes of scratch for the matrix-pipe kernel: the pre-split K image
 * (32 B per key and head), the V image (32 B) and, per (head, 32-key pair-tile), the sum of the tile's keys (float4) and one f32
 * bounding the tile's largest ||k||.  The norms let the kernel prove from ||q|| ||k|| alone that a tile holds no probability above
 * 2^-8 of its row sum (then only the f16 hi half of P is used for it); the sums give it the mean key, hence a lower bound of every
 * final row sum before a key has been seen.  NULL selects the workspace-free kernel (exact-f32 P.V on
 * v_mfma_f32_4x4x1). */
int64_t gsdd_d3pm_attention_workspace_bytes(int B, int L, int H);
/* k = v = NULL: the workspace already holds the images, key sums and norms (written by gsdd_d3pm_layer through kv_img).
 * redo_events: optional device counter (caller-owned, caller-zeroed) to which the kernel adds one per (wave, chunk) it had to
 * redo with a larger exponent offset after an f16 overflow -- the kernel's only other data-dependent cost, 0 for near-uniform
 * attention rows.  The library keeps no counter of its own (no hidden global state). */
/* mode: how the softmax probabilities P enter the P.V product on the f16 matrix pipe (errors and times: DESIGN.md section 4).
 *   GSDD_ATTN_AUTO   GSDD_ATTN_A8 for L >= 2048, GSDD_ATTN_P22 below (short rows have too few keys to average the skipped halves out)
 *   GSDD_ATTN_P22    f16 hi + lo (22 bits) in every tile: the most exact variant (output within 2e-6 of fp64)
 *   GSDD_ATTN_P11    f16 hi only (11 bits) in every tile: the fastest; relative error of a probability <= 2^-12, output error up to
 *                    ~1.4e-4 |v - o| / sqrt(effective keys per row) -- within the path's 1e-4 logits contract on every pinned case
 *                    (tests/test_gpu_parity.py, tests/test_gpu_fullsize.py), but not within this kernel's own 2e-5 bar on peaked rows
 *   GSDD_ATTN_A8     adaptive: lo half only in (16-query, 32-key) tiles that can hold a probability above 2^-8 of the row sum
 *   GSDD_ATTN_A12    the same with threshold 2^-12
 *   GSDD_ATTN_F32PV  the workspace-free kernel (exact-f32 P.V) even when a workspace is given (k, v must be given)
 *   GSDD_ATTN_KC256  development variant: 256-key chunks, hi + lo everywhere */
#define GSDD_ATTN_AUTO 0
#define GSDD_ATTN_P22 1
#define GSDD_ATTN_P11 2
#define GSDD_ATTN_A8 3
#define GSDD_ATTN_A12 4
#define GSDD_ATTN_F32PV 5
#define GSDD_ATTN_KC256 6
int gsdd_d3pm_attention(const float* q, const float* k, const float* v, int B, int L, int H,
                        float* out, void* workspace, int64_t workspace_bytes, uint64_t* redo_events, int mode, void* stream);

/* Fused post-attention half of a denoiser block (n_embd 64, hidden 256), rows updated in place:
 *   x += proj(y)+b_proj+cvec[b];  x += W2 GELU2(W1 LN2(x)+b1)+b2;  [qkv_next = Wqkv AdaLN_next(x,t)+b_qkv, head-major]
 * Replaces Block.forward's tail (transformer_utils.py:268-282: attn1 proj + residual, T_E==1 cross-attention
 * vector, ln2, mlp :258-263, residual) and the next block's AdaLayerNorm (:150-159) + query/key/value (:48-50).
 * qkv == NULL skips the next-block stage (last block). */
typedef struct {
    const float* y;             /* [M][64] attention output                         */
    float* x;                   /* [M][64] residual stream, in/out                  */
    int64_t M;
    int L;                      /* rows per batch element                           */
    int n_embd, hidden;         /* must be 64, 256                                  */
    const float* cvec;          /* optional [M/L][64] per-batch vector              */
    const float* wproj; const float* bproj;
    const float* ln2_g; const float* ln2_b;
    const float* w1; const float* b1;     /* [256][64], [256]                       */
    const float* w2; const float* b2;     /* [64][256], [64]                        */
    const float* ada;           /* next block: [T][128] (1+scale | shift) table     */
    const int64_t* t2;          /* device int64[M/L] timesteps                      */
    const float* wqkv; const float* bqkv; /* [192][64], [192]                       */
    float* qkv;                 /* [48][M][4] or NULL                               */
    const void* w2_x3;          /* gsdd_d3pm_layer_pack images (bf16x3 MFMA fragments) of this block's w2 + wproj and of the */
    const void* wqkv_x3;        /* next block's wqkv: ready-made matrix operands streamed through LDS.  One of the two image */
                                /* sets (these or layer_h2 / wqkv_h2) must be given                                           */
    int64_t kv_img_bytes;       /* size of kv_img in bytes: at least gsdd_d3pm_attention_workspace_bytes(M / L, L, 16)          */
    void* kv_img;               /* optional (L % 32 == 0): the attention workspace of the next                                  */
                                /* block.  k and v are then written there as the matrix-pipe kernel's pre-split images instead */
                                /* of f32 rows of qkv, and gsdd_d3pm_attention is called with k = v = NULL (no prep pass)      */
    const void* layer_h2;       /* optional: gsdd_d3pm_layer_pack_h2 images (f16 hi + lo MFMA fragments) of this block's w1, w2 and */
    const void* wqkv_h2;        /* wproj, and of the next block's wqkv.  Preferred over the bf16x3 images when given: all three    */
                                /* weight matrices of the block are then LDS-resident and every product is 3 matrix instructions   */
                                /* instead of 6; as accurate as an f32 GEMM with f32 accumulation (22-bit operands, exact products) */
    int variant;                /* GSDD_LAYER_AUTO (the f16 hi + lo kernel when its images are given, else the bf16x3 one), or one of */
                                /* them explicitly: GSDD_LAYER_H2 / GSDD_LAYER_X3P (an error if that kernel's images are missing)    */
    int* range_flag;            /* optional device int (caller-zeroed), used by the f16 hi + lo kernel only: its operands are 16 a as  */
                                /* f16, so an activation |a| >= 4094 overflows to inf.  The kernel sets *range_flag = 1 when a row of  */
                                /* x it writes is not finite (inf / NaN propagate there); the caller then reruns with the bf16x3       */
                                /* images, which have f32's range (d3pm.py does: checked once per sample(), outside graph capture)    */
} gsdd_layer_desc;
#define GSDD_LAYER_AUTO 0
#define GSDD_LAYER_X3P 2
#define GSDD_LAYER_H2 3
int gsdd_d3pm_layer(const gsdd_layer_desc* d, void* stream);
/* Pre-split w2 [64][256] + wproj [64][64] (-> layer_x3, GSDD_LAYER_X3_BYTES) and wqkv [192][64] (-> wqkv_x3,
 * GSDD_LAYER_WQKV_X3_BYTES; both may be NULL) into the fragment images gsdd_d3pm_layer consumes; valid until the weights change. */
#define GSDD_LAYER_X3_BYTES (40 * 3 * 1024)
#define GSDD_LAYER_WQKV_X3_BYTES (24 * 3 * 1024)
int gsdd_d3pm_layer_pack(const float* w2, const float* wproj, const float* wqkv, void* layer_x3, void* wqkv_x3, void* stream);
/* The f16 hi + lo images: w1 [256][64] + w2 [64][256] + wproj [64][64] -> layer_h2 (GSDD_LAYER_H2_BYTES), wqkv [192][64] -> wqkv_h2
 * (GSDD_LAYER_WQKV_H2_BYTES); either image may be NULL (then its sources may be too).  Weights must be below 255 in magnitude. */
#define GSDD_LAYER_H2_BYTES (72 * 2 * 1024)
#define GSDD_LAYER_WQKV_H2_BYTES (24 * 2 * 1024)
int gsdd_d3pm_layer_pack_h2(const float* w1, const float* w2, const float* wproj, const float* wqkv, void* layer_h2, void* wqkv_h2,
                            void* stream);

/* Row GEMMs of a block in the training step: out[m][:] = x[m] W'^T + bias [+ bvec[m / rows_per_batch]] [+ residual[m]], n_in -> n_out
 * one of 64 -> 64 / 128 / 192 / 256 or 128 / 192 / 256 -> 64, with W' (n_out x n_in) given as a bf16x3 fragment image of
 * GSDD_ROWS_LINEAR_IMAGE_BYTES(n_out, n_in) bytes.  Replaces the nn.Linear calls of Block / FullAttention (transformer_utils.py:48-50,
 * 60, 258-263) and their data gradients in D3PMTrainer; head_major: out is [n_out / 4][M][4] (q | k | v).
 * gsdd_rows_linear_pack_many fills n_desc images in one launch from a DEVICE array of
 *   struct { const float* w; int n_out, n_in, ld, transpose; void* image; }   (transpose: W'[n][k] = w[k * ld + n], the data gradient's
 * operand; else W'[n][k] = w[n * ld + k]); max_out / max_in: the largest n_out / n_in among them. */
#define GSDD_ROWS_LINEAR_IMAGE_BYTES(n_out, n_in) ((n_out) / 64 * ((n_in) / 64) * 8 * 3 * 1024)
int gsdd_rows_linear_pack_many(const void* descs_dev, int n_desc, int max_out, int max_in, void* stream);
int gsdd_rows_linear(const float* x, int64_t M, int n_in, const void* image, int n_out, const float* bias, const float* bvec,
                     int rows_per_batch, const float* residual, float* out, int head_major, void* stream);

/* to_logits: out[m][:] = W LayerNorm(x[m]) + bias  (nn.LayerNorm + nn.Linear, transformer_utils.py:353-356, 442);
 * x: [M][64], w: [K][64], out: [M][K] rows (the reference's (B,K,L) is a transposed view of this). */
int gsdd_d3pm_logits(const float* x, int64_t M, int n_embd, const float* ln_g, const float* ln_b, const float* w,
                     const float* bias, int K, float* out, void* stream);

/* General cross-attention (T_E condition tokens), head dim 4; q head-major [H][M][4],
 * kc/vc rows [B*Te][H*4]; out rows [M][H*4].  transformer_utils.py:95-113. */
int gsdd_d3pm_cross_attention(const float* q, const float* kc, const float* vc, int B, int L, int Te,
                              int H, float* out, void* stream);

/* One fused reverse-diffusion step on int64 tokens:
 *   predict_start (fp log_softmax, clamp) x2 -> cf guidance mix -> q_posterior -> Gumbel arg-max.
 * Replaces diffusion_transformer.py:220-249 (predict_start/cf_predict_start tails), :251-283
 * (q_posterior), :354-359 (log_sample_categorical) and the log-one-hot round trip (:44-54).
 * logits_c / logits_u: [B*L][K] rows (logits_u may be NULL -> no guidance: predict_start only).
 * sched: 8 device arrays in the order log_at, log_bt, log_ct, log_1_min_ct (T each),
 *        log_cumprod_at, log_cumprod_bt, log_cumprod_ct, log_1_min_cumprod_ct (T+1 each).
 * t_dev: device int64[B] timesteps; stream_dev: device int64[1] Philox stream id (both read on
 * device so a captured graph can be replayed).  post_dbg: optional [B][K+1][L] log-probabilities.
 * x0_dbg: optional [B][K+1][L] guided log p(x0|xt) (cf_predict_start output). */
typedef struct {
    const float* logits_c;
    const float* logits_u;
    const int64_t* tok_in;
    int64_t* tok_out;
    int B, L, K, T;
    float guidance;
    const float* sched[8];
    const int64_t* t_dev;
    uint64_t seed;
    const int64_t* stream_dev;
    int64_t row0;               /* global row offset of this shard (multi-GPU batch split)   */
    float* post_dbg;
    float* x0_dbg;
    int occupancy;              /* 0 = default; 2 / 3: waves per SIMD the K = 4096 kernel's register budget is sized for       */
                                /* (development switch: the default is 2, no scratch)                                          */
} gsdd_step_desc;
int gsdd_d3pm_step(const gsdd_step_desc* d, void* stream);

/* q_sample for training: tok_out = Gumbel-argmax(q_pred(onehot(x0), t)), diffusion_transformer.py:361-366 */
int gsdd_d3pm_q_sample(const int64_t* x0, int64_t* xt, int B, int L, int K, int T,
                       const float* const* sched, const int64_t* t_dev, uint64_t seed,
                       const int64_t* stream_dev, int64_t row0, void* stream);

/* Training objective, forward value: _train_loss + the loss tail of forward (diffusion_transformer.py:391-457, :548)
 * from the denoiser logits of x_t.  Per-position scratch (kl, nll, aux: float[B*L]; x0_recon, xt1_recon: int64[B*L]),
 * per-sample results per_sample[B][4] = (kl_loss, vb_loss, acc rate, keep rate), loss[0] = sum(vb)/(B*L);
 * Lt_history / Lt_count (float[T]) are updated in place (:432-436).  probs: optional [B][K+1][L] = exp(log_model_prob). */
typedef struct {
    const float* logits;        /* [B*L][K] denoiser output for x_t                         */
    const int64_t* x0;          /* (B,L) clean tokens                                       */
    const int64_t* xt;          /* (B,L) noised tokens (gsdd_d3pm_q_sample)                 */
    const int64_t* t_dev;       /* int64[B]                                                 */
    const float* pt;            /* float[B] sampling probability of t                       */
    int B, L, K, T;
    const float* sched[8];
    float mask_weight[2];
    float aux_weight; int adaptive_aux;
    float* kl; float* nll; float* aux; int64_t* x0_recon; int64_t* xt1_recon;
    float* Lt_history; float* Lt_count;
    float* loss; float* per_sample; float* probs;
} gsdd_train_desc;
int gsdd_d3pm_train_loss(const gsdd_train_desc* d, void* stream);

/* dlogits = d loss / d logits of the objective above (backward of predict_start -> q_posterior -> KL / NLL / aux KL). */
int gsdd_d3pm_train_loss_bwd(const gsdd_train_desc* d, float* dlogits, void* stream);
/* Both of the above in ONE pass over the logits (the training step's path; d->probs must be NULL): every output of
 * gsdd_d3pm_train_loss, bit-identical, plus dlogits. */
int gsdd_d3pm_train_loss_grad(const gsdd_train_desc* d, float* dlogits, void* stream);

/* ------------------------------------------------------------------ D3PM training step: backward building blocks
 * (autograd of transformer_utils.py:24-62, 138-159, 258-282, 353-356 and dalle_mask_image_embedding.py:59-79) */
/* out = gelu2(a) (backward=0) or out = du * gelu2'(a) (backward=1), n % 4 == 0 */
int gsdd_gelu2(const float* a, const float* du, float* out, int64_t n, int backward, void* stream);
/* LayerNorm forward of the training step over rows of 64 (nn.LayerNorm / AdaLayerNorm, transformer_utils.py:138-159): stats[row] =
 * (mean, rstd) and y = (x - mean) * rstd * gamma + beta in one pass; gamma/beta = base + sel[row / rows_per_batch] * gstride
 * (sel NULL: one shared pair).  The sampler never materialises y (it is a GEMM prologue there); the training step keeps it as the
 * weight-gradient operand. */
int gsdd_ln_fwd(const float* x, int64_t M, int C, float eps, const float* gamma, const float* beta, const int64_t* sel, int gstride,
                int rows_per_batch, float* stats, float* y, void* stream);
/* LayerNorm backward over rows of 64: dx_out = dx_in + LN'(dh); dgamma/dbeta accumulated (+=) per batch element
 * (acc_by_batch, AdaLN table rows) or globally (affine LN).  gamma = gamma_base + sel[b]*gstride. */
int gsdd_ln_bwd(const float* dh, const float* x, const float* stats, const float* gamma, const int64_t* sel, int gstride,
                int rows_per_batch, int64_t M, int C, const float* dx_in, float* dx_out, float* dgamma, float* dbeta,
                int gacc_stride, int acc_by_batch, void* stream);
/* dW[N][K] += dY^T X (contraction over the M rows), db[N] += column sums of dY (optional) */
int gsdd_wgrad(const float* dY, int ldy, const float* X, int ldx, int64_t M, int N, int K, float* dW, float* db, void* stream);
/* out[n] += sum_m Y[m][n] */
int gsdd_colsum(const float* Y, int ld, int64_t M, int N, float* out, void* stream);
/* out[b][c] = sum_l Y[b*L+l][c] */
int gsdd_batch_rowsum(const float* Y, int B, int L, int C, float* out, void* stream);
/* head-dim-4 self-attention for training: forward that also returns the log2-domain log-sum-exp per (head,row), and the
 * backward (dq|dk|dv rows [M][3*H*4]); scratch: float[H*M].  With a workspace of gsdd_d3pm_attention_workspace_bytes() and
 * L % 32 == 0 the forward runs on the matrix-pipe kernel of gsdd_d3pm_attention; workspace may be NULL (VALU kernel). */
/* mode: GSDD_ATTN_AUTO (= GSDD_ATTN_A8 for L >= 2048, else GSDD_ATTN_P22), GSDD_ATTN_P22 or GSDD_ATTN_A8; anything else is an error. */
int gsdd_d3pm_attention_train(const float* q, const float* k, const float* v, int B, int L, int H, float* out, float* lse,
                              void* workspace, int64_t workspace_bytes, int mode, void* stream);
/* With a workspace of gsdd_d3pm_attention_bwd_workspace_bytes() and L % 32 == 0 the backward runs on the bf16 matrix pipe
 * (pre-split operand images, dQ kernel + dK/dV kernel); otherwise on the VALU kernels, which need `scratch`. */
int64_t gsdd_d3pm_attention_bwd_workspace_bytes(int B, int L, int H);
/* variant: GSDD_ATTN_BWD_AUTO = the fused kernel (one pass over the scores for dQ, dK and dV); GSDD_ATTN_BWD_VALU = the vector
 * kernels whatever the shape; the rest are development variants kept as cross-checks of each other (tests/test_gpu_training.py):
 * _SPLIT = dQ kernel + dK/dV kernel, _FQC64 / _FQC128 = queries per LDS chunk of the fused kernel (default 96), _NW8 = eight waves
 * per workgroup, _DBG1 / _DBG2 = the fused kernel stopping after its dQ / dK stage. */
#define GSDD_ATTN_BWD_AUTO 0
#define GSDD_ATTN_BWD_VALU 1
#define GSDD_ATTN_BWD_SPLIT 2
#define GSDD_ATTN_BWD_FQC64 3
#define GSDD_ATTN_BWD_FQC128 4
#define GSDD_ATTN_BWD_NW8 5
#define GSDD_ATTN_BWD_DBG1 6
#define GSDD_ATTN_BWD_DBG2 7
#define GSDD_ATTN_BWD_DEV_LAST 7
int gsdd_d3pm_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* dO, const float* lse,
                            int B, int L, int H, float* dqkv, float* scratch, void* workspace, int64_t workspace_bytes,
                            int variant, void* stream);
/* demb[tok] += dx, dpos[l] += dx */
int gsdd_d3pm_embed_bwd(const float* dx, const int64_t* tok, int B, int L, int D, int n_embed, float* demb, float* dpos,
                        void* stream);
/* y = W x + b over R rows: dx (optional) = dy W ; dW += dy^T x ; db (optional) += sum dy */
int gsdd_small_linear_bwd(const float* dy, const float* x, const float* w, int R, int Cin, int Cout, float* dx, float* dw,
                          float* db, void* stream);
/* backward of gsdd_adaln_table restricted to the rows t[b]: dtab [B][2D] -> demb (+=, rows t[b]), dW (+=), db (+=) */
int gsdd_adaln_bwd(const float* dtab, const int64_t* t, int B, int D, const float* emb, const float* w, float* demb, float* dw,
                   float* db, void* stream);
/* torch.optim.Adam update (no weight decay), step >= 1 */
int gsdd_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int step,
              void* stream);
/* The same update over many tensors in one launch: table[b] = {p, g, m, v, n} (device pointers as int64, n <= 4096 elements) for
 * block b; the caller chops every parameter into such chunks. */
int gsdd_adam_multi(const int64_t* table, int n_blocks, float lr, float beta1, float beta2, float eps, int step, void* stream);
/* ... with the step count (>= 1) read from device memory at execution time: the form a captured, replayed training step uses (the
 * caller sets *step_dev before each replay, or advances it with gsdd_advance inside the graph). */
int gsdd_adam_multi_dev(const int64_t* table, int n_blocks, float lr, float beta1, float beta2, float eps, const int64_t* step_dev,
                        void* stream);

/* t[b] += dt ; stream[0] += ds   (device-side loop counters for the captured step graph) */
int gsdd_advance(int64_t* t_dev, int B, int64_t dt, int64_t* stream_dev, int64_t ds, void* stream);

/* uniform Philox floats, layout of oracle/philox.py uniform_rows (test hook) */
int gsdd_philox_uniform(uint64_t seed, int64_t stream_id, int64_t row0, int64_t n_rows, int n_cols,
                        float* out, void* stream);

/* ------------------------------------------------------------------ hipGraph capture of a step
 * (the 100-iteration loop at diffusion_transformer.py:621-626 becomes 100 replays). */
int gsdd_graph_begin(void* stream);
int gsdd_graph_end(void* stream, void** graph_exec_out);
int gsdd_graph_launch(void* graph_exec, void* stream);
int gsdd_graph_destroy(void* graph_exec);

/* ------------------------------------------------------------------ clip preprocessing (next row)
 * replaces `preprocess` (src/datamodules/datasets/ucf101_dataset.py:105-140): uint8 frames video[N][T][H][W][3] ->
 * out[N][3][t_out][R][R] float32 = centre crop (h_start, w_start) of the bilinear resize (align_corners false) to th x tw of
 * (x/255 - ImageNet mean)/std; the first t_out frames of each clip (temporal crop, :116-118). */
int gsdd_preprocess_clip(const uint8_t* video, int N, int T, int H, int W, int t_out, int th, int tw, int h_start, int w_start,
                         int R, float* out, void* stream);

/* HIP-event timing on a given stream (bench.py measures the stream the kernels run on) */
int gsdd_event_create(void** ev);
int gsdd_event_record(void* ev, void* stream);
int gsdd_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);   /* synchronises ev_stop */
int gsdd_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* GSDD_H */
