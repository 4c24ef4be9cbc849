// Fused "post-attention" half of a D3PM denoiser block, n_embd = 64, for one 32-row group per wave:
//
//   x1 = x + proj(y) + b_proj + c_cross[b]                     (attn1 residual + T_E==1 cross-attention vector)
//   x2 = x1 + W2 GELU2(W1 LN2(x1) + b1) + b2                   (MLP residual)                 -> x (in place)
//   qkv_next = Wqkv AdaLN_next(x2, t) + b_qkv                   (next block's q|k|v, head-major)   [optional]
//
// Replaces Block.forward's tail (transformer_utils.py:268-282), AdaLayerNorm (:150-159), nn.LayerNorm, the MLP
// (:258-263) and the next block's query/key/value linears (:48-50).
//
// Layout trick: every GEMM is computed TRANSPOSED on v_mfma_f32_32x32x2_f32 (A = weights, B = activations), so an
// accumulator tile has the ROW m on the lane (col = lane&31) and 16 of a tile's 32 features in its registers
// (feature = 8*(r>>2) + 4*(lane>>5) + (r&3)).  The next GEMM contracts over exactly those features, so register r
// of the accumulator IS the B operand of MFMA step r (the weight fragment is read in the same permuted k order):
// activations never leave registers between the four chained GEMMs; LayerNorm needs one cross-half shuffle.
// W1/W2 (128 KiB) live in LDS for the whole persistent workgroup, W_proj / W_qkv stream from L2.
#include "common.hpp"

namespace gsdd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int D = 64;          // n_embd
constexpr int HID = 256;       // mlp hidden
constexpr int W1P = 68;        // LDS pitch of W1 rows (k = 64): == 4 (mod 64) -> conflict-free ds_read_b128

struct LayerArgs {
    const float* y;            // [M][64] attention output
    float* x;                  // [M][64] residual stream (in/out)
    int64_t M;
    int L;                     // rows per batch element
    const float* cvec;         // [B2][64] or null
    const float* wproj; const float* bproj;
    const float* ln2_g; const float* ln2_b;
    const float* w1; const float* b1;      // [256][64], [256]
    const float* w2; const float* b2;      // [64][256], [64]
    // next block (optional)
    const float* ada;          // [T][128] = (1+scale | shift)
    const int64_t* t2;         // [B2]
    const float* wqkv; const float* bqkv;  // [192][64], [192]
    float* qkv;                // [48][M][4]
    const uint4* w2_x3;        // optional fragment images (gsdd_d3pm_layer_pack)
    const uint4* wqkv_x3;
    const uint4* lay_h2;       // optional f16 hi + lo fragment images (gsdd_d3pm_layer_pack_h2)
    const uint4* wqkv_h2;
    uint4* kimg; uint4* vimg;  // optional: the next block's attention images (k, v go there instead of qkv rows)
    float* knorm;              // with them: per (head, 32-key pair-tile) bound of ||k|| (common.hpp::kv_image_knorm)
    float4* ksum;              // and the sum of the tile's keys (common.hpp::kv_image_ksum)
    int* range_flag;           // optional (f16 hi + lo kernel): set to 1 when an updated row of x is not finite, i.e. an activation
                               // left the f16 operand range somewhere upstream (inf / NaN reach x through the residual adds)
};

// GELU2 (transformer_utils.py:115-119): v * sigmoid(1.702 v) = v / (1 + 2^(-1.702 log2(e) v)).
// 128 evaluations per row and lane make this the kernel's largest VALU item, and VALU time is not hidden behind the
// MFMAs here, so it runs on the bare v_exp_f32 / v_rcp_f32 (1 ulp each) instead of expf and an IEEE division
// (4 instructions instead of ~25).  The single rounding of the exponent argument changes the result by at most
// |v| s (1 - s) |arg| 2^-24 ln 2 <= 1.5e-8 absolute (s = the sigmoid); the reciprocal adds one ulp.  No range guards
// are needed: 2^arg -> inf gives v * rcp(inf) = v * 0, 2^arg -> 0 gives v (a Newton step on the reciprocal would
// turn the inf case into NaN, hence none).
__device__ __forceinline__ float gelu2(float v) {
    return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -2.4554669595930157f));
}

// fragment helpers: lane (m = lane&31, h = lane>>5) owns features f(t,g,e) = 32t + 8g + 4h + e in reg 16t + 4g + e
__device__ __forceinline__ void load_frag(const float* row, int h, float (&r)[32]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(row + 32 * t + 8 * g + 4 * h);
            r[16 * t + 4 * g + 0] = v.x; r[16 * t + 4 * g + 1] = v.y; r[16 * t + 4 * g + 2] = v.z; r[16 * t + 4 * g + 3] = v.w;
        }
}

// acc[nt] += act (32 rows x 64 features in registers) x W[32 nt + li][k]^T on the exact-f32 matrix instruction, W rows at `w` with pitch
// `pitch` floats: acc[nt][r] = out[row (r&3)+8(r>>2)+4h][feature 32nt + li] — the feature is on the lane, so a store instruction writes
// 2 x 128 contiguous bytes (to_logits, whose result goes straight to HBM).
__device__ __forceinline__ void gemm64_rows(const float* w, int pitch, int li, int h, const float (&act)[32], f32x16 (&acc)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 a0 = *reinterpret_cast<const float4*>(w + (int64_t)li * pitch + 32 * t + 8 * g + 4 * h);
            float4 a1 = *reinterpret_cast<const float4*>(w + (int64_t)(32 + li) * pitch + 32 * t + 8 * g + 4 * h);
            const int r = 16 * t + 4 * g;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(act[r + 0], a0.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(act[r + 0], a1.x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(act[r + 1], a0.y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(act[r + 1], a1.y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(act[r + 2], a0.z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(act[r + 2], a1.z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(act[r + 3], a0.w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(act[r + 3], a1.w, acc[1], 0, 0, 0);
        }
}

__device__ __forceinline__ void zero2(f32x16 (&acc)[2]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
}

// LayerNorm over the 64 features of each row: own 32 registers + the partner half (lane ^ 32)
__device__ __forceinline__ void row_norm(const float (&v)[32], float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
    s += __shfl_xor(s, 32);
    mean = s * (1.f / 64.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) { const float dlt = v[i] - mean; q += dlt * dlt; }
    q += __shfl_xor(q, 32);
    rstd = 1.0f / sqrtf(q * (1.f / 64.f) + 1e-5f);
}

// parameter block of the fused layer kernels (floats)
constexpr int PAR_BPROJ = 0, PAR_G2 = 64, PAR_B2LN = 128, PAR_B1 = 192, PAR_B2 = 448, PAR_BQKV = 512, PAR_N = 704;


// ------------------------------------------------------------------------------------------------------------
// The same fused layer with every GEMM on the bf16 matrix pipe: each f32 operand element is split error-free into three
// bf16 pieces (x = x1 + x2 + x3, round-to-nearest pieces, exact residuals) and a k-step of 16 is the six significant cross
// products on v_mfma_f32_32x32x16_bf16 — the scheme of gemm.hip, results indistinguishable from the f32 kernel.  Why it pays
// here: on f32 operands the MFMA and the VALU share one datapath, so this kernel's 196 k MFMA cycles and 46 k VALU cycles per
// SIMD add up; the bf16 products take 6 x 32 cycles instead of 8 x 64 per 32 x 32 x 16 block and run beside the VALU, which now
// also does the splits (weights are split once per 32-row group as they are read from LDS, activations once per GEMM input).
// The accumulator layout of the 32x32x16 MFMA is that of the 32x32x2 one, so the register chaining between the GEMMs and all
// epilogues are unchanged: accumulator registers 8s .. 8s+7 of a 32-feature tile are the B fragment of k-step s, and the
// matching weight fragment is the two float4 runs [16s + 4h, +4) and [16s + 8 + 4h, +4) of the weight row.
typedef __bf16 lbf16x8 __attribute__((ext_vector_type(8)));
struct P3 { lbf16x8 p[3]; };
__device__ __forceinline__ P3 split8(const float (&x)[8]) {
    P3 o;
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = x[j];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o.p[i][j] = (__bf16)r[j];
        if (i < 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] -= (float)o.p[i][j];
        }
    }
    return o;
}
__device__ __forceinline__ void mma6(const P3& a, const P3& b, f32x16& acc) {   // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], b.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[0], acc, 0, 0, 0);
}
// the four k-steps (t, s) of a 64-feature activation held as act[16t + 8s + j]
__device__ __forceinline__ void split_act(const float (&act)[32], P3 (&b)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act[8 * q + j];
        b[q] = split8(v);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Third variant: the weights arrive as bf16x3 *fragment images*, so no weight is ever split inside the row loop.
// An image is a sequence of fragments; fragment f holds, for each of its three pieces p and each lane l, the 16 bytes
// (8 bf16) that lane feeds to v_mfma_f32_32x32x16_bf16 as its A operand: byte offset ((3 f + p) * 64 + l) * 16.
// Fragment (q, nt) of a 64-output x 64-input block W: lane (li = l & 31, h = l >> 5), element j ->
// W[32 nt + li][16 q + 8 (j >> 2) + 4 h + (j & 3)]  (the k order of an accumulator tile used as the B operand).
//   W1    [256][64]: f = (4 c + q) * 2 + nt, block c = rows 64 c ..       split into LDS by the workgroup at start (96 KB, resident)
//   layer image (gsdd_d3pm_layer_pack, global): W2 [64][256] as 4 column blocks of 8 fragments (f = 8 c + 2 q + nt), then
//                                               Wproj [64][64] (8 fragments, f = 2 q + nt; copied to LDS at start)
//   qkv image (global): Wqkv [192][64] as 3 row blocks of 8 fragments
// W2 and Wqkv fragments are read from L2 straight into registers, requested one k-step (W2) / two k-steps (Wqkv) before use.
// (A variant that passed them through a two-slot LDS ring shared by the eight waves, one barrier per 24 KB block, measured
// slower: 0.123 vs 0.109 ms — the barriers and the extra register pressure cost more than the L2 latency they hide.)
// What is left on the VALU per row group is the splits of the four GEMM inputs and the elementwise work.
constexpr int IMG_FRAG_U4 = 3 * 64;                          // uint4 per fragment
constexpr int IMG_BLOCK_U4 = 8 * IMG_FRAG_U4;                // a stage's block: 8 fragments = 1536 uint4 = 24 KB
constexpr int X3P_W1_U4 = 32 * IMG_FRAG_U4;
constexpr int X3P_PAR_OFF = (X3P_W1_U4 + IMG_BLOCK_U4) * 4;       // float offset of the parameter block (after W1 + Wproj images)
constexpr int X3P_SCR_OFF = X3P_PAR_OFF + PAR_N;
constexpr int X3P_LDS_FLOATS = X3P_SCR_OFF + 8 * 192;

__device__ __forceinline__ P3 split8_w(const float* row) {    // row -> W[..][16 q + 4 h]: two float4 runs, 8 floats apart
    const float4 lo = *reinterpret_cast<const float4*>(row), hi = *reinterpret_cast<const float4*>(row + 8);
    const float wv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return split8(wv);
}
__device__ __forceinline__ void store_frag(uint4* img, int f, int l, const P3& a) {
#pragma unroll
    for (int i = 0; i < 3; ++i) img[(3 * f + i) * 64 + l] = __builtin_bit_cast(uint4, a.p[i]);
}
__device__ __forceinline__ P3 load_frag3(const uint4* img, int f, int l) {
    P3 a;
#pragma unroll
    for (int i = 0; i < 3; ++i) a.p[i] = __builtin_bit_cast(lbf16x8, img[(3 * f + i) * 64 + l]);
    return a;
}

// one thread per (fragment, lane): 32 fragments of W2 + 8 of Wproj -> layer image, 24 of Wqkv -> qkv image
__global__ __launch_bounds__(256) void layer_pack_kernel(const float* w2, const float* wproj, const float* wqkv, uint4* lay_x3,
                                                         uint4* wqkv_x3) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    const int nL = 40 * 64, nQ = wqkv != nullptr ? 24 * 64 : 0;
    if (u >= nL + nQ) return;
    const bool is_q = u >= nL;
    const int v = is_q ? u - nL : u;
    const int f = v >> 6, l = v & 63, li = l & 31, h = l >> 5;
    const int nt = f & 1, q = (f >> 1) & 3, c = f >> 3;
    if (is_q) store_frag(wqkv_x3, f, l, split8_w(wqkv + (int64_t)(64 * c + 32 * nt + li) * D + 16 * q + 4 * h));
    else if (f < 32) store_frag(lay_x3, f, l, split8_w(w2 + (int64_t)(32 * nt + li) * HID + 64 * c + 16 * q + 4 * h));
    else store_frag(lay_x3, f, l, split8_w(wproj + (int64_t)(32 * nt + li) * D + 16 * q + 4 * h));
}

// acc[nt] += (fragments f0 .. f0+7 of an LDS image) x the four k-steps of bp; the next fragment's three ds_read_b128 are
// issued before the current fragment's six MFMAs
template <bool SWAP = false>   // SWAP: activations as the A operand (result: row in registers, feature on the lane)
__device__ __forceinline__ void gemm_lds_img(const uint4* img, int f0, int lane, const P3 (&bp)[4], f32x16 (&acc)[2]) {
    P3 cur = load_frag3(img, f0, lane);
#pragma unroll
    for (int ts = 0; ts < 8; ++ts) {
        const P3 nxt = load_frag3(img, f0 + (ts < 7 ? ts + 1 : 0), lane);
        if (SWAP) mma6(bp[ts >> 1], cur, acc[ts & 1]);
        else mma6(cur, bp[ts >> 1], acc[ts & 1]);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);     // the three reads first,
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);     // then the six MFMAs
        cur = nxt;
    }
}

// QKV_ONLY: only the next-block stage (AdaLN + q|k|v of x as it is): block 0 of the denoiser, whose input is the embedding
template <bool HAS_QKV, bool QKV_ONLY = false>
__global__ __launch_bounds__(512, 1) void d3pm_layer_x3p_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4* iw1 = reinterpret_cast<uint4*>(lds);              // 32 fragments
    uint4* iwp = iw1 + X3P_W1_U4;                            // 8 fragments
    float* par = lds + X3P_PAR_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    float* scr = lds + X3P_SCR_OFF + wave * 192;
    const uint4* img_w2 = a.w2_x3;                           // 32 fragments, then Wproj's 8
    const uint4* img_qkv = a.wqkv_x3;                        // 24 fragments

    // ---- W1 is split by the workgroup once; Wproj's fragments are copied from the layer image
    if (!QKV_ONLY) {
        for (int u = tid; u < 32 * 64; u += 512) {
            const int f = u >> 6, l = u & 63, fl = l & 31, fh = l >> 5;
            const int nt = f & 1, q = (f >> 1) & 3, c = f >> 3;
            store_frag(iw1, f, l, split8_w(a.w1 + (int64_t)(64 * c + 32 * nt + fl) * D + 16 * q + 4 * fh));
        }
        for (int u = tid; u < IMG_BLOCK_U4; u += 512) iwp[u] = img_w2[4 * IMG_BLOCK_U4 + u];
    } else {
        for (int u = tid; u < 3 * IMG_BLOCK_U4; u += 512) iw1[u] = img_qkv[u];     // the whole Wqkv image (72 KB) is LDS-resident
    }
    for (int i = tid; i < PAR_N; i += 512) {
        float v;
        if (QKV_ONLY) v = i >= PAR_BQKV ? a.bqkv[i - PAR_BQKV] : 0.f;
        else if (i < PAR_G2) v = a.bproj[i];
        else if (i < PAR_B2LN) v = a.ln2_g[i - PAR_G2];
        else if (i < PAR_B1) v = a.ln2_b[i - PAR_B2LN];
        else if (i < PAR_B2) v = a.b1[i - PAR_B1];
        else if (i < PAR_BQKV) v = a.b2[i - PAR_B2];
        else v = HAS_QKV ? a.bqkv[i - PAR_BQKV] : 0.f;
        par[i] = v;
    }
    __syncthreads();

    const int64_t ngroups = (a.M + 31) / 32;
    const bool batch_uniform = a.L % 32 == 0;
    for (int64_t grp = (int64_t)blockIdx.x * 8 + wave; grp < ngroups; grp += (int64_t)gridDim.x * 8) {
        const int64_t m = grp * 32 + li;
        const bool valid = m < a.M;
        const bool full = grp * 32 + 32 <= a.M;
        const int64_t mc = valid ? m : a.M - 1;
        const int b = (int)((uint32_t)mc / (uint32_t)a.L);
        if (batch_uniform) {
            const int bu = (int)((uint32_t)(grp * 32) / (uint32_t)a.L);
            if (lane < 16) {
                float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.cvec != nullptr) cv = *reinterpret_cast<const float4*>(a.cvec + (int64_t)bu * D + 4 * lane);
                *reinterpret_cast<float4*>(scr + 4 * lane) = cv;
            } else if (HAS_QKV && lane < 48) {
                const float* tab = a.ada + a.t2[bu] * (2 * D);
                *reinterpret_cast<float4*>(scr + 4 * lane) = *reinterpret_cast<const float4*>(tab + 4 * (lane - 16));
            }
        }

        float act[32], x1[32];
        f32x16 acc[2];
        P3 bp[4];
        float mean, rstd;
        if (QKV_ONLY) {
            load_frag(a.x + mc * D, h, x1);
        } else {
        // W2 fragments come from L2: tile 0 of a k-step is requested during the previous k-step, tile 1 at the start of its own
        // (deeper prefetch spills: registers, not L2 latency, are the scarce resource here); the first one is in flight during
        // proj / LN / W1
        P3 w2a = load_frag3(img_w2, 0, lane);
        // ---- x1 = x + proj(y) + b_proj + cvec[b]
        load_frag(a.y + mc * D, h, act);
        load_frag(a.x + mc * D, h, x1);
        split_act(act, bp);
        zero2(acc);
        gemm_lds_img(iwp, 0, lane, bp, acc);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = 32 * t + 8 * g + 4 * h;
                const float4 bpj = *reinterpret_cast<const float4*>(par + PAR_BPROJ + f);
                float4 cv;
                if (batch_uniform) cv = *reinterpret_cast<const float4*>(scr + f);
                else cv = a.cvec != nullptr ? *reinterpret_cast<const float4*>(a.cvec + (int64_t)b * D + f) : make_float4(0.f, 0.f, 0.f, 0.f);
                const int r = 4 * g;
                x1[16 * t + r + 0] += (acc[t][r + 0] + bpj.x) + cv.x;
                x1[16 * t + r + 1] += (acc[t][r + 1] + bpj.y) + cv.y;
                x1[16 * t + r + 2] += (acc[t][r + 2] + bpj.z) + cv.z;
                x1[16 * t + r + 3] += (acc[t][r + 3] + bpj.w) + cv.w;
            }
        // ---- h = LN2(x1) * gamma + beta
        row_norm(x1, mean, rstd);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = 32 * t + 8 * g + 4 * h;
                const float4 gm = *reinterpret_cast<const float4*>(par + PAR_G2 + f);
                const float4 bt = *reinterpret_cast<const float4*>(par + PAR_B2LN + f);
                const int r = 16 * t + 4 * g;
                act[r + 0] = (x1[r + 0] - mean) * rstd * gm.x + bt.x;
                act[r + 1] = (x1[r + 1] - mean) * rstd * gm.y + bt.y;
                act[r + 2] = (x1[r + 2] - mean) * rstd * gm.z + bt.z;
                act[r + 3] = (x1[r + 3] - mean) * rstd * gm.w + bt.w;
            }
        split_act(act, bp);
        // ---- MLP in 4 chunks of 64 hidden units
        f32x16 acc3[2];
        zero2(acc3);
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            zero2(acc);
            gemm_lds_img(iw1, 8 * c, lane, bp, acc);
            // GELU2 + split of k-step q + 1 is issued between the MFMAs of k-step q (both tiles), so the matrix pipe stays busy
            // while the VALU prepares its next operand
            auto make_ub = [&](int q) {
                const int t = q >> 1, s2 = q & 1;
                const float4 b0 = *reinterpret_cast<const float4*>(par + PAR_B1 + 64 * c + 32 * t + 16 * s2 + 4 * h);
                const float4 b1 = *reinterpret_cast<const float4*>(par + PAR_B1 + 64 * c + 32 * t + 16 * s2 + 8 + 4 * h);
                const float u[8] = {gelu2(acc[t][8 * s2 + 0] + b0.x), gelu2(acc[t][8 * s2 + 1] + b0.y), gelu2(acc[t][8 * s2 + 2] + b0.z),
                                    gelu2(acc[t][8 * s2 + 3] + b0.w), gelu2(acc[t][8 * s2 + 4] + b1.x), gelu2(acc[t][8 * s2 + 5] + b1.y),
                                    gelu2(acc[t][8 * s2 + 6] + b1.z), gelu2(acc[t][8 * s2 + 7] + b1.w)};
                return split8(u);
            };
            P3 ub = make_ub(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f0 = 8 * c + 2 * q;
                const P3 w2b = load_frag3(img_w2, f0 + 1, lane);
                const P3 na = load_frag3(img_w2, f0 + 2 < 32 ? f0 + 2 : 0, lane);
                P3 ubn = ub;
                if (q < 3) ubn = make_ub(q + 1);
                mma6(w2a, ub, acc3[0]);
                mma6(w2b, ub, acc3[1]);
                if (q < 3) {
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA ...
                        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);     // ... then a slice of the next operand's VALU work
                    }
                }
                ub = ubn;
                w2a = na;
            }
        }
        // ---- x2 = x1 + mlp + b2 -> x
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = 32 * t + 8 * g + 4 * h;
                const float4 bb = *reinterpret_cast<const float4*>(par + PAR_B2 + f);
                const int r = 4 * g;
                x1[16 * t + r + 0] += acc3[t][r + 0] + bb.x;
                x1[16 * t + r + 1] += acc3[t][r + 1] + bb.y;
                x1[16 * t + r + 2] += acc3[t][r + 2] + bb.z;
                x1[16 * t + r + 3] += acc3[t][r + 3] + bb.w;
            }
        if (full) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                *reinterpret_cast<float4*>(a.x + m * D + 32 * (q >> 2) + 8 * (q & 3) + 4 * h) =
                    make_float4(x1[4 * q + 0], x1[4 * q + 1], x1[4 * q + 2], x1[4 * q + 3]);
        } else if (valid) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                *reinterpret_cast<float4*>(a.x + m * D + 32 * (q >> 2) + 8 * (q & 3) + 4 * h) =
                    make_float4(x1[4 * q + 0], x1[4 * q + 1], x1[4 * q + 2], x1[4 * q + 3]);
        }
        }
        if (HAS_QKV) {
            // Wqkv fragments come from L2 two tile-steps ahead of their use; the first two are requested before the AdaLN arithmetic
            P3 wq0, wq1;
            if (!QKV_ONLY) { wq0 = load_frag3(img_qkv, 0, lane); wq1 = load_frag3(img_qkv, 1, lane); }
            row_norm(x1, mean, rstd);
            const float* tab = a.ada + a.t2[b] * (2 * D);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f = 32 * t + 8 * g + 4 * h;
                    float4 gm, bt;
                    if (batch_uniform) {
                        gm = *reinterpret_cast<const float4*>(scr + 64 + f);
                        bt = *reinterpret_cast<const float4*>(scr + 128 + f);
                    } else {
                        gm = *reinterpret_cast<const float4*>(tab + f);
                        bt = *reinterpret_cast<const float4*>(tab + D + f);
                    }
                    const int r = 16 * t + 4 * g;
                    act[r + 0] = (x1[r + 0] - mean) * rstd * gm.x + bt.x;
                    act[r + 1] = (x1[r + 1] - mean) * rstd * gm.y + bt.y;
                    act[r + 2] = (x1[r + 2] - mean) * rstd * gm.z + bt.z;
                    act[r + 3] = (x1[r + 3] - mean) * rstd * gm.w + bt.w;
                }
            split_act(act, bp);
#pragma unroll 1
            for (int c = 0; c < 3; ++c) {
                zero2(acc);
                // The V image wants eight *rows* of one column in a lane (below), so for it the product is taken the other way
                // round (activations as the A operand): acc[nt][r] = v[row 8 (r >> 2) + 4 h + (r & 3)][feature 32 nt + li].
                const bool v_img = c == 2 && a.vimg != nullptr;                         // wave-uniform
                if (QKV_ONLY) {
                    if (v_img) gemm_lds_img<true>(iw1, 8 * c, lane, bp, acc);
                    else gemm_lds_img<false>(iw1, 8 * c, lane, bp, acc);
                } else {
#pragma unroll
                    for (int ts = 0; ts < 8; ++ts) {
                        const int fn = 8 * c + ts + 2;             // (wraps to fragment 0 / 1 after the last block: harmless)
                        const P3 wq2 = load_frag3(img_qkv, fn < 24 ? fn : fn - 24, lane);
                        if (v_img) mma6(bp[ts >> 1], wq0, acc[ts & 1]);
                        else mma6(wq0, bp[ts >> 1], acc[ts & 1]);
                        wq0 = wq1; wq1 = wq2;
                    }
                }
                if (v_img) {
                    // lane (li, h): head hd = 8 nt + (li >> 2), dim d = li & 3.  Accumulator registers 4 g2 + e and 4 (g2 + 2) + e
                    // (e = 0..3) are rows 4 g + e of the group's two 16-key tiles for key group g = 2 g2 + h: the eight f16 of
                    // image entry (pair-tile, g, column) — three entries for the three pieces of v, and lane d also writes the
                    // constant column 12 + d (1, 0, 0, 0).  (The group is one 32-key pair-tile of every head: L % 32 == 0.)
                    const int d = li & 3;
                    const uint32_t cst = d == 0 ? 0x3C003C00u : 0u;                       // f16 1.0 | 1.0
                    const int64_t m0 = grp * 32;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int hd = 8 * nt + (li >> 2);
                        const float bv = par[PAR_BQKV + 128 + 32 * nt + li];
                        uint4* dst = a.vimg + (((int64_t)hd * a.M + m0) >> 5) * 64;
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2) {
                            _Float16 p1[8], p2[8], p3[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float v = acc[nt][4 * g2 + (e & 3) + 8 * (e >> 2)] + bv;
                                const _Float16 a1 = (_Float16)v;
                                const float r1 = (v - (float)a1) * 2048.f;
                                const _Float16 a2 = (_Float16)r1;
                                p1[e] = a1; p2[e] = a2; p3[e] = (_Float16)((r1 - (float)a2) * 2048.f);
                            }
                            uint4* e0 = dst + (2 * g2 + h) * 16;
                            e0[d] = __builtin_bit_cast(uint4, p1);
                            e0[4 + d] = __builtin_bit_cast(uint4, p2);
                            e0[8 + d] = __builtin_bit_cast(uint4, p3);
                            e0[12 + d] = make_uint4(cst, cst, cst, cst);
                        }
                    }
                    continue;
                }
                float4 o[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int t = q >> 2, r = 4 * (q & 3);
                    const float4 bb = *reinterpret_cast<const float4*>(par + PAR_BQKV + 64 * c + 32 * t + 8 * (q & 3) + 4 * h);
                    o[q] = make_float4(acc[t][r + 0] + bb.x, acc[t][r + 1] + bb.y, acc[t][r + 2] + bb.z, acc[t][r + 3] + bb.w);
                }
                if (c == 1 && a.kimg != nullptr) {
                    // k of head hd = 8 t + 2 g + h as the attention kernel's pre-split image (row = hd * M + m): two 16-byte stores
                    if (valid) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int hd = 8 * (q >> 2) + 2 * (q & 3) + h;
                            const float vals[4] = {o[q].x, o[q].y, o[q].z, o[q].w};
                            kv_image_store_k(vals, (int64_t)hd * a.M + m, a.kimg);
                        }
                    }
                    kv_image_store_knorm(o, h, li, grp, a.M, a.knorm, a.ksum);      // M % 32 == 0 here: the group is one whole pair-tile
                } else if (full) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int n = 64 * c + 32 * (q >> 2) + 8 * (q & 3) + 4 * h;
                        *reinterpret_cast<float4*>(a.qkv + ((int64_t)(n >> 2) * a.M + m) * 4) = o[q];
                    }
                } else if (valid) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int n = 64 * c + 32 * (q >> 2) + 8 * (q & 3) + 4 * h;
                        *reinterpret_cast<float4*>(a.qkv + ((int64_t)(n >> 2) * a.M + m) * 4) = o[q];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Row GEMMs of the training step (gsdd_rows_linear): out[m][:] = x[m] W^T + bias [+ bvec[batch]] [+ residual[m]] for the skinny
// layers of a block (64 -> 64 / 192 / 256 and 192 / 256 -> 64, forward and data gradients).  The generic GEMM writes such outputs
// at 1.2-1.5 TB/s (4-byte stores, weights re-split by every 128-row block: 51-62 us for 64 -> 192 at 65,536 rows); here the weights
// are a bf16x3 fragment image made once per optimiser step (gsdd_rows_linear_pack), LDS-resident for the launch, the row is on the
// lane as in the fused layer kernel, and a lane stores 16 bytes of one row at a time.  bf16x3, not f16 pairs: gradients need f32's range.
// Image: fragment f = ((nb * KC + kc) * 4 + q) * 2 + nt  (64-output block nb, 64-input chunk kc, k-step q, 32-row half nt), piece p,
// lane l -> uint4 (3 f + p) * 64 + l; element j of lane (li, h): W[64 nb + 32 nt + li][64 kc + 16 q + 8 (j >> 2) + 4 h + (j & 3)].
struct RowsLinArgs {
    const float* x; int64_t M; int K, N;
    const uint4* img;
    const float* bias;         // [N] or null
    const float* bvec;         // [M / rows_per_batch][N] or null
    int rows_per_batch;
    const float* residual;     // [M][N] or null
    float* out;                // [M][N], or head-major [N / 4][M][4]
    int head_major;
};
struct RowsPackDesc {          // one weight matrix -> one image; lives in device memory (gsdd_rows_linear_pack_many)
    const float* w;            // [rows][ld] row-major
    int n_out, n_in;           // the image's matrix W' is n_out x n_in
    int ld;
    int transpose;             // 0: W'[n][k] = w[n * ld + k];  1: W'[n][k] = w[k * ld + n]  (the data-gradient GEMM's operand)
    uint4* img;
};

__global__ __launch_bounds__(256) void rows_linear_pack_kernel(const RowsPackDesc* descs, int n_desc) {
    const RowsPackDesc d = descs[blockIdx.y];
    const int KC = d.n_in >> 6;
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= (d.n_out >> 6) * KC * 8 * 64) return;
    const int f = u >> 6, l = u & 63, li = l & 31, h = l >> 5;
    const int nt = f & 1, q = (f >> 1) & 3, blk = f >> 3;
    const int nb = blk / KC, kc = blk - nb * KC;
    const int n = 64 * nb + 32 * nt + li;
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 64 * kc + 16 * q + 8 * (j >> 2) + 4 * h + (j & 3);
        wv[j] = d.transpose ? d.w[(int64_t)k * d.ld + n] : d.w[(int64_t)n * d.ld + k];
    }
    store_frag(d.img, f, l, split8(wv));
}

template <int KC, int NB>      // K = 64 KC inputs, N = 64 NB outputs; KC == 1 or NB == 1
__global__ __launch_bounds__(512, 1) void rows_linear_kernel(const RowsLinArgs a) {
    static_assert(KC == 1 || NB == 1, "one of the two dimensions is a single 64-wide block");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4* img = reinterpret_cast<uint4*>(lds);
    constexpr int IMG_U4 = NB * KC * 8 * IMG_FRAG_U4;
    float* bias = lds + IMG_U4 * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    {   // image -> LDS, every load in flight before the first store
        constexpr int NCP = IMG_U4 / 512;
        static_assert(IMG_U4 % 512 == 0, "whole rounds of the workgroup");
        uint4 tmp[NCP];
#pragma unroll
        for (int i = 0; i < NCP; ++i) tmp[i] = a.img[tid + 512 * i];
#pragma unroll
        for (int i = 0; i < NCP; ++i) img[tid + 512 * i] = tmp[i];
    }
    if (tid < 64 * NB) bias[tid] = a.bias != nullptr ? a.bias[tid] : 0.f;
    __syncthreads();

    const int64_t ngroups = (a.M + 31) / 32;
    for (int64_t grp = (int64_t)blockIdx.x * 8 + wave; grp < ngroups; grp += (int64_t)gridDim.x * 8) {
        const int64_t m = grp * 32 + li;
        const bool valid = m < a.M;
        const int64_t mc = valid ? m : a.M - 1;
        const float* bv = a.bvec != nullptr ? a.bvec + (mc / a.rows_per_batch) * a.N : nullptr;
        float act[32];
        f32x16 acc[2];
        P3 bp[4];
        auto finish = [&](int nb) {       // acc -> + bias [+ bvec] [+ residual] -> out: 8 runs of 4 features of row m
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = 64 * nb + 32 * t + 8 * g + 4 * h, r = 4 * g;
                    const float4 bb = *reinterpret_cast<const float4*>(bias + n);
                    float4 o = make_float4(acc[t][r + 0] + bb.x, acc[t][r + 1] + bb.y, acc[t][r + 2] + bb.z, acc[t][r + 3] + bb.w);
                    if (bv != nullptr) {
                        const float4 c = *reinterpret_cast<const float4*>(bv + n);
                        o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
                    }
                    if (a.residual != nullptr) {
                        const float4 c = *reinterpret_cast<const float4*>(a.residual + mc * a.N + n);
                        o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
                    }
                    if (valid) {
                        if (a.head_major) *reinterpret_cast<float4*>(a.out + ((int64_t)(n >> 2) * a.M + m) * 4) = o;
                        else *reinterpret_cast<float4*>(a.out + m * a.N + n) = o;
                    }
                }
        };
        if (KC == 1) {
            load_frag(a.x + mc * a.K, h, act);
            split_act(act, bp);
#pragma unroll 1
            for (int nb = 0; nb < NB; ++nb) {
                zero2(acc);
                gemm_lds_img(img, 8 * nb, lane, bp, acc);
                finish(nb);
            }
        } else {
            zero2(acc);
#pragma unroll 1
            for (int kc = 0; kc < KC; ++kc) {
                load_frag(a.x + mc * a.K + 64 * kc, h, act);
                split_act(act, bp);
                gemm_lds_img(img, 8 * kc, lane, bp, acc);
            }
            finish(0);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Fourth variant ("h2"): every GEMM operand is an f16 hi + lo pair (22 significant bits) instead of three bf16 pieces, the
// products hi.hi + hi.lo + lo.hi accumulate in f32 on v_mfma_f32_32x32x16_f16 (each product is exact in the accumulator's
// format; the dropped lo.lo term is 2^-22 of the product).  Measured against fp64 the result is as accurate as an f32 GEMM
// with f32 accumulation (tools/emulate_h2.py: rms error 1.3e-8 on 0.16-sized outputs vs 2.3e-8 for numpy's f32 matmul and
// 4e-9 for the bf16x3 kernel, whose error is the final f32 rounding alone).  What it buys:
//   * 3 matrix instructions per fragment instead of 6, 2 KB per fragment instead of 3;
//   * W1, W2 and Wproj (144 KB as h2 images) are ALL LDS-resident: only the next block's Wqkv (48 KB) still streams from L2,
//     so the MLP loop has no global loads at all (the bf16x3 kernel streams W2, 96 KB per 32-row group, and waits for it);
//   * 32 fewer live registers in the MLP loop (activation pieces 32 instead of 48, fragments 8 instead of 12): no scratch spills
//     (the bf16x3 kernel spills 91 dwords per lane, some inside the chunk loop).
// Subnormal f16 operands do not survive the matrix pipe (measured: unscaled lo pieces gave 3e-5 on the block output, the size of
// the dropped subnormals), so both sides are scaled by exact powers of two that keep the lo pieces normal: the weights by 2^8 when
// the images are packed (gsdd_d3pm_layer_pack_h2; lo normal for |w| >= 4.9e-4, below that at most 2.4e-7 of the weight is lost;
// |w| < 255), the activations by 2^4 -- folded into the LayerNorm gamma / beta, the GELU2 bias and exponent constant, one
// multiplication for the attention output -- (lo normal for |a| >= 7.8e-3, below that at most 3.8e-6 is lost; |a| < 4094).
// Accumulators are scaled back by the exact factor 2^-12 in the fused multiply-add that adds the bias.
typedef _Float16 lh16x8 __attribute__((ext_vector_type(8)));
struct P2 { lh16x8 p[2]; };
constexpr float H2_WSCALE = 256.f, H2_ASCALE = 16.f, H2_UNSCALE = 1.f / (256.f * 16.f);
constexpr int H2_FRAG_U4 = 2 * 64;                            // uint4 per fragment
constexpr int H2_W2_OFF = 32 * H2_FRAG_U4, H2_WP_OFF = 64 * H2_FRAG_U4, H2_IMG_U4 = 72 * H2_FRAG_U4;    // [W1 | W2 | Wproj] = 144 KB
constexpr int H2_PAR_OFF = H2_IMG_U4 * 4;                     // float offset of the parameter block
constexpr int H2_SCR_OFF = H2_PAR_OFF + PAR_N;
constexpr int H2_LDS_FLOATS = H2_SCR_OFF + 8 * 192;           // 156,416 bytes

// GELU2 of v = v16 / 16, times 16: the power-of-two factors go through the exponent constant and the product exactly
__device__ __forceinline__ float gelu2_x16(float v16) {
    return v16 * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v16 * (-2.4554669595930157f / 16.f)));
}
__device__ __forceinline__ P2 split8h(const float (&v)[8]) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    typedef float f2v __attribute__((ext_vector_type(2)));
    P2 o;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f2v a = {v[i], v[i + 1]};
        const h2v hh = __builtin_convertvector(a, h2v);                       // v_cvt_pk_f16_f32 (round to nearest even)
        const f2v r = a - __builtin_convertvector(hh, f2v);                   // exact
        const h2v ll = __builtin_convertvector(r, h2v);
        o.p[0][i] = hh.x; o.p[0][i + 1] = hh.y; o.p[1][i] = ll.x; o.p[1][i + 1] = ll.y;
    }
    return o;
}
__device__ __forceinline__ P2 split8h_w(const float* row) {   // weights: scaled by 2^8
    const float4 lo = *reinterpret_cast<const float4*>(row), hi = *reinterpret_cast<const float4*>(row + 8);
    const float wv[8] = {lo.x * H2_WSCALE, lo.y * H2_WSCALE, lo.z * H2_WSCALE, lo.w * H2_WSCALE,
                         hi.x * H2_WSCALE, hi.y * H2_WSCALE, hi.z * H2_WSCALE, hi.w * H2_WSCALE};
    return split8h(wv);
}
__device__ __forceinline__ P2 load_frag2(const uint4* img, int f, int l) {
    P2 a;
#pragma unroll
    for (int i = 0; i < 2; ++i) a.p[i] = __builtin_bit_cast(lh16x8, img[(2 * f + i) * 64 + l]);
    return a;
}
__device__ __forceinline__ void mma3(const P2& a, const P2& b, f32x16& acc) {   // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.p[1], b.p[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.p[0], b.p[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.p[0], b.p[0], acc, 0, 0, 0);
}
__device__ __forceinline__ void split_act_h2(const float (&act)[32], P2 (&b)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = act[8 * q + j];
        b[q] = split8h(v);
    }
}

// one thread per (fragment, lane): W1 (32 fragments), W2 (32), Wproj (8) -> layer image; Wqkv (24) -> qkv image.  Fragment
// numbering and element order as in layer_pack_kernel (W1: f = (4 c + q) * 2 + nt, rows 64 c + 32 nt + li).
__global__ __launch_bounds__(256) void layer_pack_h2_kernel(const float* w1, const float* w2, const float* wproj, const float* wqkv,
                                                            uint4* lay_h2, uint4* wqkv_h2) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    const int nL = lay_h2 != nullptr ? 72 * 64 : 0, nQ = wqkv_h2 != nullptr ? 24 * 64 : 0;
    if (u >= nL + nQ) return;
    const bool is_q = u >= nL;
    const int v = is_q ? u - nL : u;
    const int f = v >> 6, l = v & 63, li = l & 31, h = l >> 5;
    const int nt = f & 1, q = (f >> 1) & 3, c = (f >> 3) & 3;
    P2 a;
    if (is_q) a = split8h_w(wqkv + (int64_t)(64 * c + 32 * nt + li) * D + 16 * q + 4 * h);
    else if (f < 32) a = split8h_w(w1 + (int64_t)(64 * c + 32 * nt + li) * D + 16 * q + 4 * h);
    else if (f < 64) a = split8h_w(w2 + (int64_t)(32 * nt + li) * HID + 64 * c + 16 * q + 4 * h);
    else a = split8h_w(wproj + (int64_t)(32 * nt + li) * D + 16 * q + 4 * h);
    uint4* img = is_q ? wqkv_h2 : lay_h2;
#pragma unroll
    for (int i = 0; i < 2; ++i) img[(2 * f + i) * 64 + l] = __builtin_bit_cast(uint4, a.p[i]);
}

// acc[nt] += (fragments f0 .. f0+7 of an LDS image) x the four k-steps of bp; the next fragment's two ds_read_b128 are issued
// before the current fragment's three MFMAs
template <bool SWAP = false>   // SWAP: activations as the A operand (result: row in registers, feature on the lane)
__device__ __forceinline__ void gemm_lds_img_h2(const uint4* img, int f0, int lane, const P2 (&bp)[4], f32x16 (&acc)[2]) {
    P2 cur = load_frag2(img, f0, lane);
#pragma unroll
    for (int ts = 0; ts < 8; ++ts) {
        const P2 nxt = load_frag2(img, f0 + (ts < 7 ? ts + 1 : 0), lane);
        if (SWAP) mma3(bp[ts >> 1], cur, acc[ts & 1]);
        else mma3(cur, bp[ts >> 1], acc[ts & 1]);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        cur = nxt;
    }
}

template <bool HAS_QKV, bool QKV_ONLY = false>
__global__ __launch_bounds__(512, 1) void d3pm_layer_h2_kernel(const LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint4* iw1 = reinterpret_cast<uint4*>(lds);              // 32 fragments (QKV_ONLY: the 24 fragments of Wqkv)
    uint4* iw2 = iw1 + H2_W2_OFF;                            // 32 fragments
    uint4* iwp = iw1 + H2_WP_OFF;                            // 8 fragments
    float* par = lds + H2_PAR_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    float* scr = lds + H2_SCR_OFF + wave * 192;
    const uint4* img_qkv = a.wqkv_h2;                        // 24 fragments

    // the weight images go global -> registers -> LDS with every load in flight before the first store (a copy loop pays one L2
    // round trip per iteration: 18 of them were 5 us of an 84 us kernel)
    {
        constexpr int NCP = (QKV_ONLY ? 24 * H2_FRAG_U4 : H2_IMG_U4) / 512;
        static_assert(H2_IMG_U4 % 512 == 0 && (24 * H2_FRAG_U4) % 512 == 0, "image copy assumes whole rounds of the workgroup");
        const uint4* src = QKV_ONLY ? img_qkv : a.lay_h2;
        uint4 tmp[NCP];
#pragma unroll
        for (int i = 0; i < NCP; ++i) tmp[i] = src[tid + 512 * i];
#pragma unroll
        for (int i = 0; i < NCP; ++i) iw1[tid + 512 * i] = tmp[i];
    }
    for (int i = tid; i < PAR_N; i += 512) {
        float v;
        if (QKV_ONLY) v = i >= PAR_BQKV ? a.bqkv[i - PAR_BQKV] : 0.f;
        else if (i < PAR_G2) v = a.bproj[i];
        else if (i < PAR_B2LN) v = a.ln2_g[i - PAR_G2] * H2_ASCALE;          // LN2 output, GELU2 input: carried 16 x
        else if (i < PAR_B1) v = a.ln2_b[i - PAR_B2LN] * H2_ASCALE;
        else if (i < PAR_B2) v = a.b1[i - PAR_B1] * H2_ASCALE;
        else if (i < PAR_BQKV) v = a.b2[i - PAR_B2];
        else v = HAS_QKV ? a.bqkv[i - PAR_BQKV] : 0.f;
        par[i] = v;
    }
    __syncthreads();

    const int64_t ngroups = (a.M + 31) / 32;
    const bool batch_uniform = a.L % 32 == 0;
    for (int64_t grp = (int64_t)blockIdx.x * 8 + wave; grp < ngroups; grp += (int64_t)gridDim.x * 8) {
        const int64_t m = grp * 32 + li;
        const bool valid = m < a.M;
        const bool full = grp * 32 + 32 <= a.M;
        const int64_t mc = valid ? m : a.M - 1;
        const int b = (int)((uint32_t)mc / (uint32_t)a.L);
        if (batch_uniform) {
            const int bu = (int)((uint32_t)(grp * 32) / (uint32_t)a.L);
            if (lane < 16) {
                float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.cvec != nullptr) cv = *reinterpret_cast<const float4*>(a.cvec + (int64_t)bu * D + 4 * lane);
                *reinterpret_cast<float4*>(scr + 4 * lane) = cv;
            } else if (HAS_QKV && lane < 48) {
                const float* tab = a.ada + a.t2[bu] * (2 * D);
                const float4 tv = *reinterpret_cast<const float4*>(tab + 4 * (lane - 16));
                *reinterpret_cast<float4*>(scr + 4 * lane) =
                    make_float4(tv.x * H2_ASCALE, tv.y * H2_ASCALE, tv.z * H2_ASCALE, tv.w * H2_ASCALE);
            }
        }

        float act[32], x1[32];
        f32x16 acc[2];
        P2 bp[4];
        float mean, rstd;
        if (QKV_ONLY) {
            load_frag(a.x + mc * D, h, x1);
        } else {
        // ---- x1 = x + proj(y) + b_proj + cvec[b]
        // (Tried and measured slower, each through spills of the 256-register budget: requesting the next group's rows during this
        // group's q|k|v stage, 77 -> 108 us -- vmcnt also retires in order, so every wait for a Wqkv fragment behind those loads waits
        // for their HBM latency; holding q and k in registers so that all stores follow the last product, 194 us; requesting Wqkv a
        // whole 8-fragment block at a time ahead of the previous block's stores, 110 us.)
        load_frag(a.y + mc * D, h, act);
        load_frag(a.x + mc * D, h, x1);
#pragma unroll
        for (int i = 0; i < 32; ++i) act[i] *= H2_ASCALE;
        split_act_h2(act, bp);
        zero2(acc);
        gemm_lds_img_h2(iwp, 0, lane, bp, acc);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = 32 * t + 8 * g + 4 * h;
                const float4 bpj = *reinterpret_cast<const float4*>(par + PAR_BPROJ + f);
                float4 cv;
                if (batch_uniform) cv = *reinterpret_cast<const float4*>(scr + f);
                else cv = a.cvec != nullptr ? *reinterpret_cast<const float4*>(a.cvec + (int64_t)b * D + f) : make_float4(0.f, 0.f, 0.f, 0.f);
                const int r = 4 * g;
                x1[16 * t + r + 0] += fmaf(acc[t][r + 0], H2_UNSCALE, bpj.x) + cv.x;
                x1[16 * t + r + 1] += fmaf(acc[t][r + 1], H2_UNSCALE, bpj.y) + cv.y;
                x1[16 * t + r + 2] += fmaf(acc[t][r + 2], H2_UNSCALE, bpj.z) + cv.z;
                x1[16 * t + r + 3] += fmaf(acc[t][r + 3], H2_UNSCALE, bpj.w) + cv.w;
            }
        // ---- h = LN2(x1) * gamma + beta
        row_norm(x1, mean, rstd);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = 32 * t + 8 * g + 4 * h;
                const float4 gm = *reinterpret_cast<const float4*>(par + PAR_G2 + f);
                const float4 bt = *reinterpret_cast<const float4*>(par + PAR_B2LN + f);
                const int r = 16 * t + 4 * g;
                act[r + 0] = (x1[r + 0] - mean) * rstd * gm.x + bt.x;
                act[r + 1] = (x1[r + 1] - mean) * rstd * gm.y + bt.y;
                act[r + 2] = (x1[r + 2] - mean) * rstd * gm.z + bt.z;
                act[r + 3] = (x1[r + 3] - mean) * rstd * gm.w + bt.w;
            }
        split_act_h2(act, bp);
        // ---- MLP in 4 chunks of 64 hidden units, both weight images in LDS
        f32x16 acc3[2];
        zero2(acc3);
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {
            zero2(acc);
            gemm_lds_img_h2(iw1, 8 * c, lane, bp, acc);
            // GELU2 + split of k-step q + 1 is issued between the MFMAs of k-step q (both tiles)
            auto make_ub = [&](int q) {
                const int t = q >> 1, s2 = q & 1;
                const float4 b0 = *reinterpret_cast<const float4*>(par + PAR_B1 + 64 * c + 32 * t + 16 * s2 + 4 * h);
                const float4 b1 = *reinterpret_cast<const float4*>(par + PAR_B1 + 64 * c + 32 * t + 16 * s2 + 8 + 4 * h);
                // 16 x hidden = acc * 2^-8 + 16 b1 (both exact scalings of the unscaled value), 16 x GELU2 out of it
                constexpr float U16 = H2_UNSCALE * H2_ASCALE;
                const float u[8] = {gelu2_x16(fmaf(acc[t][8 * s2 + 0], U16, b0.x)), gelu2_x16(fmaf(acc[t][8 * s2 + 1], U16, b0.y)),
                                    gelu2_x16(fmaf(acc[t][8 * s2 + 2], U16, b0.z)), gelu2_x16(fmaf(acc[t][8 * s2 + 3], U16, b0.w)),
                                    gelu2_x16(fmaf(acc[t][8 * s2 + 4], U16, b1.x)), gelu2_x16(fmaf(acc[t][8 * s2 + 5], U16, b1.y)),
                                    gelu2_x16(fmaf(acc[t][8 * s2 + 6], U16, b1.z)), gelu2_x16(fmaf(acc[t][8 * s2 + 7], U16, b1.w))};
                return split8h(u);
            };
            P2 ub = make_ub(0);
            P2 w2a = load_frag2(iw2, 8 * c, lane);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f0 = 8 * c + 2 * q;
                const P2 w2b = load_frag2(iw2, f0 + 1, lane);
                const P2 na = load_frag2(iw2, (f0 + 2) & 31, lane);
                P2 ubn = ub;
                if (q < 3) ubn = make_ub(q + 1);
                mma3(w2a, ub, acc3[0]);
                mma3(w2b, ub, acc3[1]);
                if (q < 3) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA ...
                        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);    // ... then a slice of the next operand's VALU work
                    }
                }
                ub = ubn;
                w2a = na;
            }
        }
        // ---- x2 = x1 + mlp + b2 -> x
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f = 32 * t + 8 * g + 4 * h;
                const float4 bb = *reinterpret_cast<const float4*>(par + PAR_B2 + f);
                const int r = 4 * g;
                x1[16 * t + r + 0] += fmaf(acc3[t][r + 0], H2_UNSCALE, bb.x);
                x1[16 * t + r + 1] += fmaf(acc3[t][r + 1], H2_UNSCALE, bb.y);
                x1[16 * t + r + 2] += fmaf(acc3[t][r + 2], H2_UNSCALE, bb.z);
                x1[16 * t + r + 3] += fmaf(acc3[t][r + 3], H2_UNSCALE, bb.w);
            }
        if (a.range_flag != nullptr) {
            // range screen: operands are 16 a as f16, so |a| >= 4094 (LN output, GELU2 output, attention output) becomes inf,
            // and inf / NaN then reach this row through the products and the residual adds (a q|k|v-stage overflow: through the
            // next block's attention).  inf and NaN survive additions, so one sum and one compare per lane test the 32 values.
            float chk = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) chk += x1[i];
            if (__any(valid && !(fabsf(chk) < 3.0e38f)) && lane == 0) atomicOr(a.range_flag, 1);
        }
        if (full) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                *reinterpret_cast<float4*>(a.x + m * D + 32 * (q >> 2) + 8 * (q & 3) + 4 * h) =
                    make_float4(x1[4 * q + 0], x1[4 * q + 1], x1[4 * q + 2], x1[4 * q + 3]);
        } else if (valid) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                *reinterpret_cast<float4*>(a.x + m * D + 32 * (q >> 2) + 8 * (q & 3) + 4 * h) =
                    make_float4(x1[4 * q + 0], x1[4 * q + 1], x1[4 * q + 2], x1[4 * q + 3]);
        }
        }
        if (HAS_QKV) {
            // Wqkv fragments come from L2 three tile-steps ahead of their use; the first three are requested before the AdaLN arithmetic
            P2 wq0, wq1, wq2;
            if (!QKV_ONLY) { wq0 = load_frag2(img_qkv, 0, lane); wq1 = load_frag2(img_qkv, 1, lane); wq2 = load_frag2(img_qkv, 2, lane); }
            row_norm(x1, mean, rstd);
            const float* tab = a.ada + a.t2[b] * (2 * D);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f = 32 * t + 8 * g + 4 * h;
                    float4 gm, bt;
                    if (batch_uniform) {
                        gm = *reinterpret_cast<const float4*>(scr + 64 + f);
                        bt = *reinterpret_cast<const float4*>(scr + 128 + f);
                    } else {
                        gm = *reinterpret_cast<const float4*>(tab + f);
                        bt = *reinterpret_cast<const float4*>(tab + D + f);
                        gm = make_float4(gm.x * H2_ASCALE, gm.y * H2_ASCALE, gm.z * H2_ASCALE, gm.w * H2_ASCALE);
                        bt = make_float4(bt.x * H2_ASCALE, bt.y * H2_ASCALE, bt.z * H2_ASCALE, bt.w * H2_ASCALE);
                    }
                    const int r = 16 * t + 4 * g;
                    act[r + 0] = (x1[r + 0] - mean) * rstd * gm.x + bt.x;
                    act[r + 1] = (x1[r + 1] - mean) * rstd * gm.y + bt.y;
                    act[r + 2] = (x1[r + 2] - mean) * rstd * gm.z + bt.z;
                    act[r + 3] = (x1[r + 3] - mean) * rstd * gm.w + bt.w;
                }
            split_act_h2(act, bp);
#pragma unroll 1
            for (int c = 0; c < 3; ++c) {
                zero2(acc);
                // The V image wants eight *rows* of one column in a lane (below), so for it the product is taken the other way
                // round (activations as the A operand): acc[nt][r] = v[row 8 (r >> 2) + 4 h + (r & 3)][feature 32 nt + li].
                const bool v_img = c == 2 && a.vimg != nullptr;                         // wave-uniform
                if (QKV_ONLY) {
                    if (v_img) gemm_lds_img_h2<true>(iw1, 8 * c, lane, bp, acc);
                    else gemm_lds_img_h2<false>(iw1, 8 * c, lane, bp, acc);
                } else {
#pragma unroll
                    for (int ts = 0; ts < 8; ++ts) {
                        const int fn = 8 * c + ts + 3;             // (wraps to the first fragments after the last block: harmless)
                        const P2 wq3 = load_frag2(img_qkv, fn < 24 ? fn : fn - 24, lane);
                        if (v_img) mma3(bp[ts >> 1], wq0, acc[ts & 1]);
                        else mma3(wq0, bp[ts >> 1], acc[ts & 1]);
                        wq0 = wq1; wq1 = wq2; wq2 = wq3;
                    }
                }
                if (v_img) {
                    // lane (li, h): head hd = 8 nt + (li >> 2), dim d = li & 3.  Accumulator registers 4 g2 + e and 4 (g2 + 2) + e
                    // (e = 0..3) are rows 4 g + e of the group's two 16-key tiles for key group g = 2 g2 + h: the eight f16 of
                    // image entry (pair-tile, g, column) -- three entries for the three pieces of v, and lane d also writes the
                    // constant column 12 + d (1, 0, 0, 0).  (The group is one 32-key pair-tile of every head: L % 32 == 0.)
                    const int d = li & 3;
                    const uint32_t cst = d == 0 ? 0x3C003C00u : 0u;                       // f16 1.0 | 1.0
                    const int64_t m0 = grp * 32;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int hd = 8 * nt + (li >> 2);
                        const float bv = par[PAR_BQKV + 128 + 32 * nt + li];
                        uint4* dst = a.vimg + (((int64_t)hd * a.M + m0) >> 5) * 64;
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2) {
                            _Float16 p1[8], p2[8], p3[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float v = fmaf(acc[nt][4 * g2 + (e & 3) + 8 * (e >> 2)], H2_UNSCALE, bv);
                                const _Float16 a1 = (_Float16)v;
                                const float r1 = (v - (float)a1) * 2048.f;
                                const _Float16 a2 = (_Float16)r1;
                                p1[e] = a1; p2[e] = a2; p3[e] = (_Float16)((r1 - (float)a2) * 2048.f);
                            }
                            uint4* e0 = dst + (2 * g2 + h) * 16;
                            e0[d] = __builtin_bit_cast(uint4, p1);
                            e0[4 + d] = __builtin_bit_cast(uint4, p2);
                            e0[8 + d] = __builtin_bit_cast(uint4, p3);
                            e0[12 + d] = make_uint4(cst, cst, cst, cst);
                        }
                    }
                    continue;
                }
                float4 o[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int t = q >> 2, r = 4 * (q & 3);
                    const float4 bb = *reinterpret_cast<const float4*>(par + PAR_BQKV + 64 * c + 32 * t + 8 * (q & 3) + 4 * h);
                    o[q] = make_float4(fmaf(acc[t][r + 0], H2_UNSCALE, bb.x), fmaf(acc[t][r + 1], H2_UNSCALE, bb.y),
                                       fmaf(acc[t][r + 2], H2_UNSCALE, bb.z), fmaf(acc[t][r + 3], H2_UNSCALE, bb.w));
                }
                if (c == 1 && a.kimg != nullptr) {
                    // k of head hd = 8 t + 2 g + h as the attention kernel's pre-split image (row = hd * M + m): two 16-byte stores
                    if (valid) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int hd = 8 * (q >> 2) + 2 * (q & 3) + h;
                            const float vals[4] = {o[q].x, o[q].y, o[q].z, o[q].w};
                            kv_image_store_k(vals, (int64_t)hd * a.M + m, a.kimg);
                        }
                    }
                    kv_image_store_knorm(o, h, li, grp, a.M, a.knorm, a.ksum);      // M % 32 == 0 here: the group is one whole pair-tile
                } else if (full) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int n = 64 * c + 32 * (q >> 2) + 8 * (q & 3) + 4 * h;
                        *reinterpret_cast<float4*>(a.qkv + ((int64_t)(n >> 2) * a.M + m) * 4) = o[q];
                    }
                } else if (valid) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int n = 64 * c + 32 * (q >> 2) + 8 * (q & 3) + 4 * h;
                        *reinterpret_cast<float4*>(a.qkv + ((int64_t)(n >> 2) * a.M + m) * 4) = o[q];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// to_logits: logits[m][:] = W LN(x[m]) + b  (nn.LayerNorm(64) + nn.Linear(64 -> K), transformer_utils.py:353-356,442).
// Same transposed-GEMM register layout: a wave owns 32 rows, normalises them in registers once and sweeps all K
// output features; W streams through LDS in 256-feature chunks shared by the 8 waves (double-buffered).
struct LogitsArgs {
    const float* x; int64_t M; int K;
    const float* g; const float* b;      // LayerNorm affine
    const float* w; const float* bias;   // [K][64], [K]
    float* out;                          // [M][K]
};

constexpr int LCH = 256;                 // features per LDS chunk

__global__ __launch_bounds__(512, 1) void d3pm_logits_kernel(const LogitsArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [2][LCH][W1P]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int nch = (a.K + LCH - 1) / LCH;
    const int64_t nblocks = (a.M + 255) / 256;

    // Weight rows of the next chunk are staged a quarter per sub-chunk (8 registers): loaded before the sub-chunk's MFMAs, written
    // to the other LDS buffer after them.  The loads are branch-free (rows past K read row K-1; the classes they produce are never
    // stored) so that nothing waits on them, or on the logits stores still in flight, at the point of issue.
    float4 stage[2];
    auto load_w = [&](int c, int quarter) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 512 * (2 * quarter + i);  // 4096 float4 per chunk
            const int n = c * LCH + (idx >> 4), col = (idx & 15) * 4;
            stage[i] = *reinterpret_cast<const float4*>(a.w + (int64_t)(n < a.K ? n : a.K - 1) * D + col);
        }
    };
    auto store_w = [&](int buf, int quarter) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 512 * (2 * quarter + i);
            *reinterpret_cast<float4*>(&lds[(buf * LCH + (idx >> 4)) * W1P + (idx & 15) * 4]) = stage[i];
        }
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int64_t m = blk * 256 + wave * 32 + li;
        const int64_t mc = m < a.M ? m : a.M - 1;
        float act[32];
        {
            float xr[32];
            load_frag(a.x + mc * D, h, xr);
            float mean, rstd;
            row_norm(xr, mean, rstd);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int f = 32 * t + 8 * g + 4 * h;
                    const float4 gm = *reinterpret_cast<const float4*>(a.g + f);
                    const float4 bt = *reinterpret_cast<const float4*>(a.b + f);
                    const int r = 16 * t + 4 * g;
                    act[r + 0] = (xr[r + 0] - mean) * rstd * gm.x + bt.x;
                    act[r + 1] = (xr[r + 1] - mean) * rstd * gm.y + bt.y;
                    act[r + 2] = (xr[r + 2] - mean) * rstd * gm.z + bt.z;
                    act[r + 3] = (xr[r + 3] - mean) * rstd * gm.w + bt.w;
                }
        }
        __syncthreads();                 // previous block's readers are done with both buffers
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) { load_w(0, qd); store_w(0, qd); }
        __syncthreads();
        const int64_t row_u = blk * 256 + wave_u * 32;                   // wave-uniform first row
        const bool rows_full = row_u + 32 <= a.M;
        const uint32_t vbyte = ((uint32_t)(4 * h) * (uint32_t)a.K + (uint32_t)li) * 4u;   // lane part of the store address (bytes < 2^32)
        for (int c = 0; c < nch; ++c) {
            const int buf = c & 1;
            const bool more = c + 1 < nch;
            const float* wl = lds + buf * LCH * W1P;
#pragma unroll 1
            for (int sc = 0; sc < LCH / 64; ++sc) {
                // (plain locals rather than the staging array: the array version ends up in scratch)
                const int wi0 = tid + 512 * (2 * sc), wi1 = wi0 + 512;
                const int wn0 = (c + 1) * LCH + (wi0 >> 4), wn1 = (c + 1) * LCH + (wi1 >> 4);
                const int wr0 = more ? (wn0 < a.K ? wn0 : a.K - 1) : 0, wr1 = more ? (wn1 < a.K ? wn1 : a.K - 1) : 0;
                const float4 st0 = *reinterpret_cast<const float4*>(a.w + (int64_t)wr0 * D + (wi0 & 15) * 4);
                const float4 st1 = *reinterpret_cast<const float4*>(a.w + (int64_t)wr1 * D + (wi1 & 15) * 4);
                __builtin_amdgcn_sched_barrier(0);         // keep the two loads above the MFMA block (the scheduler sinks them)
                f32x16 acc[2];
                zero2(acc);
                gemm64_rows(wl + sc * 64 * W1P, W1P, li, h, act, acc);
                // lane = class n (li), register r = row (r&3) + 8(r>>2) + 4h of this wave's 32 rows.
                // The stores read their data registers when they execute, so those registers must not be rewritten until the
                // stores are acknowledged: the results are moved (bias added) into `o`, and the empty asm keeps `acc` alive past
                // that point so that `o` cannot share registers with the accumulators the next sub-chunk's MFMAs overwrite.
                const int nb = c * LCH + sc * 64;
                float o[2][16];
                // `o` still feeds the previous sub-chunk's stores; they were issued a whole MFMA block ago, so this wait is free
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int n = nb + 32 * t + li;
                    const float bb = a.bias[n < a.K ? n : a.K - 1];
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[t][r] = acc[t][r] + bb;
                }
                asm volatile("" :: "v"(acc[0]), "v"(acc[1]));
                // stores with a wave-uniform (SGPR) base and one 32-bit lane offset: the compiler's own addressing spends 64
                // VGPRs on the 32 addresses and then spills
                const char* tile = reinterpret_cast<const char*>(a.out + (row_u * a.K + nb));
                if (rows_full && nb + 64 <= a.K) {
                    // whole 32 x 64 tile in range (wave-uniform test): 32 unconditional stores, nothing waits between them
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const char* tr = tile + (int64_t)((r & 3) + 8 * (r >> 2)) * a.K * 4;
                        asm volatile("global_store_dword %0, %1, %2\n\tglobal_store_dword %0, %3, %2 offset:128"
                                     :: "v"(vbyte), "v"(o[0][r]), "s"(tr), "v"(o[1][r]) : "memory");
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int64_t mr = row_u + 4 * h + (r & 3) + 8 * (r >> 2);
                            if (mr < a.M && nb + 32 * t + li < a.K)
                                a.out[mr * a.K + nb + 32 * t + li] = o[t][r];
                        }
                }
                // the quarter loaded before this sub-chunk's MFMAs goes to the other LDS buffer, which is free: every wave passed
                // the barrier that ended the previous chunk
                if (more) {
                    *reinterpret_cast<float4*>(&lds[((buf ^ 1) * LCH + (wi0 >> 4)) * W1P + (wi0 & 15) * 4]) = st0;
                    *reinterpret_cast<float4*>(&lds[((buf ^ 1) * LCH + (wi1 >> 4)) * W1P + (wi1 & 15) * 4]) = st1;
                }
            }
            // LDS-only barrier: __syncthreads() is also a fence and would wait for the logits stores above to be acknowledged
            // (s_waitcnt vmcnt(0)) at every chunk
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
}

}  // namespace gsdd

using namespace gsdd;

extern "C" int gsdd_d3pm_layer(const gsdd_layer_desc* d, void* stream) {
    GSDD_CHECK_ARG(d != nullptr, "null descriptor");
    const bool qkv_only = d->y == nullptr;                     // next-block stage only (block 0): needs the packed kernel
    GSDD_CHECK_ARG(d->x && (qkv_only || (d->y && d->wproj && d->bproj && d->ln2_g && d->ln2_b && d->w1 && d->b1 && d->w2 && d->b2)),
                   "null pointer");
    GSDD_CHECK_ARG(d->M > 0 && d->M < (1ll << 31) && d->L > 0, "bad sizes");
    GSDD_CHECK_ARG(d->n_embd == 64 && d->hidden == 256, "kernel is specialised for n_embd 64, hidden 256");
    const bool has_qkv = d->qkv != nullptr;
    GSDD_CHECK_ARG(!has_qkv || (d->ada && d->t2 && d->wqkv && d->bqkv), "next-block operands missing");
    LayerArgs a;
    a.y = d->y; a.x = d->x; a.M = d->M; a.L = d->L; a.cvec = d->cvec;
    a.wproj = d->wproj; a.bproj = d->bproj; a.ln2_g = d->ln2_g; a.ln2_b = d->ln2_b;
    a.w1 = d->w1; a.b1 = d->b1; a.w2 = d->w2; a.b2 = d->b2;
    a.ada = d->ada; a.t2 = d->t2; a.wqkv = d->wqkv; a.bqkv = d->bqkv; a.qkv = d->qkv;
    a.range_flag = d->range_flag;
    const int64_t ngroups = (d->M + 31) / 32;
    const unsigned grid = (unsigned)std::min<int64_t>((ngroups + 7) / 8, 256);
    a.w2_x3 = reinterpret_cast<const uint4*>(d->w2_x3); a.wqkv_x3 = reinterpret_cast<const uint4*>(d->wqkv_x3);
    a.lay_h2 = reinterpret_cast<const uint4*>(d->layer_h2); a.wqkv_h2 = reinterpret_cast<const uint4*>(d->wqkv_h2);
    // d->variant picks the kernel (include/gsdd.h): GSDD_LAYER_AUTO = the f16 hi + lo images when the caller packed them, else the
    // bf16x3 images.  (The exact-f32 and split-on-the-fly kernels of rounds 1-2 are gone: nothing reached them but an A/B switch.)
    const bool have_x3 = (qkv_only || d->w2_x3 != nullptr) && (!has_qkv || d->wqkv_x3 != nullptr);
    const bool have_h2 = (qkv_only || d->layer_h2 != nullptr) && (!has_qkv || d->wqkv_h2 != nullptr);
    GSDD_CHECK_ARG(d->variant == GSDD_LAYER_AUTO || d->variant == GSDD_LAYER_X3P || d->variant == GSDD_LAYER_H2, "variant: one of GSDD_LAYER_*");
    const int variant = d->variant == GSDD_LAYER_AUTO ? (have_h2 ? GSDD_LAYER_H2 : GSDD_LAYER_X3P) : d->variant;
    GSDD_CHECK_ARG(variant != GSDD_LAYER_H2 || have_h2, "the f16 hi + lo kernel needs the gsdd_d3pm_layer_pack_h2 images");
    GSDD_CHECK_ARG(variant != GSDD_LAYER_X3P || have_x3, "the bf16x3 kernel needs the gsdd_d3pm_layer_pack images (or pass the f16 hi + lo ones)");
    a.kimg = a.vimg = nullptr;
    a.knorm = nullptr;
    a.ksum = nullptr;
    if (has_qkv && d->kv_img != nullptr) {
        GSDD_CHECK_ARG(d->L % 32 == 0 && d->M % d->L == 0, "kv_img needs L % 32 == 0 and whole batch elements");
        GSDD_CHECK_ARG(d->kv_img_bytes >= gsdd_d3pm_attention_workspace_bytes((int)(d->M / d->L), d->L, 16), "kv_img too small");
        a.kimg = reinterpret_cast<uint4*>(d->kv_img);
        a.vimg = a.kimg + d->M * 16 * 2;                      // K image: 2 uint4 per (row, head), 16 heads
        a.knorm = kv_image_knorm(d->kv_img, d->M * 16);
        a.ksum = kv_image_ksum(d->kv_img, d->M * 16);
    }
    GSDD_CHECK_ARG(!qkv_only || has_qkv, "y = NULL (q|k|v stage only) needs qkv");
    if (variant == GSDD_LAYER_H2) {
        const size_t ldsh = (size_t)H2_LDS_FLOATS * sizeof(float);
        GSDD_ONCE_PER_DEVICE(attr_h,
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)d3pm_layer_h2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh));
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)d3pm_layer_h2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh));
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)d3pm_layer_h2_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh));
        );
        if (qkv_only) hipLaunchKernelGGL((d3pm_layer_h2_kernel<true, true>), dim3(grid), dim3(512), ldsh, (hipStream_t)stream, a);
        else if (has_qkv) hipLaunchKernelGGL(d3pm_layer_h2_kernel<true>, dim3(grid), dim3(512), ldsh, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(d3pm_layer_h2_kernel<false>, dim3(grid), dim3(512), ldsh, (hipStream_t)stream, a);
    } else {
        const size_t ldsp = (size_t)X3P_LDS_FLOATS * sizeof(float);
        GSDD_ONCE_PER_DEVICE(attr_p,
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)d3pm_layer_x3p_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp));
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)d3pm_layer_x3p_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp));
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)d3pm_layer_x3p_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp));
        );
        if (qkv_only) hipLaunchKernelGGL((d3pm_layer_x3p_kernel<true, true>), dim3(grid), dim3(512), ldsp, (hipStream_t)stream, a);
        else if (has_qkv) hipLaunchKernelGGL(d3pm_layer_x3p_kernel<true>, dim3(grid), dim3(512), ldsp, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(d3pm_layer_x3p_kernel<false>, dim3(grid), dim3(512), ldsp, (hipStream_t)stream, a);
    }
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_rows_linear_pack_many(const void* descs_dev, int n_desc, int max_out, int max_in, void* stream) {
    GSDD_CHECK_ARG(descs_dev != nullptr && n_desc > 0 && max_out > 0 && max_in > 0 && max_out % 64 == 0 && max_in % 64 == 0, "bad args");
    const int units = (max_out >> 6) * (max_in >> 6) * 8 * 64;
    hipLaunchKernelGGL(rows_linear_pack_kernel, dim3((units + 255) / 256, n_desc), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const RowsPackDesc*>(descs_dev), n_desc);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

template <int KC, int NB>
static int rows_linear_launch(const RowsLinArgs& a, void* stream) {
    const size_t ldsb = (size_t)NB * KC * 8 * IMG_FRAG_U4 * 16 + (size_t)64 * NB * sizeof(float);
    GSDD_ONCE_PER_DEVICE(attr,
        GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)rows_linear_kernel<KC, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    );
    const int64_t ngroups = (a.M + 31) / 32;
    const unsigned grid = (unsigned)std::min<int64_t>((ngroups + 7) / 8, 256);
    hipLaunchKernelGGL((rows_linear_kernel<KC, NB>), dim3(grid), dim3(512), ldsb, (hipStream_t)stream, a);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_rows_linear(const float* x, int64_t M, int n_in, const void* image, int n_out, const float* bias, const float* bvec,
                                int rows_per_batch, const float* residual, float* out, int head_major, void* stream) {
    GSDD_CHECK_ARG(x && image && out && M > 0, "null pointer");
    GSDD_CHECK_ARG(bvec == nullptr || rows_per_batch > 0, "bvec needs rows_per_batch");
    RowsLinArgs a;
    a.x = x; a.M = M; a.K = n_in; a.N = n_out; a.img = reinterpret_cast<const uint4*>(image); a.bias = bias; a.bvec = bvec;
    a.rows_per_batch = rows_per_batch; a.residual = residual; a.out = out; a.head_major = head_major;
    if (n_in == 64 && n_out == 64) return rows_linear_launch<1, 1>(a, stream);
    if (n_in == 64 && n_out == 128) return rows_linear_launch<1, 2>(a, stream);
    if (n_in == 64 && n_out == 192) return rows_linear_launch<1, 3>(a, stream);
    if (n_in == 64 && n_out == 256) return rows_linear_launch<1, 4>(a, stream);
    if (n_in == 128 && n_out == 64) return rows_linear_launch<2, 1>(a, stream);
    if (n_in == 192 && n_out == 64) return rows_linear_launch<3, 1>(a, stream);
    if (n_in == 256 && n_out == 64) return rows_linear_launch<4, 1>(a, stream);
    GSDD_CHECK_ARG(false, "gsdd_rows_linear: shapes are 64 -> 64 / 128 / 192 / 256 and 128 / 192 / 256 -> 64");
    return GSDD_OK;
}

extern "C" int gsdd_d3pm_layer_pack_h2(const float* w1, const float* w2, const float* wproj, const float* wqkv, void* layer_h2,
                                       void* wqkv_h2, void* stream) {
    GSDD_CHECK_ARG((layer_h2 == nullptr || (w1 && w2 && wproj)) && (wqkv_h2 == nullptr || wqkv) && (layer_h2 || wqkv_h2), "null pointer");
    const int n = (layer_h2 != nullptr ? 72 * 64 : 0) + (wqkv_h2 != nullptr ? 24 * 64 : 0);
    hipLaunchKernelGGL(layer_pack_h2_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w1, w2, wproj, wqkv,
                       reinterpret_cast<uint4*>(layer_h2), reinterpret_cast<uint4*>(wqkv_h2));
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_d3pm_layer_pack(const float* w2, const float* wproj, const float* wqkv, void* layer_x3, void* wqkv_x3,
                                    void* stream) {
    GSDD_CHECK_ARG(w2 != nullptr && wproj != nullptr && layer_x3 != nullptr, "null pointer");
    GSDD_CHECK_ARG((wqkv == nullptr) == (wqkv_x3 == nullptr), "wqkv and its image come together");
    const int units = (40 + (wqkv != nullptr ? 24 : 0)) * 64;
    hipLaunchKernelGGL(layer_pack_kernel, dim3((units + 255) / 256), dim3(256), 0, (hipStream_t)stream, w2, wproj, wqkv,
                       reinterpret_cast<uint4*>(layer_x3), reinterpret_cast<uint4*>(wqkv_x3));
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_d3pm_logits(const float* x, int64_t M, int n_embd, const float* ln_g, const float* ln_b, const float* w,
                                const float* bias, int K, float* out, void* stream) {
    GSDD_CHECK_ARG(x && ln_g && ln_b && w && bias && out, "null pointer");
    GSDD_CHECK_ARG(M > 0 && n_embd == 64 && K > 0 && K % 4 == 0, "kernel is specialised for n_embd 64, K % 4 == 0");
    LogitsArgs a;
    a.x = x; a.M = M; a.K = K; a.g = ln_g; a.b = ln_b; a.w = w; a.bias = bias; a.out = out;
    const size_t lds = (size_t)2 * LCH * W1P * sizeof(float);
    GSDD_ONCE_PER_DEVICE(attr_done,
        GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)d3pm_logits_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    );
    const unsigned grid = (unsigned)std::min<int64_t>((M + 255) / 256, 256);
    hipLaunchKernelGGL(d3pm_logits_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}
