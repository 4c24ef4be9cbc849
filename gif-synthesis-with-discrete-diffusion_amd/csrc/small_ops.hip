// Small / memory-bound operators of the path: token+position embedding, AdaLayerNorm table,
// tiny linears for the condition token, axial attention of the VQ-VAE res blocks, nearest-code search.
#include "common.hpp"

namespace gsdd {

// ------------------------------------------------------------------ DalleMaskImageEmbedding (dalle_mask_image_embedding.py:59-79)
__global__ void embed_kernel(const int64_t* tok, int64_t rows, int L, int D, const float* emb, int n_embed,
                             const float* pos, int rep, float* x) {
    const int q4 = D >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * q4) return;
    const int64_t row = i / q4;
    const int c = (int)(i % q4) * 4;
    int64_t t = tok[row];
    t = t < 0 ? 0 : t;                                   // index[index < 0] = 0  (:63)
    t = t >= n_embed ? n_embed - 1 : t;                  // the reference raises; never index out of bounds here
    const int l = (int)(row % L);
    const float4 e = *reinterpret_cast<const float4*>(emb + t * D + c);
    const float4 p = *reinterpret_cast<const float4*>(pos + (int64_t)l * D + c);
    const float4 o = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
    for (int r = 0; r < rep; ++r) *reinterpret_cast<float4*>(x + ((int64_t)r * rows + row) * D + c) = o;
}

// ------------------------------------------------------------------ AdaLayerNorm table (transformer_utils.py:138-159)
__global__ void adaln_table_kernel(const float* emb, int T, int D, const float* w, const float* bias, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * 2 * D) return;
    const int t = i / (2 * D), j = i % (2 * D);
    float s = 0.f;
    for (int k = 0; k < D; ++k) {
        const float e = emb[t * D + k];
        const float si = e / (1.f + expf(-e));            // SiLU
        s = fmaf(si, w[j * D + k], s);
    }
    s += bias[j];
    out[i] = j < D ? 1.f + s : s;                          // (1 + scale) | shift
}

// one wave per output element: the lanes stride over Cin (coalesced rows of x and w), then a wave reduction
__global__ __launch_bounds__(256) void small_linear_kernel(const float* x, int R, int Cin, const float* w, const float* b, int Cout,
                                                           float* y) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= R * Cout) return;                               // wave-uniform
    const int lane = threadIdx.x & 63;
    const int r = i / Cout, j = i % Cout;
    const float* xr = x + (int64_t)r * Cin;
    const float* wr = w + (int64_t)j * Cin;
    float s = 0.f;
    for (int k = lane; k < Cin; k += 64) s = fmaf(xr[k], wr[k], s);
    s = wave_sum(s);
    if (lane == 0) y[i] = s + (b != nullptr ? b[j] : 0.f);
}

// ------------------------------------------------------------------ axial attention (model_utils.py:318-337, :586-600)
// One wave per (line, head, axis).  S = axis length (<= 64), d = C / n_head.
__global__ __launch_bounds__(64) void axial_attention_kernel(const float* qkv, int N, int T, int H, int W, int C, int n_head,
                                                             int axis /* 0: w, 1: h, 2: t */, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int head = blockIdx.y;
    const int d = C / n_head;
    const int S = axis == 0 ? W : (axis == 1 ? H : T);
    int64_t line = blockIdx.x;                            // enumerates the other two dims and n
    // position stride along the axis and line base position
    int64_t stride, basepos;
    if (axis == 0) { stride = 1; basepos = line * W; }                                        // line = (n,t,h)
    else if (axis == 1) { stride = W; const int64_t w = line % W; const int64_t nt = line / W; basepos = nt * H * W + w; }
    else { stride = (int64_t)H * W; const int64_t hw = line % ((int64_t)H * W); const int64_t n = line / ((int64_t)H * W);
           basepos = n * T * H * W + hw; }
    float* sq = smf;                  // [S][d+1]
    float* sk = sq + S * (d + 1);
    float* sv = sk + S * (d + 1);
    float* sp = sv + S * (d + 1);     // [S][S]
    const int lane = threadIdx.x;
    const int64_t rowpitch = 9 * (int64_t)C;
    const int64_t colbase = (int64_t)axis * 3 * C + head * d;
    for (int i = lane; i < S * d; i += 64) {
        const int s = i / d, e = i % d;
        const float* row = qkv + (basepos + s * stride) * rowpitch + colbase + e;
        sq[s * (d + 1) + e] = row[0];
        sk[s * (d + 1) + e] = row[C];
        sv[s * (d + 1) + e] = row[2 * C];
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)d);
    for (int i = lane; i < S * S; i += 64) {
        const int a = i / S, bb = i % S;
        float s = 0.f;
        for (int e = 0; e < d; ++e) s = fmaf(sq[a * (d + 1) + e], sk[bb * (d + 1) + e], s);
        sp[i] = s * scale;            // attn / sqrt(d)  (:591)
    }
    __syncthreads();
    for (int a = lane; a < S; a += 64) {                  // softmax rows
        float mx = -INFINITY;
        for (int j = 0; j < S; ++j) mx = fmaxf(mx, sp[a * S + j]);
        float l = 0.f;
        for (int j = 0; j < S; ++j) { const float p = expf(sp[a * S + j] - mx); sp[a * S + j] = p; l += p; }
        const float inv = 1.f / l;
        for (int j = 0; j < S; ++j) sp[a * S + j] *= inv;
    }
    __syncthreads();
    for (int i = lane; i < S * d; i += 64) {
        const int a = i / d, e = i % d;
        float o = 0.f;
        for (int j = 0; j < S; ++j) o = fmaf(sp[a * S + j], sv[j * (d + 1) + e], o);
        out[(basepos + a * stride) * (3 * (int64_t)C) + (int64_t)axis * C + head * d + e] = o;
    }
}

// ------------------------------------------------------------------ nearest code (videogpt_vq_vae.py:178-183)
// 64 z rows x 64 codes per tile, 256 threads, 4x4 register micro-tile; running per-thread arg-min, one
// 16-lane reduction at the end.  d = (|z|^2 - 2 z.e) + |e|^2 exactly in that association; first minimum wins.
constexpr int NC_T = 64;
__global__ __launch_bounds__(256) void nearest_code_kernel(const float* z, int64_t M, int E, const float* cb, int K,
                                                           int64_t* idx, float* zq) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int EP = E + 4;
    float* sz = smf;                   // [64][EP]
    float* sc = sz + NC_T * EP;        // [64][EP]
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * NC_T;
    for (int i = tid; i < NC_T * (E >> 2); i += 256) {
        const int r = i / (E >> 2), c = (i % (E >> 2)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m0 + r < M) v = *reinterpret_cast<const float4*>(z + (m0 + r) * E + c);
        *reinterpret_cast<float4*>(&sz[r * EP + c]) = v;
    }
    __syncthreads();
    float zn[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float s = 0.f;
        for (int e = 0; e < E; ++e) { const float t = sz[(ty * 4 + a) * EP + e]; s += t * t; }
        zn[a] = s;
    }
    float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int bidx[4] = {0, 0, 0, 0};
    // A 64-code chunk of the codebook is NC_T * E / 4 float4: up to NC_STG per thread, requested together and unconditionally (rows past
    // K are clamped and zeroed when staged) and one chunk ahead, so that they arrive behind the distance arithmetic.  (With the loads
    // under their bounds test inside the staging loop every chunk paid eight L2 round trips in a row -- more than its arithmetic.)
    constexpr int NC_STG = 8;                              // E <= 128 (host-checked); wider rows take the plain loop
    const int nstg = (NC_T * (E >> 2) + 255) / 256;
    const bool pipelined = nstg <= NC_STG;
    float4 stg[NC_STG];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int it = 0; it < NC_STG; ++it) {
            const int i = tid + 256 * it;
            if (it < nstg) {                               // uniform
                const int ic = i < NC_T * (E >> 2) ? i : 0;
                const int r = ic / (E >> 2), c = (ic % (E >> 2)) * 4;
                const int kr = k0 + r < K ? k0 + r : K - 1;
                stg[it] = *reinterpret_cast<const float4*>(cb + (int64_t)kr * E + c);
            }
        }
    };
    if (pipelined) load_chunk(0);
    for (int k0 = 0; k0 < K; k0 += NC_T) {
        __syncthreads();
        if (pipelined) {
#pragma unroll
            for (int it = 0; it < NC_STG; ++it) {
                const int i = tid + 256 * it;
                if (it < nstg && i < NC_T * (E >> 2)) {
                    const int r = i / (E >> 2), c = (i % (E >> 2)) * 4;
                    *reinterpret_cast<float4*>(&sc[r * EP + c]) = k0 + r < K ? stg[it] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        } else {
            for (int i = tid; i < NC_T * (E >> 2); i += 256) {
                const int r = i / (E >> 2), c = (i % (E >> 2)) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + r < K) v = *reinterpret_cast<const float4*>(cb + (int64_t)(k0 + r) * E + c);
                *reinterpret_cast<float4*>(&sc[r * EP + c]) = v;
            }
        }
        __syncthreads();
        if (pipelined && k0 + NC_T < K) load_chunk(k0 + NC_T);
        float dot[4][4], en[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) dot[a][c] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) en[c] = 0.f;
        for (int e = 0; e < E; e += 4) {
            float4 zr[4], cr[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) zr[a] = *reinterpret_cast<const float4*>(&sz[(ty * 4 + a) * EP + e]);
#pragma unroll
            for (int c = 0; c < 4; ++c) cr[c] = *reinterpret_cast<const float4*>(&sc[(tx + 16 * c) * EP + e]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                en[c] += cr[c].x * cr[c].x; en[c] += cr[c].y * cr[c].y; en[c] += cr[c].z * cr[c].z; en[c] += cr[c].w * cr[c].w;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    dot[a][c] = fmaf(zr[a].x, cr[c].x, dot[a][c]); dot[a][c] = fmaf(zr[a].y, cr[c].y, dot[a][c]);
                    dot[a][c] = fmaf(zr[a].z, cr[c].z, dot[a][c]); dot[a][c] = fmaf(zr[a].w, cr[c].w, dot[a][c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {                     // codes tx + 16c: increasing index inside a thread
            const int code = k0 + tx + 16 * c;
            if (code < K) {
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float dd = (zn[a] - 2.f * dot[a][c]) + en[c];
                    if (dd < best[a]) { best[a] = dd; bidx[a] = code; }
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float v = best[a];
        int bi = bidx[a];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o);
            const int oi = __shfl_xor(bi, o);
            if (ov < v || (ov == v && oi < bi)) { v = ov; bi = oi; }
        }
        bidx[a] = bi;
        const int64_t m = m0 + ty * 4 + a;
        if (tx == 0 && m < M) idx[m] = bi;
    }
    if (zq != nullptr) {                                  // F.embedding(encoding_indices, embeddings)  (:186)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int64_t m = m0 + ty * 4 + a;
            if (m < M)
                for (int e = tx * 4; e < E; e += 64)
                    *reinterpret_cast<float4*>(zq + m * E + e) = *reinterpret_cast<const float4*>(cb + (int64_t)bidx[a] * E + e);
        }
    }
}

// ------------------------------------------------------------------ nearest code on the matrix cores (E = 128, K % 32 == 0)
// z E^T as a GEMM on v_mfma_f32_32x32x2_f32 (exact f32: an fmaf chain, so the distances keep f32's meaning and the arg-min its
// ties) with the arg-min as the epilogue; the (M, K) distance matrix never exists.  Computed transposed -- A = a 32-code tile,
// B = 32 z rows -- so that a lane owns ONE z row and sixteen codes of the tile in increasing order: the running (best, index)
// pair is two registers per row tile and "first minimum wins" is a strict compare.
//   * z is register-resident: a wave keeps the B fragments of its 64 rows (2 x 64 VGPRs) for the whole sweep;
//   * codes stream through LDS, one 32-code x 128 tile (16 KB) per step shared by the workgroup's 4 waves (256 rows),
//     double-buffered, one barrier per tile; one ds_read_b128 feeds 4 k-steps of both row tiles (8 MFMAs);
//   * ||e||^2 comes from a K-float vector made once per call (code_norm_kernel), not once per tile;
//   * the codebook is split over `nsplit` workgroups per row block so that a 32768-row problem still fills 256 CUs; partial
//     winners meet in idx[] itself as 64-bit keys (order-preserving bits of d << 32 | code) under atomicMin, which is exactly
//     "smallest distance, then smallest index"; a last pass turns keys into indices and gathers zq.
// d = (|z|^2 - 2 z.e) + |e|^2 in that association, |z|^2 and |e|^2 summed in index order, as in nearest_code_kernel.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NCM_E = 128, NCM_EP = NCM_E + 4;

__global__ __launch_bounds__(256) void code_norm_kernel(const float* __restrict__ cb, int K, int E, float* __restrict__ en) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float s = 0.f;
    for (int e = 0; e < E; e += 4) {
        const float4 c = *reinterpret_cast<const float4*>(cb + (int64_t)k * E + e);
        s += c.x * c.x; s += c.y * c.y; s += c.z * c.z; s += c.w * c.w;
    }
    en[k] = s;
}

__global__ __launch_bounds__(256) void nearest_code_keys_init_kernel(unsigned long long* keys, int64_t M) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m < M) keys[m] = ~0ull;
}

__device__ __forceinline__ unsigned long long nc_key(float d, int code) {
    uint32_t u = __float_as_uint(d);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;                 // unsigned order == float order (negative zero below zero: harmless)
    return ((unsigned long long)u << 32) | (uint32_t)code;
}

__global__ __launch_bounds__(256, 2) void nearest_code_mfma_kernel(const float* __restrict__ z, int64_t M, const float* __restrict__ cb,
                                                                    const float* __restrict__ en, int codes_per_split, int nrb,
                                                                    unsigned long long* __restrict__ keys) {
    __shared__ __attribute__((aligned(16))) float scb[2][32][NCM_EP];
    __shared__ __attribute__((aligned(16))) float sen[2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int rb = blockIdx.x % nrb, split = blockIdx.x / nrb;
    const int64_t m0 = (int64_t)rb * 256 + wave * 64;
    const int c0 = split * codes_per_split, ntiles = codes_per_split >> 5;

    // B fragments of the wave's two row tiles: lane (row li, half lh) holds z[row][8 t + 4 lh .. + 3] for t = 0..15 -- the k-steps
    // of the t-th ds_read_b128 of a code tile contract k = 8 t + j and 8 t + 4 + j (j = 0..3)
    float4 zf[2][NCM_E / 8];
    float zn[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int64_t m = m0 + 32 * rt + li;
        const float* zr = z + (m < M ? m : M - 1) * NCM_E;
        float sacc = 0.f;                                         // |z|^2 in index order (both halves compute their row's: same value)
#pragma unroll 4
        for (int e = 0; e < NCM_E; e += 4) {
            const float4 c = *reinterpret_cast<const float4*>(zr + e);
            sacc += c.x * c.x; sacc += c.y * c.y; sacc += c.z * c.z; sacc += c.w * c.w;
        }
        zn[rt] = sacc;
    }
    __builtin_amdgcn_sched_barrier(0);                            // (keeps the 128 fragment registers below out of the loops above)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int64_t m = m0 + 32 * rt + li;
        const float* zr = z + (m < M ? m : M - 1) * NCM_E;
#pragma unroll
        for (int t = 0; t < NCM_E / 8; ++t) zf[rt][t] = *reinterpret_cast<const float4*>(zr + 8 * t + 4 * lh);
    }

    // staging: a tile is 32 codes x 32 float4 = 1024 float4, four per thread, all requested together and one tile ahead
    float4 stg[4];
    float sten = 0.f;
    auto load_tile = [&](int tile) {
        const float* src = cb + (int64_t)(c0 + 32 * tile) * NCM_E;
#pragma unroll
        for (int it = 0; it < 4; ++it) stg[it] = *reinterpret_cast<const float4*>(src + (int64_t)(tid + 256 * it) * 4);
        if (tid < 32) sten = en[c0 + 32 * tile + tid];
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = tid + 256 * it;
            *reinterpret_cast<float4*>(&scb[buf][i >> 5][(i & 31) * 4]) = stg[it];
        }
        if (tid < 32) sen[buf][tid] = sten;
    };
    float best[2] = {INFINITY, INFINITY};
    int bidx[2] = {0x7FFFFFFF, 0x7FFFFFFF};
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int buf = tile & 1;
        if (tile + 1 < ntiles) load_tile(tile + 1);
        f32x16 acc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
        // one fragment read ahead of its 8 MFMAs, pinned: left to itself the scheduler hoists all 16 reads (64 registers) and spills
        float4 a = *reinterpret_cast<const float4*>(&scb[buf][li][4 * lh]);
#pragma unroll
        for (int t = 0; t < NCM_E / 8; ++t) {
            const float4 an = *reinterpret_cast<const float4*>(&scb[buf][li][8 * (t + 1 < NCM_E / 8 ? t + 1 : t) + 4 * lh]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, zf[0][t].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, zf[1][t].x, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, zf[0][t].y, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, zf[1][t].y, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, zf[0][t].z, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, zf[1][t].z, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, zf[0][t].w, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, zf[1][t].w, acc[1], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read ...
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);      // ... then this step's eight MFMAs
            a = an;
        }
        // lane holds, of each row tile, its row li against codes 8 (r >> 2) + 4 lh + (r & 3): increasing in r
        const int cbase = c0 + 32 * tile + 4 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 e4 = *reinterpret_cast<const float4*>(&sen[buf][8 * q + 4 * lh]);
            const float ee[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int code = cbase + 8 * q + rr;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const float dd = (zn[rt] - 2.f * acc[rt][4 * q + rr]) + ee[rr];
                    if (dd < best[rt]) { best[rt] = dd; bidx[rt] = code; }
                }
            }
        }
        if (tile + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const float ob = __shfl_xor(best[rt], 32);
        const int oi = __shfl_xor(bidx[rt], 32);
        if (ob < best[rt] || (ob == best[rt] && oi < bidx[rt])) { best[rt] = ob; bidx[rt] = oi; }
        const int64_t m = m0 + 32 * rt + li;
        if (lh == 0 && m < M && bidx[rt] != 0x7FFFFFFF) atomicMin(&keys[m], nc_key(best[rt], bidx[rt]));
    }
}

__global__ __launch_bounds__(256) void nearest_code_finish_kernel(int64_t* idx, int64_t M, int E, const float* __restrict__ cb,
                                                                  float* __restrict__ zq) {
    // 16 threads per row: key -> index (a row whose distances were all NaN never won a compare: index 0, as the VALU kernel), zq gather
    const int64_t m = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (m >= M) return;
    const unsigned long long key = reinterpret_cast<const unsigned long long*>(idx)[m];
    const int code = key == ~0ull ? 0 : (int)(uint32_t)key;
    __syncthreads();
    if ((threadIdx.x & 15) == 0) idx[m] = code;
    if (zq != nullptr)
        for (int e = (threadIdx.x & 15) * 4; e < E; e += 64)
            *reinterpret_cast<float4*>(zq + m * E + e) = *reinterpret_cast<const float4*>(cb + (int64_t)code * E + e);
}

// ------------------------------------------------------------------ 3-D pooling on channels-last rows (I3D: pytorch_i3d.py:7-34, 296)
// out[n][to][ho][wo][c] = max / mean over the (kt, kh, kw) window starting at (to st - pt, ho sh - ph, wo sw - pw).  Positions outside
// the input count as ZERO in the maximum: MaxPool3dSamePadding pads with F.pad's zeros and then pools with padding = 0, so a border
// window of all-negative values yields 0, not their maximum.  The mean divides by the full window (AvgPool3d's count_include_pad
// default; the reference only uses it unpadded).  One thread per (output position, 4 channels): 16-byte loads along c.
struct PoolArgs {
    const float* in; float* out;
    int N, Di, Hi, Wi, C, in_pitch;
    int Do, Ho, Wo, out_pitch;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;
    int mode;                        // 0 max, 1 mean
};
__global__ __launch_bounds__(256) void pool3d_kernel(const PoolArgs a) {
    const int c4n = a.C >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)a.N * a.Do * a.Ho * a.Wo * c4n;
    if (i >= total) return;
    const int c4 = (int)(i % c4n);
    int64_t r = i / c4n;
    const int wo = (int)(r % a.Wo); r /= a.Wo;
    const int ho = (int)(r % a.Ho); r /= a.Ho;
    const int to = (int)(r % a.Do);
    const int n = (int)(r / a.Do);
    const bool mx = a.mode == 0;
    float4 acc = mx ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0.f, 0.f, 0.f, 0.f);
    bool padded = false;
    for (int dt = 0; dt < a.kt; ++dt) {
        const int ti = to * a.st - a.pt + dt;
        for (int dh = 0; dh < a.kh; ++dh) {
            const int hi = ho * a.sh - a.ph + dh;
            for (int dw = 0; dw < a.kw; ++dw) {
                const int wi = wo * a.sw - a.pw + dw;
                if (ti < 0 || ti >= a.Di || hi < 0 || hi >= a.Hi || wi < 0 || wi >= a.Wi) { padded = true; continue; }
                const float4 v = *reinterpret_cast<const float4*>(a.in + ((((int64_t)n * a.Di + ti) * a.Hi + hi) * a.Wi + wi) * a.in_pitch + 4 * c4);
                if (mx) { acc.x = fmaxf(acc.x, v.x); acc.y = fmaxf(acc.y, v.y); acc.z = fmaxf(acc.z, v.z); acc.w = fmaxf(acc.w, v.w); }
                else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            }
        }
    }
    if (mx) {
        if (padded) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    } else {
        const float inv = 1.f / (float)(a.kt * a.kh * a.kw);
        acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
    }
    *reinterpret_cast<float4*>(a.out + ((((int64_t)n * a.Do + to) * a.Ho + ho) * a.Wo + wo) * a.out_pitch + 4 * c4) = acc;
}

// ------------------------------------------------------------------ clip preprocessing (ucf101_dataset.py:105-140)
// uint8 THWC frames -> normalised (x/255 - mean)/std, bilinear resize of the shorter side to R (align_corners = false, PyTorch's
// source-index rule with one rounding: src = max(fma(in/out, dst + 0.5, -0.5), 0)), centre crop, CTHW float32.  One thread per
// output pixel, three channels each; stores are coalesced along x in each channel plane.
__global__ __launch_bounds__(256) void preprocess_clip_kernel(const uint8_t* __restrict__ video, int N, int T, int H, int W, int t_out,
                                                              int th, int tw, int h_start, int w_start, int R,
                                                              float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)N * t_out * R * R;
    if (i >= total) return;
    const int x = (int)(i % R), y = (int)((i / R) % R);
    const int t = (int)((i / ((int64_t)R * R)) % t_out), n = (int)(i / ((int64_t)R * R * t_out));
    const float sy = (float)H / (float)th, sx = (float)W / (float)tw;
    const float fy = fmaxf(fmaf(sy, (float)(y + h_start) + 0.5f, -0.5f), 0.f);
    const float fx = fmaxf(fmaf(sx, (float)(x + w_start) + 0.5f, -0.5f), 0.f);
    int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    y0 = y0 < H - 1 ? y0 : H - 1;
    x0 = x0 < W - 1 ? x0 : W - 1;
    const int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const uint8_t* frame = video + ((int64_t)n * T + t) * H * W * 3;
    const uint8_t* p00 = frame + ((int64_t)y0 * W + x0) * 3;
    const uint8_t* p01 = frame + ((int64_t)y0 * W + x1) * 3;
    const uint8_t* p10 = frame + ((int64_t)y1 * W + x0) * 3;
    const uint8_t* p11 = frame + ((int64_t)y1 * W + x1) * 3;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    const int64_t plane = (int64_t)t_out * R * R;
    float* o = out + (int64_t)n * 3 * plane + ((int64_t)t * R + y) * R + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a00 = ((float)p00[c] / 255.f - mean[c]) / stdv[c], a01 = ((float)p01[c] / 255.f - mean[c]) / stdv[c];
        const float a10 = ((float)p10[c] / 255.f - mean[c]) / stdv[c], a11 = ((float)p11[c] / 255.f - mean[c]) / stdv[c];
        const float top = (1.f - lx) * a00 + lx * a01, bot = (1.f - lx) * a10 + lx * a11;
        o[c * plane] = (1.f - ly) * top + ly * bot;
    }
}

}  // namespace gsdd

using namespace gsdd;

extern "C" int gsdd_d3pm_embed(const int64_t* tok, int B, int L, int D, const float* emb, int n_embed, const float* pos,
                               int rep, float* x, void* stream) {
    GSDD_CHECK_ARG(tok && emb && pos && x, "null pointer");
    GSDD_CHECK_ARG(B > 0 && L > 0 && D > 0 && D % 4 == 0 && n_embed > 0 && rep >= 1, "bad sizes");
    const int64_t rows = (int64_t)B * L, n = rows * (D / 4);
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tok, rows, L, D,
                       emb, n_embed, pos, rep, x);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_adaln_table(const float* emb, int T, int D, const float* lin_w, const float* lin_b, float* out,
                                void* stream) {
    GSDD_CHECK_ARG(emb && lin_w && lin_b && out && T > 0 && D > 0, "bad args");
    const int n = T * 2 * D;
    hipLaunchKernelGGL(adaln_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, emb, T, D, lin_w,
                       lin_b, out);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_small_linear(const float* x, int R, int Cin, const float* w, const float* b, int Cout, float* y,
                                 void* stream) {
    GSDD_CHECK_ARG(x && w && y && R > 0 && Cin > 0 && Cout > 0, "bad args");
    const int n = R * Cout;
    hipLaunchKernelGGL(small_linear_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, R, Cin, w, b, Cout,
                       y);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_axial_attention(const float* qkv, int N, int T, int H, int W, int C, int n_head, float* out,
                                    int variant, void* stream) {
    GSDD_CHECK_ARG(qkv && out, "null pointer");
    GSDD_CHECK_ARG(variant == GSDD_AXIAL_AUTO || variant == GSDD_AXIAL_VALU, "variant: GSDD_AXIAL_AUTO or GSDD_AXIAL_VALU");
    GSDD_CHECK_ARG(N > 0 && T > 0 && H > 0 && W > 0 && C > 0 && n_head > 0 && C % n_head == 0, "bad sizes");
    GSDD_CHECK_ARG(T <= 64 && H <= 64 && W <= 64, "axis length > 64 unsupported");
    const int d = C / n_head;
    hipStream_t st = (hipStream_t)stream;
    const int axes_len[3] = {W, H, T};
    const int64_t pos = (int64_t)N * T * H * W;
    const bool force_valu = variant == GSDD_AXIAL_VALU;
    for (int axis = 0; axis < 3; ++axis) {
        if (!force_valu && axial_attention_mfma_launch(qkv, N, T, H, W, C, n_head, axis, out, st)) {
            GSDD_CHECK_LAUNCH();
            continue;
        }
        const int S = axes_len[axis];
        const size_t lds = (size_t)(3 * S * (d + 1) + S * S) * sizeof(float);
        GSDD_CHECK_ARG(lds <= 64 * 1024, "line does not fit LDS");
        hipLaunchKernelGGL(axial_attention_kernel, dim3((unsigned)(pos / S), n_head, 1), dim3(64), lds, st, qkv, N, T, H, W,
                           C, n_head, axis, out);
        GSDD_CHECK_LAUNCH();
    }
    return GSDD_OK;
}

extern "C" int64_t gsdd_nearest_code_workspace_bytes(int K) { return (int64_t)K * 4; }

extern "C" int gsdd_nearest_code(const float* z, int64_t M, int E, const float* cb, int K, int64_t* idx, float* zq,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
    GSDD_CHECK_ARG(z && cb && idx, "null pointer");
    GSDD_CHECK_ARG(M > 0 && K > 0 && E > 0 && E % 4 == 0 && E <= 256, "E must be a multiple of 4, <= 256");
    hipStream_t st = (hipStream_t)stream;
    if (E == NCM_E && K % 32 == 0 && workspace != nullptr) {          // (workspace == NULL: the caller asks for the vector kernel)
        GSDD_CHECK_ARG(workspace_bytes >= gsdd_nearest_code_workspace_bytes(K), "workspace too small");
        GSDD_CHECK_ARG(M < (1ll << 31) * 16, "too many rows");
        float* en = reinterpret_cast<float*>(workspace);
        unsigned long long* keys = reinterpret_cast<unsigned long long*>(idx);
        const int nrb = (int)((M + 255) / 256);
        int nsplit = 1;                                        // enough workgroups for two rounds of 2 per CU, tiles of 32 codes
        while (nrb * nsplit < 1024 && nsplit * 2 <= K / 32 && (K / (nsplit * 2)) % 32 == 0) nsplit *= 2;
        hipLaunchKernelGGL(code_norm_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, st, cb, K, E, en);
        hipLaunchKernelGGL(nearest_code_keys_init_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, keys, M);
        hipLaunchKernelGGL(nearest_code_mfma_kernel, dim3((unsigned)(nrb * nsplit)), dim3(256), 0, st, z, M, cb, en, K / nsplit, nrb, keys);
        hipLaunchKernelGGL(nearest_code_finish_kernel, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, st, idx, M, E, cb, zq);
        GSDD_CHECK_LAUNCH();
        return GSDD_OK;
    }
    const size_t lds = (size_t)2 * NC_T * (E + 4) * sizeof(float);
    hipLaunchKernelGGL(nearest_code_kernel, dim3((unsigned)((M + NC_T - 1) / NC_T)), dim3(256), lds, st, z,
                       M, E, cb, K, idx, zq);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_pool3d(const float* in, int N, int Di, int Hi, int Wi, int C, int in_pitch, const int* kernel, const int* stride,
                           const int* pad_front, int Do, int Ho, int Wo, int mode, float* out, int out_pitch, void* stream) {
    GSDD_CHECK_ARG(in && out && kernel && stride && pad_front, "null pointer");
    GSDD_CHECK_ARG(N > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0, "bad sizes");
    GSDD_CHECK_ARG(C > 0 && C % 4 == 0 && in_pitch % 4 == 0 && out_pitch % 4 == 0 && in_pitch >= C && out_pitch >= C, "C and the pitches must be multiples of 4");
    GSDD_CHECK_ARG(mode == 0 || mode == 1, "mode: 0 max, 1 mean");
    for (int i = 0; i < 3; ++i) GSDD_CHECK_ARG(kernel[i] > 0 && stride[i] > 0 && pad_front[i] >= 0, "bad window");
    PoolArgs a;
    a.in = in; a.out = out; a.N = N; a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.C = C; a.in_pitch = in_pitch;
    a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.out_pitch = out_pitch;
    a.kt = kernel[0]; a.kh = kernel[1]; a.kw = kernel[2]; a.st = stride[0]; a.sh = stride[1]; a.sw = stride[2];
    a.pt = pad_front[0]; a.ph = pad_front[1]; a.pw = pad_front[2]; a.mode = mode;
    // every window must start inside the (virtually padded) input and touch at least one real position
    GSDD_CHECK_ARG((Do - 1) * a.st - a.pt < Di && (Ho - 1) * a.sh - a.ph < Hi && (Wo - 1) * a.sw - a.pw < Wi, "output grid outside the input");
    const int64_t total = (int64_t)N * Do * Ho * Wo * (C / 4);
    GSDD_CHECK_ARG(total < (1ll << 31) * 256, "too many outputs");
    hipLaunchKernelGGL(pool3d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_preprocess_clip(const uint8_t* video, int N, int T, int H, int W, int t_out, int th, int tw, int h_start,
                                    int w_start, int R, float* out, void* stream) {
    GSDD_CHECK_ARG(video && out, "null pointer");
    GSDD_CHECK_ARG(N > 0 && T > 0 && H > 0 && W > 0 && R > 0 && t_out > 0 && t_out <= T, "bad sizes");
    GSDD_CHECK_ARG(th >= R && tw >= R && h_start >= 0 && w_start >= 0 && h_start + R <= th && w_start + R <= tw, "crop outside the resized frame");
    const int64_t total = (int64_t)N * t_out * R * R;
    hipLaunchKernelGGL(preprocess_clip_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, video, N, T, H,
                       W, t_out, th, tw, h_start, w_start, R, out);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}
