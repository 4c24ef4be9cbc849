// Backward of the head-dim-4 self-attention on the bf16 matrix pipe (training step of the D3PM denoiser).
//
// Same formulation as the forward kernel (d3pm_attention.hip): scores are recomputed tile by tile from error-free bf16
// splits and never leave registers.  With s = c q.k - lse (log2 domain, lse from the forward), p = 2^s, dp = dO.v,
// delta = dO.O:   dS = p (dp - delta),  dQ = 1/2 dS K,  dK = 1/2 dS^T Q,  dV = P^T dO.
//   * both K=4 contractions (q.k and dO.v) are ONE v_mfma_f32_16x16x32_bf16 each: the six significant cross products of the
//     3-way splits along K = 32, plus three slots carrying 1 * (-lse) resp. 1 * (-delta), so the tile comes out as s and
//     dp - delta directly;
//   * P and dS (f32 in registers) are split hi + lo in bf16 (16 significant bits) and multiplied with the 3-way bf16 splits of
//     dO / K / Q on the matrix pipe (products exact in the f32 accumulator), like P.V in the forward.
// The shipped path is ONE fused kernel (a wave owns 64 keys, queries stream through LDS, dS crosses an LDS transpose for dQ; section
// "fused backward" below) fed by attn_bwd_prep_fused_kernel, which pre-splits only the two score-row images of Q and dO (96 B per
// (row, head)): the accumulate-side fragments are transposed LDS reads of those images, the key-side images are built by the waves that
// own the keys.  GSDD_ATTN_BWD_SPLIT selects the older pair -- dQ (a wave owns 64 queries, keys stream) and dK/dV (a wave owns 64 keys,
// queries stream) -- with all seven operand images pre-split by attn_bwd_prep_kernel (256 B per (row, head)).
//
// Reference semantics: autograd of FullAttention.forward (transformer_utils.py:46-62).
#include <stdlib.h>

#include "common.hpp"

namespace gsdd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BKC = 256;           // rows (keys or queries) per LDS chunk

__device__ __forceinline__ uint32_t bw_bf16_rn(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void bw_split3(float x, uint32_t& a, uint32_t& b, uint32_t& c) {
    a = bw_bf16_rn(x);
    const float r = x - __uint_as_float(a << 16);
    b = bw_bf16_rn(r);
    const float r2 = r - __uint_as_float(b << 16);
    c = bw_bf16_rn(r2);
}
__device__ __forceinline__ uint4 bw_pack8(const uint32_t (&lo)[4], const uint32_t (&hi)[4]) {
    return make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
}
__device__ __forceinline__ bf16x8 bw_frag(uint4 u) {
    union { uint4 u; bf16x8 v; } c;
    c.u = u;
    return c.v;
}
// x as three bf16 pieces in contraction slots 24..26 (lane group 3's fragment); the facing fragment holds ones there
__device__ __forceinline__ uint4 bw_slot_frag(float x) {
    uint32_t a, b, c;
    bw_split3(x, a, b, c);
    return make_uint4(a | (b << 16), c, 0u, 0u);
}
__device__ __forceinline__ uint4 bw_ones_frag() { return make_uint4(0x3F803F80u, 0x00003F80u, 0u, 0u); }

// "row" operand of a score MFMA (the side that comes from LDS): pieces A = [x1|x2] (faces [y1|y1] and [y2|y2]) and
// B = [x3|x1] (faces [y1|y3]).  "col" operand (the side a wave keeps in registers), per lane group: [y1|y1] [y2|y2] [y1|y3] extra
__device__ __forceinline__ void bw_row_pieces(const float (&x)[4], uint4& pa, uint4& pb) {
    uint32_t x1[4], x2[4], x3[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) bw_split3(x[d], x1[d], x2[d], x3[d]);
    pa = bw_pack8(x1, x2);
    pb = bw_pack8(x3, x1);
}
__device__ __forceinline__ uint4 bw_col_frag(const float (&y)[4], int lg, uint4 extra) {
    uint32_t y1[4], y2[4], y3[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) bw_split3(y[d], y1[d], y2[d], y3[d]);
    const uint4 f0 = bw_pack8(y1, y1), f1 = bw_pack8(y2, y2), f2 = bw_pack8(y1, y3);
    return lg == 0 ? f0 : (lg == 1 ? f1 : (lg == 2 ? f2 : extra));
}
// accumulate-side image ("V format"): per 32-row pair-tile [row group g][col j] -> 8 bf16 (tile0 rows 4g+r, tile1 rows 4g+r),
// cols = [x1 | x2 | x3 | 0]
__device__ __forceinline__ void bw_store_vformat(const float (&x)[4], int64_t row, uint4* img) {
    uint16_t col[16];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t a, b, c;
        bw_split3(x[d], a, b, c);
        col[d] = (uint16_t)a; col[4 + d] = (uint16_t)b; col[8 + d] = (uint16_t)c; col[12 + d] = 0;
    }
    const int64_t pair = row >> 5;
    const int kk = (int)(row & 31), th = kk >> 4, kt = kk & 15, g = kt >> 2, r = kt & 3;
    uint16_t* dst = reinterpret_cast<uint16_t*>(img + (pair * 4 + g) * 16) + 4 * th + r;
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[j * 8] = col[j];
}

// Split 8 f32 values into packed bf16 hi / lo fragments: hi = bf16(x) (RNE), lo = bf16(x - hi).  One asm block ending in
// s_nop 1 (hipcc does not pad the VALU-write -> MFMA-read hazard for registers written inside inline asm).  Not `volatile`: the
// block is a pure function of its inputs, and letting the scheduler move it is worth 9 % in the fused backward (2.17 -> 1.97 ms);
// the forward kernel's split helpers stay volatile (there the freedom costs 4 %).
__device__ __forceinline__ void bw_split8(const float (&x)[8], uint4& hi, uint4& lo) {
    float t0, t1, t2, t3, t4, t5, t6, t7;
    asm(
        "v_cvt_pk_bf16_f32 %0, %16, %17\n\t"
        "v_cvt_pk_bf16_f32 %1, %18, %19\n\t"
        "v_cvt_pk_bf16_f32 %2, %20, %21\n\t"
        "v_cvt_pk_bf16_f32 %3, %22, %23\n\t"
        "v_lshlrev_b32 %8, 16, %0\n\t"
        "v_and_b32 %9, 0xffff0000, %0\n\t"
        "v_lshlrev_b32 %10, 16, %1\n\t"
        "v_and_b32 %11, 0xffff0000, %1\n\t"
        "v_lshlrev_b32 %12, 16, %2\n\t"
        "v_and_b32 %13, 0xffff0000, %2\n\t"
        "v_lshlrev_b32 %14, 16, %3\n\t"
        "v_and_b32 %15, 0xffff0000, %3\n\t"
        "v_sub_f32 %8, %16, %8\n\t"
        "v_sub_f32 %9, %17, %9\n\t"
        "v_sub_f32 %10, %18, %10\n\t"
        "v_sub_f32 %11, %19, %11\n\t"
        "v_sub_f32 %12, %20, %12\n\t"
        "v_sub_f32 %13, %21, %13\n\t"
        "v_sub_f32 %14, %22, %14\n\t"
        "v_sub_f32 %15, %23, %15\n\t"
        "v_cvt_pk_bf16_f32 %4, %8, %9\n\t"
        "v_cvt_pk_bf16_f32 %5, %10, %11\n\t"
        "v_cvt_pk_bf16_f32 %6, %12, %13\n\t"
        "v_cvt_pk_bf16_f32 %7, %14, %15\n\t"
        "s_nop 1"
        : "=&v"(hi.x), "=&v"(hi.y), "=&v"(hi.z), "=&v"(hi.w), "=&v"(lo.x), "=&v"(lo.y), "=&v"(lo.z), "=&v"(lo.w),
          "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
}

// The same split in plain C (the fused kernel): hipcc emits v_cvt_pk_bf16_f32 / v_lshlrev / v_and / v_pk_add_f32 / v_cvt_pk_bf16_f32
// for it, pads the hazards itself, and -- unlike an asm block -- can interleave the pieces with the neighbouring matrix instructions.
typedef __bf16 bw_bf2 __attribute__((ext_vector_type(2)));
typedef float bw_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bw_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const bw_f2 x = {a, b};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bw_bf2));
    const bw_f2 r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bw_bf2));
}
__device__ __forceinline__ void bw_split8c(const float (&x)[8], uint4& hi, uint4& lo) {
    bw_split2(x[0], x[1], hi.x, lo.x);
    bw_split2(x[2], x[3], hi.y, lo.y);
    bw_split2(x[4], x[5], hi.z, lo.z);
    bw_split2(x[6], x[7], hi.w, lo.w);
}

struct BwdImages {                 // all indexed by the head-major row h*M + b*L + i
    uint4* kp;                     // [rows][2]  score-row image of K (slots swapped for (row&15) >= 8, as the forward's)
    uint4* vk;                     // [rows][2]  score-row image of V
    uint4* kv;                     // [rows/32][4][16] accumulate image of K
    uint4* qp;                     // [rows][3]  score-row image of c*Q, third piece = -lse
    uint4* gp;                     // [rows][3]  score-row image of dO, third piece = -delta
    uint4* qv;                     // accumulate image of Q
    uint4* gv;                     // accumulate image of dO
};

__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, const float* __restrict__ o,
                                                            const float* __restrict__ dO, const float* __restrict__ lse, int64_t M,
                                                            int H, BwdImages im, bool score_kv) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;      // h*M + m
    if (row >= M * H) return;
    const int h = (int)(row / M);
    const int64_t m = row - (int64_t)h * M;
    const float4 rq = *reinterpret_cast<const float4*>(q + row * 4);
    float4 rk = make_float4(0.f, 0.f, 0.f, 0.f), rv = rk;          // K and V images: the two-kernel variant only (the fused kernel
    if (score_kv) {                                                   // splits its own keys)
        rk = *reinterpret_cast<const float4*>(k + row * 4);
        rv = *reinterpret_cast<const float4*>(v + row * 4);
    }
    const float4 rg = *reinterpret_cast<const float4*>(dO + m * (H * 4) + h * 4);
    const float4 ro = *reinterpret_cast<const float4*>(o + m * (H * 4) + h * 4);
    const float delta = (rg.x * ro.x + rg.y * ro.y) + (rg.z * ro.z + rg.w * ro.w);
    const float c = 0.5f * 1.4426950408889634f;
    const float qs[4] = {rq.x * c, rq.y * c, rq.z * c, rq.w * c}, qr[4] = {rq.x, rq.y, rq.z, rq.w};
    const float ks[4] = {rk.x, rk.y, rk.z, rk.w}, vs[4] = {rv.x, rv.y, rv.z, rv.w}, gs[4] = {rg.x, rg.y, rg.z, rg.w};
    const int sw = (int)((row >> 3) & 1);
    uint4 pa, pb;
    if (score_kv) {          // score-row images of K and V: only the dQ kernel of the two-kernel variant reads them
        bw_row_pieces(ks, pa, pb);
        im.kp[row * 2 + sw] = pa; im.kp[row * 2 + (sw ^ 1)] = pb;
        bw_row_pieces(vs, pa, pb);
        im.vk[row * 2 + sw] = pa; im.vk[row * 2 + (sw ^ 1)] = pb;
    }
    bw_row_pieces(qs, pa, pb);
    im.qp[row * 3 + 0] = pa; im.qp[row * 3 + 1] = pb; im.qp[row * 3 + 2] = bw_slot_frag(-lse[row]);
    bw_row_pieces(gs, pa, pb);
    im.gp[row * 3 + 0] = pa; im.gp[row * 3 + 1] = pb; im.gp[row * 3 + 2] = bw_slot_frag(-delta);
    if (score_kv) {          // accumulate images: the fused kernel reads them out of the score-row images (transposed LDS reads)
        bw_store_vformat(ks, row, im.kv);
        bw_store_vformat(qr, row, im.qv);
        bw_store_vformat(gs, row, im.gv);
    }
}

// The fused kernel's pre-split: only the two score-row images.  A block takes 16 consecutive rows m of all 16 heads: dO and O are read
// as the row-major [m][h][4] arrays they are (16 threads cover one row's 256 bytes; the per-head kernel above reads 16 of every 256
// bytes per wave and fetched 2.9x the bytes), handed over through LDS, and the images are written head-major, 16 rows = 768
// contiguous bytes per head.  H == 16, M % 16 == 0.
__global__ __launch_bounds__(256) void attn_bwd_prep_fused_kernel(const float* __restrict__ q, const float* __restrict__ o,
                                                                  const float* __restrict__ dO, const float* __restrict__ lse, int64_t M,
                                                                  BwdImages im) {
    __shared__ float4 gs_s[16][17];
    __shared__ float dl_s[16][17];
    const int t = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * 16;
    {
        const int ml = t >> 4, h = t & 15;
        const float4 rg = *reinterpret_cast<const float4*>(dO + (m0 + ml) * 64 + h * 4);
        const float4 ro = *reinterpret_cast<const float4*>(o + (m0 + ml) * 64 + h * 4);
        gs_s[h][ml] = rg;
        dl_s[h][ml] = (rg.x * ro.x + rg.y * ro.y) + (rg.z * ro.z + rg.w * ro.w);
    }
    __syncthreads();
    const int h = t >> 4, ml = t & 15;
    const int64_t row = (int64_t)h * M + m0 + ml;
    const float4 rq = *reinterpret_cast<const float4*>(q + row * 4);
    const float4 rg = gs_s[h][ml];
    const float delta = dl_s[h][ml];
    const float c = 0.5f * 1.4426950408889634f;
    const float qs[4] = {rq.x * c, rq.y * c, rq.z * c, rq.w * c}, gs[4] = {rg.x, rg.y, rg.z, rg.w};
    uint4 pa, pb;
    bw_row_pieces(qs, pa, pb);
    im.qp[row * 3 + 0] = pa; im.qp[row * 3 + 1] = pb; im.qp[row * 3 + 2] = bw_slot_frag(-lse[row]);
    bw_row_pieces(gs, pa, pb);
    im.gp[row * 3 + 0] = pa; im.gp[row * 3 + 1] = pb; im.gp[row * 3 + 2] = bw_slot_frag(-delta);
}

__device__ __forceinline__ unsigned bw_xcd_remap(unsigned wg, unsigned nwg) {
    const unsigned q8 = nwg / 8, r8 = nwg % 8, xcd = wg % 8, idx = wg / 8;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
}

// ------------------------------------------------------------------ dQ: a wave owns 64 queries, keys stream through LDS
struct DqSmem {
    uint4 k[2][BKC][2];
    uint4 v[2][BKC][2];
    uint4 kv[2][BKC / 32][4][16];
    uint4 ones[16];
};

__global__ __launch_bounds__(256, 2) void attn_bwd_dq_mfma_kernel(const float* __restrict__ q, const float* __restrict__ o,
                                                               const float* __restrict__ dO, const float* __restrict__ lse,
                                                               BwdImages im, int B, int L, int H, float* __restrict__ dqkv) {
    __shared__ DqSmem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nqb = (L + 255) / 256;
    const unsigned wg = bw_xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = wg % nqb;
    const int h = (wg / nqb) % H, b = wg / (nqb * H);
    const int64_t M = (int64_t)B * L;
    const int64_t hrow0 = (int64_t)h * M + (int64_t)b * L;            // first head-major row of this (b,h)
    const int li = lane & 15, lg = lane >> 4;
    const int q0 = qblk * 256 + wave * 64;

    if (tid < 16) sm.ones[tid] = (tid == 0 || tid == 1) ? bw_ones_frag() : make_uint4(0u, 0u, 0u, 0u);

    // register operands of this wave's queries: c*q with -lse, dO with -delta (delta = dO . O)
    const float c = 0.5f * 1.4426950408889634f;
    uint4 qfrag[4], gfrag[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int qi = q0 + 16 * j + li;
        qi = qi < L ? qi : L - 1;
        const float4 qv = *reinterpret_cast<const float4*>(q + (hrow0 + qi) * 4);
        const int64_t row = (int64_t)b * L + qi;
        const float4 g = *reinterpret_cast<const float4*>(dO + row * (H * 4) + h * 4);
        const float4 ov = *reinterpret_cast<const float4*>(o + row * (H * 4) + h * 4);
        const float delta = (g.x * ov.x + g.y * ov.y) + (g.z * ov.z + g.w * ov.w);
        const float qs[4] = {qv.x * c, qv.y * c, qv.z * c, qv.w * c}, gs[4] = {g.x, g.y, g.z, g.w};
        qfrag[j] = bw_col_frag(qs, lg, bw_slot_frag(-lse[hrow0 + qi]));
        gfrag[j] = bw_col_frag(gs, lg, bw_slot_frag(-delta));
    }

    const int nchunks = (L + BKC - 1) / BKC;
    uint4 r0, r1, r2, r3, r4, r5;
    auto load_chunk = [&](int ch) {
        // unconditional loads from a clamped index (a "cond ? p[i] : zero" load turns into a flat load, which also counts on
        // lgkmcnt and stalls the chunk's first LDS wait); rows past a short last chunk are staged but never read
        const int last = 2 * min(BKC, L - ch * BKC) - 1;              // keys: a multiple of 32
        const uint4* ks = im.kp + (hrow0 + (int64_t)ch * BKC) * 2;
        const uint4* vs = im.vk + (hrow0 + (int64_t)ch * BKC) * 2;
        const uint4* kvs = im.kv + ((hrow0 + (int64_t)ch * BKC) >> 5) * 64;
        const int i0 = min(tid, last), i1 = min(tid + 256, last);
        r0 = ks[i0]; r1 = ks[i1];
        r2 = vs[i0]; r3 = vs[i1];
        r4 = kvs[i0]; r5 = kvs[i1];
    };
    auto store_chunk = [&](int buf) {
        uint4* kd = &sm.k[buf][0][0];
        uint4* vd = &sm.v[buf][0][0];
        uint4* kvd = &sm.kv[buf][0][0][0];
        kd[tid] = r0; kd[tid + 256] = r1;
        vd[tid] = r2; vd[tid + 256] = r3;
        kvd[tid] = r4; kvd[tid + 256] = r5;
    };

    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j][0] = 0.f; acc[j][1] = 0.f; acc[j][2] = 0.f; acc[j][3] = 0.f; }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    const int slot = ((lg >> 1) ^ (li >> 3)) & 1;
    const int rstep = (lg == 3) ? 0 : 32;                              // uint4 stride between consecutive 16-row tiles
    const int rbuf = (lg == 3) ? 0 : BKC * 2;
    const uint4* kbase0 = (lg == 3) ? &sm.ones[(li >= 4 && li < 12) ? 0 : 1] : &sm.k[0][li][slot];
    const uint4* vbase0 = (lg == 3) ? &sm.ones[(li >= 4 && li < 12) ? 0 : 1] : &sm.v[0][li][slot];

    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) load_chunk(ch + 1);
        const int npairs = min(BKC, L - ch * BKC) >> 5;
        const uint4* kb = kbase0 + buf * rbuf;
        const uint4* vb = vbase0 + buf * rbuf;
        for (int u = 0; u < npairs; ++u) {
            const bf16x8 kf0 = bw_frag(kb[(2 * u) * rstep]), kf1 = bw_frag(kb[(2 * u + 1) * rstep]);
            const bf16x8 vf0 = bw_frag(vb[(2 * u) * rstep]), vf1 = bw_frag(vb[(2 * u + 1) * rstep]);
            const bf16x8 kvb = bw_frag(sm.kv[buf][u][lg][li]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0, bw_frag(qfrag[j]), zero, 0, 0, 0);
                const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1, bw_frag(qfrag[j]), zero, 0, 0, 0);
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf0, bw_frag(gfrag[j]), zero, 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf1, bw_frag(gfrag[j]), zero, 0, 0, 0);
                float ds[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ds[r] = __builtin_amdgcn_exp2f(s0[r]) * d0[r];
                    ds[4 + r] = __builtin_amdgcn_exp2f(s1[r]) * d1[r];
                }
                uint4 hi, lo;
                bw_split8(ds, hi, lo);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw_frag(hi), kvb, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw_frag(lo), kvb, acc[j], 0, 0, 0);
            }
        }
        if (ch + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // epilogue: D[query 4lg+r][col li], cols = [k1 | k2 | k3] sums: dq_d = 1/2 (D[d] + D[4+d] + D[8+d])
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = acc[j][r];
            const int rowbase = lane & 48;
            const float a1 = __shfl(a, rowbase + (li & 3) + 4);
            const float a2 = __shfl(a, rowbase + (li & 3) + 8);
            const int qi = q0 + 16 * j + 4 * lg + r;
            if (li < 4 && qi < L) dqkv[((int64_t)b * L + qi) * (3 * H * 4) + h * 4 + li] = 0.5f * ((a + a1) + a2);
        }
    }
}

// ------------------------------------------------------------------ dK, dV: a wave owns 64 keys, queries stream through LDS
struct DkvSmem {
    uint4 q[2][BKC][3];
    uint4 g[2][BKC][3];
    uint4 qv[2][BKC / 32][4][16];
    uint4 gv[2][BKC / 32][4][16];
};

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_mfma_kernel(const float* __restrict__ k, const float* __restrict__ v,
                                                                BwdImages im, int B, int L, int H, float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) char dkv_raw[];
    DkvSmem& sm = *reinterpret_cast<DkvSmem*>(dkv_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nkb = (L + 255) / 256;
    const unsigned wg = bw_xcd_remap(blockIdx.x, gridDim.x);
    const int kblk = wg % nkb;
    const int h = (wg / nkb) % H, b = wg / (nkb * H);
    const int64_t M = (int64_t)B * L;
    const int64_t hrow0 = (int64_t)h * M + (int64_t)b * L;
    const int li = lane & 15, lg = lane >> 4;
    const int k0 = kblk * 256 + wave * 64;

    // register operands of this wave's keys: [k1|k1] [k2|k2] [k1|k3] ones  and the same for v
    uint4 kfrag[4], vfrag[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ki = k0 + 16 * j + li;
        ki = ki < L ? ki : L - 1;
        const float4 kk = *reinterpret_cast<const float4*>(k + (hrow0 + ki) * 4);
        const float4 vv = *reinterpret_cast<const float4*>(v + (hrow0 + ki) * 4);
        const float ks[4] = {kk.x, kk.y, kk.z, kk.w}, vs[4] = {vv.x, vv.y, vv.z, vv.w};
        kfrag[j] = bw_col_frag(ks, lg, bw_ones_frag());
        vfrag[j] = bw_col_frag(vs, lg, bw_ones_frag());
    }

    const int nchunks = (L + BKC - 1) / BKC;
    // staging registers are named scalars on purpose: an array captured by the lambdas is demoted to scratch memory (and the
    // chunk's global loads are then waited for on the spot)
    uint4 rq0, rq1, rq2, rg0, rg1, rg2, rqv0, rqv1, rgv0, rgv1;
    auto load_chunk = [&](int ch) {
        const int rows = min(BKC, L - ch * BKC);                      // multiple of 32
        const uint4* qs = im.qp + (hrow0 + (int64_t)ch * BKC) * 3;
        const uint4* gs = im.gp + (hrow0 + (int64_t)ch * BKC) * 3;
        const uint4* qvs = im.qv + ((hrow0 + (int64_t)ch * BKC) >> 5) * 64;
        const uint4* gvs = im.gv + ((hrow0 + (int64_t)ch * BKC) >> 5) * 64;
        // unconditional, clamped (see the dQ kernel)
        const int l3 = rows * 3 - 1, l2 = rows * 2 - 1;
        const int a0 = min(tid, l3), a1 = min(tid + 256, l3), a2 = min(tid + 512, l3);
        const int b0 = min(tid, l2), b1 = min(tid + 256, l2);
        rq0 = qs[a0]; rq1 = qs[a1]; rq2 = qs[a2];
        rg0 = gs[a0]; rg1 = gs[a1]; rg2 = gs[a2];
        rqv0 = qvs[b0]; rqv1 = qvs[b1];
        rgv0 = gvs[b0]; rgv1 = gvs[b1];
    };
    auto store_chunk = [&](int buf) {
        uint4* qd = &sm.q[buf][0][0];
        uint4* gd = &sm.g[buf][0][0];
        uint4* qvd = &sm.qv[buf][0][0][0];
        uint4* gvd = &sm.gv[buf][0][0][0];
        qd[tid] = rq0; qd[tid + 256] = rq1; qd[tid + 512] = rq2;
        gd[tid] = rg0; gd[tid + 256] = rg1; gd[tid + 512] = rg2;
        qvd[tid] = rqv0; qvd[tid + 256] = rqv1;
        gvd[tid] = rgv0; gvd[tid + 256] = rgv1;
    };

    f32x4 acck[4], accv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acck[j][e] = 0.f; accv[j][e] = 0.f; }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    const int piece = lg < 2 ? 0 : lg - 1;                            // lane group -> piece of the query row: A A B C

    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) load_chunk(ch + 1);
        const int npairs = min(BKC, L - ch * BKC) >> 5;
        for (int u = 0; u < npairs; ++u) {
            const bf16x8 qa0 = bw_frag(sm.q[buf][32 * u + li][piece]), qa1 = bw_frag(sm.q[buf][32 * u + 16 + li][piece]);
            const bf16x8 ga0 = bw_frag(sm.g[buf][32 * u + li][piece]), ga1 = bw_frag(sm.g[buf][32 * u + 16 + li][piece]);
            const bf16x8 qvb = bw_frag(sm.qv[buf][u][lg][li]), gvb = bw_frag(sm.gv[buf][u][lg][li]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa0, bw_frag(kfrag[j]), zero, 0, 0, 0);
                const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa1, bw_frag(kfrag[j]), zero, 0, 0, 0);
                const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga0, bw_frag(vfrag[j]), zero, 0, 0, 0);
                const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga1, bw_frag(vfrag[j]), zero, 0, 0, 0);
                float p[8], ds[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[e] = __builtin_amdgcn_exp2f(s0[e]); p[4 + e] = __builtin_amdgcn_exp2f(s1[e]);
                    ds[e] = p[e] * d0[e]; ds[4 + e] = p[4 + e] * d1[e];
                }
                uint4 phi, plo, dhi, dlo;
                bw_split8(p, phi, plo);
                bw_split8(ds, dhi, dlo);
                accv[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw_frag(phi), gvb, accv[j], 0, 0, 0);
                accv[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw_frag(plo), gvb, accv[j], 0, 0, 0);
                acck[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw_frag(dhi), qvb, acck[j], 0, 0, 0);
                acck[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw_frag(dlo), qvb, acck[j], 0, 0, 0);
            }
        }
        if (ch + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // epilogue: D[key 4lg+r][col li]: dk_d = 1/2 (D[d] + D[4+d] + D[8+d]) over [q1|q2|q3], dv_d likewise over [g1|g2|g3]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rowbase = lane & 48;
            const float ak = acck[j][e], av = accv[j][e];
            const float ak1 = __shfl(ak, rowbase + (li & 3) + 4), ak2 = __shfl(ak, rowbase + (li & 3) + 8);
            const float av1 = __shfl(av, rowbase + (li & 3) + 4), av2 = __shfl(av, rowbase + (li & 3) + 8);
            const int ki = k0 + 16 * j + 4 * lg + e;
            if (li < 4 && ki < L) {
                float* dst = dqkv + ((int64_t)b * L + ki) * (3 * H * 4);
                dst[H * 4 + h * 4 + li] = 0.5f * ((ak + ak1) + ak2);
                dst[2 * H * 4 + h * 4 + li] = (av + av1) + av2;
            }
        }
    }
}


// ------------------------------------------------------------------ fused backward: one pass, a wave owns 64 keys
// The two kernels above each recompute the scores, the exponentials and dS.  This one computes them once, with the KEY on the
// lane (as the dK/dV kernel does), and gets dQ out of the same dS:
//   * dS (bf16 hi / lo, 8 bytes per lane and 16-query tile) crosses a wave-private LDS image [32 keys][16 queries] once and is read
//     back transposed with ds_read_b64_tr_b16: 4 keys x 16 queries per 16-lane group arrive query-major, which is exactly the A
//     fragment layout of a v_mfma_f32_16x16x32_bf16 over 32 keys;
//   * the product is taken transposed, dQ^T[4 d + piece][query] (A = the wave's K image with its columns reordered, B = the dS
//     fragment -- both operands of a 16x16x32 MFMA have the same lane layout, so the transposed read serves either side): a wave
//     sums dQ over its 64 keys in the accumulator, the three K pieces of a column are registers 0..2 of one lane (two in-lane adds),
//     and each lane stores one float per 16-query tile to the wave's own slab [column][query] (LDS float atomics into one shared tile
//     cost ~100 cycles each here); after each query chunk the four slabs (256 keys) are added and written out with plain stores to
//     partial[key block][head][row][4]; attn_bwd_dq_reduce_kernel adds the L/256 partials (fixed order: bitwise reproducible,
//     unlike float atomics, and ~4x cheaper per byte -- MI355X_MICROARCH.md "Global float atomics").
//   * ONE barrier per chunk: the query-side staging and the slabs are double-buffered by chunk parity, so the next chunk's staging
//     stores and the previous chunk's slab flush run inside the compute phase instead of between two barriers (measured: the two
//     barriers with the flush and the staging stores between them cost 15 % of the kernel).
// Per 512 scores: 4 score MFMAs, 8 v_exp, 2 hi/lo splits, 6 accumulate MFMAs (dV, dK, dQ) -- against 8 + 16 + 3 + 6 for the pair.
// (Chunk = FQ queries, 96 by default.)
// FQC = queries per LDS chunk (template parameter FQ); the slab column stride in floats is FQC + 16 (the 4-byte stores of lane
// groups 0 / 1 -- columns 0 / 1 -- land on banks 0..15 / 16..31)
template <int FQC>
struct FusedStage {
    uint4 q[FQC][3];               // score-row image of c Q: [x1 | x2] [x3 | x1] [-lse slots]; its first 32 bytes are also the 16 columns
    uint4 g[FQC][3];               // [x1 | x2 | x3 | x1] the accumulate products read transposed (the fourth column group is ignored)
};
// NW waves per workgroup = NW * 64 keys.  NW = 8 (GSDD_ATTN_BWD_NW=8) lets the query-side chunk staged in LDS serve 512 keys instead
// of 256: the staging traffic per key and the number of partial dQ copies (one per workgroup and query) halve -- 307 -> 154 MB written
// and 286 -> 143 MB re-read per layer at bs 16 -- at the same 8 waves per CU (one workgroup of 107 KB of LDS instead of two of 73 KB).
// Measured (round 3, bs 16, L = 4096): the training step is 2 ms SLOWER with it (65.6-66.2 vs 63.6-64.4 ms; the pair fused + reduce
// 2.100 vs 2.088 ms in the microbenchmark): one barrier now releases eight lockstep waves instead of four, and what the bytes save is
// not where the kernel's time is (vector issue 63 %, not memory).  NW = 4 stays the default; the traffic is not the lever here.
template <int NW, int FQC>
struct FusedSmem {
    FusedStage<FQC> st[2];         // by chunk parity
    uint2 t[NW][2][2][32][4];      // [wave][hi/lo][query tile][key row of the pair][8-byte chunk, XOR-swizzled by (row >> 2) & 3]
                                   // (hi and lo of a chunk side by side, stored with one ds_write_b128 instead of two b64: 2 % slower)
    float dq[2][NW][4][FQC + 16];  // [chunk parity][wave][column][query]: each wave's own sum over its 64 keys
};

// ds_read_b64_tr_b16 through the compiler's builtin: it then places the s_waitcnt itself and orders the read after the wave's own
// LDS stores to the image (a hand-written asm read would need a manual wait, and the register moves that build the 128-bit
// fragment from two 64-bit reads could be scheduled in front of that wait)
typedef short bw_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_read_tr16(const uint2* p) {
    const bw_v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bw_v4s*)(p));
    return __builtin_bit_cast(uint2, r);
}
// (Adding the partial dQ of a (query chunk, 256-key block) straight into dqkv with float atomics -- no partial buffer, no reduction
// kernel -- was built and measured in round 4: 2.33 ms against 1.84 ms for this kernel + the reduction at B = 16, L = 4096: 67 M
// atomics per layer cost more than the 268 MB of plain stores and the 50 us reduction they replace.)
template <int DBG, int NW = 4, int FQ = 128>   // DBG 0: the kernel; 1: without the LDS hand-over of dQ; 2: without the dQ product as well (timing only)
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_fused_kernel(const float* __restrict__ k, const float* __restrict__ v, BwdImages im,
                                                                    int B, int L, int H, float* __restrict__ dqkv,
                                                                    float* __restrict__ dq_part) {
    extern __shared__ __attribute__((aligned(16))) char fused_raw[];
    FusedSmem<NW, FQ>& sm = *reinterpret_cast<FusedSmem<NW, FQ>*>(fused_raw);
    constexpr int NT = 64 * NW, KB = 64 * NW, FQC = FQ;               // threads, keys per workgroup, queries per chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nkb = (L + KB - 1) / KB;
    const unsigned wg = bw_xcd_remap(blockIdx.x, gridDim.x);
    const int kblk = wg % nkb;
    const int h = (wg / nkb) % H, b = wg / (nkb * H);
    const int64_t M = (int64_t)B * L;
    const int64_t hrow0 = (int64_t)h * M + (int64_t)b * L;
    const int li = lane & 15, lg = lane >> 4;
    const int k0 = kblk * KB + wave * 64;

    uint4 kfrag[4], vfrag[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ki = k0 + 16 * j + li;
        ki = ki < L ? ki : L - 1;
        const float4 kk = *reinterpret_cast<const float4*>(k + (hrow0 + ki) * 4);
        const float4 vv = *reinterpret_cast<const float4*>(v + (hrow0 + ki) * 4);
        const float ks[4] = {kk.x, kk.y, kk.z, kk.w}, vs[4] = {vv.x, vv.y, vv.z, vv.w};
        kfrag[j] = bw_col_frag(ks, lg, bw_ones_frag());
        vfrag[j] = bw_col_frag(vs, lg, bw_ones_frag());
        if (j == lg) {
            // row-major image of key 16 lg + li in the wave's (still unused) transpose buffer: position 4 d + piece = piece of k[d]
            uint2* row = &sm.t[wave][0][0][0][0] + (16 * lg + li) * 4;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint32_t a, b, c;
                bw_split3(ks[d], a, b, c);
                row[d] = make_uint2(a | (b << 16), c);
            }
        }
    }
    // accumulate image of this wave's keys, one 32-key pair-tile per jp: the A operand of the TRANSPOSED dQ product
    //   dQ^T[4 d + piece][query] += K^T[4 d + piece][key] . dS^T[key][query],
    // read transposed out of the row-major image above (lane li = 4 d + piece gets position li of keys 4 lg .. 4 lg + 3 of both 16-key
    // tiles; LDS instructions of a wave execute in order, so the rows written by other lanes are in place).  Row 4 lg + r of the output sits in register r of lane group lg, so a
    // lane group is one column d of dQ and its registers are the three bf16 pieces of K: the piece sum is two in-lane adds (the
    // untransposed product had the pieces on 12 different lanes: 16 DPP moves + 8 packed adds per pair-tile, 11 % of the kernel); a pair-tile past the end of
    // the sequence (L % 256 != 0; L % 32 == 0) gets a zero image: its clamped keys give finite dS, times zero they add nothing
    uint4 kvb[2];
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
        const uint2* kt = &sm.t[wave][0][0][0][0] + (32 * jp + 4 * lg + (li >> 2)) * 4 + (li & 3);
        const uint2 kt0 = lds_read_tr16(kt), kt1 = lds_read_tr16(kt + 16 * 4);
        kvb[jp] = make_uint4(kt0.x, kt0.y, kt1.x, kt1.y);
        if (k0 + 32 * jp >= L) kvb[jp] = make_uint4(0u, 0u, 0u, 0u);
    }

    const int nchunks = (L + FQC - 1) / FQC;
    uint4 rq0, rq1, rg0, rg1;
    auto load_chunk = [&](int ch) {
        const int rows = min(FQC, L - ch * FQC);                      // multiple of 32
        const uint4* qs = im.qp + (hrow0 + (int64_t)ch * FQC) * 3;
        const uint4* gs = im.gp + (hrow0 + (int64_t)ch * FQC) * 3;
        const int l3 = rows * 3 - 1;                                  // unconditional, clamped loads (see the dQ kernel)
        rq0 = qs[min(tid, l3)]; rg0 = gs[min(tid, l3)];
        if (NT < FQC * 3) { rq1 = qs[min(tid + NT, l3)]; rg1 = gs[min(tid + NT, l3)]; }
    };
    auto store_chunk = [&](int par) {
        FusedStage<FQ>& st = sm.st[par];
        uint4* qd = &st.q[0][0];
        uint4* gd = &st.g[0][0];
        if (tid < FQC * 3) { qd[tid] = rq0; gd[tid] = rg0; }                    // 384 fragments each
        if (NT < FQC * 3 && tid < FQC * 3 - NT) { qd[tid + NT] = rq1; gd[tid + NT] = rg1; }
    };
    // partial dQ of chunk `ch` (this workgroup's keys = its waves' slabs) -> dq_part[kblk][h][row][4]; thread = (query, column pair)
    auto flush_dq = [&](int ch) {
        const int q = tid >> 1, c0 = 2 * (tid & 1);
        const int qi = ch * FQC + q;
        if (tid < 2 * FQC && qi < L) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                s0 += sm.dq[ch & 1][w][c0][q];
                s1 += sm.dq[ch & 1][w][c0 + 1][q];
            }
            *reinterpret_cast<float2*>(dq_part + (((int64_t)kblk * H + h) * M + (int64_t)b * L + qi) * 4 + c0) =
                make_float2(0.5f * s0, 0.5f * s1);
        }
    };

    f32x4 acck[4], accv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { acck[j][e] = 0.f; accv[j][e] = 0.f; }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    const int piece = lg < 2 ? 0 : lg - 1;                            // lane group -> piece of the query row: A A B C
    // transpose image addresses of this lane: write = row (16 jj + li), chunk lg ^ (li >> 2);
    // transposed read = row (4 lg + (li >> 2)) [+ 16 for the second key tile], chunk (li & 3) ^ lg
    uint2* const tw = &sm.t[wave][0][0][li][lg ^ (li >> 2)];
    const uint2* const tr0 = &sm.t[wave][0][0][4 * lg + (li >> 2)][(li & 3) ^ lg];
    constexpr int T_TILE = 32 * 4, T_HL = 2 * T_TILE, T_ROW16 = 16 * 4;       // in uint2 (8-byte) units

    load_chunk(0);
    store_chunk(0);
    if (nchunks > 1) load_chunk(1);
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        // every wave is past the barrier that ended chunk ch - 1: the other staging buffer and the other slab set are free, the slabs
        // of chunk ch - 1 are complete
        if (ch + 1 < nchunks) store_chunk((ch + 1) & 1);
        if (ch + 2 < nchunks) load_chunk(ch + 2);
        if (ch > 0) flush_dq(ch - 1);
        const FusedStage<FQ>& st = sm.st[ch & 1];
        float* const slab = &sm.dq[ch & 1][wave][lg][li];
        const int npairs = min(FQC, L - ch * FQC) >> 5;
        for (int u = 0; u < npairs; ++u) {
            const bf16x8 qa0 = bw_frag(st.q[32 * u + li][piece]), qa1 = bw_frag(st.q[32 * u + 16 + li][piece]);
            const bf16x8 ga0 = bw_frag(st.g[32 * u + li][piece]), ga1 = bw_frag(st.g[32 * u + 16 + li][piece]);
            // accumulate-side fragments (B operand of dK += dS^T Q, dV += P^T dO): lane (lg, column li) needs column li of queries
            // 4 lg .. 4 lg + 3 of both 16-query tiles -- the transposed read of the row-major score image (16 lanes point at 4 rows x
            // 4 chunks of 4 columns and get one column of the 4 rows each); round 2 staged a second, pre-transposed image for this
            const uint2* const qt = reinterpret_cast<const uint2*>(&st.q[32 * u + 4 * lg + (li >> 2)][0]) + (li & 3);
            const uint2* const gt = reinterpret_cast<const uint2*>(&st.g[32 * u + 4 * lg + (li >> 2)][0]) + (li & 3);
            const uint2 qt0 = lds_read_tr16(qt), qt1 = lds_read_tr16(qt + 16 * 6), gt0 = lds_read_tr16(gt), gt1 = lds_read_tr16(gt + 16 * 6);
            const bf16x8 qvb = bw_frag(make_uint4(qt0.x, qt0.y, qt1.x, qt1.y)), gvb = bw_frag(make_uint4(gt0.x, gt0.y, gt1.x, gt1.y));
            f32x4 dq0 = zero, dq1 = zero;
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * jp + jj;
                    const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa0, bw_frag(kfrag[j]), zero, 0, 0, 0);
                    const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa1, bw_frag(kfrag[j]), zero, 0, 0, 0);
                    const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga0, bw_frag(vfrag[j]), zero, 0, 0, 0);
                    const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga1, bw_frag(vfrag[j]), zero, 0, 0, 0);
                    float p[8], ds[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        p[e] = __builtin_amdgcn_exp2f(s0[e]); p[4 + e] = __builtin_amdgcn_exp2f(s1[e]);
                        ds[e] = p[e] * d0[e]; ds[4 + e] = p[4 + e] * d1[e];
                    }
                    uint4 phi, plo, dhi, dlo;
                    bw_split8c(p, phi, plo);
                    bw_split8c(ds, dhi, dlo);
                    accv[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw_frag(phi), gvb, accv[j], 0, 0, 0);
                    acck[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw_frag(dhi), qvb, acck[j], 0, 0, 0);
                    accv[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw_frag(plo), gvb, accv[j], 0, 0, 0);
                    acck[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw_frag(dlo), qvb, acck[j], 0, 0, 0);
                    // dS of key li (tile jj of the pair), queries 4 lg .. 4 lg + 3 of query tile 0 / 1, hi and lo
                    uint2* w = tw + jj * (16 * 4);
                    w[0] = make_uint2(dhi.x, dhi.y);
                    w[T_TILE] = make_uint2(dhi.z, dhi.w);
                    w[T_HL] = make_uint2(dlo.x, dlo.y);
                    w[T_HL + T_TILE] = make_uint2(dlo.z, dlo.w);
                }
                if (DBG < 2) {
                    // A fragments of dQ += dS^T K: keys 4 lg .. 4 lg + 3 of both tiles of the pair, query li
                    const uint2 h0a = lds_read_tr16(tr0), h0b = lds_read_tr16(tr0 + T_ROW16);
                    const uint2 h1a = lds_read_tr16(tr0 + T_TILE), h1b = lds_read_tr16(tr0 + T_TILE + T_ROW16);
                    const uint2 l0a = lds_read_tr16(tr0 + T_HL), l0b = lds_read_tr16(tr0 + T_HL + T_ROW16);
                    const uint2 l1a = lds_read_tr16(tr0 + T_HL + T_TILE), l1b = lds_read_tr16(tr0 + T_HL + T_TILE + T_ROW16);
                    const bf16x8 kb = bw_frag(kvb[jp]);
                    dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb, bw_frag(make_uint4(h0a.x, h0a.y, h0b.x, h0b.y)), dq0, 0, 0, 0);
                    dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb, bw_frag(make_uint4(h1a.x, h1a.y, h1b.x, h1b.y)), dq1, 0, 0, 0);
                    dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb, bw_frag(make_uint4(l0a.x, l0a.y, l0b.x, l0b.y)), dq0, 0, 0, 0);
                    dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb, bw_frag(make_uint4(l1a.x, l1a.y, l1b.x, l1b.y)), dq1, 0, 0, 0);
                }
            }
            // the wave's 64 keys are summed; lane (li, lg) holds column lg of query li of both 16-query tiles as three pieces in
            // registers 0..2 (register 3 is the image's zero column), added in the order (k1 + k2) + k3 as before
            if (DBG >= 1) {
                if (dq0[0] + dq1[0] == 123.456f) sm.dq[0][wave][0][0] = 1.f;    // keep the products alive
            } else {
                slab[32 * u] = (dq0[0] + dq0[1]) + dq0[2];
                slab[32 * u + 16] = (dq1[0] + dq1[1]) + dq1[2];
            }
        }
        __syncthreads();
        // Leave the barrier out of step: wave w starts 16 w cycles late.  Four waves released in the same cycle run the same
        // instruction stream in exact lockstep and collide on the CU's shared LDS path at every tile (measured: 2.17 -> 1.87 ms for
        // the whole backward at B = 16, L = 4096; without any barrier -- racy, timing only -- 1.86 ms; 32 w and 64 w cycles: 1.89, 1.94).
        for (int i = 0; i < wave; ++i) asm volatile("s_nop 15");
    }
    flush_dq(nchunks - 1);

    // dK, dV of this wave's keys (as in the dK/dV kernel)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rowbase = lane & 48;
            const float ak = acck[j][e], av = accv[j][e];
            const float ak1 = __shfl(ak, rowbase + (li & 3) + 4), ak2 = __shfl(ak, rowbase + (li & 3) + 8);
            const float av1 = __shfl(av, rowbase + (li & 3) + 4), av2 = __shfl(av, rowbase + (li & 3) + 8);
            const int ki = k0 + 16 * j + 4 * lg + e;
            if (li < 4 && ki < L) {
                float* dst = dqkv + ((int64_t)b * L + ki) * (3 * H * 4);
                dst[H * 4 + h * 4 + li] = 0.6931471805599453f * ((ak + ak1) + ak2);      // 1/2 dS^T Q = (1 / (2 c)) dS^T (c Q), c = log2(e) / 2
                dst[2 * H * 4 + h * 4 + li] = (av + av1) + av2;
            }
        }
    }
}

// dq[m][h*4 + d] = sum over key blocks of dq_part[kblk][h][m][d]   (one thread per (m, h), fixed summation order)
__global__ __launch_bounds__(256) void attn_bwd_dq_reduce_kernel(const float* __restrict__ part, int nkb, int H, int64_t M,
                                                                 float* __restrict__ dqkv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // h * M + m
    if (i >= (int64_t)H * M) return;
    const int h = (int)(i / M);
    const int64_t m = i - (int64_t)h * M;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kb = 0; kb < nkb; ++kb) {
        const float4 p = *reinterpret_cast<const float4*>(part + (((int64_t)kb * H + h) * M + m) * 4);
        s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    *reinterpret_cast<float4*>(dqkv + m * (3 * H * 4) + h * 4) = s;
}

}  // namespace gsdd

using namespace gsdd;

extern "C" int64_t gsdd_d3pm_attention_bwd_workspace_bytes(int B, int L, int H) {
    // 16 uint4 per (row, head): kp 2, vk 2, kv 2, qp 3, gp 3, qv 2, gv 2 (the fused kernel uses qp and gp only; the layout is the
    // two-kernel variant's); then the fused kernel's partial dQ: one float4 per (256-key block, row, head)
    return (int64_t)B * L * H * 16 * 16 + (int64_t)((L + 255) / 256) * B * L * H * 16;
}

// Matrix-pipe backward; returns GSDD_OK with *done = 0 when the shape needs the VALU kernels (L % 32 != 0 or no workspace).
int gsdd_attention_bwd_mfma(const float* q, const float* k, const float* v, const float* o, const float* dO, const float* lse, int B,
                            int L, int H, float* dqkv, void* workspace, int64_t workspace_bytes, int variant, void* stream, int* done) {
    *done = 0;
    if (L % 32 != 0 || workspace == nullptr) return GSDD_OK;
    GSDD_CHECK_ARG(workspace_bytes >= gsdd_d3pm_attention_bwd_workspace_bytes(B, L, H), "workspace too small");
    GSDD_CHECK_ARG((int64_t)B * H * ((L + 255) / 256) < (1ll << 31), "grid too large");
    const int64_t M = (int64_t)B * L, rows = M * H;
    BwdImages im;
    uint4* w = reinterpret_cast<uint4*>(workspace);
    im.kp = w; w += rows * 2;
    im.vk = w; w += rows * 2;
    im.kv = w; w += rows * 2;
    im.qp = w; w += rows * 3;
    im.gp = w; w += rows * 3;
    im.qv = w; w += rows * 2;
    im.gv = w;
    hipStream_t st = (hipStream_t)stream;
    const bool split_kernels = variant == GSDD_ATTN_BWD_SPLIT;                      // development variant: dQ kernel + dK/dV kernel
    if (!split_kernels && H == 16 && M % 16 == 0)
        hipLaunchKernelGGL(attn_bwd_prep_fused_kernel, dim3((unsigned)(M / 16)), dim3(256), 0, st, q, o, dO, lse, M, im);
    else
        hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, q, k, v, o, dO, lse, M, H, im,
                           split_kernels);
    GSDD_CHECK_LAUNCH();
    const dim3 grid((unsigned)(B * H * ((L + 255) / 256)));
    if (!split_kernels) {
        float* dq_part = reinterpret_cast<float*>(reinterpret_cast<uint4*>(workspace) + rows * 16);
        GSDD_ONCE_PER_DEVICE(fattr_done,
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)sizeof(FusedSmem<4, 128>)));
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)sizeof(FusedSmem<4, 128>)));
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)sizeof(FusedSmem<4, 128>)));
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<0, 8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)sizeof(FusedSmem<8, 128>)));
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<0, 4, 64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)sizeof(FusedSmem<4, 64>)));
            GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_fused_kernel<0, 4, 96>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)sizeof(FusedSmem<4, 96>)));
        );
        // development variants (include/gsdd.h): debug stages of the fused kernel, queries per LDS chunk 96 (default) / 64 / 128, waves per
        // workgroup 4 (default) or 8
        const int dbgv = variant == GSDD_ATTN_BWD_DBG1 ? 1 : (variant == GSDD_ATTN_BWD_DBG2 ? 2 : 0);
        const int fqc = variant == GSDD_ATTN_BWD_FQC64 ? 64 : (variant == GSDD_ATTN_BWD_FQC128 ? 128 : 96);
        const int nw = variant == GSDD_ATTN_BWD_NW8 ? 8 : 4;
        const int kb = 64 * nw, nkb = (L + kb - 1) / kb;
        const dim3 fgrid((unsigned)(B * H * nkb));
        if (dbgv == 1) hipLaunchKernelGGL((attn_bwd_fused_kernel<1, 4>), fgrid, dim3(256), sizeof(FusedSmem<4, 128>), st, k, v, im, B, L, H, dqkv, dq_part);
        else if (dbgv == 2) hipLaunchKernelGGL((attn_bwd_fused_kernel<2, 4>), fgrid, dim3(256), sizeof(FusedSmem<4, 128>), st, k, v, im, B, L, H, dqkv, dq_part);
        else if (nw == 4 && fqc == 64)
            hipLaunchKernelGGL((attn_bwd_fused_kernel<0, 4, 64>), fgrid, dim3(256), sizeof(FusedSmem<4, 64>), st, k, v, im, B, L, H, dqkv, dq_part);
        else if (nw == 4 && fqc != 128)
            // default: 96-query chunks.  49 KB of LDS per workgroup -> three workgroups per CU (3 waves per SIMD, which the kernel's 156
            // VGPRs allow) where 128-query chunks (58 KB) fit two; 64-query chunks (38 KB) fit three as well but pay a third more chunk
            // barriers: 1.995 / 1.94 / 1.92 ms for 128 / 64 / 96 (microbenchmark incl. pre-split and reduction, one box)
            hipLaunchKernelGGL((attn_bwd_fused_kernel<0, 4, 96>), fgrid, dim3(256), sizeof(FusedSmem<4, 96>), st, k, v, im, B, L, H, dqkv, dq_part);
        else if (nw == 4) hipLaunchKernelGGL((attn_bwd_fused_kernel<0, 4>), fgrid, dim3(256), sizeof(FusedSmem<4, 128>), st, k, v, im, B, L, H, dqkv, dq_part);
        else hipLaunchKernelGGL((attn_bwd_fused_kernel<0, 8>), fgrid, dim3(512), sizeof(FusedSmem<8, 128>), st, k, v, im, B, L, H, dqkv, dq_part);
        GSDD_CHECK_LAUNCH();
        hipLaunchKernelGGL(attn_bwd_dq_reduce_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, dq_part, nkb, H, M, dqkv);
        GSDD_CHECK_LAUNCH();
        *done = 1;
        return GSDD_OK;
    }
    hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel, grid, dim3(256), 0, st, q, o, dO, lse, im, B, L, H, dqkv);
    GSDD_CHECK_LAUNCH();
    GSDD_ONCE_PER_DEVICE(attr_done,
        GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)attn_bwd_dkv_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sizeof(DkvSmem)));
    );
    hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel, grid, dim3(256), sizeof(DkvSmem), st, k, v, im, B, L, H, dqkv);
    GSDD_CHECK_LAUNCH();
    *done = 1;
    return GSDD_OK;
}
