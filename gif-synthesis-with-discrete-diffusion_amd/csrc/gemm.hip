// Generic fp32 implicit GEMM on v_mfma_f32_32x32x2_f32 (exact f32, gfx950).
//
//   out[orow(m)][n] = epi( sum_{tap} sum_{c<Cin} pro(in[src(m,tap)][c]) * w[tap][n][c] )
//
// Block tile 128 x BN x 32, 256 threads = 4 waves (2x2), each wave 64 x BN/2 as 32x32 MFMA tiles.
// Both operands are staged through LDS "index-major, k-contiguous" ([row][32+4 pad]); MFMA step s of
// lane half h consumes memory k = 16h + s for A and B alike, so one ds_read_b128 feeds 4 MFMA steps
// and the padded pitch (36 dwords) is conflict-free for the b128 lane groups.  Double-buffered LDS,
// next chunk's global loads are issued before the current chunk's 16 MFMA steps.
//
// Replaces (see include/gsdd.h): nn.Conv3d / nn.ConvTranspose3d (videogpt_vq_vae.py:289-332),
// nn.Linear (model_utils.py:223-233, transformer_utils.py:36-43,258-263,353-356), BatchNorm(eval)+ReLU
// folded as pro/epilogue (videogpt_vq_vae.py:125-133), LayerNorm apply (transformer_utils.py:157,217,354).
#include <stdlib.h>

#include "common.hpp"

namespace gsdd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BK = 32, LDP = 36;

template <int BN, int BMt = BM>
struct GemmSmem {
    float a[2][BMt][LDP];
    float b[2][BN][LDP];
};

// ---- bf16x3 variant: every f32 operand element is split error-free into three bf16 pieces (x = x1 + x2 + x3, 24 bits) when it
// is staged into LDS, and a 32x32x16 k-step is the six significant cross products a1b1 a1b2 a2b1 a1b3 a2b2 a3b1 on
// v_mfma_f32_32x32x16_bf16 (bf16 x bf16 products are exact in the f32 accumulator; the dropped a2b3 + a3b2 + a3b3 are below
// 2^-24 |a||b|): 6 x 32 cycles on the bf16 matrix pipe instead of 8 x 64 on the f32 datapath.  Same accumulator layout as
// v_mfma_f32_32x32x2_f32, so prologues, epilogues and addressing are shared.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDPB = 40;                 // bf16 per LDS row (32 + 8 pad): 80-B pitch keeps the 16-B fragment reads conflict-free
template <int BN, int BMt = BM>
struct GemmSmemX3 {
    uint16_t a[3][BMt][LDPB];
    uint16_t b[3][BN][LDPB];
};
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// 4 floats -> three pieces of 4 packed bf16 each (uint2): x = p[0] + p[1] + p[2]
__device__ __forceinline__ void split3x4(const float4 v, uint2 (&p)[3]) {
    float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t h01 = cvt_pk_bf16(x[0], x[1]), h23 = cvt_pk_bf16(x[2], x[3]);
        p[i] = make_uint2(h01, h23);
        if (i < 2) {
            x[0] -= __uint_as_float(h01 << 16); x[1] -= __uint_as_float(h01 & 0xffff0000u);
            x[2] -= __uint_as_float(h23 << 16); x[3] -= __uint_as_float(h23 & 0xffff0000u);
        }
    }
}
__device__ __forceinline__ bf16x8 as_bf8(uint4 u) {
    union { uint4 u; bf16x8 v; } c;
    c.u = u;
    return c.v;
}

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v * (1.f / (1.f + expf(-1.702f * v)));   // x * sigmoid(1.702 x)
    return v;
}

// Block tile BMt x BN, NTH threads = NTH/64 waves laid out (BMt/64) x 2, each wave 64 x BN/2.  BMt = 128 (256 threads, two
// blocks per CU) everywhere except the big bf16x3 convolutions, which take BMt = 256 (512 threads, one block per CU): the weight
// tile is staged once per 256 rows, a quarter less L2 traffic and staging work per MAC (measured: +6 %; the transposed choice,
// 128 x 256 with the activation tile staged once per 256 output channels, instantiates too but runs 4 % slower than that).
template <int BN, bool X3, int BMt = BM, int NTH = 256>
__global__ __launch_bounds__(NTH, 512 / NTH) void gemm_kernel(const gsdd_gemm_desc d, const int64_t M) {
    constexpr int WNW = BN >= 128 ? 64 : 32;   // columns per wave
    constexpr int NT = WNW / 32;         // 32-wide n tiles per wave
    constexpr int WN = BN / WNW;         // waves along n (the others stack along m, 64 rows each)
    constexpr int RSTEP = NTH / 8;       // rows staged per pass of the block's threads (8 threads per 32-k row)
    constexpr int RA = BMt / RSTEP;      // activation rows staged per thread
    constexpr int BJ = BN / RSTEP;       // weight rows staged per thread
    static_assert((NTH / 64) / WN * 64 == BMt, "wave grid must cover the tile");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    GemmSmem<BN, BMt>& sm = *reinterpret_cast<GemmSmem<BN, BMt>*>(smem_raw);
    GemmSmemX3<BN, BMt>& sx = *reinterpret_cast<GemmSmemX3<BN, BMt>*>(smem_raw);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int64_t bm0 = (int64_t)blockIdx.x * BMt;
    const int bn0 = blockIdx.y * BN;
    const int kq = tid & 7, r0 = tid >> 3;

    // ---- per-thread staged rows of the activation operand (row index and tap-free coordinates, all 32-bit)
    int ti0[RA], hi0[RA], wi0[RA], rbase[RA];
    bool rvalid[RA];
    float mu[RA], rs[RA];
    int64_t sel[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const int64_t m = bm0 + r0 + RSTEP * j;
        rvalid[j] = m < M;
        const uint32_t mm = rvalid[j] ? (uint32_t)m : 0u;          // M < 2^31 (checked on the host)
        if (d.gather != nullptr) {
            rbase[j] = rvalid[j] ? (int)d.gather[mm] : 0;
            ti0[j] = hi0[j] = wi0[j] = 0;
        } else {
            uint32_t q = mm;
            const uint32_t wo = q % (uint32_t)d.Wo; q /= (uint32_t)d.Wo;
            const uint32_t ho = q % (uint32_t)d.Ho; q /= (uint32_t)d.Ho;
            const uint32_t to = q % (uint32_t)d.Do; q /= (uint32_t)d.Do;
            ti0[j] = (int)to * d.sd; hi0[j] = (int)ho * d.sh; wi0[j] = (int)wo * d.sw;
            rbase[j] = (((int)q * d.Di + ti0[j]) * d.Hi + hi0[j]) * d.Wi + wi0[j];      // input rows < 2^31 (host check)
        }
        mu[j] = 0.f; rs[j] = 1.f; sel[j] = 0;
        if (d.ln_stats != nullptr) {
            mu[j] = d.ln_stats[2 * (int64_t)mm]; rs[j] = d.ln_stats[2 * (int64_t)mm + 1];
            if (d.ln_sel != nullptr) sel[j] = d.ln_sel[mm / (uint32_t)d.rows_per_batch];
        }
    }

    const int cchunks = (d.Cin + BK - 1) / BK;
    const int nchunks = d.ntaps * cchunks;
    const bool has_pro = d.pro_scale != nullptr;
    const float* pro_s = has_pro ? d.pro_scale : d.w;          // always-dereferenceable pointers: the loads below are
    const float* pro_b = has_pro ? d.pro_shift : d.w;          // unconditional so that nothing waits on them before the MFMAs

    // Global loads of chunk i+1 are issued before the MFMAs of chunk i and consumed after them.  They are branch-free
    // (out-of-range rows / channels read a clamped, valid address and are zeroed when staged into LDS): a load under an
    // exec-masked branch makes the compiler wait for it on the spot, which serialises memory latency with the MFMAs.
    float4 ra[RA], rb[BJ], ps, pb;
    unsigned okbits = 0;
    int cc = 0;
    // Chunks are requested in order, so (tap, channel chunk) is a running counter, and everything that depends on the tap alone
    // (source row, bounds test) is computed once per tap and re-used by its channel chunks.
    int ld_tap = 0, ld_cq = 0;
    const float* prow[RA];
    unsigned rowok = 0;
    auto load_chunk = [&](int /*chunk: always the next one*/) {
        const int tap = ld_tap;
        if (ld_cq == 0) {
            int dt = 0, dh = 0, dw = 0;
            if (d.taps != nullptr) { dt = d.taps[3 * tap]; dh = d.taps[3 * tap + 1]; dw = d.taps[3 * tap + 2]; }
            const int tapoff = (dt * d.Hi + dh) * d.Wi + dw;
            rowok = 0;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                const int ti = ti0[j] + dt, hi = hi0[j] + dh, wi = wi0[j] + dw;
                const bool ok = rvalid[j] & ((unsigned)ti < (unsigned)d.Di) & ((unsigned)hi < (unsigned)d.Hi) &
                                ((unsigned)wi < (unsigned)d.Wi);
                const int row = ok ? rbase[j] + tapoff : 0;
                prow[j] = d.in + (int64_t)row * d.in_pitch;
                rowok |= (ok ? 1u : 0u) << j;
            }
        }
        const int c = ld_cq * BK + 4 * kq;
        const bool cvalid = c < d.Cin;
        cc = cvalid ? c : 0;
        if (++ld_cq == cchunks) { ld_cq = 0; ++ld_tap; }
        ps = *reinterpret_cast<const float4*>(pro_s + cc);
        pb = *reinterpret_cast<const float4*>(pro_b + cc);
        okbits = cvalid ? rowok : 0u;
        const float* pa[RA];
        const float* pw[BJ];
#pragma unroll
        for (int j = 0; j < RA; ++j) pa[j] = prow[j] + cc;
        const int wrow0 = tap * d.Cout;
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int n = bn0 + r0 + RSTEP * j;
            const bool ok = (n < d.Cout) & cvalid;
            pw[j] = d.w + (uint32_t)((wrow0 + (ok ? n : 0)) * d.Cin + cc);           // weights hold < 2^31 elements (host check)
            okbits |= (ok ? 1u : 0u) << (8 + j);
        }
#pragma unroll
        for (int j = 0; j < RA; ++j) ra[j] = *reinterpret_cast<const float4*>(pa[j]);
#pragma unroll
        for (int j = 0; j < BJ; ++j) rb[j] = *reinterpret_cast<const float4*>(pw[j]);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            float4 v = ra[j];
            if (has_pro) {
                v.x = fmaxf(fmaf(v.x, ps.x, pb.x), 0.f); v.y = fmaxf(fmaf(v.y, ps.y, pb.y), 0.f);
                v.z = fmaxf(fmaf(v.z, ps.z, pb.z), 0.f); v.w = fmaxf(fmaf(v.w, ps.w, pb.w), 0.f);
            }
            if (d.ln_stats != nullptr) {
                const float4 g = *reinterpret_cast<const float4*>(d.ln_gamma + sel[j] * d.ln_stride + cc);
                const float4 bt = *reinterpret_cast<const float4*>(d.ln_beta + sel[j] * d.ln_stride + cc);
                v.x = (v.x - mu[j]) * rs[j] * g.x + bt.x; v.y = (v.y - mu[j]) * rs[j] * g.y + bt.y;
                v.z = (v.z - mu[j]) * rs[j] * g.z + bt.z; v.w = (v.w - mu[j]) * rs[j] * g.w + bt.w;
            }
            if (!((okbits >> j) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (X3) {
                uint2 pc[3];
                split3x4(v, pc);
#pragma unroll
                for (int i = 0; i < 3; ++i) *reinterpret_cast<uint2*>(&sx.a[i][r0 + RSTEP * j][4 * kq]) = pc[i];
            } else {
                *reinterpret_cast<float4*>(&sm.a[buf][r0 + RSTEP * j][4 * kq]) = v;
            }
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            float4 v = rb[j];
            if (!((okbits >> (8 + j)) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (X3) {
                uint2 pc[3];
                split3x4(v, pc);
#pragma unroll
                for (int i = 0; i < 3; ++i) *reinterpret_cast<uint2*>(&sx.b[i][r0 + RSTEP * j][4 * kq]) = pc[i];
            } else {
                *reinterpret_cast<float4*>(&sm.b[buf][r0 + RSTEP * j][4 * kq]) = v;
            }
        }
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const int li = lane & 31, lh = lane >> 5;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    if constexpr (X3) {
        // single LDS buffer: chunk i+1 waits in registers while chunk i's MFMAs run, and is split + staged between two barriers
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            if (chunk + 1 < nchunks) load_chunk(chunk + 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[2][3], bf[NT][3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        af[mt][i] = as_bf8(*reinterpret_cast<const uint4*>(&sx.a[i][wm * 64 + mt * 32 + li][16 * ks + 8 * lh]));
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        bf[nt][i] = as_bf8(*reinterpret_cast<const uint4*>(&sx.b[i][wn * WNW + nt * 32 + li][16 * ks + 8 * lh]));
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][2], bf[nt][0], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][1], bf[nt][1], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][0], bf[nt][2], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][1], bf[nt][0], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][0], bf[nt][1], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][0], bf[nt][0], acc[mt][nt], 0, 0, 0);
                    }
            }
            __syncthreads();                               // every wave is done reading this chunk
            if (chunk + 1 < nchunks) store_chunk(0);
            __syncthreads();
        }
    } else {
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = chunk & 1;
        if (chunk + 1 < nchunks) load_chunk(chunk + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 af[2], bf[NT];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                af[mt] = *reinterpret_cast<const float4*>(&sm.a[buf][wm * 64 + mt * 32 + li][16 * lh + 4 * q]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bf[nt] = *reinterpret_cast<const float4*>(&sm.b[buf][wn * WNW + nt * 32 + li][16 * lh + 4 * q]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].x, bf[nt].x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].y, bf[nt].y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].z, bf[nt].z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt].w, bf[nt].w, acc[mt][nt], 0, 0, 0);
                }
        }
        if (chunk + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }
    }

    // ---- epilogue: lane holds column n = li, rows (r&3)+8*(r>>2)+4*lh of each 32x32 tile
    const bool linear_rows = (d.oD == d.Do && d.oH == d.Ho && d.oW == d.Wo && d.osd == 1 && d.osh == 1 && d.osw == 1 &&
                              d.ood == 0 && d.ooh == 0 && d.oow == 0);
    const int64_t ohw = (int64_t)d.oH * d.oW;
    int ncol[NT];
    bool nok[NT];
    float es[NT], eh[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        ncol[nt] = bn0 + wn * WNW + nt * 32 + li;
        nok[nt] = ncol[nt] < d.Cout;
        es[nt] = (d.epi_scale != nullptr && nok[nt]) ? d.epi_scale[ncol[nt]] : 1.f;
        eh[nt] = (d.epi_shift != nullptr && nok[nt]) ? d.epi_shift[ncol[nt]] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = bm0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m >= M) continue;
            const uint32_t mu32 = (uint32_t)m;
            int64_t orow = m, bidx = 0, od = 0, rem = 0;
            if (d.out_mode != 2 && (!linear_rows || d.out_mode == 1)) {
                uint32_t q = mu32;
                const uint32_t wo = q % (uint32_t)d.Wo; q /= (uint32_t)d.Wo;
                const uint32_t ho = q % (uint32_t)d.Ho; q /= (uint32_t)d.Ho;
                const uint32_t to = q % (uint32_t)d.Do; q /= (uint32_t)d.Do;
                bidx = q;
                od = (int64_t)to * d.osd + d.ood;
                rem = (int64_t)((int)ho * d.osh + d.ooh) * d.oW + ((int)wo * d.osw + d.oow);
                orow = (bidx * d.oD + od) * ohw + rem;
            }
            const float* bv = d.bvec != nullptr ? d.bvec + (int64_t)(mu32 / (uint32_t)d.rows_per_batch) * d.Cout : nullptr;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (!nok[nt]) continue;
                const int n = ncol[nt];
                float v = acc[mt][nt][r];
                if (d.epi_scale != nullptr) v *= es[nt];
                v += eh[nt];
                if (bv != nullptr) v += bv[n];
                v = act_fn(v, d.act);
                int64_t addr;
                if (d.out_mode == 2) addr = ((int64_t)(n >> 2) * M + m) * 4 + (n & 3);
                else if (d.out_mode == 1) addr = ((bidx * d.Cout + n) * d.oD + od) * ohw + rem;
                else addr = orow * d.out_pitch + n;
                if (d.residual != nullptr) v += d.residual[addr];
                d.out[addr] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void row_stats_kernel(const float* x, int64_t M, int C, float eps, float* stats) {
    // 16 lanes per row (float4 each per pass); two-pass mean / biased variance like nn.LayerNorm
    const int lane16 = threadIdx.x & 15;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool ok = row < M;
    const float* p = x + (ok ? row : 0) * C;
    float s = 0.f;
    for (int c = 4 * lane16; c < C; c += 64) {
        const float4 v = *reinterpret_cast<const float4*>(p + c);
        s += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = 4 * lane16; c < C; c += 64) {
        const float4 v = *reinterpret_cast<const float4*>(p + c);
        const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, dd = v.w - mean;
        q += (a * a + b * b) + (cc * cc + dd * dd);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (ok && lane16 == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = 1.0f / sqrtf(q / (float)C + eps);
    }
}

__global__ void ncdhw_to_rows_kernel(const float* x, int N, int C, int D, int H, int W, int Cpad, int padw, float* out) {
    // out[n][d][h][w + padw][Cpad], zero-filled pad columns / channels
    const int Wp = W + 2 * padw;
    const int64_t total = (int64_t)N * D * H * Wp;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int64_t q = i;
    const int wp = (int)(q % Wp); q /= Wp;
    const int h = (int)(q % H); q /= H;
    const int dd = (int)(q % D); q /= D;
    const int64_t n = q;
    const int w = wp - padw;
    for (int c = 0; c < Cpad; ++c) {
        float v = 0.f;
        if (c < C && w >= 0 && w < W) v = x[(((n * C + c) * D + dd) * H + h) * (int64_t)W + w];
        out[i * Cpad + c] = v;
    }
}

}  // namespace gsdd

using namespace gsdd;

extern "C" int gsdd_gemm(const gsdd_gemm_desc* d, void* stream) {
    GSDD_CHECK_ARG(d != nullptr, "null descriptor");
    GSDD_CHECK_ARG(d->in && d->w && d->out, "null tensor pointer");
    GSDD_CHECK_ARG(d->N > 0 && d->Do > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0 && d->Cin > 0, "bad sizes");
    GSDD_CHECK_ARG(d->Cin % 4 == 0 && d->in_pitch % 4 == 0, "Cin and in_pitch must be multiples of 4");
    GSDD_CHECK_ARG(d->ntaps >= 1 && (d->ntaps == 1 || d->taps != nullptr), "taps table required");
    GSDD_CHECK_ARG(d->gather == nullptr || d->ntaps == 1, "gather needs ntaps == 1");
    GSDD_CHECK_ARG((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "pro_scale/pro_shift come together");
    GSDD_CHECK_ARG(d->ln_stats == nullptr || (d->ln_gamma && d->ln_beta && d->ntaps == 1 && d->gather == nullptr),
                   "LayerNorm prologue needs gamma/beta and a plain row operand");
    GSDD_CHECK_ARG((d->ln_sel == nullptr && d->bvec == nullptr) || d->rows_per_batch > 0, "rows_per_batch required");
    GSDD_CHECK_ARG(d->out_mode >= 0 && d->out_mode <= 2, "bad out_mode");
    GSDD_CHECK_ARG(d->gather != nullptr || (d->Di > 0 && d->Hi > 0 && d->Wi > 0), "bad input dims");
    GSDD_CHECK_ARG(d->out_mode == 2 || (d->oD > 0 && d->oH > 0 && d->oW > 0), "bad output dims");
    GSDD_CHECK_ARG(d->out_mode != 0 || d->out_pitch >= d->Cout, "out_pitch too small");
    const int64_t M = (int64_t)d->N * d->Do * d->Ho * d->Wo;
    GSDD_CHECK_ARG(M < (1ll << 31), "more than 2^31 rows");
    GSDD_CHECK_ARG(d->gather != nullptr || (int64_t)d->N * d->Di * d->Hi * d->Wi < (1ll << 31), "more than 2^31 input rows");
    GSDD_CHECK_ARG((int64_t)d->ntaps * d->Cout * d->Cin < (1ll << 31), "more than 2^31 weight elements");
    hipStream_t st = (hipStream_t)stream;
    unsigned gx = (unsigned)((M + BM - 1) / BM);
    // bf16x3 on the matrix pipe when the contraction is long enough to pay for the splits; flags & GSDD_GEMM_EXACT_F32: the f32 MFMA
    GSDD_CHECK_ARG((d->flags & ~GSDD_GEMM_EXACT_F32) == 0, "unknown flags");
    const bool force_f32 = (d->flags & GSDD_GEMM_EXACT_F32) != 0;
    const bool x3 = !force_f32 && (int64_t)d->ntaps * d->Cin >= 64;
    if (d->Cout > 64) {
        const dim3 grid(gx, (d->Cout + 127) / 128);
        const bool big = x3 && M >= 256 * 1024 / 2 && (int64_t)d->ntaps * d->Cin >= 1024;        // long contraction, many rows
        if (big) {
            GSDD_ONCE_PER_DEVICE(attr_done,
                GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_kernel<128, true, 256, 512>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(GemmSmemX3<128, 256>)));
            );
            const dim3 grid2((unsigned)((M + 255) / 256), (d->Cout + 127) / 128);
            hipLaunchKernelGGL((gemm_kernel<128, true, 256, 512>), grid2, dim3(512), sizeof(GemmSmemX3<128, 256>), st, *d, M);
        } else if (x3) hipLaunchKernelGGL((gemm_kernel<128, true>), grid, dim3(256), sizeof(GemmSmemX3<128>), st, *d, M);
        else hipLaunchKernelGGL((gemm_kernel<128, false>), grid, dim3(256), sizeof(GemmSmem<128>), st, *d, M);
    } else {
        const dim3 grid(gx, 1);
        // few output channels over many rows and a long contraction (the Cout = 3 transposed conv): a 256 x 32 tile spends half
        // the matrix-pipe time of the 128 x 64 one on padding columns
        const bool narrow = x3 && d->Cout <= 32 && M >= 256 * 1024 / 2 && (int64_t)d->ntaps * d->Cin >= 1024;
        if (narrow) {
            GSDD_ONCE_PER_DEVICE(attr_done,
                GSDD_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_kernel<32, true, 256, 256>,
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(GemmSmemX3<32, 256>)));
            );
            const dim3 grid2((unsigned)((M + 255) / 256), 1);
            hipLaunchKernelGGL((gemm_kernel<32, true, 256, 256>), grid2, dim3(256), sizeof(GemmSmemX3<32, 256>), st, *d, M);
        } else if (x3) hipLaunchKernelGGL((gemm_kernel<64, true>), grid, dim3(256), sizeof(GemmSmemX3<64>), st, *d, M);
        else hipLaunchKernelGGL((gemm_kernel<64, false>), grid, dim3(256), sizeof(GemmSmem<64>), st, *d, M);
    }
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_row_stats(const float* x, int64_t M, int C, float eps, float* stats, void* stream) {
    GSDD_CHECK_ARG(x && stats && M > 0 && C > 0 && C % 4 == 0, "bad args");
    const int64_t threads = M * 16;
    hipLaunchKernelGGL(row_stats_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       M, C, eps, stats);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}

extern "C" int gsdd_ncdhw_to_rows(const float* x, int N, int C, int D, int H, int W, int Cpad, int padw, float* out,
                                  void* stream) {
    GSDD_CHECK_ARG(x && out && N > 0 && C > 0 && D > 0 && H > 0 && W > 0 && Cpad >= C && padw >= 0, "bad args");
    const int64_t total = (int64_t)N * D * H * (W + 2 * padw);
    hipLaunchKernelGGL(ncdhw_to_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, N, C, D, H, W, Cpad, padw, out);
    GSDD_CHECK_LAUNCH();
    return GSDD_OK;
}
