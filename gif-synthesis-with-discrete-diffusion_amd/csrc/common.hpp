// Shared helpers for the gfx950 kernels (host error plumbing, wave64 reductions, Philox).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <string>

#include "../../include/gsdd.h"

namespace gsdd {

void set_error(const std::string& s);

#define GSDD_CHECK_ARG(cond, msg)                                                      \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            ::gsdd::set_error(std::string(__func__) + ": " + (msg) + " [" #cond "]"); \
            return GSDD_E_ARG;                                                         \
        }                                                                              \
    } while (0)

#define GSDD_CHECK_HIP(expr)                                                                   \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            ::gsdd::set_error(std::string(__func__) + ": " #expr " -> " + hipGetErrorString(e_)); \
            return GSDD_E_HIP;                                                                 \
        }                                                                                      \
    } while (0)

#define GSDD_CHECK_LAUNCH() GSDD_CHECK_HIP(hipGetLastError())

// Per-device, per-call-site set-up (hipFuncSetAttribute for dynamic LDS above 64 KB is per device).  The call site's mask holds one bit per
// device; a bit is set only AFTER every call of the set-up body has succeeded, so a failed hipFuncSetAttribute is reported by this call
// and retried by the next one instead of surfacing later as an opaque launch error.  Atomic: two host threads may race through the body
// (the attribute calls are idempotent), neither can skip it before it has succeeded once.
inline bool device_setup_pending(const std::atomic<unsigned long long>& mask, unsigned long long& bit) {
    int dev = 0;
    bit = 0ull;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    bit = 1ull << (dev & 63);
    return (mask.load(std::memory_order_acquire) & bit) == 0ull;
}
#define GSDD_ONCE_PER_DEVICE(mask_name, ...)                                   \
    do {                                                                       \
        static std::atomic<unsigned long long> mask_name{0ull};                \
        unsigned long long bit_ = 0ull;                                        \
        if (::gsdd::device_setup_pending(mask_name, bit_)) {                   \
            __VA_ARGS__                                                        \
            mask_name.fetch_or(bit_, std::memory_order_release);               \
        }                                                                      \
    } while (0)

constexpr int WAVE = 64;

// ---------------------------------------------------------------- wave64 reductions (all lanes get the result)
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------- Philox4x32-10 (same stream as oracle/philox.py)
__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one widening multiply per product (v_mad_u64_u32 issues at the rate of a single v_mul_lo/hi_u32)
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    return make_uint4(c0, c1, c2, c3);
}
// uniforms for 4 consecutive columns (col4*4 .. col4*4+3) of row `row` in a draw whose rows hold kp4 quads
__device__ __forceinline__ float4 philox_uniform4(uint64_t seed, uint32_t stream_id, uint64_t row, uint32_t kp4,
                                                  uint32_t col4) {
    const uint64_t ctr = row * (uint64_t)kp4 + col4;
    const uint4 r = philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), stream_id, 0u, (uint32_t)seed,
                                  (uint32_t)(seed >> 32));
    // top 24 bits / 2^24, written as (r with its low byte cleared) / 2^32: the same value, the mask issues at full rate
    constexpr float S = 1.0f / 4294967296.0f;
    constexpr uint32_t M = 0xFFFFFF00u;
    return make_float4((float)(r.x & M) * S, (float)(r.y & M) * S, (float)(r.z & M) * S, (float)(r.w & M) * S);
}

// ---------------------------------------------------------------- pre-split K / V images of the matrix-pipe attention kernel
// (written by d3pm_attn_prep_kernel, or directly by the fused layer kernel's q|k|v epilogue)
__device__ __forceinline__ uint32_t bf16_rn(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// x = a + b + c with a, b, c bf16 (error-free up to the last piece's rounding, < 2^-24 |x|)
__device__ __forceinline__ void split3(float x, uint32_t& a, uint32_t& b, uint32_t& c) {
    a = bf16_rn(x);
    const float r = x - __uint_as_float(a << 16);
    b = bf16_rn(r);
    const float r2 = r - __uint_as_float(b << 16);
    c = bf16_rn(r2);
}
__device__ __forceinline__ uint4 pack8(const uint32_t (&lo)[4], const uint32_t (&hi)[4]) {
    return make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
}
//   Kp[row][2] uint4 : pieces A=[k1|k2], B=[k3|k1] (bf16), slots swapped for (row & 15) >= 8   (row = h*M + b*L + key)
__device__ __forceinline__ void kv_image_store_k(const float (&ks)[4], int64_t row, uint4* kp) {
    // the same round-to-nearest pieces as split3, through the hardware conversion (v_cvt_pk_bf16_f32) instead of integer rounding
    typedef __bf16 kbf16x8 __attribute__((ext_vector_type(8)));
    kbf16x8 a, b;                                   // a = [k1 | k2], b = [k3 | k1]
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const __bf16 p1 = (__bf16)ks[d];
        const float r1 = ks[d] - (float)p1;
        const __bf16 p2 = (__bf16)r1;
        const __bf16 p3 = (__bf16)(r1 - (float)p2);
        a[d] = p1; a[4 + d] = p2; b[d] = p3; b[4 + d] = p1;
    }
    const int sw = (int)((row >> 3) & 1);
    kp[row * 2 + sw] = __builtin_bit_cast(uint4, a);
    kp[row * 2 + (sw ^ 1)] = __builtin_bit_cast(uint4, b);
}
//   Vp[row/32][4][16] uint4 : per 32-key pair-tile, [key group g][col j] -> 8 f16 (tile0 keys 4g+r, tile1 keys 4g+r),
//                             cols = [v1 | v2*2^11 | v3*2^22 | 1 | 0 0 0]
__device__ __forceinline__ void kv_image_store_v(const float (&vs)[4], int64_t row, uint4* vp) {
    _Float16 col[16];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const _Float16 a = (_Float16)vs[d];
        const float r1 = (vs[d] - (float)a) * 2048.f;
        const _Float16 b2 = (_Float16)r1;
        const float r2 = (r1 - (float)b2) * 2048.f;
        col[d] = a; col[4 + d] = b2; col[8 + d] = (_Float16)r2;
    }
    col[12] = (_Float16)1.f;
    col[13] = col[14] = col[15] = (_Float16)0.f;
    const int64_t pair = row >> 5;
    const int kk = (int)(row & 31), th = kk >> 4, kt = kk & 15, g = kt >> 2, r = kt & 3;
    _Float16* dst = reinterpret_cast<_Float16*>(vp + (pair * 4 + g) * 16) + 4 * th + r;
#pragma unroll
    for (int j = 0; j < 16; ++j) dst[j * 8] = col[j];
}

// Per-tile key norms of the attention workspace: knorm[(h*M + b*L + key) >> 5] = an upper bound of max ||k|| over the 32 keys of
// that pair-tile (f32, rounded up).  With the wave's ||q|| it bounds every score of a (query, pair-tile) by Cauchy-Schwarz, which is
// how the attention kernel proves "no probability of this tile can matter" without looking at the scores (d3pm_attention.hip).
// ksum[(h*M + b*L + key) >> 5] = the sum of the 32 keys of that pair-tile (float4, fixed summation tree: bitwise reproducible): the
// attention kernel adds the tiles of a (b, h) to the mean key, which gives it a lower bound of every FINAL row sum before it has seen a
// key (Jensen: log2 sum_j 2^s_j >= log2 L + mean_j s_j = log2 L + q . kmean).
// Workspace layout (gsdd_d3pm_attention_workspace_bytes): K image 32 B per (key, head) | V image 32 B | ksum 16 B per 32 keys | knorm 4 B
// per 32 keys.
__host__ __device__ __forceinline__ float4* kv_image_ksum(void* workspace, int64_t rows) {
    return reinterpret_cast<float4*>(reinterpret_cast<char*>(workspace) + rows * 64);
}
__host__ __device__ __forceinline__ float* kv_image_knorm(void* workspace, int64_t rows) {
    return reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + rows * 64 + ((rows + 31) / 32) * 16);
}
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)));
}
template <int CTRL>
__device__ __forceinline__ float dpp_min(float v) {
    return fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false)));
}
// maximum of a non-negative value over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15): xor 1, xor 2, half mirror, mirror
__device__ __forceinline__ float row16_max(float v) {
    v = dpp_max<0xB1>(v);        // quad_perm [1,0,3,2]
    v = dpp_max<0x4E>(v);        // quad_perm [2,3,0,1]
    v = dpp_max<0x141>(v);       // row_half_mirror
    return dpp_max<0x140>(v);    // row_mirror
}
__device__ __forceinline__ float row16_min(float v) {
    v = dpp_min<0xB1>(v);
    v = dpp_min<0x4E>(v);
    v = dpp_min<0x141>(v);
    return dpp_min<0x140>(v);
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// lane 15 of DPP rows 0 / 2 broadcast to the lanes of rows 1 / 3 (row_bcast:15 with row_mask 0b1010; the other rows read 0): the step
// that joins the two rows of a 32-lane half without leaving the vector ALU (a __shfl_xor goes through the LDS permute path)
__device__ __forceinline__ float dpp_prev_row_lane15(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));
}
// sum over the 32 lanes of a half; valid in the UPPER 16 lanes of the half (lanes 16..31 and 48..63).  A fixed tree: the same bits for
// the same inputs in every producer and every run.
__device__ __forceinline__ float half32_sum(float v) {
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    return v + dpp_prev_row_lane15(v);
}
// ||k|| bound of one pair-tile from per-lane squared norms: lanes of a 32-lane half hold the 32 keys; valid in the upper 16 lanes of
// the half (squared norms are non-negative, so the 0 the lower rows read in the joining step is harmless)
__device__ __forceinline__ float half32_norm_bound(float n2) {
    float v = row16_max(n2);
    v = fmaxf(v, dpp_prev_row_lane15(v));
    return sqrtf(v) * 1.000001f + 1e-30f;
}
// fused layer kernels: lane (li = row of the 32-row group, h) holds k of heads 8 (q >> 2) + 2 (q & 3) + h in o[q]
__device__ __forceinline__ void kv_image_store_knorm(const float4 (&o)[8], int h, int li, int64_t grp, int64_t M, float* knorm,
                                                     float4* ksum) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float n2 = (o[q].x * o[q].x + o[q].y * o[q].y) + (o[q].z * o[q].z + o[q].w * o[q].w);
        const float nb = half32_norm_bound(n2);
        const float4 sk = make_float4(half32_sum(o[q].x), half32_sum(o[q].y), half32_sum(o[q].z), half32_sum(o[q].w));
        const int hd = 8 * (q >> 2) + 2 * (q & 3) + h;
        if (li == 16) {                           // (the reductions are valid in the upper 16 lanes of each half)
            knorm[(int64_t)hd * (M >> 5) + grp] = nb;
            ksum[(int64_t)hd * (M >> 5) + grp] = sk;
        }
    }
}

// axial_attention_mfma.hip: register-resident MFMA kernels for 16-position lines; false = shape not covered
bool axial_attention_mfma_launch(const float* qkv, int N, int T, int H, int W, int C, int n_head, int axis, float* out,
                                 hipStream_t st);
bool axial_attention_bwd_mfma_launch(const float* qkv, const float* datt, int N, int T, int H, int W, int C, int n_head, int axis,
                                     float* dqkv, hipStream_t st);

}  // namespace gsdd
