"""Counter-based noise source shared by the oracle and the HIP kernels (TEST INFRASTRUCTURE).

The reference draws its Gumbel noise with ``torch.rand_like(logits)`` on a ``(B, K+1, L)``
tensor (reference: src/models/motionencoder/diffusion_transformer.py:354-359).  A torch CPU
stream cannot be reproduced inside a GPU kernel, so parity mode injects noise from a
Philox4x32-10 generator that is restated identically here (numpy) and in
``csrc/d3pm_step.hip``.  The golden-fixture harness feeds the same uniforms to the
*reference* by patching ``torch.rand_like``.

Layout of one draw ("stream" = the index of the ``rand_like`` call, e.g. the reverse step):

    row    = b * L + l                      # one row per token position
    KP     = (K+1 + 3) // 4 * 4             # row length padded to a multiple of 4
    ctr    = (row * (KP // 4) + k // 4)     # 64-bit Philox counter (words 0,1)
    word   = k % 4                          # which of the 4 outputs
    key    = (seed_lo, seed_hi)
    ctr2,3 = (stream, 0)
    u      = (out[word] >> 8) * 2**-24      # float32 in [0, 1)
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All inputs are uint32 arrays (or scalars); returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.asarray(c1, dtype=np.uint64)
    c2 = np.asarray(c2, dtype=np.uint64)
    c3 = np.asarray(c3, dtype=np.uint64)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def uniform_rows(seed, stream, n_rows, n_cols, row0=0):
    """Uniforms for rows ``row0 .. row0+n_rows-1`` and columns ``0..n_cols-1`` -> float32 (n_rows, n_cols)."""
    kp4 = (n_cols + 3) // 4
    rows = (np.arange(n_rows, dtype=np.uint64) + np.uint64(row0))[:, None]
    ctr = rows * np.uint64(kp4) + np.arange(kp4, dtype=np.uint64)[None, :]
    c0 = (ctr & _MASK).astype(np.uint32)
    c1 = (ctr >> np.uint64(32)).astype(np.uint32)
    c2 = np.full_like(c0, np.uint32(stream & 0xFFFFFFFF))
    c3 = np.zeros_like(c0)
    o = philox4x32_10(c0, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(o, axis=-1).reshape(n_rows, kp4 * 4)[:, :n_cols]
    return ((words >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def uniform_bkl(seed, stream, B, K1, L, row0=0):
    """The uniforms for one ``rand_like`` on a (B, K+1, L) tensor, in the reference's (B, K+1, L) layout."""
    u = uniform_rows(seed, stream, B * L, K1, row0=row0)          # (B*L, K1)
    return np.ascontiguousarray(u.reshape(B, L, K1).transpose(0, 2, 1))
