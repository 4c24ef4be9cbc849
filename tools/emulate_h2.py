#!/usr/bin/env python3
"""Accuracy of the fused layer kernel's operand formats, emulated in numpy against fp64: f16 hi + lo (the `h2` kernel: products
hi.hi + hi.lo + lo.hi, weights scaled by 2^8) vs three bf16 pieces (the `x3p` kernel: six products) vs numpy's own f32 matmul.
Products are summed in fp64 and rounded to f32 once, i.e. the operand format's error alone."""
import numpy as np

rng = np.random.default_rng(0)


def f16x2(x, scale=1.0):
    xs = (x * scale).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float32).astype(np.float16)
    return hi.astype(np.float64) / scale, lo.astype(np.float64) / scale


def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def bf16x3(x):
    a = bf16(x)
    r = (x - a).astype(np.float32)
    b = bf16(r)
    c = bf16((r - b).astype(np.float32))
    return a.astype(np.float64), b.astype(np.float64), c.astype(np.float64)


for K, ws, as_ in [(64, 0.02, 1.0), (256, 0.02, 0.5), (64, 0.1, 1.0), (64, 0.02, 8.0)]:
    W = (rng.standard_normal((256, K)) * ws).astype(np.float32)
    A = (rng.standard_normal((K, 512)) * as_).astype(np.float32)
    ref = W.astype(np.float64) @ A.astype(np.float64)
    wh, wl = f16x2(W, 256.0)
    ah, al = f16x2(A, 16.0)
    h2 = wh @ ah + wh @ al + wl @ ah
    w1, w2, w3 = bf16x3(W)
    a1, a2, a3 = bf16x3(A)
    x3 = w1 @ a1 + w1 @ a2 + w2 @ a1 + w1 @ a3 + w3 @ a1 + w2 @ a2
    for name, v in (("numpy f32", (W @ A).astype(np.float64)), ("f16 hi+lo", h2.astype(np.float32).astype(np.float64)),
                    ("bf16 x 3", x3.astype(np.float32).astype(np.float64))):
        e = np.abs(v - ref)
        print(f"K={K:3d} |w|~{ws} |a|~{as_}: {name:10s} max err {e.max():.2e}  rms {np.sqrt((e ** 2).mean()):.2e}  "
              f"(|out| rms {np.sqrt((ref ** 2).mean()):.2e})")
