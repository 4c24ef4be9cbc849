#!/usr/bin/env python3
"""Fused attention backward vs the two-kernel variant and fp64 autograd on small shapes (debug aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsdd_amd  # noqa: E402,F401
from gsdd_amd import ops  # noqa: E402


def run(B, L, scale, H=16):
    g = torch.Generator().manual_seed(9)
    q = (torch.randn(B, H, L, 4, generator=g) * scale).double().requires_grad_(True)
    k = (torch.randn(B, H, L, 4, generator=g) * scale).double().requires_grad_(True)
    v = torch.randn(B, H, L, 4, generator=g).double().requires_grad_(True)
    dO = torch.randn(B, H, L, 4, generator=g).double()
    o = torch.softmax((q @ k.transpose(-1, -2)) * 0.5, dim=-1) @ v
    o.backward(dO)
    hm = lambda z: z.detach().float().permute(1, 0, 2, 3).reshape(H, B * L, 4).contiguous().cuda()
    rm = lambda z: z.detach().float().permute(0, 2, 1, 3).reshape(B * L, H * 4).contiguous().cuda()
    qh, kh, vh = hm(q), hm(k), hm(v)
    out = torch.empty((B * L, H * 4), device="cuda")
    lse = torch.empty((H * B * L,), device="cuda")
    ops.d3pm_attention_train(qh, kh, vh, B, L, H, out, lse, ws=ops.d3pm_attention_workspace(B, L, H, "cuda"))
    res = {}
    for mode in ("fused", "split"):
        ws = ops.d3pm_attention_bwd_workspace(B, L, H, "cuda")
        res[mode] = ops.d3pm_attention_bwd(qh, kh, vh, out, rm(dO), lse, B, L, H, ws=ws, variant="split" if mode == "split" else None).cpu().double()
    want = {"dq": rm(q.grad).cpu().double(), "dk": rm(k.grad).cpu().double(), "dv": rm(v.grad).cpu().double()}
    for i, name in enumerate(("dq", "dk", "dv")):
        w = want[name]
        for mode in ("fused", "split"):
            got = res[mode][:, 64 * i:64 * (i + 1)]
            err = (got - w).abs().max().item() / w.abs().max().item()
            print(f"B={B} L={L} x{scale} {name} {mode}: rel max err {err:.3e}")
    if (res["fused"][:, :64] - want["dq"]).abs().max() > 1e-3 * want["dq"].abs().max():
        f, w = res["fused"][:, :64], want["dq"]
        print("dq fused[0, :8]", f[0, :8].tolist())
        print("dq want [0, :8]", w[0, :8].tolist())
        ratio = (f / w)
        print("ratio median", ratio.median().item(), "rows with small err:",
              ((f - w).abs().max(dim=1).values < 1e-4 * w.abs().max()).nonzero().flatten()[:20].tolist())


if __name__ == "__main__":
    for (B, L, s) in [(1, 32, 1.0), (1, 64, 1.0), (2, 544, 1.0), (1, 1024, 1.0)]:
        run(B, L, s)
