#!/usr/bin/env python3
"""Error of the sampler's attention variants (GSDD_ATTN_P=22: P = f16 hi + lo; 11: hi only) against fp64, head dim 4, L = 4096,
over softmax regimes from flat to peaky (q, k scaled): max and rms absolute error of the output, and the effective number of keys
N_eff = 1 / sum(softmax^2) of the rows.  One JSON line per (scale, variant)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsdd_amd  # noqa: E402,F401
from gsdd_amd import ops  # noqa: E402


def main():
    B, L, H = 1, 4096, 16
    g = torch.Generator().manual_seed(0)
    for scale in (0.05, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0):
        q = (torch.randn(B, H, L, 4, generator=g) * scale)
        k = (torch.randn(B, H, L, 4, generator=g) * scale)
        v = torch.randn(B, H, L, 4, generator=g)
        qd, kd, vd = q.double().cuda(), k.double().cuda(), v.double().cuda()
        pr = torch.softmax(qd @ kd.transpose(-1, -2) * 0.5, dim=-1)
        want = (pr @ vd).permute(0, 2, 1, 3).reshape(B * L, H * 4)
        neff = (1.0 / (pr * pr).sum(-1))
        hm = lambda z: z.float().permute(1, 0, 2, 3).reshape(H, B * L, 4).contiguous().cuda()
        qh, kh, vh = hm(q), hm(k), hm(v)
        out = torch.empty((B * L, H * 4), device="cuda")
        aws = ops.d3pm_attention_workspace(B, L, H, "cuda")
        for pbits in os.environ.get("GSDD_BENCH_PMODES", "22,11,a8,a12").split(","):
            os.environ["GSDD_ATTN_P"] = pbits
            ops.d3pm_attention(qh, kh, vh, B, L, H, out, ws=aws)
            err = (out.double() - want).abs()
            print(json.dumps({"scale": scale, "P_mode": pbits, "max_err": err.max().item(), "rms_err": err.pow(2).mean().sqrt().item(),
                              "neff_median": neff.median().item(), "neff_min": neff.min().item()}), flush=True)
    os.environ.pop("GSDD_ATTN_P", None)


if __name__ == "__main__":
    main()
