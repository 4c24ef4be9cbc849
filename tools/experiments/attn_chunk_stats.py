#!/usr/bin/env python3
"""Development tool: how the adaptive attention kernel decided its chunks on the denoiser's own q, k, v (trained-like weights, one row,
every layer asked for).  Needs a library built with -DGSDD_DEV_ATTN_STATS (GSDD_LIB_PATH): the kernel then adds, per (wave, chunk), to
redo[1] = cleared whole by the bounds (runs the hi-only loop), [2] = partly cleared, [3] = nothing cleared, [4] / [5] = (sub-tile, tile)
pairs cleared / in all, [6] = chunks of workgroups that computed centred norms, [7] = chunks run with the measured test armed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsdd_amd  # noqa: E402
from gsdd_amd import ops  # noqa: E402
from gsdd_amd.d3pm_train import D3PMTrainer  # noqa: E402
import bench  # noqa: E402


def main():
    L, K, H = 4096, 4096, 16
    torch.manual_seed(0)
    d = gsdd_amd.DalleMaskImageEmbedding(num_embed=K, spatial_size=[64, 64], embed_dim=64)
    tr = gsdd_amd.Text2ImageTransformer(dalle=d, n_layer=19, n_embd=64, n_head=16, content_seq_len=L, block_activate="GELU2",
                                        content_spatial_size=[64, 64], condition_dim=512, diffusion_step=100)
    dm = gsdd_amd.DiffusionTransformer(transformer=tr, diffusion_step=100, alpha_init_type="alpha1", auxiliary_loss_weight=5e-4,
                                       adaptive_auxiliary_loss=True, guidance_scale=2, content_seq_len=L).cuda()
    g = torch.Generator().manual_seed(1)
    x0 = torch.randint(0, K, (1, L), generator=g).cuda()
    xt = torch.where(torch.rand((1, L), generator=g).cuda() < 0.5, torch.full_like(x0, K), x0)
    cond = (torch.randn((1, 1, 512), generator=g) * 0.5).cuda()
    t = torch.tensor([50], device="cuda")
    bench.trained_like_weights(dm)
    sv = D3PMTrainer(dm, lr=1e-4)._forward(xt, cond, t)
    for li in (1, 5, 10, 18):
        qkv = sv["layers"][li]["qkv"]
        q, k, v = (qkv[i * H:(i + 1) * H].contiguous() for i in range(3))
        out = torch.empty((L, H * 4), device="cuda")
        ws = ops.d3pm_attention_workspace(1, L, H, "cuda")
        ctr = torch.zeros((256,), dtype=torch.int64, device="cuda")
        ops.d3pm_attention(q, k, v, 1, L, H, out, ws=ws, redo=ctr, mode="a8")
        c = ctr.cpu().tolist()
        tot = max(c[1] + c[2] + c[3], 1)
        print(f"layer {li:2d}: chunks cleared whole {c[1] / tot:.3f}  partly {c[2] / tot:.3f}  not at all {c[3] / tot:.3f}  pairs cleared "
              f"{c[4] / max(c[5], 1):.3f}  chunks with centred norms {c[6] / tot:.3f}  measured test armed {c[7] / tot:.3f}  redo events {c[0]}")
        import numpy as np
        bits = np.array(c[8:8 + 128], dtype=np.uint32)
        cn_dev = bits.view(np.float32)
        k0 = k[0].double()                                         # head 0: [L][4]
        kbar = k0.mean(0)
        cn_ref = (k0 - kbar).norm(dim=-1).view(128, 32).amax(1).cpu().numpy()
        kcj = np.array(c[200:204], dtype=np.uint32).view(np.float32)
        km = np.array(c[204:208], dtype=np.uint32).view(np.float32)
        qs0 = q[0, :64].double() * (0.5 * 1.4426950408889634)
        kc_ref = [(3.98 / qs0[16 * j:16 * j + 16].norm(dim=-1).max()).item() for j in range(4)]
        print("   cn dev[:6]", cn_dev[:6], "ref[:6]", cn_ref[:6], "max rel dev", float(np.abs(cn_dev / cn_ref - 1).max()))
        print("   kcj dev", kcj, "ref", kc_ref, " kmean dev", km, "ref", kbar.cpu().numpy())


if __name__ == "__main__":
    main()
