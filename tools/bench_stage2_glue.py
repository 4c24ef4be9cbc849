#!/usr/bin/env python3
"""Time the reference-shaped stage-2 training step (`python src/train.py model=discrete_diffusion`, C4 per-GPU shapes: bs 16 clips of
16x128x128, 16x16x16 tokens, 19 layers, K = 4096) through the generator glue: frozen VQ-VAE encode -> D3PM objective -> manual_backward
-> Adam, with the glue's two by-product decodes deferred (default) or computed as the reference does (GSDD_EAGER_OUTPUTS=1)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsdd_amd  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    L, K = 4096, 4096
    torch.manual_seed(0)
    d = gsdd_amd.DalleMaskImageEmbedding(num_embed=K, spatial_size=[64, 64], embed_dim=64)
    tr = gsdd_amd.Text2ImageTransformer(dalle=d, n_layer=19, n_embd=64, n_head=16, content_seq_len=L, block_activate="GELU2",
                                        content_spatial_size=[64, 64], condition_dim=512, diffusion_step=100)
    dm = gsdd_amd.DiffusionTransformer(transformer=tr, diffusion_step=100, alpha_init_type="alpha1", auxiliary_loss_weight=5e-4,
                                       adaptive_auxiliary_loss=True, guidance_scale=2, content_seq_len=L).cuda().train()
    vq = gsdd_amd.VQVAE(None, 128, K, 256, 3, [1, 8, 8], 16, 128).cuda().eval()
    for p in vq.parameters():
        p.requires_grad_(False)
    gen = gsdd_amd.DiscreteDiffusion(textencoder=lambda texts: torch.zeros(len(texts), 512), diffusion_model=dm)
    opt = torch.optim.Adam(dm.parameters(), lr=1e-4, betas=(0.5, 0.999))
    g = torch.Generator().manual_seed(1)
    batch = {"video": (torch.rand((B, 3, 16, 128, 128), generator=g) - 0.5).cuda(), "text": ["a"] * B}
    times = []
    for i in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = gen(batch, vq)
        loss = out["losses"].mean() if torch.is_tensor(out["losses"]) else out["losses"]
        opt.zero_grad()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
        pend = out.pending() if hasattr(out, "pending") else []
    tail = sorted(times[2:])
    print(f"stage-2 glue step bs {B}: median {tail[len(tail) // 2]:.1f} ms (min {tail[0]:.1f}); outputs never read: {pend}")


if __name__ == "__main__":
    main()
