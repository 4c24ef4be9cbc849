#!/usr/bin/env python3
"""Time one D3PM training step (config C4 per-rank shape: bs 16, 16x16x16 tokens, 19 layers, K = 4096) on the HIP path."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsdd_amd  # noqa: E402
from gsdd_amd.d3pm_train import D3PMTrainer  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    L, K = 4096, 4096
    torch.manual_seed(0)
    d = gsdd_amd.DalleMaskImageEmbedding(num_embed=K, spatial_size=[64, 64], embed_dim=64)
    tr = gsdd_amd.Text2ImageTransformer(dalle=d, n_layer=19, n_embd=64, n_head=16, content_seq_len=L, block_activate="GELU2",
                                        content_spatial_size=[64, 64], condition_dim=512, diffusion_step=100)
    dm = gsdd_amd.DiffusionTransformer(transformer=tr, diffusion_step=100, alpha_init_type="alpha1", auxiliary_loss_weight=5e-4,
                                       adaptive_auxiliary_loss=True, guidance_scale=2, content_seq_len=L).cuda()
    trainer = D3PMTrainer(dm, lr=1e-4)
    g = torch.Generator().manual_seed(1)
    tok = torch.randint(0, K, (B, L), generator=g).cuda()
    cond = torch.zeros(B, 1, 512).cuda()
    losses = []
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    times = []
    hosts = []
    for i in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = trainer.step(tok, cond)
        host = time.perf_counter() - t0                     # python has enqueued the whole step
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        hosts.append(host * 1e3)
        losses.append(float(loss[0]))
        times.append(dt * 1e3)
        print(f"step {i}: loss {losses[-1]:.4f}  {dt * 1e3:.1f} ms  ({B / dt:.1f} samples/s)  mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    if steps > 4:
        tail = sorted(times[2:])
        print(f"median of steps 2..{steps - 1}: {tail[len(tail) // 2]:.2f} ms  (min {tail[0]:.2f}); host enqueue time median "
              f"{sorted(hosts[2:])[len(hosts[2:]) // 2]:.2f} ms")


if __name__ == "__main__":
    main()
