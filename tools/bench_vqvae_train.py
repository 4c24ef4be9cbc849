#!/usr/bin/env python3
"""Time one VQ-VAE training step (config C2: 16x128x128 clips, n_hiddens 256, 3 res layers, downsample [1,8,8], 4096 codes)
on the HIP path: forward with batch statistics + codebook EMA, full backward, Adam."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsdd_amd  # noqa: E402
from gsdd_amd.vqvae_trainer import VQVAETrainer  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    torch.manual_seed(0)
    vq = gsdd_amd.VQVAE(None, 128, 4096, 256, 3, [1, 8, 8], 16, 128).cuda().train()
    tr = VQVAETrainer(vq, lr=4e-4)
    g = torch.Generator().manual_seed(1)
    x = (torch.rand((B, 3, 16, 128, 128), generator=g) - 0.5).cuda()
    for i in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = tr.step(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"step {i}: recon {float(losses['recon_loss']):.4f} commit {float(losses['commitment_loss']):.5f}  {dt * 1e3:.1f} ms  "
              f"({B / dt:.1f} clips/s)  mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)


if __name__ == "__main__":
    main()
