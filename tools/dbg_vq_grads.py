#!/usr/bin/env python3
"""Per-tensor gradient errors of the full-size VQ-VAE case (tests/test_gpu_vqvae_training.py::case 'full') in state_dict order."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsdd_amd  # noqa: E402
from tests.test_gpu_vqvae_training import build_vqvae, case, oracle_grads  # noqa: E402
from gsdd_amd.vqvae_trainer import VQVAETrainer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "full"
x, sd, cfg, perm = case(gsdd_amd, None, name)
out, want = oracle_grads(x, sd, cfg, perm)
if "--f64" in sys.argv:
    _, want = oracle_grads(x.double(), {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}, cfg, perm)
    want = {k: v.float() for k, v in want.items()}
m = build_vqvae(gsdd_amd, sd, cfg)
pt = torch.from_numpy(np.asarray(perm))
m.perm_source = lambda n: pt
tr = VQVAETrainer(m)
sv, losses = tr.forward(x.cuda())
got = tr.backward(sv)
print("code mismatches", int((sv["idx"].cpu() != out["encodings"].reshape(-1)).sum()))
print("z err", (sv["z"].cpu() - out["z"].permute(0, 2, 3, 4, 1).reshape(-1, out["z"].shape[1])).abs().max().item(), "z scale", out["z"].abs().max().item())
print("recon err", (sv["x_recon"].cpu() - out["pred_data"]).abs().max().item())
gmax = max(w.abs().max().item() for w in want.values())
for k, w in want.items():
    g = got[k].detach().cpu()
    scale = max(w.abs().max().item(), 1e-3 * gmax)
    err = (g - w).abs().max().item() / scale
    cos = torch.nn.functional.cosine_similarity(g.flatten().double(), w.flatten().double(), dim=0).item()
    print(f"{k:55s} |g|max {w.abs().max().item():.3e}  rel err {err:.2e}  cos {cos:.6f}  norm ratio {g.norm().item() / max(w.norm().item(), 1e-30):.5f}")
