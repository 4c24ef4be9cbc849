#!/usr/bin/env python3
"""Generate tests/golden/i3d.npz from the REFERENCE's InceptionI3d (src/models/motionencoder/pytorch_i3d.py), imported in the
build container only (the module is self-contained: torch + numpy).  The 12.3 M weights are not stored: both sides rebuild them
from a seed with oracle/i3d.py::seeded_state_dict over the reference's state_dict keys (stored, with their shapes).  The fixture
holds the inputs' seeds, the reference's logits / pooled features for two clips (16x224x224: the evaluator's shape; 32 frames:
more than one time step into the time-mean) and a strided sample of every end point's output (localises a mismatch)."""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
from oracle.i3d import ENDPOINTS, seeded_state_dict  # noqa: E402


def endpoint_sample(t):
    """A deterministic sub-sample of an end point's (B,C,T,H,W) output: every 7th channel, 3rd frame, 5th row / column."""
    return t[:, ::7, ::3, ::5, ::5].contiguous().numpy()


def main():
    from src.models.motionencoder.pytorch_i3d import InceptionI3d        # the reference, build container only
    torch.manual_seed(0)
    ref = InceptionI3d()
    keys = [(k, tuple(v.shape)) for k, v in ref.state_dict().items()]
    sd = seeded_state_dict(keys, seed=2024)
    ref.load_state_dict(sd)
    ref.eval()
    res = {"keys": np.array([k for k, _ in keys]), "shapes": np.array([list(s) + [0] * (5 - len(s)) for _, s in keys], dtype=np.int64),
           "ndims": np.array([len(s) for _, s in keys], dtype=np.int64), "weight_seed": np.int64(2024)}
    for tag, (B, T, seed) in {"a": (1, 16, 1), "b": (2, 32, 2)}.items():
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(B, 3, T, 224, 224, generator=g)
        with torch.no_grad():
            h = x
            for name, _, _ in ENDPOINTS:
                h = ref._modules[name](h)
                if tag == "a":
                    res[f"ep_{name}"] = endpoint_sample(h)
            feats = ref.extract_features(x)
            logits = ref(x)
        assert torch.equal(feats, ref.avg_pool(h))
        res[f"x_{tag}"] = np.array([B, T, seed], dtype=np.int64)
        res[f"features_{tag}"] = feats.numpy()
        res[f"logits_{tag}"] = logits.numpy()
        print(tag, tuple(feats.shape), tuple(logits.shape), float(logits.abs().max()), float(feats.abs().max()))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "i3d.npz"), **res)
    print("params", sum(int(np.prod(s)) for k, s in keys if "num_batches" not in k))


if __name__ == "__main__":
    main()
