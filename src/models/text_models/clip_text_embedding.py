"""Stand-in for the reference's frozen CLIP text tower (src/models/text_models/clip_text_embedding.py:22-38).

The real tower needs the `clip` package and a network fetch of ViT-B/32 — neither exists here — and the reference
zeroes its output anyway (src/models/networks/discrete_diffusion.py:25,49).  This provider keeps the call contract
(list[str] -> (B, clip_dim) float tensor) with a deterministic hash embedding."""
import hashlib

import torch
import torch.nn as nn


class CLIPTextEmbedding(nn.Module):
    def __init__(self, clip_dim=512, **kwargs):
        super().__init__()
        self.clip_dim = clip_dim
        self.register_buffer("_anchor", torch.zeros(1))

    @torch.no_grad()
    def forward(self, texts):
        rows = []
        for t in texts:
            seed = int.from_bytes(hashlib.sha256(t.encode()).digest()[:8], "little")
            g = torch.Generator().manual_seed(seed)
            v = torch.randn(self.clip_dim, generator=g)
            rows.append(v / v.norm())
        return torch.stack(rows).to(self._anchor.device)
