"""Text-conditioning provider with the reference's call contract (src/models/text_models/clip_text_embedding.py:11-69):
`CLIPTextEmbedding(clip_dim=512)(list[str]) -> (B, clip_dim)` float tensor.

The reference builds a frozen OpenAI CLIP ViT-B/32 text tower through the `clip` package, which downloads its weights (:22-29).
Neither the package nor the weights nor the BPE vocabulary exist offline, and the generator zeroes the embedding anyway
(src/models/networks/discrete_diffusion.py:25, :49), so by default this provider is a deterministic hash embedding that keeps the
contract.  When a LOCAL copy of the text tower is supplied (`weights=<directory>` in the Hugging Face layout: config.json,
model.safetensors, vocab.json, merges.txt of `openai/clip-vit-base-patch32`), the real tower runs instead, reproducing the reference's
recipe: tokenise with context length 22 (start + 20 + end, truncated), zero-pad the ids to 77, take the projected feature at the end-of-text
token (`clip_model.encode_text`, :56-64).  Nothing is ever fetched: the directory must exist.  **Parity unpinned**: no reference output of
this tower can be produced in the build container; the plumbing (tokenisation recipe, padding, pooling position, frozen weights) is
tested on a small randomly initialised tower of the same architecture (tests/test_host_logic.py).  It is off the hot path
(frozen, one call per batch) and runs as plain PyTorch on whatever device the module lives on."""
import hashlib
import os

import torch
import torch.nn as nn


class CLIPTextEmbedding(nn.Module):
    MAX_TEXT_LEN = 20                  # :57 (the reference hard-codes HumanML's limit)
    CONTEXT_DEFAULT = 77               # :58

    def __init__(self, clip_dim=512, weights=None, **kwargs):
        super().__init__()
        self.clip_dim = clip_dim
        self.register_buffer("_anchor", torch.zeros(1))
        self.tokenizer, self.clip_model = None, None
        if weights:
            if not os.path.isdir(weights):
                raise FileNotFoundError(f"CLIPTextEmbedding(weights={weights!r}): not a directory; a local copy of the CLIP text tower "
                                        "(config.json, model.safetensors, vocab.json, merges.txt) is needed -- nothing is downloaded")
            from transformers import CLIPTextModelWithProjection, CLIPTokenizer
            self.tokenizer = CLIPTokenizer.from_pretrained(weights, local_files_only=True)
            self.clip_model = CLIPTextModelWithProjection.from_pretrained(weights, local_files_only=True).eval()
            for p in self.clip_model.parameters():                 # load_and_freeze_clip (:27-38)
                p.requires_grad = False
            if self.clip_model.config.projection_dim != clip_dim:
                raise ValueError(f"the supplied tower projects to {self.clip_model.config.projection_dim} dimensions, clip_dim is {clip_dim}")

    def train(self, mode=True):                                    # the tower stays frozen in eval mode, as in the reference
        super().train(mode)
        if self.clip_model is not None:
            self.clip_model.eval()
        return self

    def tokenize(self, texts):
        """clip.tokenize(raw_text, context_length=22, truncate=True) + zero padding to 77 (:59-62) -> int64 (B, 77)."""
        ctx = self.MAX_TEXT_LEN + 2
        tk = self.tokenizer
        rows = []
        for t in texts:
            ids = tk(t, add_special_tokens=False)["input_ids"][:ctx - 2]
            ids = [tk.bos_token_id] + ids + [tk.eos_token_id]
            rows.append(ids + [0] * (self.CONTEXT_DEFAULT - len(ids)))
        return torch.tensor(rows, dtype=torch.int64)

    @torch.no_grad()
    def forward(self, texts, force_mask=False):
        dev = self._anchor.device
        if self.clip_model is None:
            rows = []
            for t in texts:
                seed = int.from_bytes(hashlib.sha256(t.encode()).digest()[:8], "little")
                g = torch.Generator().manual_seed(seed)
                v = torch.randn(self.clip_dim, generator=g)
                rows.append(v / v.norm())
            return torch.stack(rows).to(dev)
        ids = self.tokenize(texts).to(dev)
        # encode_text: token + positional embedding, causal transformer, ln_final, the feature AT THE END-OF-TEXT TOKEN times the text
        # projection.  (The zero padding behind it is invisible to that position under the causal mask.)
        hidden = self.clip_model.text_model(input_ids=ids, attention_mask=None).last_hidden_state
        eot = (ids == self.tokenizer.eos_token_id).int().argmax(dim=1)
        pooled = hidden[torch.arange(ids.shape[0], device=dev), eot]
        return self.clip_model.text_projection(pooled).float()
