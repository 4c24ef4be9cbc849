#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by importing the REFERENCE modules.

Runs ONLY in the build container (needs /root/reference).  Nothing here travels to the GPU box
except the resulting .npz data files.  The reference is imported read-only with three
in-harness stubs:
  * ``pytorch_lightning.LightningModule`` -> nn.Module + no-op ``save_hyperparameters`` + ``device``
  * ``hydra.utils.instantiate``           -> import-only use in the reference modules
  * ``torch.Tensor.cuda``                 -> identity (reference hard-codes ``t.cuda()``,
                                             src/models/motionencoder/transformer_utils.py:439)
Noise is injected by patching ``torch.rand_like`` with the Philox stream of oracle/philox.py, so
the same uniforms can be regenerated on the device (no noise tensors are stored).

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
import torch.nn as nn

from oracle import philox

OUT = os.path.join(REPO, "tests", "golden")


# --------------------------------------------------------------------------- stubs
def install_stubs():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl
    hydra = types.ModuleType("hydra")
    hutils = types.ModuleType("hydra.utils")
    hutils.instantiate = lambda *a, **k: None
    hydra.utils = hutils
    sys.modules["hydra"] = hydra
    sys.modules["hydra.utils"] = hutils
    torch.Tensor.cuda = lambda self, *a, **k: self


class PhiloxRand:
    """Replacement for torch.rand_like: call i draws stream ``stream0 + i``."""

    def __init__(self, seed, stream0=0):
        self.seed, self.stream = seed, stream0

    def __call__(self, x, **kw):
        B, K1, L = x.shape
        u = philox.uniform_bkl(self.seed, self.stream, B, K1, L)
        self.stream += 1
        return torch.from_numpy(u).to(x.dtype)


def sd_to_np(sd, prefix):
    return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


# --------------------------------------------------------------------------- VQ-VAE
def make_vqvae(name, cfg, B, seed):
    from src.models.networks.videogpt_vq_vae import VQVAE

    torch.manual_seed(seed)
    m = VQVAE(None, cfg["embedding_dim"], cfg["n_codes"], cfg["n_hiddens"], cfg["n_res_layers"],
              cfg["downsample"], cfg["sequence_length"], cfg["resolution"])
    # non-trivial BN statistics / affine so eval-mode BN is exercised
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm3d):
            mod.weight.data = 1.0 + 0.2 * torch.randn(mod.weight.shape, generator=g)
            mod.bias.data = 0.1 * torch.randn(mod.bias.shape, generator=g)
    x_init = torch.randn(B, 3, cfg["sequence_length"], cfg["resolution"], cfg["resolution"], generator=g)
    m.train()
    with torch.no_grad():
        for _ in range(40):         # converge the BN running stats (momentum 0.1)
            m({"video": x_init})
        m.codebook._need_init = True    # re-seed the codebook from latents at the converged statistics
        m({"video": x_init})            # (reference: videogpt_vq_vae.py:160-172,176-177)
    m.eval()
    x = torch.randn(B, 3, cfg["sequence_length"], cfg["resolution"], cfg["resolution"], generator=g)
    out = {}
    with torch.no_grad():
        h_enc = m.encoder(x)
        z = m.pre_vq_conv(h_enc)
        enc, emb = m.encode(x, include_embeddings=True)
        from src.models.utils.model_utils import shift_dim
        flat = shift_dim(z, 1, -1).flatten(end_dim=-2)
        E = m.codebook.embeddings
        d = (flat ** 2).sum(1, keepdim=True) - 2 * flat @ E.t() + (E.t() ** 2).sum(0, keepdim=True)
        top2 = torch.topk(d, 2, dim=1, largest=False).values
        rec = m.decode(enc)
        fwd = m({"video": x})
        # intermediate activations for kernel-level tests
        h = x
        acts = []
        for conv in m.encoder.convs:
            h = torch.relu(conv(h))
            acts.append(h)
        h_last = m.encoder.conv_last(h)
        blk0 = m.encoder.res_stack[0]
        ax_in = blk0.block[:8](h_last)
        ax_out = blk0.block[8](ax_in)
    out.update(sd_to_np(m.state_dict(), "sd/"))
    out.update({
        "x": x.numpy(), "enc_conv0": acts[0].numpy(), "enc_conv_last": h_last.numpy(),
        "axial_in": ax_in.numpy(), "axial_out": ax_out.numpy(),
        "h_enc": h_enc.numpy(), "z": z.numpy(), "encodings": enc.numpy(), "embeddings": emb.numpy(),
        "argmin_margin": (top2[:, 1] - top2[:, 0]).numpy(), "decoded": rec.numpy(),
        "fwd_pred": fwd["pred_data"].numpy(), "fwd_recon_loss": fwd["losses"]["recon_loss"].numpy(),
        "fwd_commitment_loss": fwd["losses"]["commitment_loss"].numpy(),
        "cfg_embedding_dim": cfg["embedding_dim"], "cfg_n_codes": cfg["n_codes"],
        "cfg_n_hiddens": cfg["n_hiddens"], "cfg_n_res_layers": cfg["n_res_layers"],
        "cfg_downsample": np.array(cfg["downsample"]), "cfg_sequence_length": cfg["sequence_length"],
        "cfg_resolution": cfg["resolution"],
    })
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "codes used:", len(np.unique(enc.numpy())), "min margin:", float(out["argmin_margin"].min()))


# --------------------------------------------------------------------------- D3PM
def build_d3pm(K, L, spatial, n_layer, cond_dim, T, seed, guidance=2.0):
    from src.models.motionencoder.dalle_mask_image_embedding import DalleMaskImageEmbedding
    from src.models.motionencoder.transformer_utils import Text2ImageTransformer
    from src.models.motionencoder.diffusion_transformer import DiffusionTransformer

    torch.manual_seed(seed)
    dalle = DalleMaskImageEmbedding(num_embed=K, spatial_size=spatial, embed_dim=64, trainable=True,
                                    pos_emb_type="embedding")
    tr = Text2ImageTransformer(dalle=dalle, condition_seq_len=77, n_layer=n_layer, n_embd=64, n_head=16,
                               content_seq_len=L, attn_pdrop=0.0, resid_pdrop=0.0, mlp_hidden_times=4,
                               block_activate="GELU2", attn_type="selfcross", content_spatial_size=spatial,
                               condition_dim=cond_dim, diffusion_step=T, timestep_type="adalayernorm")
    # the reference init (N(0,0.02), zero bias) makes attention near-uniform; rescale so the
    # fixture exercises softmax / LayerNorm affine / biases for real
    g = torch.Generator().manual_seed(seed + 1)
    for mod in tr.modules():
        if isinstance(mod, nn.Linear):
            mod.weight.data = torch.randn(mod.weight.shape, generator=g) * (1.2 / mod.in_features ** 0.5)
            if mod.bias is not None:
                mod.bias.data = 0.1 * torch.randn(mod.bias.shape, generator=g)
        elif isinstance(mod, nn.Embedding):
            mod.weight.data = torch.randn(mod.weight.shape, generator=g) * 0.5
        elif isinstance(mod, nn.LayerNorm) and mod.elementwise_affine:
            mod.weight.data = 1.0 + 0.2 * torch.randn(mod.weight.shape, generator=g)
            mod.bias.data = 0.1 * torch.randn(mod.bias.shape, generator=g)
    dm = DiffusionTransformer(transformer=tr, diffusion_step=T, alpha_init_type="alpha1",
                              auxiliary_loss_weight=5.0e-4, adaptive_auxiliary_loss=True, mask_weight=[1, 1],
                              learnable_cf=False, guidance_scale=guidance, content_seq_len=L)
    dm.eval()
    return dm


def make_d3pm(name, K, L, spatial, n_layer, cond_dim, T, B, seed, noise_seed):
    import src.models.motionencoder.diffusion_transformer as dt_mod

    dm = build_d3pm(K, L, spatial, n_layer, cond_dim, T, seed)
    g = torch.Generator().manual_seed(seed + 2)
    out = {}
    sd = {k: v for k, v in dm.state_dict().items() if not k.endswith("attn2.mask") and k != "empty_text_embed"}
    out.update(sd_to_np(sd, "sd/"))
    out.update({"cfg_K": K, "cfg_L": L, "cfg_spatial": np.array(spatial), "cfg_n_layer": n_layer,
                "cfg_cond_dim": cond_dim, "cfg_T": T, "cfg_B": B, "cfg_guidance": 2.0,
                "cfg_noise_seed": noise_seed})

    # ---- one teacher-forced step: x_t with a mix of codes and [MASK]
    xt = torch.randint(0, K, (B, L), generator=g)
    xt[torch.rand(B, L, generator=g) < 0.4] = K
    cond = torch.randn(B, 1, cond_dim, generator=g)
    cf_cond = torch.zeros(B, 1, cond_dim)
    t = torch.tensor([37, 5][:B] if B <= 2 else list(range(3, 3 + 7 * B, 7)), dtype=torch.long) % T
    with torch.no_grad():
        logits = dm.transformer(xt.clone(), cond, t)                       # (B,K,L)
        logits_u = dm.transformer(xt.clone(), cf_cond, t)
        log_xt = dt_mod.index_to_log_onehot(xt, K + 1)
        ps = dm.predict_start(log_xt, cond, t)
        cfps = dm.cf_predict_start(log_xt, cond, cf_cond, t)
        post = dm.q_posterior(cfps, log_xt, t)
        torch.rand_like = PhiloxRand(noise_seed, stream0=1000)
        samp = dm.log_sample_categorical(post)
        u = philox.uniform_bkl(noise_seed, 1000, B, K + 1, L)
        gum = -np.log(-np.log(u + np.float32(1e-30)) + np.float32(1e-30))
        top2 = np.sort((torch.from_numpy(gum) + post).numpy(), axis=1)[:, -2:, :]
        # general cross-attention: 3 condition tokens
        cond3 = torch.randn(B, 3, cond_dim, generator=g)
        logits_c3 = dm.transformer(xt.clone(), cond3, t)
    out.update({"step_xt": xt.numpy(), "step_cond": cond.numpy(), "step_t": t.numpy(),
                "step_logits": logits.numpy(), "step_logits_uncond": logits_u.numpy(),
                "step_predict_start": ps.numpy(), "step_cf_predict_start": cfps.numpy(),
                "step_posterior": post.numpy(), "step_stream": 1000,
                "step_sample": dt_mod.log_onehot_to_index(samp).numpy(),
                "step_margin": (top2[:, 1] - top2[:, 0]),
                "cond3": cond3.numpy(), "logits_cond3": logits_c3.numpy()})

    # ---- full reverse loop, recording tokens after every step (stream = call index 0..T-1)
    trace = []
    orig_p_sample = dm.p_sample

    def traced(*a, **k):
        r = orig_p_sample(*a, **k)
        trace.append(dt_mod.log_onehot_to_index(r[0]).numpy().copy())
        return r

    dm.p_sample = traced
    torch.rand_like = PhiloxRand(noise_seed, stream0=0)
    with torch.no_grad():
        res = dm.sample(["a"] * B, None, cond, cf_cond, content_token=None, filter_ratio=0)
    dm.p_sample = orig_p_sample
    out.update({"loop_tokens": res["content_token"].numpy(), "loop_trace": np.stack(trace)})

    # ---- training loss with fixed t and injected q_sample noise (stream 5000)
    x0 = torch.randint(0, K, (B, L), generator=g)
    t_tr = torch.tensor([0, 61][:B] if B <= 2 else list(range(B)), dtype=torch.long) % T
    dm.sample_time = lambda b, device, method="uniform": (t_tr, torch.ones(b) / T)
    torch.rand_like = PhiloxRand(noise_seed, stream0=5000)
    lt_h0, lt_c0 = dm.Lt_history.clone(), dm.Lt_count.clone()
    with torch.no_grad():
        o = dm({"condition_embed_token": cond, "content_token": x0}, return_loss=True)
    out.update({"train_x0": x0.numpy(), "train_t": t_tr.numpy(), "train_stream": 5000,
                "train_loss": o["loss"].numpy(), "train_logits": o["logits"].numpy(),
                "train_pred": o["pred_data"].numpy(), "train_Lt_history": dm.Lt_history.numpy(),
                "train_Lt_count": dm.Lt_count.numpy()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "step margin min:", float(out["step_margin"].min()),
          "loop tokens unique:", len(np.unique(out["loop_tokens"])), "loss:", float(o["loss"]))


def main():
    install_stubs()
    real_rand_like = torch.rand_like
    os.makedirs(OUT, exist_ok=True)
    make_vqvae("vqvae_ds188", dict(embedding_dim=8, n_codes=32, n_hiddens=16, n_res_layers=2,
                                   downsample=[1, 8, 8], sequence_length=4, resolution=32), B=2, seed=11)
    make_vqvae("vqvae_ds244", dict(embedding_dim=8, n_codes=24, n_hiddens=16, n_res_layers=1,
                                   downsample=[2, 4, 4], sequence_length=4, resolution=16), B=1, seed=12)
    make_d3pm("d3pm_L64", K=32, L=64, spatial=[8, 8], n_layer=2, cond_dim=32, T=100, B=2, seed=21,
              noise_seed=1234)
    torch.rand_like = real_rand_like
    make_vqvae_train("vqvae_train_ds188", dict(embedding_dim=8, n_codes=32, n_hiddens=16, n_res_layers=1,
                                               downsample=[1, 8, 8], sequence_length=4, resolution=32), B=2, seed=31)





# --------------------------------------------------------------------------- VQ-VAE train-mode forward
def make_vqvae_train(name, cfg, B, seed):
    """One train-mode forward (BatchNorm batch statistics + running-stat update, codebook EMA + restart) with the
    permutations drawn by the codebook captured, from a state saved *before* the step."""
    from src.models.networks.videogpt_vq_vae import VQVAE

    torch.manual_seed(seed)
    m = VQVAE(None, cfg["embedding_dim"], cfg["n_codes"], cfg["n_hiddens"], cfg["n_res_layers"],
              cfg["downsample"], cfg["sequence_length"], cfg["resolution"])
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():
        if isinstance(mod, nn.BatchNorm3d):
            mod.weight.data = 1.0 + 0.2 * torch.randn(mod.weight.shape, generator=g)
            mod.bias.data = 0.1 * torch.randn(mod.bias.shape, generator=g)
    shape = (B, 3, cfg["sequence_length"], cfg["resolution"], cfg["resolution"])
    perms = []
    real_randperm = torch.randperm

    def cap_randperm(n, *a, **k):
        p = real_randperm(n, generator=g)
        perms.append(p.numpy().copy())
        return p

    torch.randperm = cap_randperm
    m.train()
    with torch.no_grad():
        m({"video": torch.randn(shape, generator=g)})            # step 0: codebook init (consumes 2 permutations)
        perms.clear()
        before = {k: v.clone() for k, v in m.state_dict().items()}
        x = torch.randn(shape, generator=g)
        out = m({"video": x})                                     # the recorded step (1 permutation: restart draw)
    torch.randperm = real_randperm
    res = {}
    res.update(sd_to_np(before, "sd/"))
    res.update(sd_to_np(m.state_dict(), "after/"))
    res.update({"x": x.numpy(), "perm": np.stack(perms), "pred": out["pred_data"].numpy(),
                "recon_loss": out["losses"]["recon_loss"].numpy(),
                "commitment_loss": out["losses"]["commitment_loss"].numpy(),
                "cfg_embedding_dim": cfg["embedding_dim"], "cfg_n_codes": cfg["n_codes"],
                "cfg_n_hiddens": cfg["n_hiddens"], "cfg_n_res_layers": cfg["n_res_layers"],
                "cfg_downsample": np.array(cfg["downsample"]), "cfg_sequence_length": cfg["sequence_length"],
                "cfg_resolution": cfg["resolution"]})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **res)
    print(name, "perms:", len(perms), "codes >=1 usage:", int((m.codebook.N >= 1).sum()))


if __name__ == "__main__":
    main()
