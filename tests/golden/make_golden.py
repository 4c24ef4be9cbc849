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

Usage:  python tests/golden/make_golden.py [fixture names]
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
import torch.nn.functional as F

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


def make_d3pm(name, K, L, spatial, n_layer, cond_dim, T, B, seed, noise_seed, light=False):
    """light: only what pins a long-sequence run (one guided step's logits / sample and the 100-step loop with its per-step trace,
    stored as int16) -- the fixture of L = 2048, where the attention kernel's adaptive arithmetic is the default, stays under 1 MiB."""
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
        logits_c3 = None if light else dm.transformer(xt.clone(), cond3, t)
    if light:
        out.update({"step_xt": xt.numpy(), "step_cond": cond.numpy(), "step_t": t.numpy(), "step_logits": logits.numpy(), "step_stream": 1000,
                    "step_sample": dt_mod.log_onehot_to_index(samp).numpy().astype(np.int16), "step_margin": (top2[:, 1] - top2[:, 0])})
    else:
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
    if light:
        out["loop_tokens"], out["loop_trace"] = out["loop_tokens"].astype(np.int16), out["loop_trace"].astype(np.int16)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, "step margin min:", float(out["step_margin"].min()), "loop tokens unique:", len(np.unique(out["loop_tokens"])),
              "still masked at the end:", int((out["loop_tokens"] == K).sum()))
        return

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


# --------------------------------------------------------------------------- generator glue (DiscreteDiffusion.forward)
class TableTextEncoder(nn.Module):
    """Stand-in for CLIPTextEmbedding in the fixture: a fixed (B, cond_dim) row per caption, stored in the fixture."""

    def __init__(self, table):
        super().__init__()
        self.table = table
        self.dummy = nn.Parameter(torch.zeros(1))          # DiscreteDiffusion.__init__ reads next(self.parameters())

    def forward(self, texts):
        return torch.stack([self.table[t] for t in texts])


def make_glue(name, vq_name, d3pm_name, seed):
    """discrete_diffusion.py:16-83 run end to end (do_inference=True and False) on the two small fixture models: every key of the
    output dict.  Two variants: as written (text embeddings zeroed, :25/:49) and with those two `zeros_like` calls made the identity
    (the optional non-zero conditioning of SURVEY.md appendix D) -- the reference's code path otherwise untouched."""
    import src.models.networks.discrete_diffusion as dd_mod
    from src.models.networks.videogpt_vq_vae import VQVAE

    zv = np.load(os.path.join(OUT, vq_name + ".npz"))
    zd = np.load(os.path.join(OUT, d3pm_name + ".npz"))
    cfgv = {k[4:]: (zv[k].tolist() if zv[k].ndim else zv[k].item()) for k in zv.files if k.startswith("cfg_")}
    cfgd = {k[4:]: (zd[k].tolist() if zd[k].ndim else zd[k].item()) for k in zd.files if k.startswith("cfg_")}
    vq = VQVAE(None, cfgv["embedding_dim"], cfgv["n_codes"], cfgv["n_hiddens"], cfgv["n_res_layers"], cfgv["downsample"],
               cfgv["sequence_length"], cfgv["resolution"])
    vq.load_state_dict({k[3:]: torch.from_numpy(zv[k]) for k in zv.files if k.startswith("sd/")})
    vq.codebook._need_init = False
    vq.eval()
    K, L, T, B = cfgd["K"], cfgd["L"], cfgd["T"], cfgd["B"]
    assert K == cfgv["n_codes"] and L == int(np.prod(vq.latent_shape))
    dm = build_d3pm(K, L, cfgd["spatial"], cfgd["n_layer"], cfgd["cond_dim"], T, seed=0)
    dm.load_state_dict({k[3:]: torch.from_numpy(zd[k]) for k in zd.files if k.startswith("sd/")}, strict=False)
    dm.eval()
    g = torch.Generator().manual_seed(seed)
    texts = ["a person juggling", "waves on a beach"][:B]
    table = {t: torch.randn(cfgd["cond_dim"], generator=g) for t in texts + [""]}
    gen = dd_mod.DiscreteDiffusion.__new__(dd_mod.DiscreteDiffusion)
    nn.Module.__init__(gen)
    gen.textencoder = TableTextEncoder(table)
    gen.diffusion_model = dm
    x = torch.from_numpy(zv["x"])
    batch = {"video": x, "text": texts, "length": [x.shape[2]] * B}
    t_fix = torch.tensor([13, 0][:B], dtype=torch.long)
    dm.sample_time = lambda b, device, method="uniform": (t_fix, torch.ones(b) / T)
    out = {"cfg_vqvae": vq_name, "cfg_d3pm": d3pm_name, "cfg_noise_seed": cfgd["noise_seed"], "cfg_stream": 7000,
           "texts": np.array(texts), "text_table": torch.stack([table[t] for t in texts + [""]]).numpy(), "t": t_fix.numpy()}
    real_zeros_like = torch.zeros_like
    sampled, real_sample = [], dm.sample

    def sample_and_keep(*a, **k):                          # the sampled tokens, to make a decode mismatch diagnosable
        r = real_sample(*a, **k)
        sampled.append(r["content_token"].numpy().copy())
        return r

    dm.sample = sample_and_keep

    def keep_text(x_, *a, **k):                            # the two text-embedding zeroings become the identity
        if x_.dim() == 3 and x_.shape[1:] == (1, cfgd["cond_dim"]):
            return x_.clone()
        return real_zeros_like(x_, *a, **k)

    for tag, zl in (("zero", real_zeros_like), ("cond", keep_text)):
        torch.zeros_like = zl
        dm.Lt_history.zero_(); dm.Lt_count.zero_()
        torch.rand_like = PhiloxRand(cfgd["noise_seed"], stream0=7000)
        with torch.no_grad():
            o = gen(batch, vq, None, do_inference=True)
        torch.zeros_like = real_zeros_like
        assert set(o) == {"pred_data", "pred_single_step", "gt_data", "losses", "test"}
        out.update({f"{tag}/pred_data": o["pred_data"].numpy(), f"{tag}/pred_single_step": o["pred_single_step"].numpy(),
                    f"{tag}/losses": o["losses"].numpy(), f"{tag}/test": o["test"].numpy(), f"{tag}/content_token": sampled[-1],
                    f"{tag}/Lt_history": dm.Lt_history.numpy().copy(), f"{tag}/Lt_count": dm.Lt_count.numpy().copy()})
        torch.rand_like = PhiloxRand(cfgd["noise_seed"], stream0=7000)
        dm.Lt_history.zero_(); dm.Lt_count.zero_()
        torch.zeros_like = zl
        with torch.no_grad():
            o2 = gen(batch, vq, None)
        torch.zeros_like = real_zeros_like
        assert set(o2) == {"pred_data", "gt_data", "losses", "test"}
        assert torch.equal(o2["pred_data"], o["pred_single_step"]) and torch.equal(o2["losses"], o["losses"])
        print(name, tag, "loss", float(o["losses"]), "sampled-vs-gt mse", float(((o["pred_data"] - x) ** 2).mean()))
    assert torch.equal(o["gt_data"], x)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


# --------------------------------------------------------------------------- near-tie stress vectors + sample_time
def make_neartie(name, seed, noise_seed):
    """Inputs built so that the reference's two arg-extrema decide on margins of 1e-7 ... 1e-3 (the regular fixtures' minimum
    margins are 1e-3 / 1e-2, far above fp32 noise):
      * Codebook.forward (videogpt_vq_vae.py:174-222, eval mode): latents placed next to the bisector of deliberately close code
        pairs; recorded: the reference's indices, the fp64 margin d(z, 2nd)^2 - d(z, 1st)^2 and the fp64 winner;
      * p_sample (diffusion_transformer.py:304-359) with the denoiser replaced by a table of logits: one conditional logit per
        position is bisected until the top-2 gap of `gumbel + posterior` hits a target; recorded: the reference's sampled tokens,
        the gap it saw (fp32) and the gap / winner of an fp64 evaluation of the same formulas.
      * sample_time('importance') (:368-389) after every Lt_count passed 10: the reference's (t, pt) under torch.manual_seed."""
    import src.models.motionencoder.diffusion_transformer as dt_mod
    from src.models.networks.videogpt_vq_vae import Codebook
    from oracle import d3pm as od

    g = torch.Generator().manual_seed(seed)
    out = {}
    targets = torch.tensor([s * m for m in (1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 1e-3) for s in (1.0, -1.0)])

    # ---- codebook
    Kc, E = 32, 8
    cb = torch.randn(Kc, E, generator=g) * 3.0
    deltas = torch.tensor([1e-1, 1e-2, 1e-3, 1e-4])
    z_rows, pairs = [], []
    for i in range(64):
        j = i % (Kc // 2)
        d = torch.randn(E, generator=g)
        d = d / d.norm()
        if i < Kc // 2:
            cb[2 * j + 1] = cb[2 * j] + deltas[j % 4] * d                 # a close pair
        dirv = (cb[2 * j + 1] - cb[2 * j])
        dist = dirv.norm()
        mid = 0.5 * (cb[2 * j] + cb[2 * j + 1])
        # margin = |z - b|^2 - |z - a|^2 = 2 eps dist for z = mid - eps * dir/|dir| (+ an offset orthogonal to dir)
        eps = targets[i % len(targets)] / (2 * dist)
        orth = torch.randn(E, generator=g) * 0.3
        orth = orth - (orth @ dirv) / (dist * dist) * dirv
        z_rows.append(mid - eps * dirv / dist + orth)
        pairs.append((2 * j, 2 * j + 1))
    z = torch.stack(z_rows).view(1, 4, 4, 4, E).permute(0, 4, 1, 2, 3).contiguous()        # (1,E,4,4,4)
    book = Codebook(Kc, E)
    book.embeddings.data.copy_(cb)
    book._need_init = False
    book.eval()
    with torch.no_grad():
        ref_idx = book(z)["encodings"].view(-1)
    flat64 = z.permute(0, 2, 3, 4, 1).reshape(-1, E).double()
    d64 = ((flat64[:, None, :] - cb.double()[None]) ** 2).sum(-1)
    top2 = torch.topk(d64, 2, dim=1, largest=False)
    out.update({"cb/z": z.numpy(), "cb/codebook": cb.numpy(), "cb/ref_idx": ref_idx.numpy(),
                "cb/margin64": (top2.values[:, 1] - top2.values[:, 0]).numpy(), "cb/winner64": top2.indices[:, 0].numpy(),
                "cb/second64": top2.indices[:, 1].numpy()})

    # ---- Gumbel arg-max through p_sample with a logits table in place of the denoiser
    K, L, T, B = 32, 16, 100, 4
    dm = build_d3pm(K, L, [4, 4], 1, 32, T, seed=seed + 1)
    lc = torch.randn(B, K, L, generator=g) * 1.5
    lu = torch.randn(B, K, L, generator=g) * 1.5
    cond = torch.ones(B, 1, 32)
    cf_cond = torch.zeros(B, 1, 32)

    class Table(nn.Module):
        def forward(self, x_t, cond_emb, t):
            return (lc if float(cond_emb.abs().sum()) > 0 else lu).clone()
    dm.transformer = Table()
    xt = torch.randint(0, K, (B, L), generator=g)
    xt[torch.rand(B, L, generator=g) < 0.5] = K
    t = torch.tensor([80, 40, 10, 0])
    log_xt = dt_mod.index_to_log_onehot(xt, K + 1)
    stream = 9000
    u = torch.from_numpy(philox.uniform_bkl(noise_seed, stream, B, K + 1, L))
    gum = -torch.log(-torch.log(u + 1e-30) + 1e-30)                       # the expression of log_sample_categorical, same ops

    def scores():
        with torch.no_grad():
            post, _ = dm.p_pred(log_xt, cond, cf_cond, t)
        return gum + post

    sc = scores()
    top1 = sc.argmax(1)
    sc_codes = sc[:, :K].clone()
    sc_codes.scatter_(1, top1.clamp(max=K - 1).unsqueeze(1), float("-inf"))           # best code other than the winner
    k2 = torch.where(top1 == K, sc[:, :K].argmax(1), sc_codes.argmax(1))
    tgt = targets[torch.arange(B * L) % len(targets)].view(B, L)
    base = lc.gather(1, k2.unsqueeze(1)).squeeze(1).clone()
    lo, hi = torch.full((B, L), -5.0), torch.full((B, L), 60.0)

    def gap_at(shift):
        lc.scatter_(1, k2.unsqueeze(1), (base + shift).unsqueeze(1))
        sc = scores()
        mine = sc.gather(1, k2.unsqueeze(1)).squeeze(1)
        rest = sc.clone()
        rest.scatter_(1, k2.unsqueeze(1), float("-inf"))
        return mine - rest.max(1).values
    for _ in range(60):                                                   # gap is increasing in the shift
        mid = 0.5 * (lo + hi)
        too_big = gap_at(mid) > tgt
        hi = torch.where(too_big, mid, hi)
        lo = torch.where(too_big, lo, mid)
    gap_at(hi)
    torch.rand_like = PhiloxRand(noise_seed, stream0=stream)
    with torch.no_grad():
        samp, _ = dm.p_sample(log_xt, cond, cf_cond, t, [0] * B, 0)
    ref_tok = dt_mod.log_onehot_to_index(samp)
    sc = scores()
    t2 = torch.topk(sc, 2, dim=1)
    assert torch.equal(sc.gather(1, ref_tok.unsqueeze(1)).squeeze(1), t2.values[:, 0])      # (exact fp32 ties: argmax = first index)
    # fp64 evaluation of the same formulas on the same fp32 inputs
    sd64 = {k: v.double() for k, v in dm.state_dict().items() if k.startswith("log_")}

    def ps64(o):
        lp = F.log_softmax(o.double(), dim=1).clamp(-70, 0)
        return lp
    mix = ps64(lu) + 2.0 * (ps64(lc) - ps64(lu))
    mix = (mix - torch.logsumexp(mix, dim=1, keepdim=True)).clamp(-70, 0)
    rec64 = torch.cat((mix, torch.full((B, 1, L), -70.0, dtype=torch.float64)), dim=1)
    logxt64 = torch.log(F.one_hot(xt, K + 1).permute(0, 2, 1).double().clamp(min=1e-30))
    post64 = q_posterior64(rec64, logxt64, t, sd64)
    g64 = -torch.log(-torch.log(u.double() + 1e-30) + 1e-30)
    s64 = g64 + post64
    t64 = torch.topk(s64, 2, dim=1)
    out.update({"gum/logits_c": lc.numpy(), "gum/logits_u": lu.numpy(), "gum/xt": xt.numpy(), "gum/t": t.numpy(),
                "gum/stream": stream, "gum/ref_tok": ref_tok.numpy(), "gum/margin_ref": (t2.values[:, 0] - t2.values[:, 1]).numpy(),
                "gum/margin64": (t64.values[:, 0] - t64.values[:, 1]).numpy(), "gum/winner64": t64.indices[:, 0].numpy(),
                "gum/second64": t64.indices[:, 1].numpy(), "cfg_K": K, "cfg_L": L, "cfg_T": T, "cfg_B": B,
                "cfg_noise_seed": noise_seed, "cfg_guidance": 2.0})
    out.update(sd_to_np({k: v for k, v in dm.state_dict().items() if k.startswith("log_")}, "sd/"))

    # ---- sample_time('importance') once every timestep has been visited more than 10 times
    dm.Lt_count.fill_(11.0)
    dm.Lt_history.copy_(torch.rand(T, generator=g) * 50.0 + 0.01)
    torch.manual_seed(seed + 7)
    t_imp, pt_imp = dm.sample_time(64, "cpu", "importance")
    out.update({"st/Lt_history": dm.Lt_history.numpy().copy(), "st/seed": seed + 7, "st/t": t_imp.numpy(), "st/pt": pt_imp.numpy()})
    dm.Lt_count[3] = 10.0                                                 # one timestep short: falls back to uniform
    torch.manual_seed(seed + 8)
    t_uni, pt_uni = dm.sample_time(64, "cpu", "importance")
    out.update({"st/seed_uniform": seed + 8, "st/t_uniform": t_uni.numpy(), "st/pt_uniform": pt_uni.numpy()})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "codebook |margin64| min/max:", float(np.abs(out["cb/margin64"]).min()), float(np.abs(out["cb/margin64"]).max()),
          "ref==fp64:", int((out["cb/ref_idx"] == out["cb/winner64"]).sum()), "/ 64;",
          "gumbel |margin64| min/max:", float(np.abs(out["gum/margin64"]).min()), float(np.abs(out["gum/margin64"]).max()),
          "ref==fp64:", int((out["gum/ref_tok"] == out["gum/winner64"]).sum()), "/", B * L)


def q_posterior64(log_x_start, log_x_t, t, sd):
    """diffusion_transformer.py:251-283 in float64 (same formulas as oracle/d3pm.py::q_posterior, no fp32 constants)."""
    B, K1, L = log_x_start.shape
    lz = float(np.log(1e-30))
    ext = lambda a, tt: a.gather(-1, tt).reshape(-1, 1, 1)
    lae = lambda a, b: torch.max(a, b) + torch.log(torch.exp(a - torch.max(a, b)) + torch.exp(b - torch.max(a, b)))
    T = sd["log_at"].shape[0]

    def q_pred(lx, tt):
        tt = (tt + (T + 1)) % (T + 1)
        return torch.cat([lae(lx[:, :-1] + ext(sd["log_cumprod_at"], tt), ext(sd["log_cumprod_bt"], tt)),
                          lae(lx[:, -1:] + ext(sd["log_1_min_cumprod_ct"], tt), ext(sd["log_cumprod_ct"], tt))], dim=1)

    def q_one(lx, tt):
        return torch.cat([lae(lx[:, :-1] + ext(sd["log_at"], tt), ext(sd["log_bt"], tt)),
                          lae(lx[:, -1:] + ext(sd["log_1_min_ct"], tt), ext(sd["log_ct"], tt))], dim=1)
    mask = (log_x_t.argmax(1) == K1 - 1).unsqueeze(1)
    log_zero = torch.full((B, 1, L), lz, dtype=torch.float64)
    log_qt = q_pred(log_x_t, t)[:, :-1]
    log_qt = torch.where(mask, ext(sd["log_cumprod_ct"], t).expand(-1, K1 - 1, L), log_qt)
    log_q1 = torch.cat((q_one(log_x_t, t)[:, :-1], log_zero), dim=1)
    ct_vec = torch.cat((ext(sd["log_ct"], t).expand(-1, K1 - 1, L), torch.zeros(B, 1, L, dtype=torch.float64)), dim=1)
    log_q1 = torch.where(mask, ct_vec, log_q1)
    q = torch.cat((log_x_start[:, :-1] - log_qt, log_zero), dim=1)
    s_ = torch.logsumexp(q, dim=1, keepdim=True)
    q = q - s_
    return torch.clamp(q_pred(q, t - 1) + log_q1 + s_, -70, 0)


# --------------------------------------------------------------------------- VQ-VAE first training step: codebook data-init
def make_vqvae_init(name, cfg, B, seed):
    """The very first train-mode forward of a fresh VQVAE: Codebook._init_embeddings (videogpt_vq_vae.py:160-172) and, with fewer
    latents than codes, _tile's repeat + jitter (:151-158) -- in the init AND in the restart draw of the same step.  Captured:
    both permutations (torch.randperm) and both noise tensors (torch.randn_like)."""
    from src.models.networks.videogpt_vq_vae import VQVAE

    torch.manual_seed(seed)
    m = VQVAE(None, cfg["embedding_dim"], cfg["n_codes"], cfg["n_hiddens"], cfg["n_res_layers"],
              cfg["downsample"], cfg["sequence_length"], cfg["resolution"])
    g = torch.Generator().manual_seed(seed + 1)
    shape = (B, 3, cfg["sequence_length"], cfg["resolution"], cfg["resolution"])
    perms, noises = [], []
    real_randperm, real_randn_like = torch.randperm, torch.randn_like

    def cap_randperm(n, *a, **k):
        p = real_randperm(n, generator=g)
        perms.append(p.numpy().copy())
        return p

    def cap_randn_like(x, *a, **k):
        r = torch.randn(x.shape, generator=g, dtype=x.dtype)
        noises.append(r.numpy().copy())
        return r

    torch.randperm, torch.randn_like = cap_randperm, cap_randn_like
    m.train()
    assert m.codebook._need_init
    before = {k: v.clone() for k, v in m.state_dict().items()}
    x = torch.randn(shape, generator=g)
    with torch.no_grad():
        out = m({"video": x})
    torch.randperm, torch.randn_like = real_randperm, real_randn_like
    n_lat = B * int(np.prod(m.latent_shape))
    assert len(perms) == 2 and len(noises) == (2 if n_lat < cfg["n_codes"] else 0), (len(perms), len(noises), n_lat)
    res = {}
    res.update(sd_to_np(before, "sd/"))
    res.update(sd_to_np(m.state_dict(), "after/"))
    res.update({"x": x.numpy(), "perm_init": perms[0], "perm": perms[1], "pred": out["pred_data"].numpy(),
                "recon_loss": out["losses"]["recon_loss"].numpy(), "commitment_loss": out["losses"]["commitment_loss"].numpy(),
                "cfg_embedding_dim": cfg["embedding_dim"], "cfg_n_codes": cfg["n_codes"],
                "cfg_n_hiddens": cfg["n_hiddens"], "cfg_n_res_layers": cfg["n_res_layers"],
                "cfg_downsample": np.array(cfg["downsample"]), "cfg_sequence_length": cfg["sequence_length"],
                "cfg_resolution": cfg["resolution"]})
    if noises:
        res.update({"noise_init": noises[0], "noise": noises[1]})
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **res)
    print(name, "latents:", n_lat, "codes:", cfg["n_codes"], "tiled:", bool(noises), "codes with N >= 1:", int((m.codebook.N >= 1).sum()))


def main(argv):
    """No arguments: every fixture.  Otherwise the named ones (vqvae_ds188 vqvae_ds244 d3pm_L64 vqvae_train_ds188 glue_L64 ...)."""
    install_stubs()
    real_rand_like = torch.rand_like
    os.makedirs(OUT, exist_ok=True)
    jobs = {
        "vqvae_ds188": lambda: make_vqvae("vqvae_ds188", dict(embedding_dim=8, n_codes=32, n_hiddens=16, n_res_layers=2,
                                                              downsample=[1, 8, 8], sequence_length=4, resolution=32), B=2, seed=11),
        "vqvae_ds244": lambda: make_vqvae("vqvae_ds244", dict(embedding_dim=8, n_codes=24, n_hiddens=16, n_res_layers=1,
                                                              downsample=[2, 4, 4], sequence_length=4, resolution=16), B=1, seed=12),
        "d3pm_L64": lambda: make_d3pm("d3pm_L64", K=32, L=64, spatial=[8, 8], n_layer=2, cond_dim=32, T=100, B=2, seed=21,
                                      noise_seed=1234),
        # a long sequence (the attention kernel's adaptive P arithmetic is the default from L = 2048 on): the reference's own 100-step
        # loop at L = 2048, one clip, small model -- ~3 minutes of CPU
        "d3pm_L2048": lambda: make_d3pm("d3pm_L2048", K=32, L=2048, spatial=[64, 32], n_layer=2, cond_dim=32, T=100, B=1, seed=22,
                                        noise_seed=2345, light=True),
        "vqvae_train_ds188": lambda: make_vqvae_train("vqvae_train_ds188", dict(embedding_dim=8, n_codes=32, n_hiddens=16,
                                                                                n_res_layers=1, downsample=[1, 8, 8],
                                                                                sequence_length=4, resolution=32), B=2, seed=31),
        "glue_L64": lambda: make_glue("glue_L64", "vqvae_ds188", "d3pm_L64", seed=41),
        "neartie": lambda: make_neartie("neartie", seed=51, noise_seed=4321),
        # 16 latents < 24 codes: _tile repeats and jitters; 128 latents >= 32 codes: plain permutation draw
        "vqvae_init_tiled": lambda: make_vqvae_init("vqvae_init_tiled", dict(embedding_dim=8, n_codes=24, n_hiddens=16, n_res_layers=1,
                                                                            downsample=[1, 8, 8], sequence_length=4, resolution=16),
                                                    B=1, seed=61),
        "vqvae_init": lambda: make_vqvae_init("vqvae_init", dict(embedding_dim=8, n_codes=32, n_hiddens=16, n_res_layers=1,
                                                                 downsample=[1, 8, 8], sequence_length=4, resolution=32), B=2, seed=62),
    }
    for name in (argv or list(jobs)):
        torch.rand_like = real_rand_like
        jobs[name]()
    torch.rand_like = real_rand_like


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
    main(sys.argv[1:])
