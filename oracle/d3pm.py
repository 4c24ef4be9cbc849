"""CPU restatement of the reference D3PM path: denoiser transformer + discrete-diffusion math
(TEST INFRASTRUCTURE — never imported by the product path).

Pinned against the reference itself through tests/golden/d3pm_*.npz (made by importing
/root/reference, tests/golden/make_golden.py); checked in tests/test_oracle_golden.py.

``sd`` = dict name -> torch.Tensor with the reference DiffusionTransformer state_dict names
(``transformer.blocks.{i}...``, ``log_at`` ...).  Noise is injected from
oracle/philox.py instead of torch's RNG (reference: diffusion_transformer.py:354-359).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import philox

LOG_ZERO = math.log(1e-30)      # log(clamp(onehot,1e-30)), diffusion_transformer.py:50


# ----------------------------------------------------------------------------- schedule
def alpha_schedule(T, N, att_1=0.99999, att_T=0.000009, ctt_1=0.000009, ctt_T=0.99999):
    """Reference: alpha_schedule, diffusion_transformer.py:56-69 (numpy fp64)."""
    att = np.arange(0, T) / (T - 1) * (att_T - att_1) + att_1
    att = np.concatenate(([1], att))
    at = att[1:] / att[:-1]
    ctt = np.arange(0, T) / (T - 1) * (ctt_T - ctt_1) + ctt_1
    ctt = np.concatenate(([0], ctt))
    one_minus_ctt = 1 - ctt
    ct = 1 - one_minus_ctt[1:] / one_minus_ctt[:-1]
    bt = (1 - at - ct) / N
    att = np.concatenate((att[1:], [1]))
    ctt = np.concatenate((ctt[1:], [0]))
    btt = (1 - att - ctt) / N
    return at, bt, ct, att, btt, ctt


def schedule_buffers(T, K):
    """The 8 fp32 log-buffers.  Reference: diffusion_transformer.py:120-149 (fp64 logs -> float())."""
    at, bt, ct, att, btt, ctt = (torch.tensor(a.astype("float64")) for a in alpha_schedule(T, N=K))
    l1m = lambda a: torch.log(1 - a.exp() + 1e-40)
    log_ct, log_cct = torch.log(ct), torch.log(ctt)
    return {"log_at": torch.log(at).float(), "log_bt": torch.log(bt).float(), "log_ct": log_ct.float(),
            "log_cumprod_at": torch.log(att).float(), "log_cumprod_bt": torch.log(btt).float(),
            "log_cumprod_ct": log_cct.float(), "log_1_min_ct": l1m(log_ct).float(),
            "log_1_min_cumprod_ct": l1m(log_cct).float()}


# ----------------------------------------------------------------------------- denoiser
def content_emb(tok, sd, p="transformer.content_emb."):
    """Reference: DalleMaskImageEmbedding.forward, dalle_mask_image_embedding.py:59-79."""
    tok = tok.clamp(min=0)
    emb = F.embedding(tok, sd[p + "emb.weight"])
    Hs, Ws = sd[p + "height_emb.weight"].shape[0], sd[p + "width_emb.weight"].shape[0]
    pos = (sd[p + "height_emb.weight"].unsqueeze(1) + sd[p + "width_emb.weight"].unsqueeze(0)).view(Hs * Ws, -1)
    return emb + pos[: emb.shape[1]].unsqueeze(0)


def ada_layer_norm(x, t, sd, p):
    """Reference: AdaLayerNorm.forward, transformer_utils.py:150-159 (t < diffusion_step branch)."""
    e = F.linear(F.silu(F.embedding(t, sd[p + "emb.weight"])), sd[p + "linear.weight"], sd[p + "linear.bias"])
    scale, shift = torch.chunk(e.unsqueeze(1), 2, dim=2)
    return F.layer_norm(x, (x.shape[-1],)) * (1 + scale) + shift


def mha(xq, xkv, sd, p, n_head):
    """Reference: FullAttention / CrossAttention forward, transformer_utils.py:46-62 / :95-113."""
    B, T, C = xq.shape
    Te = xkv.shape[1]
    hs = C // n_head
    k = F.linear(xkv, sd[p + "key.weight"], sd[p + "key.bias"]).view(B, Te, n_head, hs).transpose(1, 2)
    q = F.linear(xq, sd[p + "query.weight"], sd[p + "query.bias"]).view(B, T, n_head, hs).transpose(1, 2)
    v = F.linear(xkv, sd[p + "value.weight"], sd[p + "value.bias"]).view(B, Te, n_head, hs).transpose(1, 2)
    att = F.softmax((q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hs)), dim=-1)
    y = (att @ v).transpose(1, 2).contiguous().view(B, T, C)
    return F.linear(y, sd[p + "proj.weight"], sd[p + "proj.bias"])


def block(x, cond, t, sd, p, n_head):
    """Reference: Block.forward 'selfcross' branch, transformer_utils.py:266-282; GELU2 :115-119."""
    h = ada_layer_norm(x, t, sd, p + "ln1.")
    x = x + mha(h, h, sd, p + "attn1.", n_head)
    x = x + mha(ada_layer_norm(x, t, sd, p + "ln1_1."), cond, sd, p + "attn2.", n_head)
    h = F.layer_norm(x, (x.shape[-1],), sd[p + "ln2.weight"], sd[p + "ln2.bias"])
    h = F.linear(h, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"])
    h = h * torch.sigmoid(1.702 * h)
    return x + F.linear(h, sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])


def n_layers(sd):
    return 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("transformer.blocks."))


def denoiser(tok, cond, t, sd, n_head=16):
    """Reference: Text2ImageTransformer.forward, transformer_utils.py:429-444 -> logits (B,K,L)."""
    x = content_emb(tok, sd)
    for i in range(n_layers(sd)):
        x = block(x, cond, t, sd, f"transformer.blocks.{i}.", n_head)
    x = F.layer_norm(x, (x.shape[-1],), sd["transformer.to_logits.0.weight"], sd["transformer.to_logits.0.bias"])
    return F.linear(x, sd["transformer.to_logits.1.weight"], sd["transformer.to_logits.1.bias"]).transpose(1, 2)


# ----------------------------------------------------------------------------- log-space helpers
def index_to_log_onehot(x, num_classes):
    """Reference: diffusion_transformer.py:44-51 -> (B, num_classes, L) in {0, log 1e-30}."""
    oh = F.one_hot(x, num_classes).permute(0, 2, 1)
    return torch.log(oh.float().clamp(min=1e-30))


def log_add_exp(a, b):
    """Reference: diffusion_transformer.py:32-34."""
    m = torch.max(a, b)
    return m + torch.log(torch.exp(a - m) + torch.exp(b - m))


def _ext(a, t):
    return a.gather(-1, t).reshape(-1, 1, 1)


def q_pred_one_timestep(log_x_t, t, sd):
    """Reference: diffusion_transformer.py:185-199."""
    return torch.cat([log_add_exp(log_x_t[:, :-1] + _ext(sd["log_at"], t), _ext(sd["log_bt"], t)),
                      log_add_exp(log_x_t[:, -1:] + _ext(sd["log_1_min_ct"], t), _ext(sd["log_ct"], t))], dim=1)


def q_pred(log_x_start, t, sd):
    """Reference: diffusion_transformer.py:201-218 (t wrapped modulo T+1)."""
    T = sd["log_at"].shape[0]
    t = (t + (T + 1)) % (T + 1)
    return torch.cat([log_add_exp(log_x_start[:, :-1] + _ext(sd["log_cumprod_at"], t), _ext(sd["log_cumprod_bt"], t)),
                      log_add_exp(log_x_start[:, -1:] + _ext(sd["log_1_min_cumprod_ct"], t),
                                  _ext(sd["log_cumprod_ct"], t))], dim=1)


def predict_start(log_x_t, cond, t, sd, n_head=16):
    """Reference: diffusion_transformer.py:220-238 (fp64 log_softmax, append -70, clamp)."""
    out = denoiser(log_x_t.argmax(1), cond, t, sd, n_head)
    return predict_start_from_logits(out)


def predict_start_from_logits(out):
    lp = F.log_softmax(out.double(), dim=1).float()
    lp = torch.cat((lp, torch.zeros(lp.shape[0], 1, lp.shape[2]) - 70), dim=1)
    return torch.clamp(lp, -70, 0)


def cf_mix(log_c, log_u, scale):
    """Reference: cf_predict_start, diffusion_transformer.py:240-249 (inputs = predict_start()[:, :-1])."""
    mix = log_u + scale * (log_c - log_u)
    mix = mix - torch.logsumexp(mix, dim=1, keepdim=True)
    mix = mix.clamp(-70, 0)
    return torch.cat((mix, torch.zeros(mix.shape[0], 1, mix.shape[2]) - 70), dim=1)


def cf_predict_start(log_x_t, cond, cf_cond, t, sd, scale, n_head=16):
    lc = predict_start(log_x_t, cond, t, sd, n_head)[:, :-1]
    lu = predict_start(log_x_t, cf_cond, t, sd, n_head)[:, :-1]
    return cf_mix(lc, lu, scale)


def q_posterior(log_x_start, log_x_t, t, sd):
    """Reference: diffusion_transformer.py:251-283."""
    B, K1, L = log_x_start.shape
    mask = (log_x_t.argmax(1) == K1 - 1).unsqueeze(1)
    log_zero = torch.full((B, 1, L), LOG_ZERO)
    log_one = torch.zeros(B, 1, 1)
    log_qt = q_pred(log_x_t, t, sd)[:, :-1]
    log_qt = (~mask) * log_qt + mask * _ext(sd["log_cumprod_ct"], t).expand(-1, K1 - 1, -1)
    log_q1 = torch.cat((q_pred_one_timestep(log_x_t, t, sd)[:, :-1], log_zero), dim=1)
    ct_vec = torch.cat((_ext(sd["log_ct"], t).expand(-1, K1 - 1, -1), log_one), dim=1)
    log_q1 = (~mask) * log_q1 + mask * ct_vec
    q = torch.cat((log_x_start[:, :-1] - log_qt, log_zero), dim=1)
    s = torch.logsumexp(q, dim=1, keepdim=True)
    q = q - s
    return torch.clamp(q_pred(q, t - 1, sd) + log_q1 + s, -70, 0)


def gumbel_argmax(logits, seed, stream, row0=0):
    """Reference: log_sample_categorical, diffusion_transformer.py:354-359, with Philox uniforms
    (seed=None: torch.rand_like as in the reference, used for CPU-baseline timing only)."""
    B, K1, L = logits.shape
    u = torch.rand_like(logits) if seed is None else torch.from_numpy(philox.uniform_bkl(seed, stream, B, K1, L, row0=row0))
    g = -torch.log(-torch.log(u + 1e-30) + 1e-30)
    return (g + logits).argmax(dim=1)


def p_sample_step(tok, cond, cf_cond, t, sd, scale, seed, stream, first=False, n_head=16, row0=0):
    """One reverse step on int tokens.  Reference: p_sample/p_pred, diffusion_transformer.py:285-352
    (prior_rule == 0 branch).  ``first``: the all-[MASK] start uses true -inf rows (:613-618)."""
    K1 = sd["transformer.content_emb.emb.weight"].shape[0]
    if first:
        log_z = torch.full((tok.shape[0], K1, tok.shape[1]), float("-inf"))
        log_z[:, -1] = 0
    else:
        log_z = index_to_log_onehot(tok, K1)
    rec = cf_predict_start(log_z, cond, cf_cond, t, sd, scale, n_head)
    post = q_posterior(rec, log_z, t, sd)
    return gumbel_argmax(post, seed, stream, row0=row0), post


def sample(B, L, cond, cf_cond, sd, scale, seed, n_head=16, trace=None, row0=0, steps=None, stream0=0):
    """Reference: DiffusionTransformer.sample with filter_ratio=0, diffusion_transformer.py:568-644.  Reverse step i draws
    Philox stream `stream0 + i`."""
    T = sd["log_at"].shape[0]
    K1 = sd["transformer.content_emb.emb.weight"].shape[0]
    tok = torch.full((B, L), K1 - 1, dtype=torch.long)
    for i, step in enumerate(range(T - 1, -1, -1)):
        if steps is not None and i >= steps:
            break
        t = torch.full((B,), step, dtype=torch.long)
        tok, _ = p_sample_step(tok, cond, cf_cond, t, sd, scale, seed, stream=stream0 + i, first=(i == 0), n_head=n_head,
                               row0=row0)
        if trace is not None:
            trace.append(tok.clone())
    return tok


def sample_from(content_token, filter_ratio, cond, cf_cond, sd, scale, seed, n_head=16, row0=0, stream0=0):
    """DiffusionTransformer.sample with filter_ratio > 0 and a content_token (diffusion_transformer.py:590-592, :626-634): start_step =
    int(T * filter_ratio); x_t = q_sample(content_token, t = start_step - 1) (:629-630, Gumbel arg-max of q_pred: one draw, stream
    `stream0`), then the start_step reverse steps t = start_step - 1 ... 0 (draws stream0 + 1 ...).  The reference's loop calls
    `self.p_sample(log_z, cond_emb, cf_cond_emb, t)` with four of p_sample's six positional parameters (:634 vs :304-305) and so
    raises TypeError as written; this restates what that branch is written to do (the upstream VQ-Diffusion behaviour: one p_sample
    per step, as in the filter_ratio = 0 branch)."""
    T = sd["log_at"].shape[0]
    K1 = sd["transformer.content_emb.emb.weight"].shape[0]
    start = int(T * filter_ratio)
    assert start > 0
    B = content_token.shape[0]
    t = torch.full((B,), start - 1, dtype=torch.long)
    tok = gumbel_argmax(q_pred(index_to_log_onehot(content_token, K1), t, sd), seed, stream0, row0=row0)
    for i, step in enumerate(range(start - 1, -1, -1)):
        t = torch.full((B,), step, dtype=torch.long)
        tok, _ = p_sample_step(tok, cond, cf_cond, t, sd, scale, seed, stream=stream0 + 1 + i, n_head=n_head, row0=row0)
    return tok


# ----------------------------------------------------------------------------- training loss
def train_loss(x0, cond, t, pt, sd, seed, stream, aux_weight=5.0e-4, adaptive_aux=True, mask_weight=(1, 1),
               n_head=16):
    """Reference: _train_loss + forward, diffusion_transformer.py:391-457, :520-565.
    Returns (loss scalar, exp(log_model_prob), x0_recon, kl_loss per sample)."""
    T = sd["log_at"].shape[0]
    K1 = sd["transformer.content_emb.emb.weight"].shape[0]
    log_x0 = index_to_log_onehot(x0, K1)
    xt = gumbel_argmax(q_pred(log_x0, t, sd), seed, stream)              # q_sample :361-366
    log_xt = index_to_log_onehot(xt, K1)
    log_x0_recon = predict_start(log_xt, cond, t, sd, n_head)
    log_model = q_posterior(log_x0_recon, log_xt, t, sd)
    log_true = q_posterior(log_x0, log_xt, t, sd)
    kl = (log_true.exp() * (log_true - log_model)).sum(dim=1)
    mregion = (xt == K1 - 1).float()
    mw = mregion * mask_weight[0] + (1.0 - mregion) * mask_weight[1]
    kl = (kl * mw).sum(-1)
    nll = -(log_x0.exp() * log_model).sum(dim=1).sum(-1)
    m0 = (t == 0).float()
    kl_loss = m0 * nll + (1.0 - m0) * kl
    vb = kl_loss / pt
    if aux_weight != 0:
        kl_aux = (log_x0[:, :-1].exp() * (log_x0[:, :-1] - log_x0_recon[:, :-1])).sum(dim=1)
        kl_aux = (kl_aux * mw).sum(-1)
        kl_aux_loss = m0 * nll + (1.0 - m0) * kl_aux
        w = (1 - t / T) + 1.0 if adaptive_aux else 1.0
        vb = vb + w * aux_weight * kl_aux_loss / pt
    loss = vb.sum() / (x0.shape[0] * x0.shape[1])
    return loss, log_model.exp(), log_x0_recon.argmax(1), kl_loss
