"""CPU restatement of the reference VQ-VAE path (TEST INFRASTRUCTURE — never imported by the product).

Pinned against the reference itself: tests/golden/vqvae_*.npz were produced by importing
/root/reference (tests/golden/make_golden.py) and tests/test_oracle_golden.py checks every
function below against them.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this package.

All functions take ``sd``: a dict name -> torch.Tensor with the reference's state_dict names
 and plain torch CPU tensors in the reference's NCDHW layout.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def same_pad(kernel, stride):
    """Per-dim (front, back) zero padding.  Reference: videogpt_vq_vae.py:298-305 (and :320-325)."""
    pads = []
    for k, s in zip(kernel, stride):
        p = k - s
        pads.append((p // 2 + p % 2, p // 2))
    return pads            # [(t0,t1),(h0,h1),(w0,w1)]


def _fpad(x, pads):
    flat = []
    for p in pads[::-1]:    # F.pad starts from the last dim
        flat += [p[0], p[1]]
    return F.pad(x, flat)


def same_pad_conv3d(x, w, b, stride):
    """Reference: SamePadConv3d.forward, videogpt_vq_vae.py:308-309."""
    k = w.shape[2:]
    return F.conv3d(_fpad(x, same_pad(k, stride)), w, b, stride=stride)


def same_pad_convT3d(x, w, b, stride):
    """Reference: SamePadConvTranspose3d.forward, videogpt_vq_vae.py:326-332 (padding = k-1)."""
    k = w.shape[2:]
    return F.conv_transpose3d(_fpad(x, same_pad(k, stride)), w, b, stride=stride,
                              padding=tuple(kk - 1 for kk in k))


def bn_eval(x, sd, p, eps=1e-5):
    """nn.BatchNorm3d in eval mode (running statistics).  Reference: videogpt_vq_vae.py:125-133,242-247."""
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        False, 0.0, eps)


def axial_mha(x_cl, sd, p, axis, n_head=2):
    """One MultiHeadAttention(attn_type='axial') on channels-last x (B,T,H,W,C).

    Reference: model_utils.py:211-289 (projections, head split), :318-337 (AxialAttention),
    :586-600 (softmax(QK^T/sqrt(d))V).  ``axis`` in {1,2,3} = attend along T, H or W.
    """
    B, T, H, W, C = x_cl.shape
    d = C // n_head
    q = F.linear(x_cl, sd[p + "w_qs.weight"]).view(B, T, H, W, n_head, d)
    k = F.linear(x_cl, sd[p + "w_ks.weight"]).view(B, T, H, W, n_head, d)
    v = F.linear(x_cl, sd[p + "w_vs.weight"]).view(B, T, H, W, n_head, d)
    # move the attended axis next to d: (..., axis, d)
    q, k, v = (z.movedim(axis, -2) for z in (q, k, v))      # (B, a, b, n_head, S, d)
    att = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d)
    att = F.softmax(att, dim=-1)
    a = torch.matmul(att, v).movedim(-2, axis)              # back to (B,T,H,W,n_head,d)
    a = a.reshape(B, T, H, W, C)
    return F.linear(a, sd[p + "fc.weight"], sd[p + "fc.bias"])


def axial_block(x, sd, p):
    """Reference: AxialBlock.forward, videogpt_vq_vae.py:115-119 (attn_w + attn_h + attn_t)."""
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous()
    y = axial_mha(x_cl, sd, p + "attn_w.", 3) + axial_mha(x_cl, sd, p + "attn_h.", 2) \
        + axial_mha(x_cl, sd, p + "attn_t.", 1)
    return y.permute(0, 4, 1, 2, 3).contiguous()


def res_block(x, sd, p):
    """Reference: AttentionResidualBlock, videogpt_vq_vae.py:122-138."""
    h = F.relu(bn_eval(x, sd, p + "block.0."))
    h = same_pad_conv3d(h, sd[p + "block.2.conv.weight"], None, (1, 1, 1))
    h = F.relu(bn_eval(h, sd, p + "block.3."))
    h = same_pad_conv3d(h, sd[p + "block.5.conv.weight"], None, (1, 1, 1))
    h = F.relu(bn_eval(h, sd, p + "block.6."))
    return x + axial_block(h, sd, p + "block.8.")


def res_stack(x, sd, p, n_res_layers):
    """Reference: the nn.Sequential at videogpt_vq_vae.py:242-247 / :261-266."""
    for i in range(n_res_layers):
        x = res_block(x, sd, f"{p}{i}.")
    return F.relu(bn_eval(x, sd, f"{p}{n_res_layers}."))


def conv_strides(downsample):
    """Per-layer strides of the down/up-sampling convs.  Reference: videogpt_vq_vae.py:231-239 / :268-277."""
    n = [int(math.log2(d)) for d in downsample]
    out = []
    for _ in range(max(n)):
        out.append(tuple(2 if d > 0 else 1 for d in n))
        n = [d - 1 for d in n]
    return out


def encoder(x, sd, cfg):
    """Reference: Encoder.forward, videogpt_vq_vae.py:249-255."""
    h = x
    for i, s in enumerate(conv_strides(cfg["downsample"])):
        h = F.relu(same_pad_conv3d(h, sd[f"encoder.convs.{i}.conv.weight"], sd[f"encoder.convs.{i}.conv.bias"], s))
    h = same_pad_conv3d(h, sd["encoder.conv_last.conv.weight"], sd["encoder.conv_last.conv.bias"], (1, 1, 1))
    return res_stack(h, sd, "encoder.res_stack.", cfg["n_res_layers"])


def decoder(h, sd, cfg):
    """Reference: Decoder.forward, videogpt_vq_vae.py:279-285."""
    h = res_stack(h, sd, "decoder.res_stack.", cfg["n_res_layers"])
    strides = conv_strides(cfg["downsample"])
    for i, s in enumerate(strides):
        h = same_pad_convT3d(h, sd[f"decoder.convts.{i}.convt.weight"], sd[f"decoder.convts.{i}.convt.bias"], s)
        if i < len(strides) - 1:
            h = F.relu(h)
    return h


def nearest_code(z, E):
    """Reference: Codebook.forward, videogpt_vq_vae.py:178-188 (same association of the 3 terms)."""
    flat = z.permute(0, 2, 3, 4, 1).reshape(-1, z.shape[1])
    d = (flat ** 2).sum(dim=1, keepdim=True) - 2 * flat @ E.t() + (E.t() ** 2).sum(dim=0, keepdim=True)
    idx = torch.argmin(d, dim=1)
    return idx.view(z.shape[0], *z.shape[2:]), d


def pre_vq(x, sd, cfg):
    h = encoder(x, sd, cfg)
    return same_pad_conv3d(h, sd["pre_vq_conv.conv.weight"], sd["pre_vq_conv.conv.bias"], (1, 1, 1))


def encode(x, sd, cfg):
    """Reference: VQVAE.encode, videogpt_vq_vae.py:45-51 -> int64 (B,t,h,w)."""
    z = pre_vq(x, sd, cfg)
    idx, _ = nearest_code(z, sd["codebook.embeddings"])
    return idx


def decode(encodings, sd, cfg):
    """Reference: VQVAE.decode, videogpt_vq_vae.py:53-56."""
    h = F.embedding(encodings, sd["codebook.embeddings"]).permute(0, 4, 1, 2, 3).contiguous()
    h = same_pad_conv3d(h, sd["post_vq_conv.conv.weight"], sd["post_vq_conv.conv.bias"], (1, 1, 1))
    return decoder(h, sd, cfg)


def forward_eval(x, sd, cfg):
    """Reference: VQVAE.forward in eval mode, videogpt_vq_vae.py:58-72 + Codebook.forward :174-222."""
    z = pre_vq(x, sd, cfg)
    idx, _ = nearest_code(z, sd["codebook.embeddings"])
    emb = F.embedding(idx, sd["codebook.embeddings"]).permute(0, 4, 1, 2, 3).contiguous()
    commitment = 0.25 * F.mse_loss(z, emb)
    emb_st = (emb - z) + z
    h = same_pad_conv3d(emb_st, sd["post_vq_conv.conv.weight"], sd["post_vq_conv.conv.bias"], (1, 1, 1))
    rec = decoder(h, sd, cfg)
    return {"pred_data": rec, "gt_data": x,
            "losses": {"recon_loss": F.mse_loss(rec, x) / 0.06, "commitment_loss": commitment},
            "encodings": idx}


# ----------------------------------------------------------------------------- train-mode forward (autograd-friendly:
# tests differentiate recon_loss + commitment_loss through it for the gradient reference)
def bn_train(x, sd, p, out_sd, eps=1e-5, momentum=0.1):
    """nn.BatchNorm3d in train mode: batch statistics, running stats updated with the unbiased variance."""
    rm, rv = sd[p + "running_mean"].clone(), sd[p + "running_var"].clone()
    y = F.batch_norm(x, rm, rv, sd[p + "weight"], sd[p + "bias"], True, momentum, eps)
    out_sd[p + "running_mean"], out_sd[p + "running_var"] = rm, rv
    out_sd[p + "num_batches_tracked"] = sd[p + "num_batches_tracked"] + 1
    return y


def _res_stack_train(x, sd, p, n, out_sd):
    for i in range(n):
        q = f"{p}{i}."
        h = F.relu(bn_train(x, sd, q + "block.0.", out_sd))
        h = same_pad_conv3d(h, sd[q + "block.2.conv.weight"], None, (1, 1, 1))
        h = F.relu(bn_train(h, sd, q + "block.3.", out_sd))
        h = same_pad_conv3d(h, sd[q + "block.5.conv.weight"], None, (1, 1, 1))
        h = F.relu(bn_train(h, sd, q + "block.6.", out_sd))
        x = x + axial_block(h, sd, q + "block.8.")
    return F.relu(bn_train(x, sd, f"{p}{n}.", out_sd))


def tile(flat, n_codes, noise=None):
    """Reference: Codebook._tile, videogpt_vq_vae.py:151-158: with fewer latents than codes the rows are repeated
    ceil(n_codes / d) times and jittered by N(0, (0.01 / sqrt(E))^2); `noise` (tiled shape) replaces torch.randn_like."""
    d, ew = flat.shape
    if d < n_codes:
        n_repeats = (n_codes + d - 1) // d
        std = 0.01 / np.sqrt(ew)
        flat = flat.repeat(n_repeats, 1)
        assert noise is not None and tuple(noise.shape) == tuple(flat.shape), "tiling draws noise: inject it"
        flat = flat + noise * std
    return flat


def init_embeddings(flat, n_codes, perm, noise=None):
    """Reference: Codebook._init_embeddings, videogpt_vq_vae.py:160-172 -> (embeddings, z_avg, N)."""
    y = tile(flat, n_codes, noise)
    k_rand = y[torch.as_tensor(perm).long()][:n_codes]
    return k_rand.clone(), k_rand.clone(), torch.ones(n_codes)


def forward_train(x, sd, cfg, perm, init_perm=None, init_noise=None, noise=None):
    """Reference: VQVAE.forward with self.training (videogpt_vq_vae.py:58-72) and Codebook.forward (:174-222):
    data-init of the codebook first when `init_perm` is given (_need_init, :176-177), then the EMA branch (:193-214);
    `perm` / `init_perm` replace torch.randperm (:206, :165), `noise` / `init_noise` the randn_like of _tile.
    Returns the output dict and the updated buffers (same names as the state_dict)."""
    new = {}
    h = x
    for i, s in enumerate(conv_strides(cfg["downsample"])):
        h = F.relu(same_pad_conv3d(h, sd[f"encoder.convs.{i}.conv.weight"], sd[f"encoder.convs.{i}.conv.bias"], s))
    h = same_pad_conv3d(h, sd["encoder.conv_last.conv.weight"], sd["encoder.conv_last.conv.bias"], (1, 1, 1))
    h = _res_stack_train(h, sd, "encoder.res_stack.", cfg["n_res_layers"], new)
    z = same_pad_conv3d(h, sd["pre_vq_conv.conv.weight"], sd["pre_vq_conv.conv.bias"], (1, 1, 1))
    E = sd["codebook.embeddings"]
    K = E.shape[0]
    flat = z.permute(0, 2, 3, 4, 1).reshape(-1, z.shape[1])
    N0, zavg0 = sd["codebook.N"], sd["codebook.z_avg"]
    if init_perm is not None:
        E, zavg0, N0 = init_embeddings(flat.detach(), K, init_perm, init_noise)
    idx, _ = nearest_code(z, E)
    emb = F.embedding(idx, E).permute(0, 4, 1, 2, 3).contiguous()
    commitment = 0.25 * F.mse_loss(z, emb.detach())                   # :190
    onehot = F.one_hot(idx.view(-1), K).type_as(flat)
    n_total = onehot.sum(dim=0)
    encode_sum = flat.t() @ onehot
    N = N0 * 0.99 + 0.01 * n_total
    z_avg = zavg0 * 0.99 + 0.01 * encode_sum.t()
    n = N.sum()
    weights = (N + 1e-7) / (n + K * 1e-7) * n
    newE = z_avg / weights.unsqueeze(1)
    k_rand = tile(flat.detach(), K, noise)[torch.as_tensor(perm).long()][:K]
    usage = (N.view(K, 1) >= 1).float()
    newE = newE * usage + k_rand * (1 - usage)
    new.update({"codebook.N": N, "codebook.z_avg": z_avg, "codebook.embeddings": newE})
    emb_st = (emb - z).detach() + z                                   # straight-through estimator (:216)
    h = same_pad_conv3d(emb_st, sd["post_vq_conv.conv.weight"], sd["post_vq_conv.conv.bias"], (1, 1, 1))
    h = _res_stack_train(h, sd, "decoder.res_stack.", cfg["n_res_layers"], new)
    strides = conv_strides(cfg["downsample"])
    for i, s in enumerate(strides):
        h = same_pad_convT3d(h, sd[f"decoder.convts.{i}.convt.weight"], sd[f"decoder.convts.{i}.convt.bias"], s)
        if i < len(strides) - 1:
            h = F.relu(h)
    avg = onehot.mean(dim=0)
    perplexity = torch.exp(-torch.sum(avg * torch.log(avg + 1e-10)))
    out = {"pred_data": h, "gt_data": x, "losses": {"recon_loss": F.mse_loss(h, x) / 0.06, "commitment_loss": commitment},
           "perplexity": perplexity, "encodings": idx, "z": z}
    return out, new
