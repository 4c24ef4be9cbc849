"""VQ-VAE shell with the reference's constructor, method names and state_dict keys
(reference: src/models/networks/videogpt_vq_vae.py), computing on gfx950 through the C ABI.

The nn.Module tree below exists only to *own parameters under the reference's names* (SURVEY.md
appendix C) so reference checkpoints load unchanged; none of its torch forward()s is ever called.
Inference (encode / decode / forward in eval mode) runs channels-last on the HIP kernels:

  SamePadConv3d / SamePadConvTranspose3d -> gsdd_gemm implicit GEMM (transposed conv = sub-pixel phases)
  BatchNorm3d(eval)+ReLU                 -> folded into the consuming GEMM's prologue / producing epilogue
  AxialBlock                             -> one fused q|k|v GEMM (9C outputs), gsdd_axial_attention, one fc GEMM (K = 3C)
  Codebook nearest code                  -> gsdd_nearest_code
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import GsddError


# ----------------------------------------------------------------------------- parameter containers
class SamePadConv3d(nn.Module):
    """Parameter holder for videogpt_vq_vae.py:289-309."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=True):
        super().__init__()
        k = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        s = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        self.kernel_size, self.stride = k, s
        self.pad_front = tuple((kk - ss) // 2 + (kk - ss) % 2 for kk, ss in zip(k, s))
        self.conv = nn.Conv3d(in_channels, out_channels, k, stride=s, padding=0, bias=bias)


class SamePadConvTranspose3d(nn.Module):
    """Parameter holder for videogpt_vq_vae.py:312-332."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=True):
        super().__init__()
        k = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        s = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        self.kernel_size, self.stride = k, s
        self.pad_front = tuple((kk - ss) // 2 + (kk - ss) % 2 for kk, ss in zip(k, s))
        self.convt = nn.ConvTranspose3d(in_channels, out_channels, k, stride=s, bias=bias,
                                        padding=tuple(kk - 1 for kk in k))


class MultiHeadAttention(nn.Module):
    """Parameter holder for model_utils.py:211-233 (axial variant, no bias on q/k/v)."""

    def __init__(self, dim, n_head, n_layer=1):
        super().__init__()
        self.n_head = n_head
        self.w_qs = nn.Linear(dim, dim, bias=False)
        self.w_ks = nn.Linear(dim, dim, bias=False)
        self.w_vs = nn.Linear(dim, dim, bias=False)
        self.fc = nn.Linear(dim, dim, bias=True)
        for lin in (self.w_qs, self.w_ks, self.w_vs):
            lin.weight.data.normal_(std=1.0 / np.sqrt(dim))
        self.fc.weight.data.normal_(std=1.0 / np.sqrt(dim * n_layer))


class AxialBlock(nn.Module):
    def __init__(self, n_hiddens, n_head):
        super().__init__()
        self.attn_w = MultiHeadAttention(n_hiddens, n_head)
        self.attn_h = MultiHeadAttention(n_hiddens, n_head)
        self.attn_t = MultiHeadAttention(n_hiddens, n_head)


class AttentionResidualBlock(nn.Module):
    def __init__(self, n_hiddens):
        super().__init__()
        self.block = nn.Sequential(
            nn.BatchNorm3d(n_hiddens), nn.ReLU(),
            SamePadConv3d(n_hiddens, n_hiddens // 2, 3, bias=False),
            nn.BatchNorm3d(n_hiddens // 2), nn.ReLU(),
            SamePadConv3d(n_hiddens // 2, n_hiddens, 1, bias=False),
            nn.BatchNorm3d(n_hiddens), nn.ReLU(),
            AxialBlock(n_hiddens, 2))


def _layer_strides(sample):
    n = np.array([int(math.log2(d)) for d in sample])
    out = []
    for _ in range(n.max()):
        out.append(tuple(2 if d > 0 else 1 for d in n))
        n -= 1
    return out


class Encoder(nn.Module):
    def __init__(self, n_hiddens, n_res_layers, downsample):
        super().__init__()
        self.convs = nn.ModuleList()
        in_channels = 3
        for i, stride in enumerate(_layer_strides(downsample)):
            in_channels = 3 if i == 0 else n_hiddens
            self.convs.append(SamePadConv3d(in_channels, n_hiddens, 4, stride=stride))
        self.conv_last = SamePadConv3d(in_channels, n_hiddens, kernel_size=3)   # (sic) reference :240
        self.res_stack = nn.Sequential(*[AttentionResidualBlock(n_hiddens) for _ in range(n_res_layers)],
                                       nn.BatchNorm3d(n_hiddens), nn.ReLU())


class Decoder(nn.Module):
    def __init__(self, n_hiddens, n_res_layers, upsample):
        super().__init__()
        self.res_stack = nn.Sequential(*[AttentionResidualBlock(n_hiddens) for _ in range(n_res_layers)],
                                       nn.BatchNorm3d(n_hiddens), nn.ReLU())
        strides = _layer_strides(upsample)
        self.convts = nn.ModuleList()
        for i, us in enumerate(strides):
            out_channels = 3 if i == len(strides) - 1 else n_hiddens
            self.convts.append(SamePadConvTranspose3d(n_hiddens, out_channels, 4, stride=us))


class Codebook(nn.Module):
    def __init__(self, n_codes, embedding_dim):
        super().__init__()
        self.register_buffer("embeddings", torch.randn(n_codes, embedding_dim))
        self.register_buffer("N", torch.zeros(n_codes))
        self.register_buffer("z_avg", self.embeddings.data.clone())
        self.n_codes, self.embedding_dim = n_codes, embedding_dim
        self._need_init = True


# ----------------------------------------------------------------------------- weight repacking (host logic, CPU-testable)
def conv_taps(kernel, stride, pad_front):
    """Tap offsets (dt,dh,dw) of SamePadConv3d in kernel order (kt,kh,kw): input = o*stride + k - pad_front."""
    kt, kh, kw = kernel
    return [(a - pad_front[0], b - pad_front[1], c - pad_front[2]) for a in range(kt) for b in range(kh)
            for c in range(kw)]


def pack_conv_weight(w):
    """(Cout,Cin,kt,kh,kw) -> [taps][Cout][Cin]"""
    co, ci = w.shape[:2]
    return w.permute(2, 3, 4, 0, 1).reshape(-1, co, ci).contiguous()


def pack_conv0_weight(w, cpad=4):
    """First conv (Cin=3): merge kw into the channel axis -> [kt*kh][Cout][kw*cpad] over a W-padded NDHWC4 input."""
    co, ci, kt, kh, kw = w.shape
    wp = torch.zeros((co, cpad, kt, kh, kw), dtype=w.dtype, device=w.device)
    wp[:, :ci] = w
    return wp.permute(2, 3, 0, 4, 1).reshape(kt * kh, co, kw * cpad).contiguous()


def convT_phases(kernel, stride, pad_front):
    """Sub-pixel decomposition of SamePadConvTranspose3d (padding = k-1 on a front/back padded input).

    For one dim: out o = o'*s + p gets input x = o' + (p + k - 1 - kk)/s - pad_front for the kk with
    (p + k - 1 - kk) % s == 0.  Returns [(phase(pt,ph,pw), [(kt,kh,kw)], [(dt,dh,dw)])]."""
    per_dim = []
    for k, s, pf in zip(kernel, stride, pad_front):
        phases = []
        for p in range(s):
            taps = [(kk, (p + k - 1 - kk) // s - pf) for kk in range(k) if (p + k - 1 - kk) % s == 0]
            phases.append(taps)
        per_dim.append(phases)
    out = []
    for pt, tt in enumerate(per_dim[0]):
        for ph, th in enumerate(per_dim[1]):
            for pw, tw in enumerate(per_dim[2]):
                ks = [(a[0], b[0], c[0]) for a in tt for b in th for c in tw]
                offs = [(a[1], b[1], c[1]) for a in tt for b in th for c in tw]
                out.append(((pt, ph, pw), ks, offs))
    return out


def pack_convT_weight(w, ks):
    """(Cin,Cout,kt,kh,kw) restricted to taps ks -> [taps][Cout][Cin]"""
    return torch.stack([w[:, :, a, b, c].t() for (a, b, c) in ks]).contiguous()


def fold_bn(bn):
    """eval-mode BatchNorm as y = x*scale + shift (videogpt_vq_vae.py:125-133 use running stats in eval)."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    return scale.contiguous(), shift.contiguous()


# ----------------------------------------------------------------------------- the module
class VQVAE(nn.Module):
    """Drop-in for src.models.networks.videogpt_vq_vae.VQVAE (constructor :15-16, encode :45-51,
    decode :53-56, forward :58-72, latent_shape :38-43)."""

    def __init__(self, checkpoint_path, embedding_dim, n_codes, n_hiddens, n_res_layers, downsample,
                 sequence_length, resolution, **kwargs):
        super().__init__()
        self.embedding_dim, self.n_codes = embedding_dim, n_codes
        self.n_hiddens, self.n_res_layers = n_hiddens, n_res_layers
        self.downsample = list(downsample)
        self.sequence_length, self.resolution = sequence_length, resolution
        self.encoder = Encoder(n_hiddens, n_res_layers, self.downsample)
        self.decoder = Decoder(n_hiddens, n_res_layers, self.downsample)
        self.pre_vq_conv = SamePadConv3d(n_hiddens, embedding_dim, 1)
        self.post_vq_conv = SamePadConv3d(embedding_dim, n_hiddens, 1)
        self.codebook = Codebook(n_codes, embedding_dim)
        self._packed = None
        self._packed_key = None

    @property
    def device(self):
        return self.codebook.embeddings.device

    @property
    def latent_shape(self):
        shape = (self.sequence_length, self.resolution, self.resolution)
        return tuple(s // d for s, d in zip(shape, self.downsample))

    # ------------------------------------------------------------------ packed weights
    def _state_key(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def packed(self):
        key = self._state_key()
        if self._packed is None or key != self._packed_key:
            with torch.no_grad():
                self._packed = self._pack()
            self._packed_key = key
        return self._packed

    def _pack_res_stack(self, stack, dev):
        blocks = []
        C_ = self.n_hiddens
        for i in range(self.n_res_layers):
            b = stack[i].block
            ax = b[8]
            wqkv = torch.cat([torch.cat([a.w_qs.weight, a.w_ks.weight, a.w_vs.weight], 0)
                              for a in (ax.attn_w, ax.attn_h, ax.attn_t)], 0).contiguous()          # [9C][C]
            wfc = torch.cat([ax.attn_w.fc.weight, ax.attn_h.fc.weight, ax.attn_t.fc.weight], 1).contiguous()  # [C][3C]
            bfc = (ax.attn_w.fc.bias + ax.attn_h.fc.bias + ax.attn_t.fc.bias).contiguous()
            blocks.append(dict(bn0=fold_bn(b[0]), w3=pack_conv_weight(b[2].conv.weight),
                               taps3=ops.taps_tensor(conv_taps((3, 3, 3), (1, 1, 1), (1, 1, 1)), dev),
                               bn3=fold_bn(b[3]), w1=pack_conv_weight(b[5].conv.weight), bn6=fold_bn(b[6]),
                               wqkv=wqkv.view(1, 9 * C_, C_), wfc=wfc.view(1, C_, 3 * C_), bfc=bfc))
        return dict(blocks=blocks, bn_out=fold_bn(stack[self.n_res_layers]))

    def _pack(self):
        dev = self.device
        p = {}
        enc = []
        for i, c in enumerate(self.encoder.convs):
            k, s, pf = c.kernel_size, c.stride, c.pad_front
            if i == 0:
                taps = [(a - pf[0], b - pf[1], 0) for a in range(k[0]) for b in range(k[1])]
                enc.append(dict(w=pack_conv0_weight(c.conv.weight), taps=ops.taps_tensor(taps, dev), stride=s,
                                bias=c.conv.bias.contiguous(), padw=pf[2], first=True))
            else:
                enc.append(dict(w=pack_conv_weight(c.conv.weight), taps=ops.taps_tensor(conv_taps(k, s, pf), dev),
                                stride=s, bias=c.conv.bias.contiguous(), first=False))
        p["enc_convs"] = enc
        cl = self.encoder.conv_last
        p["enc_last"] = dict(w=pack_conv_weight(cl.conv.weight),
                             taps=ops.taps_tensor(conv_taps(cl.kernel_size, cl.stride, cl.pad_front), dev),
                             bias=cl.conv.bias.contiguous())
        p["enc_res"] = self._pack_res_stack(self.encoder.res_stack, dev)
        p["dec_res"] = self._pack_res_stack(self.decoder.res_stack, dev)
        p["pre_w"] = pack_conv_weight(self.pre_vq_conv.conv.weight)
        p["pre_b"] = self.pre_vq_conv.conv.bias.contiguous()
        p["post_w"] = pack_conv_weight(self.post_vq_conv.conv.weight)
        p["post_b"] = self.post_vq_conv.conv.bias.contiguous()
        p["codebook"] = self.codebook.embeddings.contiguous()
        dec = []
        for c in self.decoder.convts:
            phases = []
            for (ph, ks, offs) in convT_phases(c.kernel_size, c.stride, c.pad_front):
                phases.append(dict(phase=ph, w=pack_convT_weight(c.convt.weight, ks), taps=ops.taps_tensor(offs, dev)))
            dec.append(dict(phases=phases, stride=c.stride, bias=c.convt.bias.contiguous(),
                            cout=c.convt.weight.shape[1]))
        p["dec_convts"] = dec
        return p

    # ------------------------------------------------------------------ HIP pipelines (channels-last rows)
    def _res_stack(self, h, dims, rp, stack=None):
        """AttentionResidualBlock x n (videogpt_vq_vae.py:122-138); the trailing BN+ReLU is returned as a
        prologue for the consumer.  h: rows [M][C].  `stack` (the nn.Sequential) switches to train mode: BatchNorm
        uses batch statistics (gsdd_bn_train, running stats updated) and is applied as the consumer's prologue."""
        N, T, H, W = dims
        M, C_ = h.shape
        dev = h.device
        train = stack is not None
        for i, blk in enumerate(rp["blocks"]):
            a = torch.empty((M, C_ // 2), dtype=torch.float32, device=dev)
            b = torch.empty((M, C_), dtype=torch.float32, device=dev)
            qkv = torch.empty((M, 9 * C_), dtype=torch.float32, device=dev)
            if train:
                mods = stack[i].block
                ops.gemm(h, blk["w3"], a, in_dims=dims, out_grid=(T, H, W), taps=blk["taps3"], ntaps=27,
                         pro=ops.bn_train(h, mods[0]))
                ops.gemm(a, blk["w1"], b, in_dims=dims, out_grid=(T, H, W), pro=ops.bn_train(a, mods[3]))
                ops.gemm(b, blk["wqkv"], qkv, in_dims=dims, out_grid=(T, H, W), pro=ops.bn_train(b, mods[6]))
            else:
                ops.gemm(h, blk["w3"], a, in_dims=dims, out_grid=(T, H, W), taps=blk["taps3"], ntaps=27,
                         pro=blk["bn0"], epi_scale=blk["bn3"][0], epi_shift=blk["bn3"][1], act=ops.ACT_RELU)
                ops.gemm(a, blk["w1"], b, in_dims=dims, out_grid=(T, H, W), epi_scale=blk["bn6"][0],
                         epi_shift=blk["bn6"][1], act=ops.ACT_RELU)
                ops.gemm(b, blk["wqkv"], qkv, in_dims=dims, out_grid=(T, H, W))
            att = torch.empty((M, 3 * C_), dtype=torch.float32, device=dev)
            ops.axial_attention(qkv, dims, C_, 2, att)
            hn = torch.empty((M, C_), dtype=torch.float32, device=dev)
            ops.gemm(att, blk["wfc"], hn, in_dims=dims, out_grid=(T, H, W), epi_shift=blk["bfc"], residual=h)
            h = hn
        if train:
            return h, ops.bn_train(h, stack[self.n_res_layers])
        return h, rp["bn_out"]

    def _encode_rows(self, x, train=False):
        """x (B,3,T,H,W) on the GPU -> (z rows [M][E], latent dims)."""
        if not x.is_cuda:
            raise GsddError("VQVAE runs on the HIP path only: move the module and the input to a ROCm device")
        p = self.packed()
        x = x.contiguous().float()
        B, _, T, H, W = x.shape
        dims = None
        h = None
        for i, c in enumerate(p["enc_convs"]):
            s = c["stride"]
            if c["first"]:
                xr = ops.ncdhw_to_rows(x, 4, c["padw"])                    # (B,T,H,W+2p,4)
                Wp = W + 2 * c["padw"]
                To, Ho, Wo = T // s[0], H // s[1], W // s[2]
                h = torch.empty((B * To * Ho * Wo, self.n_hiddens), dtype=torch.float32, device=x.device)
                ops.gemm(xr, c["w"], h, in_dims=(B, T, H, Wp), out_grid=(To, Ho, Wo), stride=s, taps=c["taps"],
                         ntaps=c["w"].shape[0], cin=c["w"].shape[2], in_pitch=4, epi_shift=c["bias"], act=ops.ACT_RELU)
            else:
                Ti, Hi, Wi = dims[1:]
                To, Ho, Wo = Ti // s[0], Hi // s[1], Wi // s[2]
                hn = torch.empty((B * To * Ho * Wo, self.n_hiddens), dtype=torch.float32, device=x.device)
                ops.gemm(h, c["w"], hn, in_dims=dims, out_grid=(To, Ho, Wo), stride=s, taps=c["taps"],
                         ntaps=c["w"].shape[0], epi_shift=c["bias"], act=ops.ACT_RELU)
                h = hn
            dims = (B, To, Ho, Wo)
        cl = p["enc_last"]
        hn = torch.empty_like(h)
        ops.gemm(h, cl["w"], hn, in_dims=dims, out_grid=dims[1:], taps=cl["taps"], ntaps=27, epi_shift=cl["bias"])
        h, bn_out = self._res_stack(hn, dims, p["enc_res"], self.encoder.res_stack if train else None)
        z = torch.empty((h.shape[0], self.embedding_dim), dtype=torch.float32, device=x.device)
        ops.gemm(h, p["pre_w"], z, in_dims=dims, out_grid=dims[1:], pro=bn_out, epi_shift=p["pre_b"])
        return z, dims

    def _decode_rows(self, src, dims, gather=None, train=False):
        """post_vq_conv + Decoder on rows.  src: codebook [K][E] with gather=codes, or rows [M][E]."""
        p = self.packed()
        B, T, H, W = dims
        M = B * T * H * W
        dev = src.device
        h = torch.empty((M, self.n_hiddens), dtype=torch.float32, device=dev)
        ops.gemm(src, p["post_w"], h, in_dims=dims, out_grid=(T, H, W), gather=gather, epi_shift=p["post_b"])
        h, pro = self._res_stack(h, dims, p["dec_res"], self.decoder.res_stack if train else None)
        n_up = len(p["dec_convts"])
        for i, ct in enumerate(p["dec_convts"]):
            s = ct["stride"]
            To, Ho, Wo = T * s[0], H * s[1], W * s[2]
            last = i == n_up - 1
            if last:
                out = torch.empty((B, ct["cout"], To, Ho, Wo), dtype=torch.float32, device=dev)
            else:
                out = torch.empty((B * To * Ho * Wo, ct["cout"]), dtype=torch.float32, device=dev)
            for ph in ct["phases"]:
                ops.gemm(h, ph["w"], out, in_dims=(B, T, H, W), out_grid=(T, H, W), taps=ph["taps"],
                         ntaps=ph["w"].shape[0], pro=pro, epi_shift=ct["bias"],
                         act=ops.ACT_NONE if last else ops.ACT_RELU, out_dims=(To, Ho, Wo), out_step=s,
                         out_off=ph["phase"], out_mode=1 if last else 0)
            h, pro = out, None
            T, H, W = To, Ho, Wo
        return h

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def encode(self, x, include_embeddings=False):
        """videogpt_vq_vae.py:45-51.  In train mode the reference's encode runs BatchNorm on batch statistics and the codebook's
        data-init / EMA update / restart (Codebook.forward :174-222 is mode-dependent) -- so does this one."""
        if self.training:
            z, dims, idx, zq, _, _ = self._quantise_train(x)
        else:
            z, dims = self._encode_rows(x)
            idx = torch.empty((z.shape[0],), dtype=torch.int64, device=z.device)
            zq = torch.empty_like(z) if include_embeddings else None
            ops.nearest_code(z, self.packed()["codebook"], idx, zq)
        enc = idx.view(dims)
        if include_embeddings:
            emb_st = (zq - z) + z                                   # straight-through value (:216)
            return enc, emb_st.view(*dims, -1).permute(0, 4, 1, 2, 3).contiguous()
        return enc

    @torch.no_grad()
    def decode(self, encodings):
        """videogpt_vq_vae.py:53-56 (BatchNorm follows the module's mode, as there)."""
        if not encodings.is_cuda:
            raise GsddError("VQVAE runs on the HIP path only: move the module and the input to a ROCm device")
        enc = encodings.contiguous().long()
        out = self._decode_rows(self.packed()["codebook"], tuple(enc.shape), gather=enc.view(-1), train=self.training)
        if self.training:
            self._packed = None                                     # BN running stats changed in place
        return out

    def _tile(self, x):
        """Codebook._tile (videogpt_vq_vae.py:151-158): repeat + jitter when there are fewer latents than codes."""
        d, ew = x.shape
        if d < self.n_codes:
            n_repeats = (self.n_codes + d - 1) // d
            x = x.repeat(n_repeats, 1)
            src = getattr(self, "noise_source", None)          # tests inject the jitter (a callable shape -> tensor)
            noise = src(tuple(x.shape)).to(x) if src is not None else torch.randn_like(x)
            x = x + noise * (0.01 / math.sqrt(ew))
        return x

    def _draw_rows(self, z):
        """`y[randperm][:n_codes]` (:165, :206-207) with the C1/C3 broadcast from rank 0 (:168-169, :210-211).
        tests inject `self.perm_source` (a callable n -> permutation) in place of torch.randperm."""
        import torch.distributed as dist
        y = self._tile(z)
        src = getattr(self, "perm_source", None)
        perm = (src(y.shape[0]) if src is not None else torch.randperm(y.shape[0], device=z.device)).to(z.device)
        if src is not None and (perm.numel() < self.n_codes or int(perm.max()) >= y.shape[0] or int(perm.min()) < 0):
            raise GsddError(f"perm_source must return at least {self.n_codes} indices below {y.shape[0]}")
        perm = perm[: self.n_codes].long().contiguous()
        if dist.is_available() and dist.is_initialized():
            rows = y[perm].contiguous()
            dist.broadcast(rows, 0)
            return rows, torch.arange(self.n_codes, device=z.device)
        return y.contiguous(), perm

    @torch.no_grad()
    def _quantise_train(self, x):
        """Encoder + Codebook.forward in train mode (videogpt_vq_vae.py:174-222): BatchNorm batch statistics + running-stat
        update, codebook data-init on the first call, EMA update with all-reduced statistics (C2) and dead-code restart.
        -> (z rows, dims, idx, zq rows [the pre-update code vectors], commitment loss, (., perplexity))."""
        import torch.distributed as dist
        cb = self.codebook
        z, dims = self._encode_rows(x, train=True)
        if cb._need_init:                                              # _init_embeddings (:160-172)
            cb._need_init = False
            rows, perm = self._draw_rows(z)
            k_rand = rows[perm]
            cb.embeddings.data.copy_(k_rand)
            cb.z_avg.data.copy_(k_rand)
            cb.N.data.fill_(1.0)
        idx = torch.empty((z.shape[0],), dtype=torch.int64, device=z.device)
        zq = torch.empty_like(z)
        ops.nearest_code(z, cb.embeddings.contiguous(), idx, zq)
        commitment = ops.mse(z, zq, 0.25)
        n_local, encode_sum = ops.codebook_ema_stats(z, idx, self.n_codes)
        n_total = n_local
        if dist.is_available() and dist.is_initialized():               # C2 (:196-198)
            n_total = n_local.clone()
            dist.all_reduce(n_total)
            dist.all_reduce(encode_sum)
        rows, perm = self._draw_rows(z)
        scal = ops.codebook_ema_update(rows, idx, perm, cb.N, cb.z_avg, cb.embeddings, n_total, encode_sum,
                                       n_local=n_local, m_local=z.shape[0])
        self._packed = None                                             # BN running stats / codebook changed in place
        return z, dims, idx, zq, commitment, scal

    @torch.no_grad()
    def _forward_train(self, x):
        """Train-mode forward value (videogpt_vq_vae.py:58-72).  The backward lives in vqvae_trainer.py."""
        z, dims, idx, zq, commitment, scal = self._quantise_train(x)
        emb_st = ((zq - z) + z).contiguous()
        x_recon = self._decode_rows(emb_st, dims, train=True)
        recon = ops.mse(x_recon, x, 1.0 / 0.06)
        self._packed = None
        return {"pred_data": x_recon, "gt_data": x,
                "losses": {"recon_loss": recon, "commitment_loss": commitment}, "perplexity": scal[1], "encodings": idx.view(dims)}

    def forward(self, batch, do_inference=False):
        x = batch["video"].to(self.device).contiguous().float()
        if self.training:
            if torch.is_grad_enabled():                                 # losses carry a grad_fn into the HIP backward
                from .vqvae_trainer import train_forward
                recon, commitment, x_recon = train_forward(self, x)
                return {"pred_data": x_recon, "gt_data": x, "losses": {"recon_loss": recon, "commitment_loss": commitment}}
            out = self._forward_train(x)
            out.pop("encodings")
            out.pop("perplexity")
            return out
        with torch.no_grad():
            z, dims = self._encode_rows(x)
            idx = torch.empty((z.shape[0],), dtype=torch.int64, device=z.device)
            zq = torch.empty_like(z)
            ops.nearest_code(z, self.packed()["codebook"], idx, zq)
            commitment = ops.mse(z, zq, 0.25)
            emb_st = ((zq - z) + z).contiguous()
            x_recon = self._decode_rows(emb_st, dims)
            recon = ops.mse(x_recon, x, 1.0 / 0.06)
        return {"pred_data": x_recon, "gt_data": x,
                "losses": {"recon_loss": recon, "commitment_loss": commitment}}
