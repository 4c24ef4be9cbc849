"""Tensor-level wrappers over the C ABI (include/gsdd.h).  torch is used only to own device memory
and streams; every computation below is a HIP kernel in libgsdd.so."""
import ctypes as C
import os

import torch

from . import _lib as abi
from ._lib import GemmDesc, GsddError, LayerDesc, StepDesc, TrainDesc, check, lib, ptr, stream_ptr

ACT_NONE, ACT_RELU, ACT_GELU2 = 0, 1, 2


# ----------------------------------------------------------------------------- variant / precision selection
# The C ABI takes the arithmetic mode / kernel variant as an argument of each call (include/gsdd.h); libgsdd.so reads no environment.
# The debugging switches the tests and the A/B tools have always used are environment variables: they are translated HERE, per call, into
# those arguments (an explicit keyword argument of the wrapper wins over the environment).
_ATTN_MODES = {"auto": abi.ATTN_AUTO, "22": abi.ATTN_P22, "11": abi.ATTN_P11, "a8": abi.ATTN_A8, "a12": abi.ATTN_A12,
               "f32pv": abi.ATTN_F32PV, "kc256": abi.ATTN_KC256}
_LAYER_VARIANTS = {"auto": abi.LAYER_AUTO, "h2": abi.LAYER_H2, "x3p": abi.LAYER_X3P}
_ATTN_BWD_VARIANTS = {"auto": abi.ATTN_BWD_AUTO, "valu": abi.ATTN_BWD_VALU, "split": abi.ATTN_BWD_SPLIT, "fqc64": abi.ATTN_BWD_FQC64,
                      "fqc128": abi.ATTN_BWD_FQC128, "nw8": abi.ATTN_BWD_NW8, "dbg1": abi.ATTN_BWD_DBG1, "dbg2": abi.ATTN_BWD_DBG2}


def _pick(table, value, what):
    if isinstance(value, int):
        return value
    try:
        return table[str(value).lower()]
    except KeyError:
        raise GsddError(f"{what}: {value!r} is not one of {sorted(table)}") from None


def attn_mode(mode=None):
    """GSDD_ATTN_P = 22 | 11 | a8 | a12 (the sampler's P format), GSDD_ATTN_V3=1 (exact-f32 P.V kernel), GSDD_ATTN_KC=256."""
    if mode is not None:
        return _pick(_ATTN_MODES, mode, "attention mode")
    if os.environ.get("GSDD_ATTN_V3"):
        return abi.ATTN_F32PV
    if os.environ.get("GSDD_ATTN_KC") == "256":
        return abi.ATTN_KC256
    return _pick(_ATTN_MODES, os.environ.get("GSDD_ATTN_P", "auto"), "GSDD_ATTN_P")


def attn_train_mode(mode=None):
    """GSDD_ATTN_TRAIN_P = 22 | a8 (explicit values only; anything else is an error)."""
    v = mode if mode is not None else os.environ.get("GSDD_ATTN_TRAIN_P", "auto")
    m = _pick(_ATTN_MODES, v, "GSDD_ATTN_TRAIN_P")
    if m not in (abi.ATTN_AUTO, abi.ATTN_P22, abi.ATTN_A8):
        raise GsddError(f"GSDD_ATTN_TRAIN_P: {v!r} is not one of '22', 'a8'")
    return m


def attn_bwd_variant(variant=None):
    """GSDD_ATTN_BWD = valu | split | fqc64 | fqc128 | nw8 | dbg1 | dbg2 (development variants of the attention backward)."""
    return _pick(_ATTN_BWD_VARIANTS, variant if variant is not None else os.environ.get("GSDD_ATTN_BWD", "auto"), "GSDD_ATTN_BWD")


def layer_variant(variant=None):
    """GSDD_LAYER = h2 | x3p."""
    return _pick(_LAYER_VARIANTS, variant if variant is not None else os.environ.get("GSDD_LAYER", "auto"), "GSDD_LAYER")


def gemm_flags(exact_f32=None):
    """GSDD_GEMM_F32=1: the exact-f32 MFMA instead of the bf16x3 matrix pipe."""
    if exact_f32 is None:
        exact_f32 = bool(os.environ.get("GSDD_GEMM_F32"))
    return abi.GEMM_EXACT_F32 if exact_f32 else 0


def axial_variant(valu=None):
    """GSDD_AXIAL_VALU=1: the LDS / vector kernel for every axis."""
    if valu is None:
        valu = bool(os.environ.get("GSDD_AXIAL_VALU"))
    return abi.AXIAL_VALU if valu else abi.AXIAL_AUTO


def taps_tensor(taps, device):
    """int32 device table [(dt,dh,dw), ...]"""
    return torch.tensor(taps, dtype=torch.int32, device=device).contiguous()


def gemm(inp, w, out, *, in_dims, out_grid, stride=(1, 1, 1), taps=None, ntaps=1, cin=None, in_pitch=None,
         gather=None, pro=None, ln=None, epi_scale=None, epi_shift=None, bvec=None, rows_per_batch=0, act=ACT_NONE,
         residual=None, out_dims=None, out_step=(1, 1, 1), out_off=(0, 0, 0), out_pitch=None, out_mode=0, cout=None,
         exact_f32=None, stream=None, _desc_only=False):
    """out[orow(m)][n] = epi(sum_tap sum_c pro(in[src(m,tap)][c]) * w[tap][n][c]).

    in_dims = (N, Di, Hi, Wi); out_grid = (Do, Ho, Wo); w: [ntaps][Cout][Cin];
    pro = (scale, shift) per input channel with ReLU; ln = (stats, gamma, beta, sel, stride)."""
    d = GemmDesc()
    N, Di, Hi, Wi = in_dims
    d.in_ = ptr(inp)
    d.N, d.Di, d.Hi, d.Wi = N, Di, Hi, Wi
    d.Cin = cin if cin is not None else w.shape[-1]
    d.in_pitch = in_pitch if in_pitch is not None else d.Cin
    d.Do, d.Ho, d.Wo = out_grid
    d.sd, d.sh, d.sw = stride
    d.ntaps = ntaps
    d.taps = ptr(taps)
    d.gather = ptr(gather)
    d.w = ptr(w)
    d.Cout = cout if cout is not None else w.shape[-2]
    if pro is not None:
        d.pro_scale, d.pro_shift = ptr(pro[0]), ptr(pro[1])
    if ln is not None:
        stats, gamma, beta, sel, ln_stride = ln
        d.ln_stats, d.ln_gamma, d.ln_beta, d.ln_sel, d.ln_stride = ptr(stats), ptr(gamma), ptr(beta), ptr(sel), ln_stride
    d.rows_per_batch = rows_per_batch
    d.epi_scale, d.epi_shift, d.bvec = ptr(epi_scale), ptr(epi_shift), ptr(bvec)
    d.act = act
    d.residual = ptr(residual)
    d.out = ptr(out)
    od = out_dims if out_dims is not None else out_grid
    d.oD, d.oH, d.oW = od
    d.osd, d.osh, d.osw = out_step
    d.ood, d.ooh, d.oow = out_off
    d.out_pitch = out_pitch if out_pitch is not None else d.Cout
    d.out_mode = out_mode
    d.flags = gemm_flags(exact_f32)
    if _desc_only:
        return d
    check(lib().gsdd_gemm(C.byref(d), stream_ptr(stream)))
    return out


def linear(x, w, out, *, bias=None, ln=None, rows_per_batch=0, act=ACT_NONE, residual=None, bvec=None, out_mode=0,
           stream=None):
    """Row GEMM: out[m][:] = act(pro(x[m]) @ w.T + bias [+ bvec[batch]]) [+ residual].  x: [M][Cin], w: [Cout][Cin]."""
    M = x.shape[0]
    return gemm(x, w, out, in_dims=(1, 1, 1, M), out_grid=(1, 1, M), ln=ln, epi_shift=bias, bvec=bvec,
                rows_per_batch=rows_per_batch, act=act, residual=residual, out_mode=out_mode, stream=stream)


def row_stats(x, stats, eps=1e-5, stream=None):
    check(lib().gsdd_row_stats(ptr(x), x.shape[0], x.shape[1], eps, ptr(stats), stream_ptr(stream)))
    return stats


def ncdhw_to_rows(x, cpad, padw, out=None, stream=None):
    N, Cc, D, H, W = x.shape
    if out is None:
        out = torch.empty((N, D, H, W + 2 * padw, cpad), dtype=torch.float32, device=x.device)
    check(lib().gsdd_ncdhw_to_rows(ptr(x), N, Cc, D, H, W, cpad, padw, ptr(out), stream_ptr(stream)))
    return out


def axial_attention(qkv, dims, C_, n_head, out, valu=None, stream=None):
    N, T, H, W = dims
    check(lib().gsdd_axial_attention(ptr(qkv), N, T, H, W, C_, n_head, ptr(out), axial_variant(valu), stream_ptr(stream)))
    return out


def pool3d(x, dims, C_, kernel, stride, pad_front, out_grid, out, *, mode="max", in_pitch=None, out_pitch=None, stream=None):
    """Channels-last 3-D pooling (gsdd_pool3d).  x: rows [N*Di*Hi*Wi][in_pitch] (C_ pooled channels), out: rows of out_pitch floats
    (may be a channel slice view's flat tail of a wider buffer)."""
    N, Di, Hi, Wi = dims
    k, s_, p = (C.c_int * 3)(*kernel), (C.c_int * 3)(*stride), (C.c_int * 3)(*pad_front)
    check(lib().gsdd_pool3d(ptr(x), N, Di, Hi, Wi, C_, in_pitch if in_pitch is not None else C_, k, s_, p, out_grid[0], out_grid[1],
                            out_grid[2], 0 if mode == "max" else 1, ptr(out), out_pitch if out_pitch is not None else C_,
                            stream_ptr(stream)))
    return out


def nearest_code(z, cb, idx, zq=None, stream=None, matrix=True):
    """matrix=False: no workspace -> the register-tiled vector kernel whatever the shape (tests compare the two)."""
    ws = None
    if matrix and not os.environ.get("GSDD_NEAREST_VALU"):
        ws = torch.empty((lib().gsdd_nearest_code_workspace_bytes(cb.shape[0]) // 4,), dtype=torch.float32, device=z.device)
    check(lib().gsdd_nearest_code(ptr(z), z.shape[0], z.shape[1], ptr(cb), cb.shape[0], ptr(idx), ptr(zq), ptr(ws),
                                  0 if ws is None else ws.numel() * 4, stream_ptr(stream)))
    return idx


def bn_train(x, bn, momentum=0.1, update_running=True, want_stats=False, stream=None):
    """Batch statistics of rows x[M][C] -> (scale, shift) for the consuming GEMM's prologue; updates bn's running stats.
    want_stats: also return the per-channel (mean, rstd) pairs the backward needs."""
    M, C_ = x.shape
    n = lib().gsdd_bn_train_workspace_bytes(M, C_)
    ws = torch.empty((n // 8,), dtype=torch.float64, device=x.device)
    scale = torch.empty((C_,), dtype=torch.float32, device=x.device)
    shift = torch.empty_like(scale)
    rm, rv = (bn.running_mean, bn.running_var) if update_running else (None, None)
    mr = torch.empty((C_, 2), dtype=torch.float32, device=x.device) if want_stats else None
    check(lib().gsdd_bn_train(ptr(x), M, C_, ptr(bn.weight.detach()), ptr(bn.bias.detach()), bn.eps, momentum, ptr(rm), ptr(rv),
                              ptr(scale), ptr(shift), ptr(mr), ptr(ws), n, stream_ptr(stream)))
    if update_running:
        bn.num_batches_tracked += 1
    if want_stats:
        return (scale, shift), mr
    return scale, shift


def codebook_ema_stats(z, idx, K, stream=None):
    M, E = z.shape
    n_total = torch.empty((K,), dtype=torch.float32, device=z.device)
    encode_sum = torch.empty((K, E), dtype=torch.float32, device=z.device)
    dummy = torch.empty((2,), dtype=torch.float32, device=z.device)
    check(lib().gsdd_codebook_ema(ptr(z), ptr(idx), M, E, K, 0.99, None, ptr(n_total), ptr(n_total), ptr(n_total),
                                  ptr(n_total), ptr(encode_sum), ptr(dummy), 0, stream_ptr(stream)))
    return n_total, encode_sum


def codebook_ema_update(z, idx, perm, N, z_avg, emb, n_total, encode_sum, decay=0.99, n_local=None, m_local=None, stream=None):
    """z: the restart-candidate rows (tiled / broadcast).  scalars[1] = perplexity of the local one-hot mean: pass this rank's
    own counts `n_local` over its `m_local` latents whenever n_total was all-reduced or z was tiled."""
    M, E = z.shape
    scalars = torch.empty((2,), dtype=torch.float32, device=z.device)
    check(lib().gsdd_codebook_ema(ptr(z), ptr(idx), M, E, emb.shape[0], decay, ptr(perm), ptr(N), ptr(z_avg), ptr(emb),
                                  ptr(n_total), ptr(encode_sum), ptr(scalars), 1, stream_ptr(stream)))
    if n_local is not None:
        check(lib().gsdd_code_perplexity(ptr(n_local), emb.shape[0], m_local, ptr(scalars[1:]), stream_ptr(stream)))
    return scalars


def mse(a, b, scale=1.0, stream=None):
    out = torch.empty((1,), dtype=torch.float32, device=a.device)
    ws = torch.empty((1024,), dtype=torch.float64, device=a.device)
    check(lib().gsdd_mse(ptr(a), ptr(b), a.numel(), scale, ptr(out), ptr(ws), 8192, stream_ptr(stream)))
    return out[0]


def d3pm_embed(tok, emb, pos, x, rep=1, stream=None):
    B, L = tok.shape
    check(lib().gsdd_d3pm_embed(ptr(tok), B, L, emb.shape[1], ptr(emb), emb.shape[0], ptr(pos), rep, ptr(x),
                                stream_ptr(stream)))
    return x


def adaln_table(emb_w, lin_w, lin_b, out=None, stream=None):
    T, D = emb_w.shape
    if out is None:
        out = torch.empty((T, 2 * D), dtype=torch.float32, device=emb_w.device)
    check(lib().gsdd_adaln_table(ptr(emb_w), T, D, ptr(lin_w), ptr(lin_b), ptr(out), stream_ptr(stream)))
    return out


def small_linear(x, w, b, out=None, stream=None):
    R, Cin = x.shape
    if out is None:
        out = torch.empty((R, w.shape[0]), dtype=torch.float32, device=x.device)
    check(lib().gsdd_small_linear(ptr(x), R, Cin, ptr(w), ptr(b), w.shape[0], ptr(out), stream_ptr(stream)))
    return out


def d3pm_attention_workspace(B, L, H, device):
    n = lib().gsdd_d3pm_attention_workspace_bytes(B, L, H)
    return torch.empty(((n + 3) // 4,), dtype=torch.float32, device=device)


def d3pm_attention(q, k, v, B, L, H, out, ws=None, redo=None, mode=None, stream=None):
    """ws: scratch from d3pm_attention_workspace (matrix-pipe kernel); None -> workspace-free exact-f32 P.V kernel.
    redo: optional int64[1] device counter of the kernel's chunk-redo events (caller-owned; see include/gsdd.h).
    mode: 'auto' | '22' | '11' | 'a8' | 'a12' | 'f32pv' | 'kc256' (GSDD_ATTN_* of include/gsdd.h); None -> the environment, else auto."""
    nbytes = 0 if ws is None else ws.numel() * 4
    check(lib().gsdd_d3pm_attention(ptr(q), ptr(k), ptr(v), B, L, H, ptr(out), ptr(ws), nbytes, ptr(redo), attn_mode(mode),
                                    stream_ptr(stream)))
    return out


def d3pm_layer(y, x, L, lay, cvec=None, nxt=None, t2=None, qkv=None, kv_img=None, range_flag=None, variant=None, stream=None):
    """Fused post-attention half of a block (+ the next block's AdaLN/qkv when `nxt` is given).
    y = lay = None: only the next-block stage on x as it is (block 0, whose input is the embedding)."""
    d = LayerDesc()
    d.y, d.x, d.M, d.L, d.n_embd, d.hidden = ptr(y), ptr(x), x.shape[0], L, x.shape[1], 256
    d.cvec = ptr(cvec)
    if lay is not None:
        d.hidden = lay["w1"].shape[0]
        d.wproj, d.bproj, d.ln2_g, d.ln2_b = ptr(lay["wproj"]), ptr(lay["bproj"]), ptr(lay["g2"]), ptr(lay["b2"])
        d.w1, d.b1, d.w2, d.b2 = ptr(lay["w1"]), ptr(lay["bb1"]), ptr(lay["w2"]), ptr(lay["bb2"])
        d.w2_x3 = ptr(lay.get("w2_x3"))
        d.layer_h2 = ptr(lay.get("lay_h2"))
    if nxt is not None:
        d.ada, d.t2, d.wqkv, d.bqkv, d.qkv = ptr(nxt["ada1"]), ptr(t2), ptr(nxt["wqkv"]), ptr(nxt["bqkv"]), ptr(qkv)
        d.wqkv_x3 = ptr(nxt.get("wqkv_x3"))
        d.wqkv_h2 = ptr(nxt.get("wqkv_h2"))
        d.kv_img = ptr(kv_img)
        d.kv_img_bytes = 0 if kv_img is None else kv_img.numel() * kv_img.element_size()
    d.range_flag = ptr(range_flag)
    d.variant = layer_variant(variant)
    check(lib().gsdd_d3pm_layer(C.byref(d), stream_ptr(stream)))


LAYER_X3_BYTES, WQKV_X3_BYTES = 40 * 3 * 1024, 24 * 3 * 1024


def d3pm_layer_pack(w2, wproj, wqkv, stream=None):
    """bf16x3 fragment images for the fused layer kernel: (w2 + wproj of a block, wqkv of a block); valid until the weights change."""
    lay_x3 = torch.empty((LAYER_X3_BYTES,), dtype=torch.uint8, device=w2.device)
    wqkv_x3 = torch.empty((WQKV_X3_BYTES,), dtype=torch.uint8, device=w2.device)
    check(lib().gsdd_d3pm_layer_pack(ptr(w2), ptr(wproj), ptr(wqkv), ptr(lay_x3), ptr(wqkv_x3), stream_ptr(stream)))
    return lay_x3, wqkv_x3


LAYER_H2_BYTES, WQKV_H2_BYTES = 72 * 2 * 1024, 24 * 2 * 1024


def d3pm_layer_pack_h2(w1, w2, wproj, wqkv, stream=None):
    """f16 hi + lo fragment images for the fused layer kernel: (w1 + w2 + wproj of a block, wqkv of a block)."""
    lay_h2 = torch.empty((LAYER_H2_BYTES,), dtype=torch.uint8, device=w2.device)
    wqkv_h2 = torch.empty((WQKV_H2_BYTES,), dtype=torch.uint8, device=w2.device)
    check(lib().gsdd_d3pm_layer_pack_h2(ptr(w1), ptr(w2), ptr(wproj), ptr(wqkv), ptr(lay_h2), ptr(wqkv_h2), stream_ptr(stream)))
    return lay_h2, wqkv_h2


def rows_linear_image_bytes(n_out, n_in):
    return (n_out // 64) * (n_in // 64) * 8 * 3 * 1024


def rows_linear_pack_many(table, n_desc, max_out, max_in, stream=None):
    """table: device bytes holding n_desc descriptors {w ptr, n_out, n_in, ld, transpose, image ptr} (include/gsdd.h)."""
    check(lib().gsdd_rows_linear_pack_many(ptr(table), n_desc, max_out, max_in, stream_ptr(stream)))


def rows_linear(x, image, n_out, out, *, bias=None, bvec=None, rows_per_batch=0, residual=None, head_major=False, stream=None):
    """out[m][:] = x[m] W'^T + bias [+ bvec[batch]] [+ residual[m]] with W' as a gsdd_rows_linear_pack_many image (training step)."""
    check(lib().gsdd_rows_linear(ptr(x), x.shape[0], x.shape[1], ptr(image), n_out, ptr(bias), ptr(bvec), rows_per_batch,
                                 ptr(residual), ptr(out), int(head_major), stream_ptr(stream)))
    return out


def d3pm_logits(x, g, b, w, bias, out, stream=None):
    check(lib().gsdd_d3pm_logits(ptr(x), x.shape[0], x.shape[1], ptr(g), ptr(b), ptr(w), ptr(bias), w.shape[0], ptr(out),
                                 stream_ptr(stream)))
    return out


def d3pm_cross_attention(q, kc, vc, B, L, Te, H, out, stream=None):
    check(lib().gsdd_d3pm_cross_attention(ptr(q), ptr(kc), ptr(vc), B, L, Te, H, ptr(out), stream_ptr(stream)))
    return out


def d3pm_step(logits_c, logits_u, tok_in, tok_out, sched, t_dev, stream_dev, *, K, T, guidance, seed, row0=0,
              post_dbg=None, x0_dbg=None, stream=None):
    B, L = tok_in.shape
    d = StepDesc()
    d.occupancy = int(os.environ.get("GSDD_STEP_OCC", "0"))
    d.logits_c, d.logits_u = ptr(logits_c), ptr(logits_u)
    d.tok_in, d.tok_out = ptr(tok_in), ptr(tok_out)
    d.B, d.L, d.K, d.T = B, L, K, T
    d.guidance = guidance
    for i in range(8):
        d.sched[i] = ptr(sched[i])
    d.t_dev, d.seed, d.stream_dev, d.row0 = ptr(t_dev), seed, ptr(stream_dev), row0
    d.post_dbg, d.x0_dbg = ptr(post_dbg), ptr(x0_dbg)
    check(lib().gsdd_d3pm_step(C.byref(d), stream_ptr(stream)))
    return tok_out


def d3pm_q_sample(x0, xt, sched, t_dev, stream_dev, *, K, T, seed, row0=0, stream=None):
    B, L = x0.shape
    arr = (C.c_void_p * 8)(*[ptr(s) for s in sched])
    check(lib().gsdd_d3pm_q_sample(ptr(x0), ptr(xt), B, L, K, T, arr, ptr(t_dev), seed, ptr(stream_dev), row0,
                                   stream_ptr(stream)))
    return xt


def d3pm_train_loss(logits, x0, xt, t_dev, pt, sched, Lt_history, Lt_count, *, K, T, mask_weight, aux_weight, adaptive_aux,
                    want_probs=True, stream=None):
    """-> dict(loss [1], per_sample [B][4], x0_recon (B,L), xt1_recon (B,L), probs (B,K+1,L) or None)"""
    B, L = x0.shape
    dev = x0.device
    f = dict(dtype=torch.float32, device=dev)
    out = {"loss": torch.empty((1,), **f), "per_sample": torch.empty((B, 4), **f),
           "x0_recon": torch.empty((B, L), dtype=torch.int64, device=dev),
           "xt1_recon": torch.empty((B, L), dtype=torch.int64, device=dev),
           "probs": torch.empty((B, K + 1, L), **f) if want_probs else None}
    scratch = torch.empty((3, B * L), **f)
    d = TrainDesc()
    d.logits, d.x0, d.xt, d.t_dev, d.pt = ptr(logits), ptr(x0), ptr(xt), ptr(t_dev), ptr(pt)
    d.B, d.L, d.K, d.T = B, L, K, T
    for i in range(8):
        d.sched[i] = ptr(sched[i])
    d.mask_weight[0], d.mask_weight[1] = float(mask_weight[0]), float(mask_weight[1])
    d.aux_weight, d.adaptive_aux = float(aux_weight), int(bool(adaptive_aux))
    d.kl, d.nll, d.aux = ptr(scratch[0]), ptr(scratch[1]), ptr(scratch[2])
    d.x0_recon, d.xt1_recon = ptr(out["x0_recon"]), ptr(out["xt1_recon"])
    d.Lt_history, d.Lt_count = ptr(Lt_history), ptr(Lt_count)
    d.loss, d.per_sample, d.probs = ptr(out["loss"]), ptr(out["per_sample"]), ptr(out["probs"])
    check(lib().gsdd_d3pm_train_loss(C.byref(d), stream_ptr(stream)))
    return out


def d3pm_train_loss_grad(logits, x0, xt, t_dev, pt, sched, Lt_history, Lt_count, *, K, T, mask_weight, aux_weight, adaptive_aux,
                         stream=None):
    """d3pm_train_loss (without probs) and d3pm_train_loss_bwd in one pass over the logits -> (forward dict, dlogits)."""
    B, L = x0.shape
    dev = x0.device
    f = dict(dtype=torch.float32, device=dev)
    out = {"loss": torch.empty((1,), **f), "per_sample": torch.empty((B, 4), **f),
           "x0_recon": torch.empty((B, L), dtype=torch.int64, device=dev),
           "xt1_recon": torch.empty((B, L), dtype=torch.int64, device=dev), "probs": None}
    scratch = torch.empty((3, B * L), **f)
    d = TrainDesc()
    d.logits, d.x0, d.xt, d.t_dev, d.pt = ptr(logits), ptr(x0), ptr(xt), ptr(t_dev), ptr(pt)
    d.B, d.L, d.K, d.T = B, L, K, T
    for i in range(8):
        d.sched[i] = ptr(sched[i])
    d.mask_weight[0], d.mask_weight[1] = float(mask_weight[0]), float(mask_weight[1])
    d.aux_weight, d.adaptive_aux = float(aux_weight), int(bool(adaptive_aux))
    d.kl, d.nll, d.aux = ptr(scratch[0]), ptr(scratch[1]), ptr(scratch[2])
    d.x0_recon, d.xt1_recon = ptr(out["x0_recon"]), ptr(out["xt1_recon"])
    d.Lt_history, d.Lt_count = ptr(Lt_history), ptr(Lt_count)
    d.loss, d.per_sample, d.probs = ptr(out["loss"]), ptr(out["per_sample"]), None
    dlogits = torch.empty_like(logits)
    check(lib().gsdd_d3pm_train_loss_grad(C.byref(d), ptr(dlogits), stream_ptr(stream)))
    return out, dlogits


def advance(t_dev, dt, stream_dev, ds, stream=None):
    B = 0 if t_dev is None else t_dev.numel()
    check(lib().gsdd_advance(ptr(t_dev), B, dt, ptr(stream_dev), ds, stream_ptr(stream)))


def philox_uniform(seed, stream_id, n_rows, n_cols, device, row0=0):
    out = torch.empty((n_rows, n_cols), dtype=torch.float32, device=device)
    check(lib().gsdd_philox_uniform(seed, stream_id, row0, n_rows, n_cols, ptr(out), stream_ptr()))
    return out


class Graph:
    """A captured hipGraph of one reverse step (gsdd_graph_* in include/gsdd.h)."""

    def __init__(self):
        self.handle = C.c_void_p()

    def begin(self, stream):
        check(lib().gsdd_graph_begin(stream_ptr(stream)))

    def end(self, stream):
        check(lib().gsdd_graph_end(stream_ptr(stream), C.byref(self.handle)))

    def launch(self, stream):
        check(lib().gsdd_graph_launch(self.handle, stream_ptr(stream)))

    def __del__(self):
        try:
            if self.handle:
                lib().gsdd_graph_destroy(self.handle)
        except Exception:
            pass


class Event:
    """hipEvent recorded on an explicit stream (torch.cuda.Event only sees torch's current stream)."""

    def __init__(self):
        self.h = C.c_void_p()
        check(lib().gsdd_event_create(C.byref(self.h)))

    def record(self, stream=None):
        check(lib().gsdd_event_record(self.h, stream_ptr(stream)))

    def elapsed_ms(self, stop):
        ms = C.c_float()
        check(lib().gsdd_event_elapsed_ms(self.h, stop.h, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            lib().gsdd_event_destroy(self.h)
        except Exception:
            pass


# ----------------------------------------------------------------------------- training-step building blocks
def d3pm_train_loss_bwd(logits, x0, xt, t_dev, pt, sched, *, K, T, mask_weight, aux_weight, adaptive_aux, stream=None):
    B, L = x0.shape
    d = TrainDesc()
    d.logits, d.x0, d.xt, d.t_dev, d.pt = ptr(logits), ptr(x0), ptr(xt), ptr(t_dev), ptr(pt)
    d.B, d.L, d.K, d.T = B, L, K, T
    for i in range(8):
        d.sched[i] = ptr(sched[i])
    d.mask_weight[0], d.mask_weight[1] = float(mask_weight[0]), float(mask_weight[1])
    d.aux_weight, d.adaptive_aux = float(aux_weight), int(bool(adaptive_aux))
    dlogits = torch.empty_like(logits)
    check(lib().gsdd_d3pm_train_loss_bwd(C.byref(d), ptr(dlogits), stream_ptr(stream)))
    return dlogits


def gelu2(a, du=None, stream=None):
    out = torch.empty_like(a)
    check(lib().gsdd_gelu2(ptr(a), ptr(du), ptr(out), a.numel(), 0 if du is None else 1, stream_ptr(stream)))
    return out


def ln_fwd(x, gamma, beta, *, sel=None, gstride=0, rows_per_batch=1, eps=1e-5, stream=None):
    """-> (stats [M][2], y [M][64]): LayerNorm statistics and the normalised, scaled rows in one pass."""
    stats = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    check(lib().gsdd_ln_fwd(ptr(x), x.shape[0], x.shape[1], eps, ptr(gamma), ptr(beta), ptr(sel), gstride, rows_per_batch,
                            ptr(stats), ptr(y), stream_ptr(stream)))
    return stats, y


def ln_bwd(dh, x, stats, gamma, *, sel=None, gstride=0, rows_per_batch=1, dx_in=None, dgamma=None, dbeta=None,
           gacc_stride=0, acc_by_batch=False, stream=None):
    dx = torch.empty_like(x)
    check(lib().gsdd_ln_bwd(ptr(dh), ptr(x), ptr(stats), ptr(gamma), ptr(sel), gstride, rows_per_batch, x.shape[0], x.shape[1],
                            ptr(dx_in), ptr(dx), ptr(dgamma), ptr(dbeta), gacc_stride, int(acc_by_batch), stream_ptr(stream)))
    return dx


def wgrad(dY, X, dW, db=None, stream=None):
    """dW[N][K] += dY^T X ; db[N] += colsum(dY)"""
    check(lib().gsdd_wgrad(ptr(dY), dY.shape[1], ptr(X), X.shape[1], dY.shape[0], dY.shape[1], X.shape[1], ptr(dW), ptr(db),
                           stream_ptr(stream)))


def colsum(Y, out, stream=None):
    check(lib().gsdd_colsum(ptr(Y), Y.shape[1], Y.shape[0], Y.shape[1], ptr(out), stream_ptr(stream)))


def batch_rowsum(Y, B, L, out=None, stream=None):
    if out is None:
        out = torch.empty((B, Y.shape[1]), dtype=torch.float32, device=Y.device)
    check(lib().gsdd_batch_rowsum(ptr(Y), B, L, Y.shape[1], ptr(out), stream_ptr(stream)))
    return out


def d3pm_attention_train(q, k, v, B, L, H, out, lse, ws=None, mode=None, stream=None):
    check(lib().gsdd_d3pm_attention_train(ptr(q), ptr(k), ptr(v), B, L, H, ptr(out), ptr(lse), ptr(ws),
                                          0 if ws is None else ws.numel() * ws.element_size(), attn_train_mode(mode), stream_ptr(stream)))


def d3pm_attention_bwd_workspace(B, L, H, device):
    n = lib().gsdd_d3pm_attention_bwd_workspace_bytes(B, L, H)
    return torch.empty((n // 4,), dtype=torch.float32, device=device)


def d3pm_attention_bwd(q, k, v, o, dO, lse, B, L, H, ws=None, variant=None, stream=None):
    dqkv = torch.empty((B * L, 3 * H * 4), dtype=torch.float32, device=q.device)
    scratch = torch.empty((H * B * L,), dtype=torch.float32, device=q.device)
    check(lib().gsdd_d3pm_attention_bwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(dO), ptr(lse), B, L, H, ptr(dqkv), ptr(scratch), ptr(ws),
                                        0 if ws is None else ws.numel() * ws.element_size(), attn_bwd_variant(variant), stream_ptr(stream)))
    return dqkv


def d3pm_embed_bwd(dx, tok, demb, dpos, stream=None):
    B, L = tok.shape
    check(lib().gsdd_d3pm_embed_bwd(ptr(dx), ptr(tok), B, L, dx.shape[1], demb.shape[0], ptr(demb), ptr(dpos), stream_ptr(stream)))


def small_linear_bwd(dy, x, w, dw, db=None, want_dx=True, stream=None):
    R, Cin = x.shape
    dx = torch.empty_like(x) if want_dx else None
    check(lib().gsdd_small_linear_bwd(ptr(dy), ptr(x), ptr(w), R, Cin, w.shape[0], ptr(dx), ptr(dw), ptr(db), stream_ptr(stream)))
    return dx


def adaln_bwd(dtab, t, emb, w, demb, dw, db, stream=None):
    check(lib().gsdd_adaln_bwd(ptr(dtab), ptr(t), dtab.shape[0], emb.shape[1], ptr(emb), ptr(w), ptr(demb), ptr(dw), ptr(db),
                               stream_ptr(stream)))


def adam(p, g, m, v, lr, beta1, beta2, eps, step, stream=None):
    check(lib().gsdd_adam(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, step, stream_ptr(stream)))


# ----------------------------------------------------------------------------- VQ-VAE training-step building blocks
def conv_wgrad(x, dY, dW, *, in_dims, out_grid, stride=(1, 1, 1), taps=None, ntaps=1, cin, cout, in_pitch=None, pro=None,
               out_dims=None, out_step=(1, 1, 1), out_off=(0, 0, 0), stream=None):
    """dW[tap][cout][cin] += sum_m dY[orow(m)] (x) pro(x[src(m,tap)])  (same geometry arguments as the forward gemm)."""
    d = gemm(x, dW, dW, in_dims=in_dims, out_grid=out_grid, stride=stride, taps=taps, ntaps=ntaps, cin=cin, in_pitch=in_pitch,
             pro=pro, out_dims=out_dims, out_step=out_step, out_off=out_off, cout=cout, _desc_only=True)
    check(lib().gsdd_conv_wgrad(C.byref(d), ptr(dY), dY.shape[-1], ptr(dW), stream_ptr(stream)))


def bn_relu_bwd(da, x, mean_rstd, bn, dgamma, dbeta, dx_in=None, stream=None):
    M, C_ = x.shape
    n = lib().gsdd_bn_relu_bwd_workspace_bytes(M, C_)
    ws = torch.empty(((n + 7) // 8,), dtype=torch.float64, device=x.device)
    dx = torch.empty_like(x)
    check(lib().gsdd_bn_relu_bwd(ptr(da), ptr(x), M, C_, ptr(mean_rstd), ptr(bn.weight.detach()), ptr(bn.bias.detach()),
                                 ptr(dx_in), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws), n, stream_ptr(stream)))
    return dx


def relu_mask(dout, out, stream=None):
    dpre = torch.empty_like(dout)
    check(lib().gsdd_relu_mask(ptr(dout), ptr(out), ptr(dpre), dout.numel(), stream_ptr(stream)))
    return dpre


def lincomb(a, b, c, alpha, stream=None):
    out = torch.empty_like(b)
    check(lib().gsdd_lincomb(ptr(a), ptr(b), ptr(c), alpha, ptr(out), b.numel(), stream_ptr(stream)))
    return out


def axial_attention_bwd(qkv, datt, dims, C_, n_head, valu=None, stream=None):
    N, T, H, W = dims
    dqkv = torch.empty_like(qkv)
    check(lib().gsdd_axial_attention_bwd(ptr(qkv), ptr(datt), N, T, H, W, C_, n_head, ptr(dqkv), axial_variant(valu), stream_ptr(stream)))
    return dqkv
