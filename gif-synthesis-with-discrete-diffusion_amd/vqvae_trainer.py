"""VQ-VAE training step on the HIP path: train-mode forward with saved activations, analytic backward through the
3-D conv / transposed-conv stacks, BatchNorm(train)+ReLU, axial attention and the 1x1 convs, Adam, data-parallel
gradient all-reduce.  Correctness-first; parity-tested against torch.autograd of the CPU oracle.

Reference: TextMotionModel.allsplit_step (src/models/text_motion_model.py:93-144: loss = mean(commitment + recon),
automatic optimisation with Adam(lr 4e-4, betas (0.5, 0.999))) around VQVAE.forward (videogpt_vq_vae.py:58-72) with
the codebook in EMA mode (no gradient reaches the codebook; the straight-through estimator passes d(emb) to z, :216)."""
import torch
import torch.distributed as dist

from . import ops
from ._optim import MultiAdam
from ._lib import GsddError
from .parallel import GradReducer, broadcast_module, world_size
from .vqvae import conv_taps, convT_phases, pack_conv0_weight


def conv_dgrad_phases(kernel, stride, pad_front):
    """Data gradient of a strided SamePadConv3d as sub-pixel phases over the INPUT grid.
    forward: out[o] += W[k] in[o*s + k - pf].  For input index i = i'*s + p the contributing taps have (p - (k - pf)) % s == 0
    and read dOut[i' + (p - (k - pf)) // s].  -> [(phase, [(kt,kh,kw)], [(dt,dh,dw)])]"""
    per_dim = []
    for k, s, pf in zip(kernel, stride, pad_front):
        per_dim.append([[(kk, (p - (kk - pf)) // s) for kk in range(k) if (p - (kk - pf)) % s == 0] for p in range(s)])
    out = []
    for pt, tt in enumerate(per_dim[0]):
        for ph, th in enumerate(per_dim[1]):
            for pw, tw in enumerate(per_dim[2]):
                out.append(((pt, ph, pw), [(a[0], b[0], c[0]) for a in tt for b in th for c in tw],
                            [(a[1], b[1], c[1]) for a in tt for b in th for c in tw]))
    return out


class VQVAETrainer:
    def __init__(self, vq, lr=4e-4, betas=(0.5, 0.999), eps=1e-8):
        self.vq, self.lr, self.betas, self.eps = vq, lr, betas, eps
        self.step_count = 0
        self.state = {}
        self.reducer = GradReducer()
        self._synced = False

    def _sync_start(self):
        """Once, before the first data-parallel step: every rank takes rank 0's parameters and buffers (DDP's start-up broadcast)."""
        if not self._synced and world_size() > 1:
            broadcast_module(self.vq)
        self._synced = True

    # ================================================================== forward with saved activations
    def _res_stack_fwd(self, h, dims, rp, stack):
        N, T, H, W = dims
        M, C_ = h.shape
        f = dict(dtype=torch.float32, device=h.device)
        saves = []
        for i, blk in enumerate(rp["blocks"]):
            mods = stack[i].block
            s = {"r_in": h}
            pro0, s["mr0"] = ops.bn_train(h, mods[0], want_stats=True)
            s["pro0"] = pro0
            s["a"] = ops.gemm(h, blk["w3"], torch.empty((M, C_ // 2), **f), in_dims=dims, out_grid=(T, H, W), taps=blk["taps3"],
                              ntaps=27, pro=pro0)
            pro3, s["mr3"] = ops.bn_train(s["a"], mods[3], want_stats=True)
            s["pro3"] = pro3
            s["b"] = ops.gemm(s["a"], blk["w1"], torch.empty((M, C_), **f), in_dims=dims, out_grid=(T, H, W), pro=pro3)
            pro6, s["mr6"] = ops.bn_train(s["b"], mods[6], want_stats=True)
            s["pro6"] = pro6
            s["qkv"] = ops.gemm(s["b"], blk["wqkv"], torch.empty((M, 9 * C_), **f), in_dims=dims, out_grid=(T, H, W), pro=pro6)
            s["att"] = ops.axial_attention(s["qkv"], dims, C_, 2, torch.empty((M, 3 * C_), **f))
            h = ops.gemm(s["att"], blk["wfc"], torch.empty((M, C_), **f), in_dims=dims, out_grid=(T, H, W), epi_shift=blk["bfc"],
                         residual=h)
            saves.append(s)
        pro, mr = ops.bn_train(h, stack[self.vq.n_res_layers], want_stats=True)
        return h, pro, mr, saves

    def _forward(self, x):
        vq = self.vq
        p = vq.packed()
        cb = vq.codebook
        f = dict(dtype=torch.float32, device=x.device)
        B, _, T, H, W = x.shape
        sv = {"x": x, "enc": []}
        dims, h = None, None
        for c in p["enc_convs"]:
            s = c["stride"]
            if c["first"]:
                xr = ops.ncdhw_to_rows(x, 4, c["padw"])
                in_dims = (B, T, H, W + 2 * c["padw"])
                To, Ho, Wo = T // s[0], H // s[1], W // s[2]
                out = ops.gemm(xr, c["w"], torch.empty((B * To * Ho * Wo, vq.n_hiddens), **f), in_dims=in_dims,
                               out_grid=(To, Ho, Wo), stride=s, taps=c["taps"], ntaps=c["w"].shape[0], cin=c["w"].shape[2],
                               in_pitch=4, epi_shift=c["bias"], act=ops.ACT_RELU)
                sv["enc"].append(dict(inp=xr, in_dims=in_dims, out=out, out_grid=(To, Ho, Wo)))
            else:
                Ti, Hi, Wi = dims[1:]
                To, Ho, Wo = Ti // s[0], Hi // s[1], Wi // s[2]
                out = ops.gemm(h, c["w"], torch.empty((B * To * Ho * Wo, vq.n_hiddens), **f), in_dims=dims, out_grid=(To, Ho, Wo),
                               stride=s, taps=c["taps"], ntaps=c["w"].shape[0], epi_shift=c["bias"], act=ops.ACT_RELU)
                sv["enc"].append(dict(inp=h, in_dims=dims, out=out, out_grid=(To, Ho, Wo)))
            h, dims = out, (B, To, Ho, Wo)
        cl = p["enc_last"]
        sv["last_in"] = h
        hl = ops.gemm(h, cl["w"], torch.empty_like(h), in_dims=dims, out_grid=dims[1:], taps=cl["taps"], ntaps=27, epi_shift=cl["bias"])
        r, pro, mr, sv["enc_res"] = self._res_stack_fwd(hl, dims, p["enc_res"], vq.encoder.res_stack)
        sv["enc_r"], sv["enc_pro"], sv["enc_mr"], sv["dims"] = r, pro, mr, dims
        z = ops.gemm(r, p["pre_w"], torch.empty((r.shape[0], vq.embedding_dim), **f), in_dims=dims, out_grid=dims[1:], pro=pro,
                     epi_shift=p["pre_b"])
        if cb._need_init:
            cb._need_init = False
            rows, perm = vq._draw_rows(z)
            k_rand = rows[perm]
            cb.embeddings.data.copy_(k_rand); cb.z_avg.data.copy_(k_rand); cb.N.data.fill_(1.0)
        idx = torch.empty((z.shape[0],), dtype=torch.int64, device=z.device)
        zq = torch.empty_like(z)
        ops.nearest_code(z, cb.embeddings.contiguous(), idx, zq)
        commitment = ops.mse(z, zq, 0.25)
        n_local, encode_sum = ops.codebook_ema_stats(z, idx, vq.n_codes)
        n_total = n_local
        if dist.is_available() and dist.is_initialized():
            n_total = n_local.clone()
            dist.all_reduce(n_total); dist.all_reduce(encode_sum)
        rows, perm = vq._draw_rows(z)
        scal = ops.codebook_ema_update(rows, idx, perm, cb.N, cb.z_avg, cb.embeddings, n_total, encode_sum,
                                       n_local=n_local, m_local=z.shape[0])
        self.last_perplexity = scal[1]                      # of this rank's latents (videogpt_vq_vae.py:218-219)
        emb_st = ((zq - z) + z).contiguous()
        sv["z"], sv["zq"], sv["emb_st"], sv["idx"] = z, zq, emb_st, idx
        hp = ops.gemm(emb_st, p["post_w"], torch.empty((z.shape[0], vq.n_hiddens), **f), in_dims=dims, out_grid=dims[1:],
                      epi_shift=p["post_b"])
        r, pro, mr, sv["dec_res"] = self._res_stack_fwd(hp, dims, p["dec_res"], vq.decoder.res_stack)
        sv["dec_r"], sv["dec_pro"], sv["dec_mr"] = r, pro, mr
        sv["dec"] = []
        h = r
        Bc, Tc, Hc, Wc = dims
        n_up = len(p["dec_convts"])
        for i, ct in enumerate(p["dec_convts"]):
            s = ct["stride"]
            To, Ho, Wo = Tc * s[0], Hc * s[1], Wc * s[2]
            last = i == n_up - 1
            out = torch.empty((B, ct["cout"], To, Ho, Wo), **f) if last else torch.empty((B * To * Ho * Wo, ct["cout"]), **f)
            for ph in ct["phases"]:
                ops.gemm(h, ph["w"], out, in_dims=(B, Tc, Hc, Wc), out_grid=(Tc, Hc, Wc), taps=ph["taps"], ntaps=ph["w"].shape[0],
                         pro=pro, epi_shift=ct["bias"], act=ops.ACT_NONE if last else ops.ACT_RELU, out_dims=(To, Ho, Wo),
                         out_step=s, out_off=ph["phase"], out_mode=1 if last else 0)
            sv["dec"].append(dict(inp=h, in_dims=(B, Tc, Hc, Wc), out=out, fine=(To, Ho, Wo), pro=pro))
            h, pro = out, None
            Tc, Hc, Wc = To, Ho, Wo
        sv["x_recon"] = h
        recon = ops.mse(h, x, 1.0 / 0.06)
        vq._packed = None
        return sv, {"recon_loss": recon, "commitment_loss": commitment}

    # ================================================================== backward
    def _res_stack_bwd(self, dr, dims, rp, stack, saves, g, prefix):
        """dr: gradient w.r.t. the stack's last residual output r (before the trailing BN).  -> gradient w.r.t. its input."""
        N, T, H, W = dims
        C_ = self.vq.n_hiddens
        f = dict(dtype=torch.float32, device=dr.device)
        M = dr.shape[0]

        def z(name, like):
            g[name] = torch.zeros_like(like)
            return g[name]

        for i in reversed(range(len(rp["blocks"]))):
            blk, s, mods = rp["blocks"][i], saves[i], stack[i].block
            pre = f"{prefix}{i}.block."
            # r_out = att Wfc^T + bfc + r_in
            dwfc = torch.zeros((C_, 3 * C_), **f)
            dbfc = torch.zeros((C_,), **f)
            ops.wgrad(dr, s["att"], dwfc, dbfc)
            datt = ops.gemm(dr, blk["wfc"][0].t().contiguous().view(1, 3 * C_, C_), torch.empty((M, 3 * C_), **f), in_dims=dims,
                            out_grid=(T, H, W))
            dqkv = ops.axial_attention_bwd(s["qkv"], datt, dims, C_, 2)
            dwqkv = torch.zeros((1, 9 * C_, C_), **f)
            ops.conv_wgrad(s["b"], dqkv, dwqkv, in_dims=dims, out_grid=(T, H, W), cin=C_, cout=9 * C_, pro=s["pro6"])
            for ai, ax in enumerate(("attn_w", "attn_h", "attn_t")):
                for wi, wn in enumerate(("w_qs", "w_ks", "w_vs")):
                    g[f"{pre}8.{ax}.{wn}.weight"] = dwqkv[0, (3 * ai + wi) * C_:(3 * ai + wi + 1) * C_].contiguous()
                g[f"{pre}8.{ax}.fc.weight"] = dwfc[:, ai * C_:(ai + 1) * C_].contiguous()
                g[f"{pre}8.{ax}.fc.bias"] = dbfc.clone()
            dbact = ops.gemm(dqkv, blk["wqkv"][0].t().contiguous().view(1, C_, 9 * C_), torch.empty((M, C_), **f), in_dims=dims,
                             out_grid=(T, H, W))
            db_raw = ops.bn_relu_bwd(dbact, s["b"], s["mr6"], mods[6], z(pre + "6.weight", mods[6].weight), z(pre + "6.bias", mods[6].bias))
            # conv1 (1x1, C/2 -> C)
            dw1 = torch.zeros((1, C_, C_ // 2), **f)
            ops.conv_wgrad(s["a"], db_raw, dw1, in_dims=dims, out_grid=(T, H, W), cin=C_ // 2, cout=C_, pro=s["pro3"])
            g[pre + "5.conv.weight"] = dw1[0].view(C_, C_ // 2, 1, 1, 1).contiguous()
            daact = ops.gemm(db_raw, blk["w1"][0].t().contiguous().view(1, C_ // 2, C_), torch.empty((M, C_ // 2), **f), in_dims=dims,
                             out_grid=(T, H, W))
            da_raw = ops.bn_relu_bwd(daact, s["a"], s["mr3"], mods[3], z(pre + "3.weight", mods[3].weight), z(pre + "3.bias", mods[3].bias))
            # conv3 (3x3x3, C -> C/2)
            dw3 = torch.zeros((27, C_ // 2, C_), **f)
            ops.conv_wgrad(s["r_in"], da_raw, dw3, in_dims=dims, out_grid=(T, H, W), taps=blk["taps3"], ntaps=27, cin=C_,
                           cout=C_ // 2, pro=s["pro0"])
            g[pre + "2.conv.weight"] = dw3.view(3, 3, 3, C_ // 2, C_).permute(3, 4, 0, 1, 2).contiguous()
            neg_taps = ops.taps_tensor([(-a, -b, -c) for (a, b, c) in conv_taps((3, 3, 3), (1, 1, 1), (1, 1, 1))], dr.device)
            dract = ops.gemm(da_raw, blk["w3"].transpose(1, 2).contiguous(), torch.empty((M, C_), **f), in_dims=dims,
                             out_grid=(T, H, W), taps=neg_taps, ntaps=27)
            dr = ops.bn_relu_bwd(dract, s["r_in"], s["mr0"], mods[0], z(pre + "0.weight", mods[0].weight), z(pre + "0.bias", mods[0].bias),
                                 dx_in=dr)
        return dr

    @torch.no_grad()
    def forward(self, x):
        """Train-mode forward (BatchNorm batch statistics, codebook EMA) -> (saved activations, losses)."""
        vq = self.vq
        if not x.is_cuda:
            raise GsddError("the HIP path needs tensors on a ROCm device (no CPU fallback)")
        if not vq.training:
            raise GsddError("VQVAETrainer needs the module in train() mode (BatchNorm batch statistics, codebook EMA)")
        self._sync_start()
        x = x.contiguous().float()
        p = vq.packed()                     # weights of THIS step (the forward invalidates the cache at its end)
        sv, losses = self._forward(x)
        sv["p"] = p
        return sv, losses

    def loss_and_grads(self, x, w_recon=1.0, w_commit=1.0):
        sv, losses = self.forward(x)
        return losses, self.backward(sv, w_recon, w_commit)

    @torch.no_grad()
    def backward(self, sv, w_recon=1.0, w_commit=1.0, reduce=False):
        """Gradients of w_recon*recon_loss + w_commit*commitment_loss w.r.t. every parameter, keyed by state_dict name.
        reduce=True: averaged over the data-parallel group, in two buckets -- the decoder half (56 MiB at the full config) is
        all-reduced while the encoder half's backward is still being enqueued, the encoder half at the end."""
        vq, p, x = self.vq, sv["p"], sv["x"]
        dev = x.device
        f = dict(dtype=torch.float32, device=dev)
        g = {}
        red = self.reducer if (reduce and self.reducer.active()) else None
        flushed = set()

        def flush():
            names = [n for n in g if n not in flushed]
            if red is None or not names:
                return
            flat = torch.cat([g[n].reshape(-1) for n in names])
            red.add(flat)
            off = 0
            for n in names:
                k = g[n].numel()
                g[n] = flat[off:off + k].view(g[n].shape)
                off += k
            flushed.update(names)
        C_, E = vq.n_hiddens, vq.embedding_dim
        dims = sv["dims"]
        B = x.shape[0]

        def z(name, like):
            g[name] = torch.zeros_like(like)
            return g[name]

        # ---- d total / d x_recon, total = w_recon * mse(x_recon, x)/0.06 + w_commit * 0.25 mse(z, emb)
        dxr = ops.lincomb(None, sv["x_recon"], x, w_recon * 2.0 / (0.06 * x.numel()))
        n_up = len(p["dec_convts"])
        dfine = None
        # ---- decoder transposed convs, last to first
        for i in reversed(range(n_up)):
            ct, s = p["dec_convts"][i], sv["dec"][i]
            mod = vq.decoder.convts[i]
            stride = ct["stride"]
            last = i == n_up - 1
            Bc, Tc, Hc, Wc = s["in_dims"]
            cin, cout = C_, ct["cout"]
            if last and s["pro"] is None and cout <= 4:
                # Cout = 3: the adjoint of the transposed conv is a strided conv over d(out), dIn[i] = sum_kk W[kk]^T dOut[i*s + kk - c]
                # with c = (k-1) - s*pad_front.  On W-padded NDHWC4 rows the kw taps x 4 channels of one (kt, kh) are 16 contiguous
                # floats, so both gradients run with K = 16 per tap (the same merge as the encoder's first conv) instead of
                # 4-wide / 4-deep tiles padded to the 64 x 32 MFMA tile: data gradient = gsdd_gemm, weight gradient =
                # gsdd_conv_wgrad with the operand roles swapped (its "input" is d(out), its "dY" the layer input).
                k, pf = mod.kernel_size, mod.pad_front
                cc = [kk - 1 - st * pp for kk, st, pp in zip(k, stride, pf)]
                padw = max(cc[2], k[2] - stride[2] - cc[2], 0)
                dYp = ops.ncdhw_to_rows(dxr, 4, padw)                          # (B, Tf, Hf, Wf + 2 padw, 4), zero pads
                Tf, Hf, Wf = s["fine"]
                fine_dims = (Bc, Tf, Hf, Wf + 2 * padw)
                taps_m = ops.taps_tensor([(a - cc[0], b_ - cc[1], padw - cc[2]) for a in range(k[0]) for b_ in range(k[1])], dev)
                db = torch.zeros((4,), **f)
                ops.colsum(dYp.view(-1, 4), db)
                g[f"decoder.convts.{i}.convt.bias"] = db[:cout].contiguous()
                dwm = torch.zeros((k[0] * k[1], cin, k[2] * 4), **f)
                ops.conv_wgrad(dYp, s["inp"], dwm, in_dims=fine_dims, out_grid=(Tc, Hc, Wc), stride=stride, taps=taps_m,
                               ntaps=k[0] * k[1], cin=k[2] * 4, cout=cin, in_pitch=4)
                g[f"decoder.convts.{i}.convt.weight"] = dwm.view(k[0], k[1], cin, k[2], 4).permute(2, 4, 0, 1, 3)[:, :cout].contiguous()
                dact = ops.gemm(dYp, pack_conv0_weight(mod.convt.weight), torch.empty((Bc * Tc * Hc * Wc, cin), **f), in_dims=fine_dims,
                                out_grid=(Tc, Hc, Wc), stride=stride, taps=taps_m, ntaps=k[0] * k[1], cin=k[2] * 4, in_pitch=4)
                dfine = dact                                                  # (i > 0: a single convT has the BatchNorm prologue)
                continue
            if last:
                dfine = ops.ncdhw_to_rows(dxr, 4, 0).view(-1, 4)              # rows of the finest grid, channel padded 3 -> 4
            cpad = 4 if last else cout
            dY = dfine if last else ops.relu_mask(dfine, s["out"])
            db = torch.zeros((cpad,), **f)
            ops.colsum(dY, db)
            g[f"decoder.convts.{i}.convt.bias"] = db[:cout].contiguous()
            dw_full = torch.zeros_like(mod.convt.weight)                     # (Cin, Cout, kt, kh, kw)
            pairs_w, pairs_off = [], []
            for (phase, ks, offs), ph in zip(convT_phases(mod.kernel_size, stride, mod.pad_front), ct["phases"]):
                dwp = torch.zeros((len(ks), cpad, cin), **f)
                ops.conv_wgrad(s["inp"], dY, dwp, in_dims=s["in_dims"], out_grid=(Tc, Hc, Wc), taps=ph["taps"], ntaps=len(ks),
                               cin=cin, cout=cpad, pro=s["pro"], out_dims=s["fine"], out_step=stride, out_off=phase)
                for j, (a, b_, c) in enumerate(ks):
                    dw_full[:, :, a, b_, c] = dwp[j, :cout].t()
                    wt = torch.zeros((cin, cpad), **f)
                    wt[:, :cout] = ph["w"][j].t()
                    pairs_w.append(wt)
                    pairs_off.append(tuple(pp - o * st for pp, o, st in zip(phase, offs[j], stride)))
            g[f"decoder.convts.{i}.convt.weight"] = dw_full
            # data gradient: dIn[i] = sum_{phase,tap} W^T dOut[(i - off)*s + phase]
            wT = torch.stack(pairs_w).contiguous()                           # [pairs][Cin][cpad]
            dact = ops.gemm(dY, wT, torch.empty((Bc * Tc * Hc * Wc, cin), **f), in_dims=(Bc,) + tuple(s["fine"]),
                            out_grid=(Tc, Hc, Wc), stride=stride, taps=ops.taps_tensor(pairs_off, dev), ntaps=len(pairs_off))
            if i > 0:
                dfine = dact                                                  # grad w.r.t. the previous convT's ReLU output
            else:
                dr = ops.bn_relu_bwd(dact, sv["dec_r"], sv["dec_mr"], vq.decoder.res_stack[vq.n_res_layers],
                                     z(f"decoder.res_stack.{vq.n_res_layers}.weight", vq.decoder.res_stack[vq.n_res_layers].weight),
                                     z(f"decoder.res_stack.{vq.n_res_layers}.bias", vq.decoder.res_stack[vq.n_res_layers].bias))
        dhp = self._res_stack_bwd(dr, dims, p["dec_res"], vq.decoder.res_stack, sv["dec_res"], g, "decoder.res_stack.")
        # ---- post_vq_conv (1x1, E -> C) and the straight-through estimator
        dwpost = torch.zeros((C_, E), **f)
        ops.wgrad(dhp, sv["emb_st"], dwpost, z("post_vq_conv.conv.bias", vq.post_vq_conv.conv.bias))
        g["post_vq_conv.conv.weight"] = dwpost.view(C_, E, 1, 1, 1)
        demb = ops.gemm(dhp, p["post_w"][0].t().contiguous().view(1, E, C_), torch.empty((dhp.shape[0], E), **f), in_dims=dims,
                        out_grid=dims[1:])
        dz = ops.lincomb(demb, sv["z"], sv["zq"], w_commit * 0.25 * 2.0 / sv["z"].numel())
        flush()                                                               # decoder half on its way
        # ---- pre_vq_conv (1x1, C -> E) with the encoder stack's trailing BN+ReLU as prologue
        dwpre = torch.zeros((1, E, C_), **f)
        ops.conv_wgrad(sv["enc_r"], dz, dwpre, in_dims=dims, out_grid=dims[1:], cin=C_, cout=E, pro=sv["enc_pro"])
        g["pre_vq_conv.conv.weight"] = dwpre[0].view(E, C_, 1, 1, 1).contiguous()
        ops.colsum(dz, z("pre_vq_conv.conv.bias", vq.pre_vq_conv.conv.bias))
        dact = ops.gemm(dz, p["pre_w"][0].t().contiguous().view(1, C_, E), torch.empty((dz.shape[0], C_), **f), in_dims=dims,
                        out_grid=dims[1:])
        n = vq.n_res_layers
        dr = ops.bn_relu_bwd(dact, sv["enc_r"], sv["enc_mr"], vq.encoder.res_stack[n], z(f"encoder.res_stack.{n}.weight", vq.encoder.res_stack[n].weight),
                             z(f"encoder.res_stack.{n}.bias", vq.encoder.res_stack[n].bias))
        dhl = self._res_stack_bwd(dr, dims, p["enc_res"], vq.encoder.res_stack, sv["enc_res"], g, "encoder.res_stack.")
        # ---- conv_last (3x3x3, bias, no activation)
        cl = p["enc_last"]
        dwl = torch.zeros((27, C_, C_), **f)
        ops.conv_wgrad(sv["last_in"], dhl, dwl, in_dims=dims, out_grid=dims[1:], taps=cl["taps"], ntaps=27, cin=C_, cout=C_)
        g["encoder.conv_last.conv.weight"] = dwl.view(3, 3, 3, C_, C_).permute(3, 4, 0, 1, 2).contiguous()
        ops.colsum(dhl, z("encoder.conv_last.conv.bias", vq.encoder.conv_last.conv.bias))
        neg_taps = ops.taps_tensor([(-a, -b, -c) for (a, b, c) in conv_taps((3, 3, 3), (1, 1, 1), (1, 1, 1))], dev)
        dh = ops.gemm(dhl, cl["w"].transpose(1, 2).contiguous(), torch.empty_like(dhl), in_dims=dims, out_grid=dims[1:], taps=neg_taps,
                      ntaps=27)
        # ---- strided encoder convs, last to first
        for i in reversed(range(len(p["enc_convs"]))):
            c, s = p["enc_convs"][i], sv["enc"][i]
            mod = vq.encoder.convs[i]
            k, stride, pf = mod.kernel_size, c["stride"], mod.pad_front
            dY = ops.relu_mask(dh, s["out"])
            ops.colsum(dY, z(f"encoder.convs.{i}.conv.bias", mod.conv.bias))
            if c["first"]:
                dwp = torch.zeros_like(c["w"])                                # [kt*kh][Cout][kw*4]
                ops.conv_wgrad(s["inp"], dY, dwp, in_dims=s["in_dims"], out_grid=s["out_grid"], stride=stride, taps=c["taps"],
                               ntaps=c["w"].shape[0], cin=c["w"].shape[2], cout=C_, in_pitch=4)
                g[f"encoder.convs.{i}.conv.weight"] = dwp.view(k[0], k[1], C_, k[2], 4).permute(2, 4, 0, 1, 3)[:, :3].contiguous()
            else:
                dwp = torch.zeros_like(c["w"])
                ops.conv_wgrad(s["inp"], dY, dwp, in_dims=s["in_dims"], out_grid=s["out_grid"], stride=stride, taps=c["taps"],
                               ntaps=c["w"].shape[0], cin=C_, cout=C_)
                g[f"encoder.convs.{i}.conv.weight"] = dwp.view(k[0], k[1], k[2], C_, C_).permute(3, 4, 0, 1, 2).contiguous()
                Bc, Ti, Hi, Wi = s["in_dims"]
                dh = torch.empty((Bc * Ti * Hi * Wi, C_), **f)
                wfull = c["w"].view(k[0], k[1], k[2], C_, C_)
                for (phase, ks, offs) in conv_dgrad_phases(k, stride, pf):
                    wT = torch.stack([wfull[a, b_, cc].t() for (a, b_, cc) in ks]).contiguous()     # [taps][Cin][Cout]
                    ops.gemm(dY, wT, dh, in_dims=(Bc,) + tuple(s["out_grid"]), out_grid=(Ti // stride[0], Hi // stride[1], Wi // stride[2]),
                             taps=ops.taps_tensor(offs, dev), ntaps=len(ks), out_dims=(Ti, Hi, Wi), out_step=stride, out_off=phase)
        flush()
        if red is not None:
            red.finish()
        return g

    # ================================================================== optimiser step
    @torch.no_grad()
    def step(self, x):
        sv, losses = self.forward(x)
        grads = self.backward(sv, 1.0, 1.0, reduce=True)
        self.step_count += 1
        if getattr(self, "_adam", None) is None:
            self._adam = MultiAdam(list(self.vq.named_parameters()), self.lr, self.betas, self.eps)
        self._adam.lr = self.lr
        self._adam.step(grads)                  # all parameters in one launch (gsdd_adam_multi)
        self.vq._packed = None
        return losses


class _TrainForward(torch.autograd.Function):
    """Bridges the HIP training step into torch.autograd so that the reference's Lightning automatic optimisation
    (loss.backward(); optimizer.step(), text_motion_model.py:93-144) works unchanged: forward runs the HIP forward and keeps
    the activations, backward runs the HIP backward weighted by the incoming loss gradients and hands each parameter its .grad."""

    @staticmethod
    def forward(ctx, trainer, x, *params):
        sv, losses = trainer.forward(x)
        ctx.trainer, ctx.sv = trainer, sv
        pred = sv["x_recon"]
        ctx.mark_non_differentiable(pred)
        return losses["recon_loss"].clone(), losses["commitment_loss"].clone(), pred

    @staticmethod
    def backward(ctx, g_recon, g_commit, _g_pred):
        wr = 0.0 if g_recon is None else float(g_recon)
        wc = 0.0 if g_commit is None else float(g_commit)
        grads = ctx.trainer.backward(ctx.sv, wr, wc, reduce=True)          # DDP semantics: .grad = mean over the group
        ctx.sv = None
        return (None, None) + tuple(grads[n] for n, _ in ctx.trainer.vq.named_parameters())


def train_forward(vq, x):
    """VQVAE.forward in train mode with autograd enabled -> (recon_loss, commitment_loss, x_recon)."""
    tr = getattr(vq, "_hip_trainer", None)
    if tr is None:
        tr = VQVAETrainer(vq)
        object.__setattr__(vq, "_hip_trainer", tr)
    return _TrainForward.apply(tr, x, *[prm for _, prm in vq.named_parameters()])
