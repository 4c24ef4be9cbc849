"""D3PM training step on the HIP path: forward with saved activations, analytic backward, Adam, data-parallel gradient
all-reduce.  Correctness-first (one launch per operator, f32 MFMA GEMMs, VALU attention backward): parity-tested against
torch.autograd of the CPU oracle; the sampling path does not depend on anything here.

Reference: MultistageTextMotionModel.allsplit_step (src/models/multistage_text_motion_model.py:170-206: zero_grad ->
manual_backward -> Adam(lr 1e-4, betas (0.5, 0.999))) around DiffusionTransformer._train_loss
(src/models/motionencoder/diffusion_transformer.py:391-457); DDP gradient averaging = configs/trainer/default.yaml:8."""
import os

import torch
import torch.distributed as dist

from . import ops
from ._optim import GradArena, MultiAdam
from ._lib import GsddError
from .parallel import GradReducer, broadcast_module, world_size

BUCKET_LAYERS = 5          # blocks per gradient bucket: 19 blocks + head + embeddings -> 5 all-reduces of ~2.7 MB each


class _LinearImages:
    """bf16x3 fragment images (gsdd_rows_linear) of every block's weight matrices and of their transposes (the data-gradient
    operands), all refreshed by ONE launch per optimiser step.  The descriptor table points at the parameters themselves (the
    optimisers update them in place); `refresh` rebuilds it if a parameter has moved."""
    NAMES = ("qkv", "qkv_t", "proj", "proj_t", "w1", "w1_t", "w2", "w2_t")

    def __init__(self, tr):
        self.tr = tr
        self.ptrs = None

    @staticmethod
    def _pieces(blk):
        a, m = blk.attn1, blk.mlp
        q, k, v = a.query.weight, a.key.weight, a.value.weight
        # name -> (n_out, n_in, [(weight, rows of the image matrix it fills, columns, transpose, fragment offset)])
        return {
            "qkv": (192, 64, [(w, 64, 64, 0, 8 * j) for j, w in enumerate((q, k, v))]),
            "qkv_t": (64, 192, [(w, 64, 64, 1, 8 * j) for j, w in enumerate((q, k, v))]),
            "proj": (64, 64, [(a.proj.weight, 64, 64, 0, 0)]),
            "proj_t": (64, 64, [(a.proj.weight, 64, 64, 1, 0)]),
            "w1": (256, 64, [(m[0].weight, 256, 64, 0, 0)]),
            "w1_t": (64, 256, [(m[0].weight, 64, 256, 1, 0)]),
            "w2": (64, 256, [(m[2].weight, 64, 256, 0, 0)]),
            "w2_t": (256, 64, [(m[2].weight, 256, 64, 1, 0)]),
        }

    def _build(self):
        import numpy as np
        dev = self.tr.blocks[0].mlp[0].weight.device
        per_block = sum(ops.rows_linear_image_bytes(no, ni) for no, ni, _ in self._pieces(self.tr.blocks[0]).values())
        self.buf = torch.empty((len(self.tr.blocks) * per_block,), dtype=torch.uint8, device=dev)
        rows, self.images, off = [], [], 0
        for blk in self.tr.blocks:
            imgs = {}
            for name, (no, ni, pieces) in self._pieces(blk).items():
                nbytes = ops.rows_linear_image_bytes(no, ni)
                imgs[name] = self.buf[off:off + nbytes]
                for w, po, pi, tr_, frag in pieces:
                    assert w.is_contiguous() and w.dtype == torch.float32
                    rows.append((w.data_ptr(), po, pi, w.shape[1], tr_, self.buf.data_ptr() + off + frag * 3 * 1024))
                off += nbytes
            self.images.append(imgs)
        table = np.array(rows, dtype=np.dtype([("w", "<u8"), ("n_out", "<i4"), ("n_in", "<i4"), ("ld", "<i4"), ("transpose", "<i4"),
                                               ("img", "<u8")]))
        self.table = torch.from_numpy(table.view(np.uint8).copy()).to(dev)
        self.n_desc = len(rows)
        self.ptrs = [r[0] for r in rows]

    def _current_ptrs(self):
        return [w.data_ptr() for blk in self.tr.blocks for _, _, pieces in self._pieces(blk).values() for w, *_ in pieces]

    def refresh(self):
        if self.ptrs is None or self.ptrs != self._current_ptrs():
            self._build()
        ops.rows_linear_pack_many(self.table, self.n_desc, 256, 256)
        return self.images


class D3PMTrainer:
    def __init__(self, dm, lr=1e-4, betas=(0.5, 0.999), eps=1e-8):
        self.dm, self.lr, self.betas, self.eps = dm, lr, betas, eps
        self.step_count = 0
        self.state = {}
        self.reducer = GradReducer()
        self._synced = False
        self._images = None

    def _sync_start(self):
        """Once, before the first data-parallel step: every rank takes rank 0's parameters and buffers (what DDP does at wrap time)."""
        if not self._synced and world_size() > 1:
            broadcast_module(self.dm)
        self._synced = True

    # ------------------------------------------------------------------ forward with saved activations
    def _forward(self, xt, cond, t):
        dm, tr = self.dm, self.dm.transformer
        p = tr.packed()
        B, L = xt.shape
        D, H = tr.n_embd, tr.n_head
        M = B * L
        dev = xt.device
        f = dict(dtype=torch.float32, device=dev)
        cond = cond.float().contiguous()
        if cond.shape[1] != 1:
            raise NotImplementedError("the training step is built for one condition token (the reference call site)")
        flat = cond.reshape(B, -1).contiguous()
        sv = {"xt": xt, "t": t, "cond": flat, "layers": []}
        x = torch.empty((M, D), **f)
        ops.d3pm_embed(xt, p["emb"], p["pos"], x)
        aws = ops.d3pm_attention_workspace(B, L, H, dev) if L % 32 == 0 else None     # pre-split K/V images (matrix-pipe forward)
        # the block's row GEMMs run on gsdd_rows_linear (weights as fragment images, refreshed once per step) when the block has
        # the north-star widths; other widths take the generic GEMM
        fast = D == 64 and p["layers"] and p["layers"][0]["w1"].shape[0] == 256 and os.environ.get("GSDD_TRAIN_LINEAR") != "gemm"
        if fast:
            if self._images is None:
                self._images = _LinearImages(tr)
            imgs = self._images.refresh()
        sv["imgs"] = imgs if fast else None
        for li_, lay in enumerate(p["layers"]):
            s = {"x_in": x}
            if fast:
                im = imgs[li_]
                s["stats1"], s["hn"] = ops.ln_fwd(x, lay["ada1"].view(-1), lay["ada1"].view(-1)[D:], sel=t, gstride=2 * D,
                                                  rows_per_batch=L)
                s["qkv"] = ops.rows_linear(s["hn"], im["qkv"], 3 * D, torch.empty((3 * H, M, 4), **f), bias=lay["bqkv"], head_major=True)
                s["y"], s["lse"] = torch.empty((M, D), **f), torch.empty((H * M,), **f)
                ops.d3pm_attention_train(s["qkv"][0:H], s["qkv"][H:2 * H], s["qkv"][2 * H:], B, L, H, s["y"], s["lse"], ws=aws)
                s["v2"] = ops.small_linear(flat, lay["wv2"], lay["bv2"])
                cvec = ops.small_linear(s["v2"], lay["wproj2"], lay["bproj2"])
                s["x1"] = ops.rows_linear(s["y"], im["proj"], D, torch.empty((M, D), **f), bias=lay["bproj"], bvec=cvec,
                                          rows_per_batch=L, residual=x)
                s["stats2"], s["h2"] = ops.ln_fwd(s["x1"], lay["g2"], lay["b2"])
                s["a"] = ops.rows_linear(s["h2"], im["w1"], 4 * D, torch.empty((M, 4 * D), **f), bias=lay["bb1"])
                s["u"] = ops.gelu2(s["a"])
                x = ops.rows_linear(s["u"], im["w2"], D, torch.empty((M, D), **f), bias=lay["bb2"], residual=s["x1"])
                sv["layers"].append(s)
                continue
            s["stats1"], s["hn"] = ops.ln_fwd(x, lay["ada1"].view(-1), lay["ada1"].view(-1)[D:], sel=t, gstride=2 * D,
                                              rows_per_batch=L)          # AdaLayerNorm: statistics + normalised rows in one pass
            s["qkv"] = ops.linear(s["hn"], lay["wqkv"], torch.empty((3 * H, M, 4), **f), bias=lay["bqkv"], out_mode=2)
            s["y"], s["lse"] = torch.empty((M, D), **f), torch.empty((H * M,), **f)
            ops.d3pm_attention_train(s["qkv"][0:H], s["qkv"][H:2 * H], s["qkv"][2 * H:], B, L, H, s["y"], s["lse"], ws=aws)
            s["v2"] = ops.small_linear(flat, lay["wv2"], lay["bv2"])
            cvec = ops.small_linear(s["v2"], lay["wproj2"], lay["bproj2"])
            s["x1"] = ops.linear(s["y"], lay["wproj"], torch.empty((M, D), **f), bias=lay["bproj"], bvec=cvec, rows_per_batch=L,
                                 residual=x)
            s["stats2"], s["h2"] = ops.ln_fwd(s["x1"], lay["g2"], lay["b2"])
            s["a"] = ops.linear(s["h2"], lay["w1"], torch.empty((M, lay["w1"].shape[0]), **f), bias=lay["bb1"])
            s["u"] = ops.gelu2(s["a"])
            x = ops.linear(s["u"], lay["w2"], torch.empty((M, D), **f), bias=lay["bb2"], residual=s["x1"])
            sv["layers"].append(s)
        sv["x_out"] = x
        sv["statsf"], sv["hf"] = ops.ln_fwd(x, p["gf"], p["bf"])
        sv["logits"] = ops.linear(sv["hf"], p["wl"], torch.empty((M, p["wl"].shape[0]), **f), bias=p["bl"])
        return sv

    # ------------------------------------------------------------------ loss + gradients
    @torch.no_grad()
    def loss_and_grads(self, x0, cond, t=None, pt=None, want_probs=False, reduce=False, sid=None):
        """-> (loss tensor [1], {state_dict name: gradient}) for the transformer's parameters.  The gradients are views of one
        arena that the next call re-uses; `self.last_fwd` keeps the forward's dict (x0_recon, per_sample, probs if asked).
        reduce=True: the gradients come back averaged over the data-parallel group; the all-reduce runs in buckets of
        BUCKET_LAYERS blocks issued while the backward of the earlier blocks is still being enqueued.
        sid: int64[1] device tensor holding the Philox stream id of this step's q_sample draw (the captured step keeps it on the device);
        None: made from dm.noise_stream."""
        self._sync_start()
        dm, tr = self.dm, self.dm.transformer
        if not x0.is_cuda:
            raise GsddError("the HIP path needs tensors on a ROCm device (no CPU fallback)")
        p = tr.packed()
        B, L = x0.shape
        D, H = tr.n_embd, tr.n_head
        K, T = dm.num_classes - 1, dm.num_timesteps
        dev = x0.device
        f = dict(dtype=torch.float32, device=dev)
        if t is None:
            t, pt = dm.sample_time(B, dev, "importance")
        t, pt = t.to(dev).long().contiguous(), pt.to(dev).float().contiguous()
        if sid is None:
            sid = torch.tensor([dm.noise_stream], dtype=torch.int64, device=dev)
        dm.noise_stream += 1
        sched = dm._sched()
        x0 = x0.contiguous().long()
        xt = torch.empty_like(x0)
        ops.d3pm_q_sample(x0, xt, sched, t, sid, K=K, T=T, seed=dm.noise_seed, row0=dm.row_offset * L)
        sv = self._forward(xt, cond, t)
        kw = dict(K=K, T=T, mask_weight=dm.mask_weight, aux_weight=dm.auxiliary_loss_weight,
                  adaptive_aux=dm.adaptive_auxiliary_loss)
        if want_probs or os.environ.get("GSDD_TRAIN_LOSS_SPLIT"):      # (the switch: A/B and the equality test of the two paths)
            fwd = ops.d3pm_train_loss(sv["logits"], x0, xt, t, pt, sched, dm.Lt_history, dm.Lt_count, want_probs=want_probs, **kw)
            dlogits = ops.d3pm_train_loss_bwd(sv["logits"], x0, xt, t, pt, sched, **kw)
        else:                                                          # one pass over the logits for the loss and its gradient
            fwd, dlogits = ops.d3pm_train_loss_grad(sv["logits"], x0, xt, t, pt, sched, dm.Lt_history, dm.Lt_count, **kw)
        fwd["t"], fwd["xt"] = t, xt
        self.last_fwd = fwd

        g = {}
        if getattr(self, "_arena", None) is None:
            self._arena = GradArena(dev)
        self._arena.reset()                       # every gradient below is a zero-filled view of one buffer: one fill per step
        red = self.reducer if (reduce and self.reducer.active()) else None
        mark = [0]

        def bucket():                             # all-reduce what the arena handed out since the last bucket (in place, async)
            a = self._arena
            if red is not None and a.buf is not None and mark[0] < a.off <= a.buf.numel():
                red.add(a.buf[mark[0]:a.off])
                mark[0] = a.off

        def z(name, like):
            g[name] = self._arena.zeros_like(like)
            return g[name]

        def tw(w):                              # transposed copy for the data-gradient GEMMs
            return w.t().contiguous()

        # ---- to_logits
        ops.wgrad(dlogits, sv["hf"], z("to_logits.1.weight", p["wl"]), z("to_logits.1.bias", p["bl"]))
        dhf = ops.linear(dlogits, tw(p["wl"]), torch.empty((B * L, D), **f))
        dx = ops.ln_bwd(dhf, sv["x_out"], sv["statsf"], p["gf"], dgamma=z("to_logits.0.weight", p["gf"]),
                        dbeta=z("to_logits.0.bias", p["bf"]), gacc_stride=D)
        del dlogits
        bucket()
        bws = ops.d3pm_attention_bwd_workspace(B, L, H, dx.device) if L % 32 == 0 else None     # operand images (matrix-pipe backward)
        for i in reversed(range(len(p["layers"]))):
            lay, s = p["layers"][i], sv["layers"][i]
            pre = f"blocks.{i}."
            blk = tr.blocks[i]
            # ---- MLP
            im = sv["imgs"][i] if sv.get("imgs") is not None else None
            lin_t = (lambda dy_, name, w_, n_out: ops.rows_linear(dy_, im[name], n_out, torch.empty((B * L, n_out), **f))) if im is not None \
                else (lambda dy_, name, w_, n_out: ops.linear(dy_, tw(w_), torch.empty((B * L, n_out), **f)))
            ops.wgrad(dx, s["u"], z(pre + "mlp.2.weight", lay["w2"]), z(pre + "mlp.2.bias", lay["bb2"]))
            du = lin_t(dx, "w2_t", lay["w2"], s["u"].shape[1])
            da = ops.gelu2(s["a"], du)
            ops.wgrad(da, s["h2"], z(pre + "mlp.0.weight", lay["w1"]), z(pre + "mlp.0.bias", lay["bb1"]))
            dh2 = lin_t(da, "w1_t", lay["w1"], D)
            dx1 = ops.ln_bwd(dh2, s["x1"], s["stats2"], lay["g2"], dx_in=dx, dgamma=z(pre + "ln2.weight", lay["g2"]),
                             dbeta=z(pre + "ln2.bias", lay["b2"]), gacc_stride=D)
            # ---- attention output projection + the broadcast cross-attention vector
            ops.wgrad(dx1, s["y"], z(pre + "attn1.proj.weight", lay["wproj"]), z(pre + "attn1.proj.bias", lay["bproj"]))
            dcvec = ops.batch_rowsum(dx1, B, L)
            dv2 = ops.small_linear_bwd(dcvec, s["v2"], lay["wproj2"], z(pre + "attn2.proj.weight", lay["wproj2"]),
                                       z(pre + "attn2.proj.bias", lay["bproj2"]))
            ops.small_linear_bwd(dv2, sv["cond"], lay["wv2"], z(pre + "attn2.value.weight", lay["wv2"]),
                                 z(pre + "attn2.value.bias", lay["bv2"]), want_dx=False)
            for nm, ref in (("attn2.key.weight", lay["wk2"]), ("attn2.key.bias", lay["bk2"]),
                            ("attn2.query.weight", lay["wq2"]), ("attn2.query.bias", lay["bq2"]),
                            ("ln1_1.emb.weight", blk.ln1_1.emb.weight), ("ln1_1.linear.weight", blk.ln1_1.linear.weight),
                            ("ln1_1.linear.bias", blk.ln1_1.linear.bias)):
                z(pre + nm, ref)                # softmax over a single key: exactly zero gradient
            dy = lin_t(dx1, "proj_t", lay["wproj"], D)
            # ---- self-attention
            qkv = s["qkv"]
            dqkv = ops.d3pm_attention_bwd(qkv[0:H], qkv[H:2 * H], qkv[2 * H:], s["y"], dy, s["lse"], B, L, H, ws=bws)
            wq = z(pre + "_wqkv", lay["wqkv"])
            bq = z(pre + "_bqkv", lay["bqkv"])
            ops.wgrad(dqkv, s["hn"], wq, bq)
            for j, nm in enumerate(("query", "key", "value")):        # row blocks of the fused gradient: views, already contiguous
                g[pre + f"attn1.{nm}.weight"] = wq[j * D:(j + 1) * D]
                g[pre + f"attn1.{nm}.bias"] = bq[j * D:(j + 1) * D]
            del g[pre + "_wqkv"], g[pre + "_bqkv"]
            dhn = lin_t(dqkv, "qkv_t", lay["wqkv"], D)
            dtab = torch.zeros((B, 2 * D), **f)           # (not in the arena: the arena's size must not depend on the batch size)
            dx = ops.ln_bwd(dhn, s["x_in"], s["stats1"], lay["ada1"].view(-1), sel=t, gstride=2 * D, rows_per_batch=L,
                            dx_in=dx1, dgamma=dtab, dbeta=dtab.view(-1)[D:], gacc_stride=2 * D, acc_by_batch=True)
            ops.adaln_bwd(dtab, t, blk.ln1.emb.weight.contiguous(), blk.ln1.linear.weight.contiguous(),
                          z(pre + "ln1.emb.weight", blk.ln1.emb.weight), z(pre + "ln1.linear.weight", blk.ln1.linear.weight),
                          z(pre + "ln1.linear.bias", blk.ln1.linear.bias))
            if (len(p["layers"]) - i) % BUCKET_LAYERS == 0:
                bucket()
        # ---- embeddings
        ce = tr.content_emb
        Hs, Ws = ce.spatial_size
        dpos = self._arena.zeros((Hs * Ws, D))
        ops.d3pm_embed_bwd(dx, xt, z("content_emb.emb.weight", ce.emb.weight), dpos)
        ops.batch_rowsum(dpos, Hs, Ws, out=z("content_emb.height_emb.weight", ce.height_emb.weight))
        dw_ = self._arena.zeros((Ws * D,))
        ops.colsum(dpos.view(Hs, Ws * D), dw_)
        g["content_emb.width_emb.weight"] = dw_.view(Ws, D)
        if red is not None:
            bucket()
            a = self._arena
            lo = a.buf.data_ptr() if a.buf is not None else 0
            done = lambda t: lo and lo <= t.data_ptr() < lo + 4 * mark[0]         # inside a bucket that has been issued
            rest = [n for n, _ in tr.named_parameters() if not done(g[n])]
            if rest:                              # the first step (arena not sized yet): whatever lives outside goes in one bucket
                flat = torch.cat([g[n].reshape(-1) for n in rest])
                red.add(flat)
                off = 0
                for n in rest:
                    k = g[n].numel()
                    g[n] = flat[off:off + k].view(g[n].shape)
                    off += k
            red.finish()
        return fwd["loss"], g

    # ------------------------------------------------------------------ optimiser step (Adam) with DP averaging
    # ------------------------------------------------------------------ the step as one captured graph
    def _graph_usable(self, x0, cond):
        """Single-process steps only: with a data-parallel group the bucketed all-reduces run between the backward's launches through
        torch.distributed, outside any capture (the eager path; its launch gaps are covered by the collectives' own latency)."""
        return (os.environ.get("GSDD_TRAIN_GRAPH", "1") != "0" and not getattr(self, "_graph_failed", False) and x0.is_cuda
                and not self.reducer.active() and world_size() == 1 and cond.shape[1] == 1 and x0.shape[1] % 32 == 0)

    def _capture(self, x0, cond):
        """Everything of a step that runs on the device -- re-pack of the weight images and AdaLN tables, q_sample, forward, loss +
        gradient, backward, Adam -- recorded once into a hipGraph and replayed: ~1000 small dependent launches leave ~2 ms of gaps per
        step when they are enqueued one by one.  Nothing step-dependent is baked in: x_0, condition, t, p(t) live in static buffers,
        the Philox stream id and Adam's step count are device words the host sets before each replay; the timesteps are drawn
        by DiffusionTransformer.sample_time before each replay, exactly as the eager step draws them.  torch's graph-private
        allocator pool keeps every activation of the captured step at its address (torch.cuda.graph is the plumbing; every node of
        the graph is a libgsdd kernel or a fill / copy)."""
        dm, dev = self.dm, x0.device
        B = x0.shape[0]
        st = {"shape": (tuple(x0.shape), tuple(cond.shape)), "lr": self.lr,
              "x0": x0.clone().long().contiguous(), "cond": cond.clone().float().contiguous(),
              "t": torch.zeros((B,), dtype=torch.int64, device=dev), "pt": torch.full((B,), 1.0 / dm.num_timesteps, dtype=torch.float32, device=dev),
              "sid": torch.tensor([dm.noise_stream], dtype=torch.int64, device=dev),
              "adam_step": torch.tensor([self._adam.step_count + 1], dtype=torch.int64, device=dev)}
        tr = dm.transformer
        keep = (dm.noise_stream, self._adam.step_count)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            tr._packed = None                                    # the re-pack is part of the graph: every replay sees the current weights
            loss, grads = self.loss_and_grads(st["x0"], st["cond"], st["t"], st["pt"], reduce=False, sid=st["sid"])
            self._adam.lr = self.lr
            self._adam.step(grads, step_dev=st["adam_step"])
        dm.noise_stream, self._adam.step_count = keep             # capture executed nothing
        st["graph"], st["loss"], st["packed"] = graph, loss, tr._packed
        tr._packed = None
        return st

    @torch.no_grad()
    def step(self, x0, cond, t=None, pt=None):
        self._sync_start()
        if self._graph_usable(x0, cond):
            return self._step_graphed(x0, cond, t, pt)
        return self._step_eager(x0, cond, t, pt)

    def _step_graphed(self, x0, cond, t, pt):
        """Two eager steps first (arenas, images and tables reach their final addresses), then capture, then replays."""
        self._eager_steps = getattr(self, "_eager_steps", 0)
        # one captured graph per (batch shape, learning rate), the two most recent kept: an epoch's short last batch must not make
        # every epoch capture twice (each graph owns its activations' pool: ~7 GB at bs 16, L = 4096)
        graphs = self.__dict__.setdefault("_graphs", {})
        key = (tuple(x0.shape), tuple(cond.shape), float(self.lr))
        st = self._graph = graphs.get(key)
        if st is None and self._eager_steps < 2:
            self._eager_steps += 1
            return self._step_eager(x0, cond, t, pt)
        dm = self.dm
        if st is None:
            keep = (dm.noise_stream, self._adam.step_count)
            try:
                st = self._graph = self._capture(x0, cond)
                while len(graphs) >= 2:
                    graphs.pop(next(iter(graphs)))
                graphs[key] = st
            except Exception as e:                                # noqa: BLE001  (a capture that cannot be made must not cost the step)
                import warnings
                warnings.warn(f"D3PMTrainer: the training step could not be captured as a graph ({type(e).__name__}: {e}); running launch by launch")
                dm.noise_stream, self._adam.step_count = keep
                self._graph_failed = True
                dm.transformer._packed = None
                torch.cuda.synchronize()
                return self._step_eager(x0, cond, t, pt)
        if t is None:
            # the reference's own timestep sampler (importance sampling once every Lt_count exceeds 10: its check reads device memory,
            # i.e. waits for the previous step -- one host round trip per step, ~0.1 ms beside a 54 ms replay)
            t, pt = dm.sample_time(x0.shape[0], x0.device, "importance")
        st["x0"].copy_(x0, non_blocking=True)
        st["cond"].copy_(cond, non_blocking=True)
        st["t"].copy_(t, non_blocking=True)
        st["pt"].copy_(pt, non_blocking=True)
        # the two device words are set from the host's counts before every replay (two fills): whoever else draws from the noise stream
        # between steps -- the validation loop's _train_loss, set_noise, a resumed checkpoint -- moves dm.noise_stream, and the replay
        # must use the stream id the eager step would
        st["sid"].fill_(dm.noise_stream)
        st["adam_step"].fill_(self._adam.step_count + 1)
        st["graph"].replay()
        dm.noise_stream += 1
        self._adam.step_count += 1
        self.step_count += 1
        dm.transformer._packed = None                            # the packed views any eager caller holds predate this update
        self.last_fwd = None
        return st["loss"].clone()

    def _step_eager(self, x0, cond, t=None, pt=None):
        loss, grads = self.loss_and_grads(x0, cond, t, pt, reduce=True)
        tr = self.dm.transformer
        params = dict(tr.named_parameters())
        self.step_count += 1
        if getattr(self, "_adam", None) is None:
            self._adam = MultiAdam(list(params.items()), self.lr, self.betas, self.eps)
            self._adam.load_state(getattr(self, "_pending_adam", None))
            self._pending_adam = None
        self._adam.lr = self.lr
        self._adam.step(grads)                  # all parameters in one launch
        tr._packed = None                       # parameters changed in place through raw pointers
        return loss


    # ------------------------------------------------------------------ optimiser state for checkpoints (exact resume)
    def optimizer_state(self):
        a = getattr(self, "_adam", None)
        return a.state() if a is not None else getattr(self, "_pending_adam", None)

    def load_optimizer_state(self, state):
        self._graph, self._graphs = None, {}                    # (captured steps bake the addresses of the Adam state they were made with)
        if state is None:
            return
        if getattr(self, "_adam", None) is not None:
            self._adam.load_state(state)
        else:
            self._pending_adam = state


class _TrainForward(torch.autograd.Function):
    """Bridges the HIP training objective into torch.autograd so that the reference's stage-2 loop (zero_grad, manual_backward(loss),
    optimizer.step(): multistage_text_motion_model.py:186-197) runs unchanged: forward evaluates loss and gradients on the HIP path
    (one fused pass: the activations never outlive it), backward hands each transformer parameter its gradient scaled by the
    incoming d(loss)."""

    @staticmethod
    def forward(ctx, trainer, x0, cond, want_probs, *params):
        loss, grads = trainer.loss_and_grads(x0, cond, want_probs=want_probs, reduce=True)    # DDP semantics: .grad = group mean
        names = [n for n, _ in trainer.dm.transformer.named_parameters()]
        ctx.grads = [grads[n].clone() for n in names]          # the arena is re-used by the next forward
        return loss[0].clone()

    @staticmethod
    def backward(ctx, g_loss):
        out = tuple(g * g_loss for g in ctx.grads)
        ctx.grads = None
        return (None, None, None, None) + out


def train_forward(dm, x0, cond, want_probs=False):
    """DiffusionTransformer.forward in train mode with autograd enabled -> (loss with grad_fn, forward dict)."""
    tr = getattr(dm, "_hip_trainer", None)
    if tr is None:
        tr = D3PMTrainer(dm)
        object.__setattr__(dm, "_hip_trainer", tr)
    loss = _TrainForward.apply(tr, x0, cond, want_probs, *[p for _, p in dm.transformer.named_parameters()])
    return loss, tr.last_fwd
