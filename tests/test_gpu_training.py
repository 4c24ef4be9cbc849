"""D3PM training step on the HIP path: loss gradient w.r.t. every denoiser parameter against torch.autograd of the CPU
oracle (which is itself pinned to the reference's _train_loss fixture), then an Adam step against torch.optim.Adam."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gsdd_amd
    assert torch.cuda.is_available()
    gsdd_amd.lib()
    return gsdd_amd


def build(G, sd, cfg):
    d = G.DalleMaskImageEmbedding(num_embed=cfg["K"], spatial_size=cfg["spatial"], embed_dim=64)
    tr = G.Text2ImageTransformer(dalle=d, n_layer=cfg["n_layer"], n_embd=64, n_head=16, content_seq_len=cfg["L"],
                                 block_activate="GELU2", content_spatial_size=cfg["spatial"], condition_dim=cfg["cond_dim"],
                                 diffusion_step=cfg["T"])
    dm = G.DiffusionTransformer(transformer=tr, diffusion_step=cfg["T"], alpha_init_type="alpha1", auxiliary_loss_weight=5e-4,
                                adaptive_auxiliary_loss=True, guidance_scale=cfg["guidance"], content_seq_len=cfg["L"])
    dm.load_state_dict(sd, strict=False)
    return dm.cuda()


def oracle_grads(sd, a, cfg, t):
    from oracle import d3pm as od
    leaf = {k: (v.clone().requires_grad_(True) if k.startswith("transformer.") and v.dtype.is_floating_point else v)
            for k, v in sd.items()}
    pt = torch.ones(cfg["B"]) / cfg["T"]
    loss, _, _, _ = od.train_loss(torch.from_numpy(a["train_x0"]), torch.from_numpy(a["step_cond"]), t, pt, leaf,
                                  cfg["noise_seed"], int(a["train_stream"]))
    loss.backward()
    return loss.item(), {k[len("transformer."):]: (v.grad if v.grad is not None else torch.zeros_like(v))
                         for k, v in leaf.items() if k.startswith("transformer.") and v.dtype.is_floating_point}


@pytest.mark.parametrize("tvals", [[0, 61], [37, 99]])
def test_loss_gradients_match_autograd_of_oracle(G, golden, tvals):
    from gsdd_amd.d3pm_train import D3PMTrainer
    sd, a, cfg = golden("d3pm_L64")
    t = torch.tensor(tvals, dtype=torch.long)
    want_loss, want = oracle_grads(sd, a, cfg, t)
    dm = build(G, sd, cfg)
    dm.set_noise(cfg["noise_seed"], stream=int(a["train_stream"]))
    tr = D3PMTrainer(dm)
    loss, got = tr.loss_and_grads(torch.from_numpy(a["train_x0"]).cuda(), torch.from_numpy(a["step_cond"]).cuda(), t=t.cuda(),
                                  pt=(torch.ones(cfg["B"]) / cfg["T"]).cuda())
    np.testing.assert_allclose(loss.item(), want_loss, rtol=2e-5)
    assert set(got) == set(want), set(got) ^ set(want)
    worst = ("", 0.0)
    gmax = max(w.abs().max().item() for w in want.values())
    for k, w in want.items():
        gk = got[k].cpu()
        assert gk.shape == w.shape, k
        # gradients that are mathematically zero (e.g. attn1.key.bias: softmax is shift invariant) are pure rounding noise -- sums of
        # terms of the size of the large gradients that cancel -- so they are held to 2e-6 of the largest gradient of the model
        # (measured: 2e-7); every other tensor to 2e-3 of its own largest entry
        scale = max(w.abs().max().item(), 1e-3 * gmax)
        err = (gk - w).abs().max().item() / scale
        if err > worst[1]:
            worst = (k, err)
        assert err < 2e-3, f"{k}: relative max error {err:.3e} (|g|max {scale:.3e})"
    print("worst relative gradient error:", worst)


@pytest.mark.parametrize("K", [32, 4096])
def test_fused_loss_and_gradient_pass_equals_the_two_kernels(G, K):
    """gsdd_d3pm_train_loss_grad (one pass over the logits: loss terms, arg-max tokens, Lt statistics AND dlogits) against
    gsdd_d3pm_train_loss + gsdd_d3pm_train_loss_bwd: every output bit for bit, incl. t = 0 (the NLL branch) and [MASK] positions."""
    torch.manual_seed(K)
    B, L, T = 3, 96, 100
    d = G.DalleMaskImageEmbedding(num_embed=K, spatial_size=[16, 8], embed_dim=64)
    tr = G.Text2ImageTransformer(dalle=d, n_layer=1, n_embd=64, n_head=16, content_seq_len=L, block_activate="GELU2",
                                 content_spatial_size=[16, 8], diffusion_step=T)
    dm = G.DiffusionTransformer(transformer=tr, diffusion_step=T, alpha_init_type="alpha1", auxiliary_loss_weight=5e-4,
                                adaptive_auxiliary_loss=True, guidance_scale=2, content_seq_len=L).cuda()
    logits = (3.0 * torch.randn(B * L, K)).cuda()
    x0 = torch.randint(0, K, (B, L)).cuda()
    xt = x0.clone()
    xt[torch.rand(B, L, device="cuda") < 0.5] = K
    xt[0, :5] = torch.randint(0, K, (5,)).cuda()                  # a few unmasked positions that differ from x0
    t = torch.tensor([0, 37, 99]).cuda()
    pt = torch.tensor([0.01, 0.02, 0.005]).cuda()
    kw = dict(K=K, T=T, mask_weight=[1.0, 0.7], aux_weight=5e-4, adaptive_aux=True)
    h0, c0 = torch.rand(T).cuda(), torch.randint(0, 20, (T,)).float().cuda()
    ha, ca, hb, cb = h0.clone(), c0.clone(), h0.clone(), c0.clone()
    fa = G.ops.d3pm_train_loss(logits, x0, xt, t, pt, dm._sched(), ha, ca, want_probs=False, **kw)
    ga = G.ops.d3pm_train_loss_bwd(logits, x0, xt, t, pt, dm._sched(), **kw)
    fb, gb = G.ops.d3pm_train_loss_grad(logits, x0, xt, t, pt, dm._sched(), hb, cb, **kw)
    for k_ in ("loss", "per_sample", "x0_recon", "xt1_recon"):
        assert torch.equal(fa[k_], fb[k_]), k_
    assert torch.equal(ga, gb) and torch.equal(ha, hb) and torch.equal(ca, cb)
    assert bool(torch.isfinite(gb).all()) and float(gb.abs().max()) > 0


def test_loss_gradients_full_size_two_layers(G):
    """Gradient parity at the workload's own sequence and class counts (L = 4096 tokens, K = 4096 codes, one clip, TWO of the 19
    layers: the oracle's autograd keeps 16 x 4096^2 scores per layer): every parameter gradient against torch.autograd of the CPU
    oracle, same bar as the fixture-sized test.  Weights of a trained-like magnitude, so that the softmax rows are not flat."""
    from gsdd_amd.d3pm_train import D3PMTrainer
    from oracle import d3pm as od
    from tests.conftest import parity_report
    torch.manual_seed(7)
    cfg = dict(K=4096, L=4096, spatial=[64, 64], n_layer=2, cond_dim=512, T=100, guidance=2, B=1)
    d = G.DalleMaskImageEmbedding(num_embed=cfg["K"], spatial_size=cfg["spatial"], embed_dim=64)
    tr = G.Text2ImageTransformer(dalle=d, n_layer=2, n_embd=64, n_head=16, content_seq_len=cfg["L"], block_activate="GELU2",
                                 content_spatial_size=cfg["spatial"], condition_dim=512, diffusion_step=100)
    g = torch.Generator().manual_seed(8)
    for mod in tr.modules():
        if isinstance(mod, torch.nn.Linear):
            mod.weight.data = torch.randn(mod.weight.shape, generator=g) * (1.0 / mod.in_features ** 0.5)
            mod.bias.data = 0.1 * torch.randn(mod.bias.shape, generator=g)
        elif isinstance(mod, torch.nn.Embedding):
            mod.weight.data = torch.randn(mod.weight.shape, generator=g) * 0.5
    dm = G.DiffusionTransformer(transformer=tr, diffusion_step=100, alpha_init_type="alpha1", auxiliary_loss_weight=5e-4,
                                adaptive_auxiliary_loss=True, guidance_scale=2, content_seq_len=cfg["L"])
    sd = {k: v.detach().clone() for k, v in dm.state_dict().items()}
    x0 = torch.randint(0, cfg["K"], (1, cfg["L"]), generator=g)
    cond = torch.randn(1, 1, 512, generator=g)
    t = torch.tensor([41], dtype=torch.long)
    pt = torch.ones(1) / 100
    leaf = {k: (v.clone().requires_grad_(True) if k.startswith("transformer.") and v.dtype.is_floating_point else v) for k, v in sd.items()}
    want_loss, _, _, _ = od.train_loss(x0, cond, t, pt, leaf, 21, 0)
    want_loss.backward()
    want = {k[len("transformer."):]: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()
            if k.startswith("transformer.") and v.dtype.is_floating_point}
    dm = dm.cuda()
    dm.set_noise(21, stream=0)
    loss, got = D3PMTrainer(dm).loss_and_grads(x0.cuda(), cond.cuda(), t=t.cuda(), pt=pt.cuda())
    np.testing.assert_allclose(loss.item(), want_loss.item(), rtol=2e-5)
    assert set(got) == set(want), set(got) ^ set(want)
    gmax = max(w.abs().max().item() for w in want.values())
    worst = ("", 0.0)
    for k, w in want.items():
        scale = max(w.abs().max().item(), 1e-3 * gmax)
        err = (got[k].cpu() - w).abs().max().item() / scale
        if err > worst[1]:
            worst = (k, err)
    parity_report("d3pm_full_size_two_layer_gradients", {"parameters": len(want), "worst_relative_error": worst[1], "worst_parameter": worst[0],
                                                         "loss": loss.item(), "oracle_loss": want_loss.item()})
    assert worst[1] < 2e-3, worst


def test_training_forward_attention_modes_agree_at_long_sequences(G, monkeypatch):
    """The training forward's default attention arithmetic at L >= 2048 is the adaptive one (lo half only where a tile can hold a
    probability above 2^-8 of its row sum); GSDD_ATTN_TRAIN_P=22 forces hi + lo everywhere.  Loss and every parameter gradient of the
    two must agree to the documented bar (output error <= 2e-5 of the row scale: far below what the gradient parity tests resolve) at
    a length where the default actually is adaptive; anything but '22' / 'a8' is refused."""
    from gsdd_amd.d3pm_train import D3PMTrainer
    from tests.conftest import parity_report
    torch.manual_seed(17)
    K, L, B = 64, 2048, 2
    d = G.DalleMaskImageEmbedding(num_embed=K, spatial_size=[64, 32], embed_dim=64)
    tr = G.Text2ImageTransformer(dalle=d, n_layer=2, n_embd=64, n_head=16, content_seq_len=L, block_activate="GELU2",
                                 content_spatial_size=[64, 32], condition_dim=512, diffusion_step=100)
    g = torch.Generator().manual_seed(18)
    for mod in tr.modules():
        if isinstance(mod, torch.nn.Linear):
            mod.weight.data = torch.randn(mod.weight.shape, generator=g) * (1.0 / mod.in_features ** 0.5)
            mod.bias.data = 0.1 * torch.randn(mod.bias.shape, generator=g)
        elif isinstance(mod, torch.nn.Embedding):
            mod.weight.data = torch.randn(mod.weight.shape, generator=g) * 0.5
    dm = G.DiffusionTransformer(transformer=tr, diffusion_step=100, alpha_init_type="alpha1", auxiliary_loss_weight=5e-4,
                                adaptive_auxiliary_loss=True, guidance_scale=2, content_seq_len=L).cuda()
    x0 = torch.randint(0, K, (B, L), generator=g).cuda()
    cond = torch.randn(B, 1, 512, generator=g).cuda()
    t, pt = torch.tensor([30, 77]).cuda(), (torch.ones(B) / 100).cuda()
    res = {}
    for mode in (None, "22", "a8"):
        if mode is None:
            monkeypatch.delenv("GSDD_ATTN_TRAIN_P", raising=False)
        else:
            monkeypatch.setenv("GSDD_ATTN_TRAIN_P", mode)
        dm.set_noise(5, stream=0)
        loss, grads = D3PMTrainer(dm).loss_and_grads(x0, cond, t=t, pt=pt)
        res[mode] = (loss.item(), {k: v.clone() for k, v in grads.items()})
    # the default IS a8 at this length: same loss bit for bit (the forward is deterministic); the gradients agree to the order of the
    # weight-gradient kernels' float atomics
    assert res[None][0] == res["a8"][0]
    gmax8 = max(v.abs().max().item() for v in res["a8"][1].values())
    for k, w in res["a8"][1].items():                # (mathematically zero gradients -- attn1.key.bias -- are rounding noise: absolute bar)
        torch.testing.assert_close(res[None][1][k], w, atol=1e-6 * gmax8, rtol=1e-5)
    gmax = max(v.abs().max().item() for v in res["22"][1].values())
    worst = max(((res[None][1][k] - w).abs().max().item() / max(w.abs().max().item(), 1e-3 * gmax), k) for k, w in res["22"][1].items())
    parity_report("train_attention_default_vs_hi_lo_L2048", {"loss_default": res[None][0], "loss_hi_lo": res["22"][0],
                                                             "worst_relative_gradient_difference": worst[0], "worst_parameter": worst[1]})
    assert abs(res[None][0] - res["22"][0]) <= 2e-5 * abs(res["22"][0])
    assert worst[0] < 5e-4, worst
    monkeypatch.setenv("GSDD_ATTN_TRAIN_P", "11")
    with pytest.raises(G.GsddError):
        D3PMTrainer(dm).loss_and_grads(x0, cond, t=t, pt=pt)
    monkeypatch.delenv("GSDD_ATTN_TRAIN_P")


def test_captured_training_step_equals_the_eager_steps(G, golden, monkeypatch):
    """D3PMTrainer.step replays one captured hipGraph from its third step on (re-pack, q_sample, forward, loss + gradient, backward,
    Adam; the Philox stream id and Adam's step count are device words set before each replay).  Seven steps on changing batches and
    timesteps must give the losses and the weights of the eager trainer (GSDD_TRAIN_GRAPH=0): nothing step-dependent may be baked in.
    (Adam's bias corrections are evaluated in double on the device and in float on the host: the weights agree to 1e-6.)"""
    from gsdd_amd.d3pm_train import D3PMTrainer
    sd, a, cfg = golden("d3pm_L64")
    B, L, K, T = cfg["B"], cfg["L"], cfg["K"], cfg["T"]
    g = torch.Generator().manual_seed(41)
    batches = [(torch.randint(0, K, (B, L), generator=g).cuda(), torch.randn(B, 1, cfg["cond_dim"], generator=g).cuda(),
                torch.randint(0, T, (B,), generator=g).cuda(), torch.full((B,), 1.0 / T).cuda()) for _ in range(7)]
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("GSDD_TRAIN_GRAPH", mode)
        dm = build(G, sd, cfg).train()
        dm.set_noise(cfg["noise_seed"], stream=3)
        tr = D3PMTrainer(dm, lr=1e-3)
        losses = []
        for i, (x0, cond, t, pt) in enumerate(batches):
            if i == 4:                                # somebody else draws from the noise stream between two steps (a validation pass does)
                dm.noise_stream += 3
            losses.append(tr.step(x0, cond, t=t, pt=pt)[0].item())
        assert (getattr(tr, "_graph", None) is not None) == (mode == "1")
        assert dm.noise_stream == 3 + 3 + len(batches) and tr._adam.step_count == len(batches)
        runs[mode] = (losses, {k: v.detach().clone() for k, v in dm.transformer.state_dict().items()}, dm.Lt_count.clone(), tr.optimizer_state())
    monkeypatch.delenv("GSDD_TRAIN_GRAPH")
    # a short batch (an epoch's last one) gets its own captured graph, and the full-size graph is still there afterwards
    x0s, conds, ts, pts = (z[:1].contiguous() for z in batches[0])
    g_full = tr._graph
    tr.step(x0s, conds, t=ts, pt=pts)
    assert tr._graph is not g_full and len(tr._graphs) == 2
    tr.step(*batches[1][:2], t=batches[1][2], pt=batches[1][3])
    assert tr._graph is g_full
    np.testing.assert_allclose(runs["1"][0], runs["0"][0], rtol=2e-5)
    for k, w in runs["0"][1].items():
        if k.endswith("attn1.key.bias"):              # softmax is shift invariant: this gradient is mathematically zero, what arrives is
            continue                                  # rounding noise in atomics order, and Adam normalises noise to full-size updates
        torch.testing.assert_close(runs["1"][1][k], w, atol=2e-6, rtol=1e-5, msg=lambda m, k=k: f"{k}: {m}")
    assert torch.equal(runs["1"][2], runs["0"][2])
    assert runs["1"][3]["step"] == runs["0"][3]["step"] == len(batches)
    m0 = runs["0"][3]["m"]                           # (entries fed by mathematically zero gradients hold atomics-order noise: absolute bar)
    torch.testing.assert_close(runs["1"][3]["m"], m0, atol=1e-5 * m0.abs().max().item(), rtol=1e-4)


def test_adam_step_matches_torch(G, golden):
    from gsdd_amd.d3pm_train import D3PMTrainer
    sd, a, cfg = golden("d3pm_L64")
    t = torch.tensor([5, 77], dtype=torch.long)
    _, want_g = oracle_grads(sd, a, cfg, t)
    # torch.optim.Adam on CPU copies with the oracle's gradients
    ref = {k[len("transformer."):]: v.clone() for k, v in sd.items() if k.startswith("transformer.") and v.dtype.is_floating_point}
    params = [torch.nn.Parameter(v) for v in ref.values()]
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.5, 0.999))
    for prm, k in zip(params, ref):
        prm.grad = want_g[k]
    opt.step()
    dm = build(G, sd, cfg)
    dm.set_noise(cfg["noise_seed"], stream=int(a["train_stream"]))
    tr = D3PMTrainer(dm, lr=1e-4, betas=(0.5, 0.999))
    tr.step(torch.from_numpy(a["train_x0"]).cuda(), torch.from_numpy(a["step_cond"]).cuda(), t=t.cuda(),
            pt=(torch.ones(cfg["B"]) / cfg["T"]).cuda())
    got = dict(dm.transformer.named_parameters())
    gmax = max(v.abs().max().item() for v in want_g.values())
    for prm, k in zip(params, ref):
        if want_g[k].abs().max().item() < 1e-4 * gmax:
            continue        # mathematically zero gradient (rounding noise): Adam turns noise into +-lr steps, not comparable
        # Adam normalises the update to ~lr, so compare the parameter DELTAS
        d_ref = (prm.detach() - ref[k] if False else prm.detach() - sd["transformer." + k])
        d_got = got[k].detach().cpu() - sd["transformer." + k]
        big = want_g[k].abs() > 1e-3 * want_g[k].abs().max().clamp(min=1e-12)      # ignore sign flips of ~zero gradients
        assert torch.allclose(d_got[big], d_ref[big], atol=2e-6, rtol=2e-2), k
    # the sampler must see the updated weights (packed cache invalidated)
    assert dm.transformer._packed is None


def _dp_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import gsdd_amd
    from gsdd_amd.d3pm_train import D3PMTrainer
    from gsdd_amd.parallel import shard_batch
    from tests.conftest import load_golden
    dist.init_process_group("gloo")                  # rehearsal backend: both ranks share the box's single GPU
    torch.cuda.set_device(0)
    sd, a, cfg = load_golden("d3pm_L64")
    dm = build(gsdd_amd, sd, cfg)
    start, count = shard_batch(cfg["B"], world, rank)
    dm.set_noise(cfg["noise_seed"], stream=int(a["train_stream"]), row_offset=start)
    t = torch.tensor([5, 77])[start:start + count].cuda()
    tr = D3PMTrainer(dm)
    tr.step(torch.from_numpy(a["train_x0"])[start:start + count].cuda(), torch.from_numpy(a["step_cond"])[start:start + count].cuda(),
            t=t, pt=(torch.ones(count) / cfg["T"]).cuda())
    if rank == 0:
        q.put({k: v.detach().cpu().numpy() for k, v in dm.transformer.named_parameters()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_equals_global_batch(G, golden):
    """2 ranks x batch 1 with gradient all-reduce == 1 process x batch 2 (the loss is a mean over B*L)."""
    import os
    import torch.multiprocessing as mp
    from gsdd_amd.d3pm_train import D3PMTrainer
    sd, a, cfg = golden("d3pm_L64")
    dm = build(G, sd, cfg)
    dm.set_noise(cfg["noise_seed"], stream=int(a["train_stream"]))
    tr = D3PMTrainer(dm)
    tr.step(torch.from_numpy(a["train_x0"]).cuda(), torch.from_numpy(a["step_cond"]).cuda(), t=torch.tensor([5, 77]).cuda(),
            pt=(torch.ones(2) / cfg["T"]).cuda())
    single = {k: v.detach().cpu() for k, v in dm.transformer.named_parameters()}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _, want_g = oracle_grads(sd, a, cfg, torch.tensor([5, 77]))
    gmax = max(v.abs().max().item() for v in want_g.values())
    for k, v in single.items():
        if want_g[k].abs().max().item() < 1e-4 * gmax:
            continue
        d1, d2 = v - sd["transformer." + k], torch.from_numpy(got[k]) - sd["transformer." + k]
        big = want_g[k].abs() > 1e-2 * want_g[k].abs().max()
        assert torch.allclose(d1[big], d2[big], atol=5e-6, rtol=5e-2), k


def _rccl_one_rank_worker(port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", GSDD_REDUCER_FORCE="1")
    import torch.distributed as dist
    import gsdd_amd
    from gsdd_amd.d3pm_train import D3PMTrainer
    from tests.conftest import load_golden
    torch.cuda.set_device(0)
    dist.init_process_group("nccl")                  # RCCL, on the real device: a group of one
    sd, a, cfg = load_golden("d3pm_L64")
    x0, cond = torch.from_numpy(a["train_x0"]).cuda(), torch.from_numpy(a["step_cond"]).cuda()
    t, pt = torch.tensor([5, 77]).cuda(), (torch.ones(2) / cfg["T"]).cuda()
    grads = {}
    for reduce in (False, True):
        dm = build(gsdd_amd, sd, cfg)
        dm.set_noise(cfg["noise_seed"], stream=int(a["train_stream"]))
        tr = D3PMTrainer(dm)
        _, g = tr.loss_and_grads(x0, cond, t=t, pt=pt, reduce=reduce)
        torch.cuda.synchronize()
        grads[reduce] = {k: v.detach().cpu().clone() for k, v in g.items()}
        stats = tr.reducer.stats()
    probe = torch.arange(8, dtype=torch.float32, device="cuda")
    dist.all_reduce(probe)
    # (not bit-identical: the weight-gradient kernels add their row slabs with float atomics, whose order varies from run to run)
    worst = max(((grads[False][k] - grads[True][k]).abs().max() / grads[False][k].abs().max().clamp(min=1e-20)).item() for k in grads[False]
                if grads[False][k].abs().max() > 0)
    q.put({"same": worst, "stats": stats,
           "probe": probe.cpu().tolist(), "backend": dist.get_backend()})
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_initialises_and_carries_the_bucketed_exchange_on_one_gpu(G):
    """No multi-GPU box is to be had for these tests, so this is what can be shown on hardware: RCCL (backend "nccl") initialises
    on the MI355X, and the data-parallel gradient exchange -- buckets all-reduced asynchronously on RCCL's stream while the backward
    is still being enqueued, the compute stream waiting for them at the end -- runs through it with a group of one and leaves the
    gradients those of the run without it.  (Two ranks on one device are not possible with RCCL; the 2-rank tests use gloo.)"""
    import os
    import torch.multiprocessing as mp
    from tests.conftest import parity_report
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(29700 + (os.getpid() % 200), q))
    p.start()
    got = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    parity_report("rccl_one_rank_rehearsal", {"backend": got["backend"], "worst_relative_gradient_difference": got["same"], **got["stats"]})
    assert got["backend"] == "nccl" and got["same"] < 1e-5 and got["probe"] == [float(i) for i in range(8)]
    assert got["stats"].get("allreduce_buckets", 0) >= 1 and got["stats"]["allreduce_mib"] > 0


# ----------------------------------------------------------------------------- attention backward kernels vs fp64 autograd
@pytest.mark.parametrize("mode", ["fused", "fused_q64", "fused_q128", "fused_w8", "split", "valu"])
@pytest.mark.parametrize("B,L,scale", [(2, 64, 1.0), (1, 320, 1.5), (1, 1024, 1.0), (2, 96, 3.0), (2, 544, 1.0)])
def test_attention_backward_matches_fp64_autograd(G, B, L, scale, mode, monkeypatch):
    """dq | dk | dv of softmax(q k^T / 2) v for head dim 4 against torch.autograd in fp64: the fused matrix-pipe kernel (one pass,
    dS through an LDS transpose, partial dQ per 256-key block + reduction; default 96-query chunks, and the selectable 64- / 128-query
    chunks and 8-wave workgroups), the two-kernel matrix-pipe variant (variant "split") and the VALU kernels (no workspace).
    L = 320 / 544 cover partial key blocks and query chunks (544 = 2 key blocks + 32, 5 query chunks + 64), scale 3 peaky attention."""
    valu = mode == "valu"
    variant = {"fused": None, "fused_q64": "fqc64", "fused_q128": "fqc128", "fused_w8": "nw8", "split": "split", "valu": "valu"}[mode]
    H = 16
    g = torch.Generator().manual_seed(9)
    q = (torch.randn(B, H, L, 4, generator=g) * scale).double().requires_grad_(True)
    k = (torch.randn(B, H, L, 4, generator=g) * scale).double().requires_grad_(True)
    v = torch.randn(B, H, L, 4, generator=g).double().requires_grad_(True)
    dO = torch.randn(B, H, L, 4, generator=g).double()
    o = torch.softmax((q @ k.transpose(-1, -2)) * 0.5, dim=-1) @ v
    o.backward(dO)
    hm = lambda z: z.detach().float().permute(1, 0, 2, 3).reshape(H, B * L, 4).contiguous().cuda()          # head-major rows
    rm = lambda z: z.detach().float().permute(0, 2, 1, 3).reshape(B * L, H * 4).contiguous().cuda()         # [M][H*4]
    qh, kh, vh = hm(q), hm(k), hm(v)
    out = torch.empty((B * L, H * 4), device="cuda")
    lse = torch.empty((H * B * L,), device="cuda")
    G.ops.d3pm_attention_train(qh, kh, vh, B, L, H, out, lse, ws=G.ops.d3pm_attention_workspace(B, L, H, "cuda"))
    torch.testing.assert_close(out.cpu().double(), rm(o).cpu().double(), atol=2e-5, rtol=0)
    ws = None if valu else G.ops.d3pm_attention_bwd_workspace(B, L, H, "cuda")
    dqkv = G.ops.d3pm_attention_bwd(qh, kh, vh, out, rm(dO), lse, B, L, H, ws=ws, variant=variant).cpu().double()
    for name, got, want in (("dq", dqkv[:, :64], rm(q.grad)), ("dk", dqkv[:, 64:128], rm(k.grad)), ("dv", dqkv[:, 128:], rm(v.grad))):
        want = want.cpu().double()
        err = (got - want).abs().max().item() / want.abs().max().item()
        assert err < 2e-5, f"{name}: relative max error {err:.3e}"


def test_d3pm_forward_backward_through_autograd_bridge(G, golden):
    """The reference's stage-2 loop calls manual_backward(loss) on DiffusionTransformer.forward's loss
    (multistage_text_motion_model.py:186-197): in train mode with autograd enabled the loss must carry a grad_fn that fills
    every transformer .grad with the HIP gradients (scaled by the incoming d(loss))."""
    from gsdd_amd.d3pm_train import D3PMTrainer
    sd, a, cfg = golden("d3pm_L64")
    x0, cond = torch.from_numpy(a["train_x0"]).cuda(), torch.from_numpy(a["step_cond"]).cuda()
    dm = build(G, sd, cfg).train()
    dm.set_noise(cfg["noise_seed"], stream=int(a["train_stream"]))
    torch.manual_seed(0)
    out = dm({"content_token": x0, "condition_embed_token": cond}, return_loss=True)
    assert out["loss"].requires_grad and tuple(out["logits"].shape) == (cfg["B"], cfg["K"] + 1, cfg["L"])
    (2.0 * out["loss"]).backward()
    t_used = dm.last_train_stats["t"]
    ref = build(G, sd, cfg).train()
    ref.set_noise(cfg["noise_seed"], stream=int(a["train_stream"]))
    pt = torch.ones(cfg["B"], device="cuda") / cfg["T"]
    loss, want = D3PMTrainer(ref).loss_and_grads(x0, cond, t=t_used, pt=pt)
    np.testing.assert_allclose(out["loss"].item(), loss.item(), rtol=1e-6)
    for name, prm in dm.transformer.named_parameters():
        assert prm.grad is not None, name
        torch.testing.assert_close(prm.grad, 2.0 * want[name], rtol=1e-5, atol=1e-8, msg=lambda s, n=name: f"{n}: {s}")


@pytest.mark.parametrize("n_in,n_out", [(64, 64), (64, 128), (64, 192), (64, 256), (128, 64), (192, 64), (256, 64)])
def test_rows_linear_matches_fp64(G, n_in, n_out):
    """gsdd_rows_linear (the block's row GEMMs in the training step, weights as bf16x3 fragment images): every shape, plain and
    transposed images, bias / per-batch vector / residual epilogues, head-major output, a row count that is not a multiple of 32 --
    against an fp64 product.  Gradients run through it too, hence operands far outside f16's range."""
    import numpy as np
    ops = G.ops
    torch.manual_seed(n_in + n_out)
    Lb, Bn = 40, 3                                  # 120 rows: three full 32-row groups and one of 24
    M = Lb * Bn
    x = torch.randn(M, n_in, device="cuda") * 1e-6  # gradient-sized values
    x[::7] *= 1e9
    w = torch.randn(n_out, n_in, device="cuda") * 0.1
    wt = w.t().contiguous()                          # [n_in][n_out]: the image of its transpose is W again
    bias = torch.randn(n_out, device="cuda")
    bvec = torch.randn(Bn, n_out, device="cuda")
    res = torch.randn(M, n_out, device="cuda")
    nbytes = ops.rows_linear_image_bytes(n_out, n_in)
    imgs = torch.empty((2, nbytes), dtype=torch.uint8, device="cuda")
    rows = [(w.data_ptr(), n_out, n_in, n_in, 0, imgs[0].data_ptr()), (wt.data_ptr(), n_out, n_in, n_out, 1, imgs[1].data_ptr())]
    table = np.array(rows, dtype=np.dtype([("w", "<u8"), ("n_out", "<i4"), ("n_in", "<i4"), ("ld", "<i4"), ("transpose", "<i4"), ("img", "<u8")]))
    ops.rows_linear_pack_many(torch.from_numpy(table.view(np.uint8).copy()).cuda(), 2, n_out, n_in)
    assert torch.equal(imgs[0], imgs[1])
    want = x.double() @ w.double().t()
    scale = float(want.abs().max())
    out = ops.rows_linear(x, imgs[0], n_out, torch.empty(M, n_out, device="cuda"))
    assert float((out.double() - want).abs().max()) < 1e-6 * scale
    full = want + bias.double() + bvec.double().repeat_interleave(Lb, 0) + res.double()
    out = ops.rows_linear(x, imgs[1], n_out, torch.empty(M, n_out, device="cuda"), bias=bias, bvec=bvec, rows_per_batch=Lb, residual=res)
    assert float((out.double() - full).abs().max()) < 1e-6 * max(scale, float(full.abs().max()))
    hm = ops.rows_linear(x, imgs[0], n_out, torch.empty(n_out // 4, M, 4, device="cuda"), bias=bias, head_major=True)
    ref = (want + bias.double()).reshape(M, n_out // 4, 4).permute(1, 0, 2)
    assert float((hm.double() - ref).abs().max()) < 1e-6 * max(scale, float(ref.abs().max()))
