"""The generator glue and the stage-2 wrapper as drop-ins (SURVEY.md section 8a last row, 8b):

* DiscreteDiffusion.forward against the output dict of the REFERENCE's DiscreteDiffusion.forward on the two small fixture models
  (tests/golden/glue_L64.npz, written by tests/golden/make_golden.py::make_glue from the imported reference): every key, both
  with the text embeddings zeroed (discrete_diffusion.py:25, :49 as written) and with them live (`zero_text_emb=False`);
* `out['losses'].backward()` through the glue fills the transformer's .grad with the HIP gradients;
* the reference-shaped manual-optimisation loop (generator_step -> ComputeLosses.update -> zero_grad -> manual_backward -> both
  optimisers' step, multistage_text_motion_model.py:170-200) for two steps == two D3PMTrainer.step calls."""
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


class TableText(torch.nn.Module):
    """The fixture's text encoder: one stored row per caption."""

    def __init__(self, texts, table):
        super().__init__()
        self.rows = {t: torch.from_numpy(table[i]) for i, t in enumerate(list(texts) + [""])}

    def forward(self, texts):
        return torch.stack([self.rows[t] for t in texts])


def build(G, golden, zero_text_emb=True):
    _, a, cfg = golden("glue_L64")
    sdv, av, cfgv = golden(str(cfg["vqvae"]))
    sdd, ad, cfgd = golden(str(cfg["d3pm"]))
    vq = G.VQVAE(None, cfgv["embedding_dim"], cfgv["n_codes"], cfgv["n_hiddens"], cfgv["n_res_layers"], cfgv["downsample"],
                 cfgv["sequence_length"], cfgv["resolution"])
    vq.load_state_dict(sdv)
    vq.codebook._need_init = False
    vq = vq.cuda().eval()
    d = G.DalleMaskImageEmbedding(num_embed=cfgd["K"], spatial_size=cfgd["spatial"], embed_dim=64)
    tr = G.Text2ImageTransformer(dalle=d, n_layer=cfgd["n_layer"], n_embd=64, n_head=16, content_seq_len=cfgd["L"],
                                 block_activate="GELU2", content_spatial_size=cfgd["spatial"], condition_dim=cfgd["cond_dim"],
                                 diffusion_step=cfgd["T"])
    dm = G.DiffusionTransformer(transformer=tr, diffusion_step=cfgd["T"], alpha_init_type="alpha1", auxiliary_loss_weight=5e-4,
                                adaptive_auxiliary_loss=True, guidance_scale=cfgd["guidance"], content_seq_len=cfgd["L"])
    dm.load_state_dict(sdd, strict=False)
    gen = G.DiscreteDiffusion(TableText(a["texts"], a["text_table"]), dm.cuda(), zero_text_emb=zero_text_emb)
    batch = {"video": torch.from_numpy(av["x"]).cuda(), "text": [str(t) for t in a["texts"]],
             "length": [av["x"].shape[2]] * av["x"].shape[0]}
    return gen, vq, batch, a, cfg, cfgd


def pin_time(dm, a, T):
    t_fix = torch.from_numpy(a["t"]).cuda()
    dm.sample_time = lambda b, device, method="uniform": (t_fix, torch.ones(b, device="cuda") / T)


@pytest.mark.parametrize("tag", ["zero", "cond"])
def test_glue_forward_matches_reference_output_dict(G, golden, tag):
    gen, vq, batch, a, cfg, cfgd = build(G, golden, zero_text_emb=(tag == "zero"))
    dm = gen.diffusion_model.eval()
    pin_time(dm, a, cfgd["T"])
    dm.set_noise(cfg["noise_seed"], stream=int(cfg["stream"]))
    dm.Lt_history.zero_(); dm.Lt_count.zero_()
    with torch.no_grad():
        out = gen(batch, vq, None, do_inference=True)
    assert set(out) == {"pred_data", "pred_single_step", "gt_data", "losses", "test"}
    np.testing.assert_allclose(out["losses"].item(), a[f"{tag}/losses"], rtol=2e-5)
    assert np.array_equal(gen.last_content_token.cpu().numpy(), a[f"{tag}/content_token"]), "sampled tokens differ from the reference's"
    for key in ("pred_data", "pred_single_step", "test"):
        torch.testing.assert_close(out[key].cpu(), torch.from_numpy(a[f"{tag}/{key}"]), atol=1e-4, rtol=1e-4, msg=lambda s, k=key: f"{k}: {s}")
    assert torch.equal(out["gt_data"], batch["video"])
    np.testing.assert_allclose(dm.Lt_history.cpu().numpy(), a[f"{tag}/Lt_history"], rtol=1e-4)
    assert np.array_equal(dm.Lt_count.cpu().numpy(), a[f"{tag}/Lt_count"])
    # do_inference=False: same loss and single-step decode, no sampling (discrete_diffusion.py:76-81)
    dm.set_noise(cfg["noise_seed"], stream=int(cfg["stream"]))
    dm.Lt_history.zero_(); dm.Lt_count.zero_()
    with torch.no_grad():
        out2 = gen(batch, vq, None)
    assert set(out2) == {"pred_data", "gt_data", "losses", "test"}
    torch.testing.assert_close(out2["pred_data"], out["pred_single_step"], atol=0, rtol=0)
    assert out2["losses"].item() == out["losses"].item()


def test_generator_defers_unread_decodes(G, golden, monkeypatch):
    """The glue's two decoded by-products are computed when read (LazyOutputs): a training-shaped call that reads `losses` only leaves
    them pending; reading them gives the tensors of an eager call (GSDD_EAGER_OUTPUTS=1); copying the mapping computes them."""
    gen, vq, batch, a, cfg, cfgd = build(G, golden)
    dm = gen.diffusion_model.eval()
    pin_time(dm, a, cfgd["T"])
    outs = {}
    for mode in ("lazy", "eager"):
        if mode == "eager":
            monkeypatch.setenv("GSDD_EAGER_OUTPUTS", "1")
        dm.set_noise(cfg["noise_seed"], stream=int(cfg["stream"]))
        dm.Lt_history.zero_(); dm.Lt_count.zero_()
        with torch.no_grad():
            out = gen(batch, vq, None)
        assert out.pending() == (["pred_data", "test"] if mode == "lazy" else [])
        _ = out["losses"]
        assert out.pending() == (["pred_data", "test"] if mode == "lazy" else [])
        outs[mode] = out
    copied = dict(outs["lazy"])
    assert outs["lazy"].pending() == [] and set(copied) == {"pred_data", "gt_data", "losses", "test"}
    for key in ("pred_data", "test"):
        assert torch.equal(copied[key], outs["eager"][key])
    # a VQ-VAE in train mode decodes at once (its BatchNorm statistics move with every decode, as in the reference)
    monkeypatch.delenv("GSDD_EAGER_OUTPUTS")
    vq.train()
    with torch.no_grad():
        assert gen(batch, vq, None).pending() == []
    vq.eval()
    # a deferred decode is refused once the VQ-VAE's weights have moved (it would no longer be the decode of this forward)
    with torch.no_grad():
        out = gen(batch, vq, None)
        next(vq.parameters()).mul_(1.0)                  # an in-place update through torch bumps the version counter
    with pytest.raises(G.GsddError):
        out["test"]
    _ = out["losses"], out["gt_data"]                    # what was computed inside forward stays readable


def test_glue_loss_backward_fills_transformer_grads(G, golden):
    """generator(batch, autoencoder)['losses'].backward() -- the call path of multistage_text_motion_model.py:170-197."""
    from gsdd_amd.d3pm_train import D3PMTrainer
    gen, vq, batch, a, cfg, cfgd = build(G, golden)
    dm = gen.diffusion_model.train()
    pin_time(dm, a, cfgd["T"])
    dm.set_noise(cfg["noise_seed"], stream=int(cfg["stream"]))
    out = gen(batch, vq, None)
    assert out["losses"].requires_grad and out["losses"].grad_fn is not None
    assert not out["pred_data"].requires_grad and not out["test"].requires_grad
    out["losses"].backward()
    # the same objective straight from the trainer
    gen2, vq2, _, _, _, _ = build(G, golden)
    dm2 = gen2.diffusion_model.train()
    dm2.set_noise(cfg["noise_seed"], stream=int(cfg["stream"]))
    with torch.no_grad():
        tokens = vq2.encode(batch["video"]).view(len(batch["text"]), -1)
    cond = torch.zeros(len(batch["text"]), 1, cfgd["cond_dim"], device="cuda")
    loss, want = D3PMTrainer(dm2).loss_and_grads(tokens, cond, t=torch.from_numpy(a["t"]).cuda(),
                                                 pt=torch.ones(len(batch["text"]), device="cuda") / cfgd["T"])
    np.testing.assert_allclose(out["losses"].item(), loss.item(), rtol=1e-6)
    for name, prm in dm.transformer.named_parameters():
        assert prm.grad is not None, name
        # same kernels on the same inputs; the weight gradients accumulate with float atomics, so not bit-for-bit
        scale = max(want[name].abs().max().item(), 1e-12)
        assert (prm.grad - want[name]).abs().max().item() <= 1e-5 * scale, name
    assert all(p.grad is None for p in vq.parameters())          # arg-min / arg-max cut the graph, as in the reference


def _stage2(G, golden, **flags):
    import src  # noqa: F401
    from src.models.multistage_text_motion_model import MultistageTextMotionModel
    gen, vq, batch, a, cfg, cfgd = build(G, golden)
    model = MultistageTextMotionModel(generator=gen, autoencoder=vq, lr_args={"gen_lr": 1e-4, "auto_lr": 1e-6}, devices=[0], **flags)
    dm = gen.diffusion_model
    pin_time(dm, a, cfgd["T"])
    dm.set_noise(cfg["noise_seed"], stream=int(cfg["stream"]))
    return model.cuda().train(), batch


def test_reference_shaped_manual_optimisation_matches_native_step(G, golden):
    from src.tasks.runner import Trainer
    model, batch = _stage2(G, golden)
    tr = Trainer(max_epochs=1)
    model.trainer = tr
    tr.optimizers = list(model.configure_optimizers())
    assert len(tr.optimizers) == 2 and not model.autoencoder.training          # eval() survives model.train()
    before = {k: v.detach().clone() for k, v in model.autoencoder.state_dict().items()}
    losses = []
    for i in range(2):
        ld = model.training_step(batch, i)
        losses.append(float(ld["generator_loss"]))
    native, _ = _stage2(G, golden, native_step=True)
    nl = [float(native.training_step(batch, i)["generator_loss"]) for i in range(2)]
    np.testing.assert_allclose(losses, nl, rtol=1e-5)
    assert losses[1] != losses[0]
    pa = dict(model.generator.diffusion_model.transformer.named_parameters())
    pb = dict(native.generator.diffusion_model.transformer.named_parameters())
    sd0 = golden(str(golden("glue_L64")[2]["d3pm"]))[0]
    # which entries have a real gradient: Adam turns the rounding noise of a mathematically zero gradient (attn1.key.bias, the
    # single-key cross-attention's q/k, ...) into +-lr steps that no two implementations agree on
    from gsdd_amd.d3pm_train import D3PMTrainer
    probe, pbatch = _stage2(G, golden)
    with torch.no_grad():
        tokens = probe.autoencoder.encode(pbatch["video"]).view(len(pbatch["text"]), -1)
    _, g0 = D3PMTrainer(probe.generator.diffusion_model).loss_and_grads(tokens, probe.generator._text(pbatch["text"], "cuda"))
    gmax = max(v.abs().max().item() for v in g0.values())
    moved = 0
    for k in pa:
        if g0[k].abs().max().item() < 1e-4 * gmax:
            continue
        da, db = pa[k].detach() - sd0["transformer." + k].cuda(), pb[k].detach() - sd0["transformer." + k].cuda()
        big = g0[k].abs() > 1e-2 * g0[k].abs().max()
        moved += int(big.sum())
        assert torch.allclose(da[big], db[big], atol=5e-6, rtol=2e-2), k      # deltas are ~2e-4 after two steps of lr 1e-4
    assert moved > 1000
    for k, v in model.autoencoder.state_dict().items():                          # opt_auto.step() ran and changed nothing
        assert torch.equal(v, before[k]), k
    # epoch-end keys of multistage_text_motion_model.py:208-238
    model.training_epoch_end([])
    assert set(model._logged) == {"total/train", "l/dummy/train", "epoch", "step"}
    np.testing.assert_allclose(model._logged["total/train"], np.mean(losses), rtol=1e-5)


def test_autoencoder_train_mode_flag_reproduces_the_unfrozen_reference(G, golden):
    """autoencoder_train_mode=True: encode() runs BatchNorm on batch statistics and the codebook EMA, as the reference's
    never-frozen stage-2 autoencoder does (multistage_text_motion_model.py:104 is commented out)."""
    from src.tasks.runner import Trainer
    model, batch = _stage2(G, golden, autoencoder_train_mode=True)
    assert model.autoencoder.training
    model.autoencoder.perm_source = lambda n: torch.arange(n)
    tr = Trainer(max_epochs=1)
    model.trainer = tr
    tr.optimizers = list(model.configure_optimizers())
    before = {k: v.detach().clone() for k, v in model.autoencoder.state_dict().items()}
    model.training_step(batch, 0)
    after = model.autoencoder.state_dict()
    assert not torch.equal(after["codebook.N"], before["codebook.N"])
    assert not torch.equal(after["encoder.res_stack.0.block.0.running_mean"], before["encoder.res_stack.0.block.0.running_mean"])
    assert torch.equal(after["encoder.convs.0.conv.weight"], before["encoder.convs.0.conv.weight"])     # no gradient reaches it


def test_lanes_after_weight_change_give_single_lane_tokens(G, golden):
    """sample() right after the packed weights were invalidated (train -> sample): the packed tables and fragment images are built
    on the caller's stream before the lanes fork, so every lane reads finished weights."""
    sd, a, cfg = golden("d3pm_L64")
    from tests.test_gpu_parity import build_d3pm
    B = 8
    cond = torch.randn(B, 1, cfg["cond_dim"], generator=torch.Generator().manual_seed(5)).cuda()
    toks = []
    for lanes in (1, 2):
        dm = build_d3pm(G, sd, cfg)
        with torch.no_grad():
            for prm in dm.transformer.parameters():
                prm.mul_(1.0)                                   # bumps the version: the cache key changes
        dm.transformer._packed = None
        dm.set_noise(77)
        toks.append(dm.sample(["x"] * B, None, cond, torch.zeros_like(cond), filter_ratio=0, lanes=lanes)["content_token"].cpu())
    assert torch.equal(toks[0], toks[1])
