"""Pin the oracle (our CPU restatement) against fixtures produced by the REFERENCE itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import d3pm, philox, vqvae


def test_philox_known_answers():
    # Random123 kat_vectors for philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        got = philox.philox4x32_10(*c, *k)
        assert tuple(int(x) for x in got) == want


@pytest.mark.parametrize("name", ["vqvae_ds188", "vqvae_ds244"])
def test_vqvae_oracle_matches_reference(golden, name):
    sd, a, cfg = golden(name)
    x = torch.from_numpy(a["x"])
    with torch.no_grad():
        z = vqvae.pre_vq(x, sd, cfg)
        idx = vqvae.encode(x, sd, cfg)
        rec = vqvae.decode(torch.from_numpy(a["encodings"]), sd, cfg)
        fwd = vqvae.forward_eval(x, sd, cfg)
        h_enc = vqvae.encoder(x, sd, cfg)
    np.testing.assert_allclose(h_enc.numpy(), a["h_enc"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(z.numpy(), a["z"], atol=2e-5, rtol=1e-5)
    assert np.array_equal(idx.numpy(), a["encodings"])                  # code indices bit-exact
    np.testing.assert_allclose(rec.numpy(), a["decoded"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(fwd["pred_data"].numpy(), a["fwd_pred"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(fwd["losses"]["recon_loss"].item(), a["fwd_recon_loss"], rtol=1e-5)
    np.testing.assert_allclose(fwd["losses"]["commitment_loss"].item(), a["fwd_commitment_loss"], rtol=1e-5)


def test_schedule_matches_reference(golden):
    sd, a, cfg = golden("d3pm_L64")
    buf = d3pm.schedule_buffers(cfg["T"], cfg["K"])
    for k, v in buf.items():
        assert torch.equal(v, sd[k]), k


def test_denoiser_and_posterior_match_reference(golden):
    sd, a, cfg = golden("d3pm_L64")
    xt = torch.from_numpy(a["step_xt"])
    cond = torch.from_numpy(a["step_cond"])
    t = torch.from_numpy(a["step_t"])
    K1 = cfg["K"] + 1
    with torch.no_grad():
        logits = d3pm.denoiser(xt, cond, t, sd)
        logits_u = d3pm.denoiser(xt, torch.zeros_like(cond), t, sd)
        logits3 = d3pm.denoiser(xt, torch.from_numpy(a["cond3"]), t, sd)
        log_xt = d3pm.index_to_log_onehot(xt, K1)
        ps = d3pm.predict_start(log_xt, cond, t, sd)
        cf = d3pm.cf_predict_start(log_xt, cond, torch.zeros_like(cond), t, sd, cfg["guidance"])
        post = d3pm.q_posterior(cf, log_xt, t, sd)
        samp = d3pm.gumbel_argmax(post, cfg["noise_seed"], int(a["step_stream"]))
    np.testing.assert_allclose(logits.numpy(), a["step_logits"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(logits_u.numpy(), a["step_logits_uncond"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(logits3.numpy(), a["logits_cond3"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(ps.numpy(), a["step_predict_start"], atol=1e-5)
    np.testing.assert_allclose(cf.numpy(), a["step_cf_predict_start"], atol=2e-5)
    np.testing.assert_allclose(post.numpy(), a["step_posterior"], atol=2e-5)
    assert np.array_equal(samp.numpy(), a["step_sample"])               # tokens bit-exact


def test_full_reverse_loop_matches_reference(golden):
    sd, a, cfg = golden("d3pm_L64")
    cond = torch.from_numpy(a["step_cond"])
    trace = []
    with torch.no_grad():
        tok = d3pm.sample(cfg["B"], cfg["L"], cond, torch.zeros_like(cond), sd, cfg["guidance"],
                          cfg["noise_seed"], trace=trace)
    assert np.array_equal(np.stack([x.numpy() for x in trace]), a["loop_trace"])
    assert np.array_equal(tok.numpy(), a["loop_tokens"])


def test_long_sequence_fixture_first_steps_match_reference(golden):
    """`d3pm_L2048`: the reference's own 100-step loop at L = 2048 (where the device's attention kernel switches to its adaptive P
    arithmetic).  The oracle reproduces the teacher-forced step's logits and sample and the first three steps of the loop trace (the
    whole loop is 3 minutes of CPU; the device test runs all 100 steps)."""
    sd, a, cfg = golden("d3pm_L2048")
    xt, cond, t = torch.from_numpy(a["step_xt"]), torch.from_numpy(a["step_cond"]), torch.from_numpy(a["step_t"])
    with torch.no_grad():
        logits = d3pm.denoiser(xt, cond, t, sd)
        tok, _ = d3pm.p_sample_step(xt, cond, torch.zeros_like(cond), t, sd, cfg["guidance"], cfg["noise_seed"], int(a["step_stream"]))
        trace = []
        d3pm.sample(cfg["B"], cfg["L"], cond, torch.zeros_like(cond), sd, cfg["guidance"], cfg["noise_seed"], trace=trace, steps=3)
    np.testing.assert_allclose(logits.numpy(), a["step_logits"], atol=2e-5, rtol=1e-5)
    assert np.array_equal(tok.numpy(), a["step_sample"])
    assert np.array_equal(np.stack([x.numpy() for x in trace]), a["loop_trace"][:3])
    assert int((a["loop_tokens"] == cfg["K"]).sum()) == 0 and a["loop_trace"].shape == (cfg["T"], cfg["B"], cfg["L"])


def test_train_loss_matches_reference(golden):
    sd, a, cfg = golden("d3pm_L64")
    cond = torch.from_numpy(a["step_cond"])
    x0 = torch.from_numpy(a["train_x0"])
    t = torch.from_numpy(a["train_t"])
    pt = torch.ones(cfg["B"]) / cfg["T"]
    with torch.no_grad():
        loss, probs, pred, kl_loss = d3pm.train_loss(x0, cond, t, pt, sd, cfg["noise_seed"], int(a["train_stream"]))
    np.testing.assert_allclose(loss.item(), a["train_loss"], rtol=1e-5)
    np.testing.assert_allclose(probs.numpy(), a["train_logits"], atol=1e-5)
    assert np.array_equal(pred.numpy(), a["train_pred"])


def test_vqvae_train_forward_matches_reference():
    from tests.conftest import GOLDEN
    import os
    z = np.load(os.path.join(GOLDEN, "vqvae_train_ds188.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    after = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("after/")}
    cfg = {k[4:]: (z[k].tolist() if z[k].ndim else z[k].item()) for k in z.files if k.startswith("cfg_")}
    with torch.no_grad():
        out, new = vqvae.forward_train(torch.from_numpy(z["x"]), sd, cfg, z["perm"][0])
    np.testing.assert_allclose(out["pred_data"].numpy(), z["pred"], atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(out["losses"]["recon_loss"].item(), z["recon_loss"], rtol=1e-5)
    np.testing.assert_allclose(out["losses"]["commitment_loss"].item(), z["commitment_loss"], rtol=1e-5)
    for k, v in new.items():
        if v.dtype.is_floating_point:
            np.testing.assert_allclose(v.numpy(), after[k].numpy(), atol=2e-5, rtol=1e-4, err_msg=k)
        else:
            assert torch.equal(v, after[k]), k


def test_preprocess_oracle_matches_reference_function():
    """oracle/preprocess.py against outputs of the reference's `preprocess` (ucf101_dataset.py:105-140) on uint8 clips:
    landscape / portrait / square, up- and down-scaling, odd sizes, with and without the temporal crop.  fp32 tolerance 1e-6
    (the reference's vectorised bilinear kernel associates the four products differently: 1-3 ulp)."""
    import os
    from tests.conftest import GOLDEN
    from oracle import preprocess as op
    z = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    for i in range(5):
        r, sl = (int(v) for v in z[f"cfg{i}"])
        out = op.preprocess(z[f"in{i}"], r, None if sl < 0 else sl)
        assert out.shape == z[f"out{i}"].shape
        np.testing.assert_allclose(out, z[f"out{i}"], atol=1e-6, rtol=0)


def test_frechet_distance_matches_reference_function():
    """gsdd_amd.metrics.frechet_distance against values returned by the reference's frechet_distance (evaluator.py:166-179) on
    three feature-set pairs (different sample counts, dimension == sample count, shifted means)."""
    import os
    from tests.conftest import GOLDEN
    from gsdd_amd.metrics import frechet_distance
    z = np.load(os.path.join(GOLDEN, "frechet.npz"))
    for i in range(3):
        got = frechet_distance(torch.from_numpy(z[f"a{i}"]), torch.from_numpy(z[f"b{i}"])).item()
        np.testing.assert_allclose(got, float(z[f"fd{i}"]), rtol=2e-4)      # fp32 SVDs of ill-conditioned covariance products
    x = torch.from_numpy(z["a1"])
    assert abs(frechet_distance(x, x.clone()).item()) < 1e-2 * float(z["fd1"])


def _load_state_fixture(name):
    import os
    from tests.conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    after = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("after/")}
    cfg = {k[4:]: (z[k].tolist() if z[k].ndim else z[k].item()) for k in z.files if k.startswith("cfg_")}
    return z, sd, after, cfg


@pytest.mark.parametrize("name", ["vqvae_init_tiled", "vqvae_init"])
def test_vqvae_codebook_data_init_matches_reference(name):
    """The first train-mode forward of a fresh model: Codebook._init_embeddings (videogpt_vq_vae.py:160-172) + _tile's repeat and
    jitter (:151-158, the `_tiled` fixture has 16 latents for 24 codes) with the reference's permutations and noise injected."""
    z, sd, after, cfg = _load_state_fixture(name)
    noise = {k: torch.from_numpy(z[k]) for k in ("noise_init", "noise") if k in z.files}
    with torch.no_grad():
        out, new = vqvae.forward_train(torch.from_numpy(z["x"]), sd, cfg, z["perm"], init_perm=z["perm_init"],
                                       init_noise=noise.get("noise_init"), noise=noise.get("noise"))
    np.testing.assert_allclose(out["pred_data"].numpy(), z["pred"], atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(out["losses"]["recon_loss"].item(), z["recon_loss"], rtol=1e-5)
    np.testing.assert_allclose(out["losses"]["commitment_loss"].item(), z["commitment_loss"], rtol=1e-5)
    for k in ("codebook.embeddings", "codebook.N", "codebook.z_avg"):
        np.testing.assert_allclose(new[k].numpy(), after[k].numpy(), atol=2e-6, rtol=1e-5, err_msg=k)
    assert (new["codebook.N"] < 1).any() == (name == "vqvae_init_tiled")          # the tiled case restarts dead codes


def test_near_tie_vectors_oracle_follows_the_reference(golden):
    """tests/golden/neartie.npz: arg-min / Gumbel arg-max decided on margins of 1e-7 ... 1e-3.  The oracle runs the same torch-CPU
    ops as the reference, so it must reproduce the reference's choice on EVERY vector (ties included); the fp64 evaluation shows
    which of those choices fp32 rounding made."""
    sd, a, cfg = golden("neartie")
    z = torch.from_numpy(a["cb/z"])
    idx, _ = vqvae.nearest_code(z, torch.from_numpy(a["cb/codebook"]))
    assert np.array_equal(idx.view(-1).numpy(), a["cb/ref_idx"])
    flips = a["cb/ref_idx"] != a["cb/winner64"]
    assert flips.any() and np.abs(a["cb/margin64"][flips]).max() < 2e-5           # the reference itself is fp32-noisy below that
    K = cfg["K"]
    lc, lu = torch.from_numpy(a["gum/logits_c"]), torch.from_numpy(a["gum/logits_u"])
    xt, t = torch.from_numpy(a["gum/xt"]), torch.from_numpy(a["gum/t"])
    rec = d3pm.cf_mix(d3pm.predict_start_from_logits(lc)[:, :-1], d3pm.predict_start_from_logits(lu)[:, :-1], cfg["guidance"])
    post = d3pm.q_posterior(rec, d3pm.index_to_log_onehot(xt, K + 1), t, sd)
    tok = d3pm.gumbel_argmax(post, cfg["noise_seed"], int(a["gum/stream"]))
    assert np.array_equal(tok.numpy(), a["gum/ref_tok"])
    flips = a["gum/ref_tok"] != a["gum/winner64"]
    assert np.abs(a["gum/margin64"][flips]).max() < 5e-6
    assert (np.abs(a["gum/margin64"]) < 1e-4).sum() >= 20                          # the stress actually happens


def test_product_schedule_and_sample_time_match_reference(golden):
    """The PRODUCT's own schedule buffers (gsdd_amd.d3pm.DiffusionTransformer.__init__, not the oracle's) bit-equal the reference's
    (diffusion_transformer.py:115-149), and its sample_time('importance') (:368-389) draws the reference's (t, pt) under the same
    torch seed once every Lt_count passed 10 -- and falls back to uniform while one has not."""
    import gsdd_amd
    sd, a, cfg = golden("d3pm_L64")
    d = gsdd_amd.DalleMaskImageEmbedding(num_embed=cfg["K"], spatial_size=cfg["spatial"], embed_dim=64)
    tr = gsdd_amd.Text2ImageTransformer(dalle=d, n_layer=1, n_embd=64, n_head=16, content_seq_len=cfg["L"], block_activate="GELU2",
                                        content_spatial_size=cfg["spatial"], condition_dim=cfg["cond_dim"], diffusion_step=cfg["T"])
    dm = gsdd_amd.DiffusionTransformer(transformer=tr, diffusion_step=cfg["T"], alpha_init_type="alpha1", guidance_scale=2,
                                       content_seq_len=cfg["L"])
    for name in gsdd_amd.d3pm.SCHED_ORDER:
        assert torch.equal(getattr(dm, name), sd[name]), name
    assert dm.Lt_history.shape == sd["Lt_history"].shape and dm.empty_text_embed.dtype == torch.float64
    _, n, ncfg = golden("neartie")
    dm.Lt_count.fill_(11.0)
    dm.Lt_history.copy_(torch.from_numpy(n["st/Lt_history"]))
    torch.manual_seed(int(n["st/seed"]))
    t, pt = dm.sample_time(64, "cpu", "importance")
    assert np.array_equal(t.numpy(), n["st/t"]) and np.array_equal(pt.numpy(), n["st/pt"])
    assert len(np.unique(n["st/pt"])) > 10                                         # genuinely non-uniform
    dm.Lt_count[3] = 10.0
    torch.manual_seed(int(n["st/seed_uniform"]))
    t, pt = dm.sample_time(64, "cpu", "importance")
    assert np.array_equal(t.numpy(), n["st/t_uniform"]) and np.array_equal(pt.numpy(), n["st/pt_uniform"])


@pytest.mark.parametrize("tag", ["zero", "cond"])
def test_glue_oracle_matches_reference(golden, tag):
    """discrete_diffusion.py:16-83 restated with the oracle's pieces against the reference's own output dict (glue_L64.npz)."""
    _, a, cfg = golden("glue_L64")
    sdv, av, cfgv = golden(str(cfg["vqvae"]))
    sdd, ad, cfgd = golden(str(cfg["d3pm"]))
    x = torch.from_numpy(av["x"])
    B = x.shape[0]
    rows = {str(t): torch.from_numpy(a["text_table"][i]) for i, t in enumerate(list(a["texts"]) + [""])}
    emb = torch.stack([rows[str(t)] for t in a["texts"]]).unsqueeze(1)
    cf = torch.stack([rows[""]] * B).unsqueeze(1)
    if tag == "zero":
        emb, cf = torch.zeros_like(emb), torch.zeros_like(cf)
    t = torch.from_numpy(a["t"])
    with torch.no_grad():
        quant = vqvae.encode(x, sdv, cfgv)
        loss, _, pred, _ = d3pm.train_loss(quant.view(B, -1), emb, t, torch.ones(B) / cfgd["T"], sdd, cfg["noise_seed"], int(cfg["stream"]))
        tok = d3pm.sample(B, cfgd["L"], emb, cf, sdd, cfgd["guidance"], cfg["noise_seed"], stream0=int(cfg["stream"]) + 1)
        single = vqvae.decode(pred.view(quant.shape), sdv, cfgv)
        clips = vqvae.decode(tok.view(quant.shape), sdv, cfgv)
        test = vqvae.decode(quant, sdv, cfgv)
    np.testing.assert_allclose(loss.item(), a[f"{tag}/losses"], rtol=1e-5)
    assert np.array_equal(tok.numpy(), a[f"{tag}/content_token"])
    np.testing.assert_allclose(single.numpy(), a[f"{tag}/pred_single_step"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(clips.numpy(), a[f"{tag}/pred_data"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(test.numpy(), a[f"{tag}/test"], atol=2e-5, rtol=1e-5)




# ----------------------------------------------------------------------------- I3D (FVD feature extractor)
def i3d_fixture():
    """-> (npz, seeded state_dict over the reference's keys): the weights are rebuilt from the fixture's seed, not stored."""
    import os
    from oracle import i3d as oi
    from tests.conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "i3d.npz"), allow_pickle=False)
    keys = [(str(k), tuple(int(v) for v in s[:n])) for k, s, n in zip(z["keys"], z["shapes"], z["ndims"])]
    return z, oi.seeded_state_dict(keys, int(z["weight_seed"]))


def test_i3d_oracle_matches_reference_outputs():
    """oracle/i3d.py against outputs of the reference's InceptionI3d (tests/golden/make_golden_i3d.py): every end point, the pooled
    features and the time-averaged logits of a 16-frame 224x224 clip (the evaluator's shape)."""
    from oracle import i3d as oi
    z, sd = i3d_fixture()
    B, T, seed = z["x_a"].tolist()
    x = torch.randn(B, 3, T, 224, 224, generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        eps = {}
        feats = oi.extract_features(x, sd, eps)
        logits = oi.forward(x, sd)
    for name, _, _ in oi.ENDPOINTS:
        torch.testing.assert_close(eps[name][:, ::7, ::3, ::5, ::5], torch.from_numpy(z["ep_" + name]), atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(feats, torch.from_numpy(z["features_a"]), atol=1e-4, rtol=1e-5)
    torch.testing.assert_close(logits, torch.from_numpy(z["logits_a"]), atol=1e-4, rtol=1e-5)


def test_i3d_state_dict_keys_are_the_references():
    """The product module owns parameters under the reference's names, in the reference's order and shapes (a checkpoint made
    for src/models/motionencoder/pytorch_i3d.py loads unchanged); host logic only, no GPU."""
    import gsdd_amd
    from src.models.motionencoder.pytorch_i3d import InceptionI3d
    assert InceptionI3d is gsdd_amd.InceptionI3d
    z, sd = i3d_fixture()
    m = InceptionI3d()
    assert list(m.state_dict().keys()) == [str(k) for k in z["keys"]]
    m.load_state_dict(sd)                                                   # strict
    with pytest.raises(gsdd_amd.GsddError):
        m.eval()(torch.zeros(1, 3, 16, 224, 224))                           # no CPU fallback
