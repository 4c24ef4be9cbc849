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
