"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
(1) the committed golden fixtures produced by the reference and (2) the CPU oracle on seeded inputs.
Bars: token / code indices bit-exact, fp32 activations / logits within 1e-4 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from tests.conftest import parity_report

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4


@pytest.fixture(scope="module")
def G():
    import gsdd_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    gsdd_amd.lib()          # fail loudly if libgsdd.so is missing
    return gsdd_amd


def dev(x):
    return torch.as_tensor(x).cuda()


# ----------------------------------------------------------------------------- noise source
def test_philox_device_matches_oracle(G):
    from oracle import philox
    for (rows, cols, row0, stream) in [(7, 33, 0, 0), (128, 4097, 5, 99), (3, 4, 1 << 33, 7)]:
        got = G.ops.philox_uniform(1234, stream, rows, cols, "cuda", row0=row0).cpu().numpy()
        want = philox.uniform_rows(1234, stream, rows, cols, row0=row0)
        assert np.array_equal(got, want)


# ----------------------------------------------------------------------------- implicit GEMM vs torch conv
@pytest.mark.parametrize("cin,cout,k,stride", [(16, 16, 4, (1, 2, 2)), (32, 48, 4, (2, 2, 2)), (16, 8, 3, (1, 1, 1)),
                                                (64, 136, 1, (1, 1, 1)), (40, 200, 3, (1, 1, 1))])
def test_gemm_conv3d(G, cin, cout, k, stride):
    from oracle import vqvae as ov
    V = __import__("gsdd_amd").vqvae
    g = torch.Generator().manual_seed(0)
    B, T, H, W = 2, 4, 8, 8
    x = torch.randn(B, cin, T, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, k, generator=g) / (cin * k ** 3) ** 0.5
    b = torch.randn(cout, generator=g)
    want = ov.same_pad_conv3d(x, w, b, stride)
    pf = tuple((k - s) // 2 + (k - s) % 2 for s in stride)
    xr = dev(x.permute(0, 2, 3, 4, 1).contiguous())
    To, Ho, Wo = T // stride[0], H // stride[1], W // stride[2]
    out = torch.empty((B * To * Ho * Wo, cout), device="cuda")
    taps = G.ops.taps_tensor(V.conv_taps((k, k, k), stride, pf), "cuda")
    G.ops.gemm(xr, dev(V.pack_conv_weight(w)), out, in_dims=(B, T, H, W), out_grid=(To, Ho, Wo), stride=stride,
               taps=taps, ntaps=k ** 3, epi_shift=dev(b))
    got = out.view(B, To, Ho, Wo, cout).permute(0, 4, 1, 2, 3).cpu()
    torch.testing.assert_close(got, want, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("stride", [(1, 2, 2), (2, 2, 2)])
def test_gemm_conv_transpose3d(G, stride):
    from oracle import vqvae as ov
    V = __import__("gsdd_amd").vqvae
    g = torch.Generator().manual_seed(1)
    B, cin, cout, T, H, W = 2, 16, 24, 3, 4, 5
    x = torch.randn(B, cin, T, H, W, generator=g)
    w = torch.randn(cin, cout, 4, 4, 4, generator=g) / (cin * 16) ** 0.5
    b = torch.randn(cout, generator=g)
    want = ov.same_pad_convT3d(x, w, b, stride)
    pf = tuple((4 - s) // 2 + (4 - s) % 2 for s in stride)
    xr = dev(x.permute(0, 2, 3, 4, 1).contiguous())
    To, Ho, Wo = T * stride[0], H * stride[1], W * stride[2]
    out = torch.zeros((B * To * Ho * Wo, cout), device="cuda")
    for (ph, ks, offs) in V.convT_phases((4, 4, 4), stride, pf):
        G.ops.gemm(xr, dev(V.pack_convT_weight(w, ks)), out, in_dims=(B, T, H, W), out_grid=(T, H, W),
                   taps=G.ops.taps_tensor(offs, "cuda"), ntaps=len(ks), epi_shift=dev(b), out_dims=(To, Ho, Wo),
                   out_step=stride, out_off=ph)
    got = out.view(B, To, Ho, Wo, cout).permute(0, 4, 1, 2, 3).cpu()
    torch.testing.assert_close(got, want, atol=2e-5, rtol=1e-5)


# ----------------------------------------------------------------------------- VQ-VAE vs reference fixtures
def build_vqvae(G, sd, cfg):
    m = G.VQVAE(None, cfg["embedding_dim"], cfg["n_codes"], cfg["n_hiddens"], cfg["n_res_layers"], cfg["downsample"],
                cfg["sequence_length"], cfg["resolution"])
    m.load_state_dict(sd)
    return m.cuda().eval()


@pytest.mark.parametrize("name", ["vqvae_ds188", "vqvae_ds244"])
def test_vqvae_matches_reference(G, golden, name):
    sd, a, cfg = golden(name)
    m = build_vqvae(G, sd, cfg)
    x = dev(a["x"])
    z, dims = m._encode_rows(x)
    z_ref = torch.from_numpy(a["z"]).permute(0, 2, 3, 4, 1).reshape(-1, cfg["embedding_dim"])
    torch.testing.assert_close(z.cpu(), z_ref, atol=5e-5, rtol=1e-4)
    enc, emb = m.encode(x, include_embeddings=True)
    assert enc.dtype == torch.int64 and tuple(enc.shape) == a["encodings"].shape
    mism = (enc.cpu().numpy() != a["encodings"])
    # a differing index is only tolerated at a genuine near-tie of the reference's own distances
    assert not (mism.reshape(-1) & (a["argmin_margin"] > 1e-4)).any()
    assert mism.sum() == 0, f"{mism.sum()} code indices differ (all near-ties)"
    torch.testing.assert_close(emb.cpu(), torch.from_numpy(a["embeddings"]), atol=1e-5, rtol=1e-5)
    rec = m.decode(dev(a["encodings"]))
    torch.testing.assert_close(rec.cpu(), torch.from_numpy(a["decoded"]), atol=1e-4, rtol=1e-4)
    out = m({"video": x})
    torch.testing.assert_close(out["pred_data"].cpu(), torch.from_numpy(a["fwd_pred"]), atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(out["losses"]["recon_loss"].item(), a["fwd_recon_loss"], rtol=1e-4)
    np.testing.assert_allclose(out["losses"]["commitment_loss"].item(), a["fwd_commitment_loss"], rtol=1e-4)


def test_nearest_code_vs_oracle(G):
    from oracle import vqvae as ov
    g = torch.Generator().manual_seed(3)
    M, E, K = 1000, 128, 4096
    z = torch.randn(M, E, generator=g)
    cb = torch.randn(K, E, generator=g)
    idx = torch.empty(M, dtype=torch.int64, device="cuda")
    zq = torch.empty(M, E, device="cuda")
    G.ops.nearest_code(dev(z), dev(cb), idx, zq)
    want, d = ov.nearest_code(z.view(M, E, 1, 1, 1), cb)
    want = want.view(-1)
    top2 = torch.topk(d, 2, dim=1, largest=False).values
    margin = top2[:, 1] - top2[:, 0]
    mism = idx.cpu() != want
    parity_report("nearest_code_1000x4096x128", {"mismatches": int(mism.sum()), "min_margin": margin.min().item(),
                                                 "mismatch_margins": margin[mism].tolist()})
    assert all(mm < 2e-4 for mm in margin[mism].tolist())          # distances ~256 here: a near-tie of the oracle's own fp32 values
    assert mism.sum() == 0, margin[mism].tolist()
    assert torch.equal(zq.cpu()[~mism], cb[want[~mism]])
    # exact ties: duplicated codebook rows -> first index wins (torch.argmin rule)
    cb2 = torch.cat([cb[:8], cb[:8]], 0)
    idx2 = torch.empty(M, dtype=torch.int64, device="cuda")
    G.ops.nearest_code(dev(z), dev(cb2), idx2, None)
    assert (idx2 < 8).all()
    # the same on the matrix-core kernel (K % 32 == 0): the duplicate halves sit in different codebook splits / workgroups
    cb3 = torch.cat([cb[:2048], cb[:2048]], 0)
    idx3 = torch.empty(M, dtype=torch.int64, device="cuda")
    G.ops.nearest_code(dev(z), dev(cb3), idx3, None)
    assert (idx3 < 2048).all()


@pytest.mark.parametrize("M", [32768, 4096 + 77, 64])
def test_nearest_code_matrix_kernel(G, M):
    """The matrix-core kernel (z E^T on v_mfma_f32_32x32x2_f32, arg-min epilogue, codebook split over workgroups, 64-bit-key merge)
    against the register-tiled vector kernel and fp64: C2's shape (32768 latents x 4096 codes x 128), a ragged row count, a tiny one.
    The two kernels sum the 128 products in different orders, so they may differ where the fp32 distances are a near-tie."""
    g = torch.Generator().manual_seed(M)
    E, K = 128, 4096
    z = torch.randn(M, E, generator=g)
    cb = torch.randn(K, E, generator=g) * 0.7 + 0.3 * z[torch.randint(0, M, (K,), generator=g)]     # codes near latents: tight races
    zd, cd = dev(z), dev(cb)
    idx_m = torch.empty(M, dtype=torch.int64, device="cuda")
    idx_v = torch.empty(M, dtype=torch.int64, device="cuda")
    zq = torch.empty(M, E, device="cuda")
    G.ops.nearest_code(zd, cd, idx_m, zq)
    G.ops.nearest_code(zd, cd, idx_v, None, matrix=False)
    d64 = (zd.double() ** 2).sum(1, keepdim=True) - 2 * zd.double() @ cd.double().t() + (cd.double() ** 2).sum(1)[None]
    top2 = torch.topk(d64, 2, dim=1, largest=False)
    margin = (top2.values[:, 1] - top2.values[:, 0]).cpu()
    want = top2.indices[:, 0]
    mm, mv = (idx_m != want).cpu(), (idx_v != want).cpu()
    parity_report(f"nearest_code_matrix_kernel[M={M}]", {"mismatches_vs_fp64": int(mm.sum()), "vector_kernel_mismatches_vs_fp64": int(mv.sum()),
                                                         "kernels_differ": int((idx_m != idx_v).sum()), "min_margin": margin.min().item(),
                                                         "mismatch_margins": margin[mm].tolist()})
    assert 0 <= int(idx_m.min()) and int(idx_m.max()) < K
    assert all(x < 2e-4 for x in margin[mm].tolist()), margin[mm].tolist()     # distances ~2e2: an fp32 ulp is 1.5e-5
    assert torch.equal(zq, cd[idx_m])


# ----------------------------------------------------------------------------- D3PM vs reference fixtures
def build_d3pm(G, sd, cfg):
    d = G.DalleMaskImageEmbedding(num_embed=cfg["K"], spatial_size=cfg["spatial"], embed_dim=64)
    tr = G.Text2ImageTransformer(dalle=d, n_layer=cfg["n_layer"], n_embd=64, n_head=16, content_seq_len=cfg["L"],
                                 block_activate="GELU2", content_spatial_size=cfg["spatial"],
                                 condition_dim=cfg["cond_dim"], diffusion_step=cfg["T"])
    dm = G.DiffusionTransformer(transformer=tr, diffusion_step=cfg["T"], alpha_init_type="alpha1",
                                auxiliary_loss_weight=5e-4, adaptive_auxiliary_loss=True,
                                guidance_scale=cfg["guidance"], content_seq_len=cfg["L"])
    missing = dm.load_state_dict(sd, strict=False)
    assert missing.missing_keys == ["empty_text_embed"] and not missing.unexpected_keys
    return dm.cuda().eval()


def test_denoiser_logits_match_reference(G, golden):
    sd, a, cfg = golden("d3pm_L64")
    dm = build_d3pm(G, sd, cfg)
    xt, cond, t = dev(a["step_xt"]), dev(a["step_cond"]), dev(a["step_t"])
    logits = dm.transformer(xt, cond, t)
    assert tuple(logits.shape) == a["step_logits"].shape
    torch.testing.assert_close(logits.cpu(), torch.from_numpy(a["step_logits"]), atol=LOGIT_TOL, rtol=0)
    lu = dm.transformer(xt, torch.zeros_like(cond), t)
    torch.testing.assert_close(lu.cpu(), torch.from_numpy(a["step_logits_uncond"]), atol=LOGIT_TOL, rtol=0)
    l3 = dm.transformer(xt, dev(a["cond3"]), t)                        # general cross-attention, Te = 3
    torch.testing.assert_close(l3.cpu(), torch.from_numpy(a["logits_cond3"]), atol=LOGIT_TOL, rtol=0)


def test_reverse_step_matches_reference(G, golden):
    sd, a, cfg = golden("d3pm_L64")
    dm = build_d3pm(G, sd, cfg)
    dm.set_noise(cfg["noise_seed"])
    B, L, K1 = cfg["B"], cfg["L"], cfg["K"] + 1
    xt, cond, t = dev(a["step_xt"]), dev(a["step_cond"]), dev(a["step_t"])
    post = torch.empty((B, K1, L), device="cuda")
    x0 = torch.empty((B, K1, L), device="cuda")
    tok = dm.p_sample_tokens(xt, cond, torch.zeros_like(cond), t, int(a["step_stream"]), post_dbg=post, x0_dbg=x0)
    torch.testing.assert_close(x0.cpu(), torch.from_numpy(a["step_cf_predict_start"]), atol=2e-4, rtol=0)
    torch.testing.assert_close(post.cpu(), torch.from_numpy(a["step_posterior"]), atol=2e-4, rtol=0)
    mism = tok.cpu().numpy() != a["step_sample"]
    assert not (mism & (a["step_margin"] > 1e-3)).any()
    assert mism.sum() == 0


def test_step_kernel_alone_matches_oracle(G, golden):
    """The fused posterior/Gumbel kernel fed with the reference's own logits: isolates it from the denoiser."""
    from oracle import d3pm as od
    sd, a, cfg = golden("d3pm_L64")
    dm = build_d3pm(G, sd, cfg)
    B, L, K = cfg["B"], cfg["L"], cfg["K"]
    lc = dev(np.ascontiguousarray(a["step_logits"].transpose(0, 2, 1))).view(B * L, K)
    lu = dev(np.ascontiguousarray(a["step_logits_uncond"].transpose(0, 2, 1))).view(B * L, K)
    xt, t = dev(a["step_xt"]), dev(a["step_t"])
    sid = torch.tensor([int(a["step_stream"])], dtype=torch.int64, device="cuda")
    out = torch.empty_like(xt)
    post = torch.empty((B, K + 1, L), device="cuda")
    G.ops.d3pm_step(lc, lu, xt, out, dm._sched(), t, sid, K=K, T=cfg["T"], guidance=2.0, seed=cfg["noise_seed"],
                    post_dbg=post)
    torch.testing.assert_close(post.cpu(), torch.from_numpy(a["step_posterior"]), atol=2e-5, rtol=0)
    assert np.array_equal(out.cpu().numpy(), a["step_sample"])
    # unguided variant (predict_start only) against the oracle
    G.ops.d3pm_step(lc, None, xt, out, dm._sched(), t, sid, K=K, T=cfg["T"], guidance=2.0, seed=cfg["noise_seed"],
                    post_dbg=post)
    sd_cpu = {k: v for k, v in sd.items()}
    log_xt = od.index_to_log_onehot(torch.from_numpy(a["step_xt"]), K + 1)
    ps = od.predict_start_from_logits(torch.from_numpy(a["step_logits"]))
    want = od.q_posterior(ps, log_xt, torch.from_numpy(a["step_t"]), sd_cpu)
    torch.testing.assert_close(post.cpu(), want, atol=2e-5, rtol=0)


@pytest.mark.parametrize("K", [4096, 768])
def test_step_kernel_full_width_matches_oracle(G, K):
    """K = 4096 takes the kernel instantiation without validity selects and without test hooks (the one the sampler
    runs at the benchmark size); K = 768 a partially filled register grid.  Same inputs through the hooked
    instantiation (posterior values) and through the oracle: tokens must agree bit for bit, guided and unguided,
    with masked and unmasked x_t and t = 0 in the batch."""
    from oracle import d3pm as od
    from gsdd_amd.d3pm import SCHED_ORDER
    B, L, T, seed, stream = 3, 8, 100, 4321, 7
    g = torch.Generator().manual_seed(K)
    lc = torch.randn(B, K, L, generator=g) * 3.0
    lu = lc + torch.randn(B, K, L, generator=g)
    xt = torch.randint(0, K, (B, L), generator=g)
    xt[:, ::3] = K                                             # [MASK]
    t = torch.tensor([57, 0, 99])
    sd = od.schedule_buffers(T, K)
    sched = [dev(sd[n]) for n in SCHED_ORDER]
    rows = lambda x: dev(np.ascontiguousarray(x.numpy().transpose(0, 2, 1))).view(B * L, K)
    sid = torch.tensor([stream], dtype=torch.int64, device="cuda")
    log_xt = od.index_to_log_onehot(xt, K + 1)
    for guided in (True, False):
        rec = od.cf_mix(od.predict_start_from_logits(lc)[:, :-1], od.predict_start_from_logits(lu)[:, :-1], 2.0) \
            if guided else od.predict_start_from_logits(lc)
        want_post = od.q_posterior(rec, log_xt, t, sd)
        want_tok = od.gumbel_argmax(want_post, seed, stream)
        post = torch.empty((B, K + 1, L), device="cuda")
        hooked, plain = torch.empty_like(xt).cuda(), torch.empty_like(xt).cuda()
        args = (rows(lc), rows(lu) if guided else None, dev(xt))
        kw = dict(K=K, T=T, guidance=2.0, seed=seed)
        G.ops.d3pm_step(*args, hooked, sched, dev(t), sid, post_dbg=post, **kw)
        G.ops.d3pm_step(*args, plain, sched, dev(t), sid, **kw)
        torch.testing.assert_close(post.cpu(), want_post, atol=2e-5, rtol=0)
        assert np.array_equal(hooked.cpu().numpy(), want_tok.numpy())
        assert np.array_equal(plain.cpu().numpy(), want_tok.numpy())


def test_full_reverse_loop_tokens_bit_exact(G, golden):
    sd, a, cfg = golden("d3pm_L64")
    dm = build_d3pm(G, sd, cfg)
    cond = dev(a["step_cond"])
    B = cfg["B"]
    # eager loop with a per-step trace against the reference's trace
    dm.set_noise(cfg["noise_seed"])
    trace = []
    out = dm.sample(["a"] * B, None, cond, torch.zeros_like(cond), filter_ratio=0, trace=trace)
    got = np.stack([x.cpu().numpy() for x in trace])
    first_bad = np.nonzero((got != a["loop_trace"]).reshape(got.shape[0], -1).any(1))[0]
    assert first_bad.size == 0, f"token trace diverges at reverse step {first_bad[0]}"
    assert np.array_equal(out["content_token"].cpu().numpy(), a["loop_tokens"])
    # hipGraph-captured loop must give the same tokens
    dm.set_noise(cfg["noise_seed"])
    out_g = dm.sample(["a"] * B, None, cond, torch.zeros_like(cond), filter_ratio=0, use_graph=True)
    assert np.array_equal(out_g["content_token"].cpu().numpy(), a["loop_tokens"])


def test_identical_guidance_copies_run_once(G, golden, monkeypatch):
    """The reference's shipped inference path zeroes both the conditional and the unconditional embedding (discrete_diffusion.py:25, :49):
    the two guidance copies are then the same computation, and sample() runs one of them, feeding its logits to both sides of the guided
    mix.  Tokens must equal the two-copy run's (GSDD_CFG_DEDUPE=0) over the whole captured 100-step loop; different embeddings must
    keep both copies."""
    sd, a, cfg = golden("d3pm_L64")
    dm = build_d3pm(G, sd, cfg)
    B = cfg["B"]
    zero = torch.zeros_like(dev(a["step_cond"]))
    toks = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("GSDD_CFG_DEDUPE", mode)
        dm.set_noise(cfg["noise_seed"])
        toks[mode] = dm.sample(["a"] * B, None, zero, zero.clone(), filter_ratio=0, use_graph=True)["content_token"].cpu().numpy()
        assert dm._last_cfg_dedupe == (mode == "1")
    assert np.array_equal(toks["1"], toks["0"])
    monkeypatch.setenv("GSDD_CFG_DEDUPE", "1")
    dm.set_noise(cfg["noise_seed"])
    out = dm.sample(["a"] * B, None, dev(a["step_cond"]), zero, filter_ratio=0, use_graph=True)
    assert not dm._last_cfg_dedupe and np.array_equal(out["content_token"].cpu().numpy(), a["loop_tokens"])


@pytest.mark.parametrize("attention_mode", [None, "11"])
def test_long_sequence_loop_tokens_bit_exact_vs_reference(G, golden, attention_mode):
    """The reference's own 100-step reverse loop at L = 2048 (`d3pm_L2048`: one clip, two layers, K = 32, trained-like weights),
    where the attention kernel runs its default adaptive P arithmetic (f16 hi, lo where the norm bound / the measured test ask for
    it): the eager per-step trace and the captured hipGraph's final tokens equal the reference's, token for token, and the
    teacher-forced step's logits are within 1e-4.  This is the reference-generated pin of that arithmetic over a whole chain -- and,
    with attention_mode '11' (P as f16 hi only in every tile), of the documented fast mode."""
    sd, a, cfg = golden("d3pm_L2048")
    dm = build_d3pm(G, sd, cfg)
    dm.transformer.attention_mode = attention_mode
    cond = dev(a["step_cond"])
    B = cfg["B"]
    logits = dm.transformer(dev(a["step_xt"]), cond, dev(a["step_t"]))
    lerr = (logits.cpu() - torch.from_numpy(a["step_logits"])).abs().max().item()
    dm.set_noise(cfg["noise_seed"])
    trace = []
    out = dm.sample(["a"] * B, None, cond, torch.zeros_like(cond), filter_ratio=0, trace=trace)
    got = np.stack([x.cpu().numpy() for x in trace])
    bad = (got != a["loop_trace"]).reshape(got.shape[0], -1)
    parity_report("long_sequence_loop_L2048" + ("" if attention_mode is None else f"[attention_mode={attention_mode}]"), {"logits_err": lerr, "steps": int(got.shape[0]), "positions": int(got.shape[1] * got.shape[2]),
                                               "trace_mismatches": int(bad.sum()), "first_bad_step": int(np.nonzero(bad.any(1))[0][0]) if bad.any() else -1,
                                               "min_step_margin": float(a["step_margin"].min()), "redo_events": dm.attention_redo_events()})
    assert lerr < LOGIT_TOL, lerr
    assert not bad.any(), f"token trace diverges at reverse step {np.nonzero(bad.any(1))[0][0]} ({int(bad.sum())} tokens in all)"
    assert np.array_equal(out["content_token"].cpu().numpy(), a["loop_tokens"])
    dm.set_noise(cfg["noise_seed"])
    out_g = dm.sample(["a"] * B, None, cond, torch.zeros_like(cond), filter_ratio=0, use_graph=True)
    assert np.array_equal(out_g["content_token"].cpu().numpy(), a["loop_tokens"])


# ----------------------------------------------------------------------------- attention kernel vs fp64
def attention_ref(q, k, v):
    att = torch.softmax((q.double() @ k.double().transpose(-1, -2)) * 0.5, dim=-1)
    return (att @ v.double())


@pytest.mark.parametrize("use_ws", [True, False])
@pytest.mark.parametrize("B,L,spike", [(2, 64, False), (1, 1024, False), (2, 4096, False), (1, 512, True), (1, 48, False),
                                       (2, 37, False), (1, 301, False),             # ragged lengths take the VALU kernel
                                       (1, 2048, False), (1, 2080, False),          # adaptive mode with a full / partial last chunk
                                       (1, 2304, True), (1, 4096, "grow")])         # overflow redo, then the pre-scan path, in adaptive mode
def test_d3pm_attention(G, B, L, spike, use_ws):
    H = 16
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B, H, L, 4, generator=g) * 1.5
    k = torch.randn(B, H, L, 4, generator=g) * 1.5
    v = torch.randn(B, H, L, 4, generator=g)
    if spike == "grow":   # keys that grow along the sequence: every chunk holds a new row maximum (redo once, then pre-scans)
        k = k * torch.linspace(0.2, 6.0, L).view(1, 1, L, 1)
    elif spike:     # force the rare running-max raise: one key far above the first tile's maximum
        k[:, :, L // 2 + 3] = q[:, :, 7] * 40.0
        k[:, :, 5] *= 0.01
    want = attention_ref(q, k, v).permute(0, 2, 1, 3).reshape(B * L, H * 4)
    hm = lambda z: dev(z.permute(1, 0, 2, 3).reshape(H, B * L, 4).contiguous())
    out = torch.empty((B * L, H * 4), device="cuda")
    ws = G.ops.d3pm_attention_workspace(B, L, H, "cuda") if use_ws else None   # matrix-pipe kernel / exact-f32 P.V kernel
    G.ops.d3pm_attention(hm(q), hm(k), hm(v), B, L, H, out, ws=ws)
    err = (out.cpu().double() - want).abs().max().item()
    assert err < 2e-5, err


@pytest.mark.parametrize("case", ["flat", "hot_tile", "hot_query", "growing_norms", "zero_q", "mean_shift", "two_clusters", "unit_scale",
                                  "common_mean", "common_mean_hot_tile", "common_mean_drift"])
def test_attention_norm_bound(G, case, monkeypatch):
    """The adaptive mode's bounds (kernel note in d3pm_attention.hip): tiles whose ||q|| ||k|| bound keeps every probability below 2^-8
    of the row sum -- the row sum so far, or the lower bound of the final one from the mean key (Jensen) -- take the f16 hi half only,
    decided without looking at the scores; where the bound does not decide, the measured test compares with the same row sums.  Near-flat rows (the reference init) clear every
    chunk after the first; a tile of large keys, or a query of large norm inside a sub-tile, must not be cleared: the result has to
    stay within the kernel's 2e-5 of fp64 and within the adaptive mode's budget of the all-hi+lo result."""
    B, L, H = 1, 4096, 16
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, H, L, 4, generator=g) * 0.15
    k = torch.randn(B, H, L, 4, generator=g) * 0.15
    v = torch.randn(B, H, L, 4, generator=g)
    if case == "hot_tile":          # one late pair-tile of keys 15x the others, aligned with some queries: probabilities of a few
        k[:, :, 77 * 32:78 * 32] = q[:, :, 100:132] * 100.0       # percent of their rows there, 1/4096 everywhere else
    elif case == "hot_query":       # one query per sub-tile with a norm 30x its neighbours'
        q[:, :, 5::16] *= 30.0
    elif case == "growing_norms":
        k = k * torch.linspace(0.5, 12.0, L).view(1, 1, L, 1)
    elif case == "zero_q":
        q[:, :, :64] = 0.0
    elif case == "mean_shift":      # keys with a large common component: q . kmean is +-6 bits, the row-sum lower bound (Jensen) moves
        k = k + torch.tensor([3.0, -2.0, 1.0, 0.5])               # every query's threshold, up for aligned queries and down for opposed ones
        q = q * 4.0
    elif case == "two_clusters":    # half of the keys far from the other half: log-mean-exp is far above the mean score, the bound is
        k[:, :, ::2] += torch.tensor([4.0, 0.0, 0.0, 0.0])        # loose (never wrong) and the measured test has to do the work
        q = q * 6.0
    elif case == "unit_scale":      # structureless unit-scale q, k (trained-like spread): a quarter of the tiles hold a probability
        q, k = q / 0.15, k / 0.15                                 # above the threshold
    elif case.startswith("common_mean"):
        # the trained-like regime: keys = a large common vector + a small spread, queries of unit scale.  ||q'|| ||k|| is ~10 bits, so
        # the norm bound proves nothing although no tile holds a large probability (||q'|| ||k - kmean|| ~1 bit): the measured test decides
        k = k * 3.0 + torch.tensor([4.0, -3.0, 2.0, 1.0])
        q = q * 12.0
        if case == "common_mean_hot_tile":   # ... except one tile whose keys sit far from the mean along some queries: it must keep its lo half
            k[:, :, 77 * 32:78 * 32] += q[:, :, 100:132] * 2.0
        elif case == "common_mean_drift":    # ... and a mean that drifts along the row
            k = k + torch.linspace(-1.5, 1.5, L).view(1, 1, L, 1) * torch.tensor([1.0, 0.5, -0.5, 0.25])
    want = attention_ref(q, k, v).permute(0, 2, 1, 3).reshape(B * L, H * 4)
    hm = lambda z: dev(z.permute(1, 0, 2, 3).reshape(H, B * L, 4).contiguous())
    outs = {}
    for mode in ("a8", "22"):
        monkeypatch.setenv("GSDD_ATTN_P", mode)
        out = torch.empty((B * L, H * 4), device="cuda")
        G.ops.d3pm_attention(hm(q), hm(k), hm(v), B, L, H, out, ws=G.ops.d3pm_attention_workspace(B, L, H, "cuda"))
        outs[mode] = out.cpu().double()
    err = (outs["a8"] - want).abs().max().item()
    dev22 = (outs["a8"] - outs["22"]).abs().max().item()
    parity_report(f"attention_norm_bound_{case}", {"err_vs_fp64": err, "max_dev_from_hi_lo": dev22})
    assert err < 2e-5 and dev22 < 2e-5, (err, dev22)


def test_missing_cpu_fallback_is_loud(G):
    m = G.VQVAE(None, 8, 32, 16, 1, [1, 4, 4], 4, 16).eval()
    with pytest.raises(G.GsddError):
        m.encode(torch.randn(1, 3, 4, 16, 16))


# ----------------------------------------------------------------------------- training objective (forward value)
def test_train_loss_matches_reference(G, golden):
    sd, a, cfg = golden("d3pm_L64")
    dm = build_d3pm(G, sd, cfg)
    B, T = cfg["B"], cfg["T"]
    t_fix = dev(a["train_t"])
    dm.sample_time = lambda b, device, method="uniform": (t_fix, torch.ones(b, device="cuda") / T)
    dm.set_noise(cfg["noise_seed"], stream=int(a["train_stream"]))
    dm.Lt_history.zero_(); dm.Lt_count.zero_()
    out = dm({"condition_embed_token": dev(a["step_cond"]), "content_token": dev(a["train_x0"])}, return_loss=True)
    np.testing.assert_allclose(out["loss"].item(), a["train_loss"], rtol=2e-5)
    torch.testing.assert_close(out["logits"].cpu(), torch.from_numpy(a["train_logits"]), atol=2e-5, rtol=1e-4)
    assert np.array_equal(out["pred_data"].cpu().numpy(), a["train_pred"])
    np.testing.assert_allclose(dm.Lt_history.cpu().numpy(), a["train_Lt_history"], rtol=1e-4)
    assert np.array_equal(dm.Lt_count.cpu().numpy(), a["train_Lt_count"])
    # q_sample alone against the oracle (tokens exact)
    from oracle import d3pm as od
    x0 = torch.from_numpy(a["train_x0"])
    want_xt = od.gumbel_argmax(od.q_pred(od.index_to_log_onehot(x0, cfg["K"] + 1), torch.from_numpy(a["train_t"]), sd),
                               cfg["noise_seed"], int(a["train_stream"]))
    assert torch.equal(dm.last_train_stats["xt"].cpu(), want_xt)


def test_discrete_diffusion_glue_forward(G, golden):
    """DiscreteDiffusion.forward end to end on a tiny config: output-dict keys / shapes of the reference
    (discrete_diffusion.py:66-81); values are covered by the per-module tests."""
    sdv, av, cfgv = golden("vqvae_ds188")
    vq = build_vqvae(G, sdv, cfgv)
    K, L = cfgv["n_codes"], 4 * 4 * 4
    d = G.DalleMaskImageEmbedding(num_embed=K, spatial_size=[8, 8], embed_dim=64)
    tr = G.Text2ImageTransformer(dalle=d, n_layer=2, n_embd=64, n_head=16, content_seq_len=L, block_activate="GELU2",
                                 content_spatial_size=[8, 8], condition_dim=512, diffusion_step=20)
    dm = G.DiffusionTransformer(transformer=tr, diffusion_step=20, alpha_init_type="alpha1", auxiliary_loss_weight=5e-4,
                                adaptive_auxiliary_loss=True, guidance_scale=2, content_seq_len=L).cuda().eval()
    import src  # noqa: F401
    from src.models.text_models.clip_text_embedding import CLIPTextEmbedding
    gen = G.DiscreteDiffusion(CLIPTextEmbedding(512).cuda(), dm)
    batch = {"video": dev(av["x"]), "text": ["a", "b"]}
    out = gen(batch, vq, None, do_inference=True)
    assert set(out) == {"pred_data", "pred_single_step", "gt_data", "losses", "test"}
    assert tuple(out["pred_data"].shape) == tuple(av["x"].shape) and torch.isfinite(out["pred_data"]).all()
    assert out["losses"].ndim == 0 and torch.isfinite(out["losses"])
    torch.testing.assert_close(out["test"].cpu(), torch.from_numpy(av["decoded"]), atol=1e-4, rtol=1e-4)


# ----------------------------------------------------------------------------- VQ-VAE train-mode forward value
def test_vqvae_train_forward_matches_reference(G):
    import os
    from tests.conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "vqvae_train_ds188.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    after = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("after/")}
    cfg = {k[4:]: (z[k].tolist() if z[k].ndim else z[k].item()) for k in z.files if k.startswith("cfg_")}
    m = build_vqvae(G, sd, cfg)
    m.train()
    m.codebook._need_init = False
    perm = torch.from_numpy(z["perm"][0])
    m.perm_source = lambda n: perm
    with torch.no_grad():                             # (with grad enabled: tests/test_gpu_vqvae_training.py)
        out = m({"video": dev(z["x"])})
    torch.testing.assert_close(out["pred_data"].cpu(), torch.from_numpy(z["pred"]), atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(out["losses"]["recon_loss"].item(), z["recon_loss"], rtol=1e-4)
    np.testing.assert_allclose(out["losses"]["commitment_loss"].item(), z["commitment_loss"], rtol=1e-4)
    got = {k: v.cpu() for k, v in m.state_dict().items()}
    for k, v in after.items():
        if v.dtype.is_floating_point:
            torch.testing.assert_close(got[k], v, atol=5e-5, rtol=2e-4, msg=lambda s, k=k: f"{k}: {s}")
        else:
            assert torch.equal(got[k], v), k
    # the eval path must see the updated statistics (packed-weight cache invalidated)
    m.eval()
    rec = m.decode(torch.zeros((1, 4, 4, 4), dtype=torch.long, device="cuda"))
    from oracle import vqvae as ov
    with torch.no_grad():
        want = ov.decode(torch.zeros((1, 4, 4, 4), dtype=torch.long), after, cfg)
    torch.testing.assert_close(rec.cpu(), want, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("name", ["vqvae_init_tiled", "vqvae_init"])
def test_vqvae_codebook_data_init_matches_reference(G, name):
    """The first train-mode forward of a fresh model against the reference's (tests/golden/vqvae_init*.npz): data-init of the
    codebook (videogpt_vq_vae.py:160-172) and, with 16 latents for 24 codes, _tile's repeat + jitter (:151-158) in the init and
    in the restart draw; permutations and noise are the reference's (injected through perm_source / noise_source)."""
    import os
    from tests.conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    after = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("after/")}
    cfg = {k[4:]: (z[k].tolist() if z[k].ndim else z[k].item()) for k in z.files if k.startswith("cfg_")}
    m = G.VQVAE(None, cfg["embedding_dim"], cfg["n_codes"], cfg["n_hiddens"], cfg["n_res_layers"], cfg["downsample"],
                cfg["sequence_length"], cfg["resolution"])
    m.load_state_dict(sd)
    m = m.cuda().train()
    assert m.codebook._need_init
    perms = [torch.from_numpy(z["perm_init"]), torch.from_numpy(z["perm"])]
    noises = [torch.from_numpy(z[k]) for k in ("noise_init", "noise") if k in z.files]
    m.perm_source = lambda n: perms.pop(0)
    m.noise_source = lambda shape: noises.pop(0)
    with torch.no_grad():
        out = m({"video": dev(z["x"])})
    assert not perms and not noises and not m.codebook._need_init
    torch.testing.assert_close(out["pred_data"].cpu(), torch.from_numpy(z["pred"]), atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(out["losses"]["recon_loss"].item(), z["recon_loss"], rtol=1e-4)
    np.testing.assert_allclose(out["losses"]["commitment_loss"].item(), z["commitment_loss"], rtol=1e-4)
    got = {k: v.cpu() for k, v in m.state_dict().items()}
    for k in ("codebook.embeddings", "codebook.N", "codebook.z_avg"):
        torch.testing.assert_close(got[k], after[k], atol=2e-5, rtol=1e-4, msg=lambda s, k=k: f"{k}: {s}")


def test_near_tie_stress_vectors(G, golden):
    """tests/golden/neartie.npz: latents next to the bisector of close code pairs and logits whose top-2 Gumbel scores are
    bisected together, margins 1e-7 ... 1e-3 (recorded in fp64 next to the reference's choice).  Bar: wherever the fp64 margin
    exceeds the fp32 noise of the REFERENCE's own evaluation (it flips against fp64 below 2e-5 resp. 5e-6 itself), the device
    picks the fp64 winner = the reference's choice; inside that band it picks one of the two contenders.  Counts are reported."""
    sd, a, cfg = golden("neartie")
    # ---- nearest code
    z = torch.from_numpy(a["cb/z"]).permute(0, 2, 3, 4, 1).reshape(-1, a["cb/z"].shape[1]).contiguous()
    idx = torch.empty(z.shape[0], dtype=torch.int64, device="cuda")
    G.ops.nearest_code(dev(z), dev(a["cb/codebook"]), idx, None)
    idx = idx.cpu().numpy()
    m64, w64, s64, ref = np.abs(a["cb/margin64"]), a["cb/winner64"], a["cb/second64"], a["cb/ref_idx"]
    clear = m64 > 2e-5
    assert (ref[clear] == w64[clear]).all()
    assert (idx[clear] == w64[clear]).all(), m64[clear & (idx != w64)]
    assert ((idx == w64) | (idx == s64)).all()
    rec = {"codebook_vectors": len(idx), "codebook_clear": int(clear.sum()), "codebook_eq_reference": int((idx == ref).sum()),
           "codebook_eq_fp64": int((idx == w64).sum()), "codebook_reference_eq_fp64": int((ref == w64).sum()),
           "codebook_largest_margin_flipped_vs_fp64": float(m64[idx != w64].max()) if (idx != w64).any() else 0.0}
    # ---- Gumbel arg-max through the fused step kernel (guided: conditional + unconditional logits)
    B, L, K, T = cfg["B"], cfg["L"], cfg["K"], cfg["T"]
    rows = lambda x: torch.from_numpy(x).permute(0, 2, 1).reshape(B * L, K).contiguous().cuda()
    sched = [sd[n].cuda() for n in G.d3pm.SCHED_ORDER]
    tok_in = dev(a["gum/xt"])
    tok_out = torch.empty_like(tok_in)
    t2 = torch.cat([dev(a["gum/t"]), dev(a["gum/t"])]).contiguous()
    sid = torch.tensor([int(a["gum/stream"])], dtype=torch.int64, device="cuda")
    G.ops.d3pm_step(rows(a["gum/logits_c"]), rows(a["gum/logits_u"]), tok_in, tok_out, sched, t2, sid, K=K, T=T,
                    guidance=float(cfg["guidance"]), seed=cfg["noise_seed"])
    tok = tok_out.cpu().numpy()
    m64, w64, s64, ref = np.abs(a["gum/margin64"]), a["gum/winner64"], a["gum/second64"], a["gum/ref_tok"]
    clear = m64 > 5e-6
    assert (ref[clear] == w64[clear]).all()
    assert (tok[clear] == w64[clear]).all(), m64[clear & (tok != w64)]
    assert ((tok == w64) | (tok == s64)).all()
    rec.update({"gumbel_vectors": tok.size, "gumbel_clear": int(clear.sum()), "gumbel_below_1e-4": int((m64 < 1e-4).sum()),
                "gumbel_eq_reference": int((tok == ref).sum()), "gumbel_eq_fp64": int((tok == w64).sum()),
                "gumbel_reference_eq_fp64": int((ref == w64).sum()),
                "gumbel_largest_margin_flipped_vs_fp64": float(m64[tok != w64].max()) if (tok != w64).any() else 0.0})
    parity_report("near_tie_stress", rec)
    # 61 of the 64 land on the reference's fp32 choice (62 before the sum terms of the log-sum-exp reductions went to one product +
    # v_exp_f32: vector 2, fp64 margin 9.2e-8); a further slip would mean the arg-max path's arithmetic moved
    assert rec["gumbel_eq_reference"] >= 61 and rec["gumbel_eq_fp64"] >= 61, rec


def test_fused_layer_variants_agree(G, monkeypatch):
    """gsdd_d3pm_layer with the f16 hi + lo weight images (default when Text2ImageTransformer packs them), with the bf16x3 images
    (variant 'x3p'), and writing k/v as attention images instead of f32 rows: same
    block output, each also against an fp64 evaluation of the block; the image path is checked through the attention kernel that
    consumes it."""
    torch.manual_seed(11)
    B2, L, H, Dm = 4, 64, 16, 64
    d = G.DalleMaskImageEmbedding(num_embed=33, spatial_size=[8, 8], embed_dim=Dm)
    tr = G.Text2ImageTransformer(dalle=d, n_layer=2, n_embd=Dm, n_head=H, content_seq_len=L, block_activate="GELU2",
                                 content_spatial_size=[8, 8], diffusion_step=10).cuda()
    with torch.no_grad():
        for p_ in tr.parameters():
            p_.mul_(8.0)                      # N(0, 0.02) init would make every block nearly the identity
    lay0, lay1 = (dict(l) for l in tr.packed()["layers"])
    x3_0, x3_1 = dict(lay0), dict(lay1)
    for lay in (lay0, lay1):                  # the sampler makes these on first use
        lay["lay_h2"], lay["wqkv_h2"] = G.ops.d3pm_layer_pack_h2(lay["w1"], lay["w2"], lay["wproj"], lay["wqkv"])
    for lay in (x3_0, x3_1):
        lay["w2_x3"], lay["wqkv_x3"] = G.ops.d3pm_layer_pack(lay["w2"], lay["wproj"], lay["wqkv"])
    M = B2 * L
    y = torch.randn(M, Dm, device="cuda")
    x_in = torch.randn(M, Dm, device="cuda")
    cv = torch.randn(B2, Dm, device="cuda")
    t2 = torch.tensor([0, 3, 9, 5], dtype=torch.int64, device="cuda")
    # fp64 evaluation of the same block tail (Block.forward's tail + the next block's AdaLN and q|k|v)
    f = lambda k_, l_: l_[k_].detach().double()
    x1 = x_in.double() + y.double() @ f("wproj", lay0).T + f("bproj", lay0) + cv.double().repeat_interleave(L, 0)
    ln = torch.nn.functional.layer_norm(x1, (Dm,), f("g2", lay0), f("b2", lay0), 1e-5)
    hid = ln @ f("w1", lay0).T + f("bb1", lay0)
    x2 = x1 + (hid * torch.sigmoid(1.702 * hid)) @ f("w2", lay0).T + f("bb2", lay0)
    tab = f("ada1", lay1)[t2].repeat_interleave(L, 0)
    an = torch.nn.functional.layer_norm(x2, (Dm,), None, None, 1e-5) * tab[:, :Dm] + tab[:, Dm:]
    qkv64 = (an @ f("wqkv", lay1).T + f("bqkv", lay1)).reshape(M, 3 * H, 4).permute(1, 0, 2)
    outs, errs = [], {}
    for name, l0, l1 in (("h2", lay0, lay1), ("x3p", x3_0, x3_1)):
        x = x_in.clone()
        qkv = torch.zeros(3 * H, M, 4, device="cuda")
        G.ops.d3pm_layer(y, x, L, l0, cvec=cv, nxt=l1, t2=t2, qkv=qkv)
        outs.append((x, qkv))
        errs[name] = {"x": float((x.double() - x2).abs().max()), "qkv": float((qkv.double() - qkv64).abs().max())}
    errs["max_abs"] = {"x": float(x2.abs().max()), "hidden": float(hid.abs().max()), "qkv": float(qkv64.abs().max())}
    parity_report("fused_layer_vs_fp64", errs)
    # (weights x 8: hidden units reach ~30 and GELU2 runs on the bare v_exp_f32 / v_rcp_f32, 1 ulp each -- the x error of every
    # variant is that, not the operand format; q|k|v of the normalised stream is the sharper check)
    assert all(e["x"] < 6e-5 and e["qkv"] < 1e-5 for n_, e in errs.items() if n_ != "max_abs"), errs
    for o in outs[1:]:                        # (|x| reaches 75 here: 1e-6 relative is one ulp)
        torch.testing.assert_close(outs[0][0], o[0], atol=4e-5, rtol=1e-6)
        torch.testing.assert_close(outs[0][1], o[1], atol=2e-5, rtol=0)
    # k, v as images: attention on (q rows, images) == attention on the f32 q, k, v rows -- for both image kernels
    for (l0, l1), ref, variant in (((lay0, lay1), outs[0], None), ((x3_0, x3_1), outs[1], "x3p")):
        ws = G.ops.d3pm_attention_workspace(B2, L, H, torch.device("cuda"))
        x = x_in.clone()
        qkv2 = torch.zeros(3 * H, M, 4, device="cuda")
        G.ops.d3pm_layer(y, x, L, l0, cvec=cv, nxt=l1, t2=t2, qkv=qkv2, kv_img=ws, variant=variant)
        torch.testing.assert_close(x, ref[0], atol=0, rtol=0)
        assert torch.equal(qkv2[:H], ref[1][:H]) and not qkv2[H:].any()          # q rows written, k/v rows untouched
        a_img = torch.empty(M, Dm, device="cuda")
        a_ref = torch.empty(M, Dm, device="cuda")
        G.ops.d3pm_attention(qkv2[:H], None, None, B2, L, H, a_img, ws=ws)
        q, k, v = ref[1][:H], ref[1][H:2 * H], ref[1][2 * H:]
        ws_ref = G.ops.d3pm_attention_workspace(B2, L, H, torch.device("cuda"))
        G.ops.d3pm_attention(q, k, v, B2, L, H, a_ref, ws=ws_ref)
        torch.testing.assert_close(a_img, a_ref, atol=1e-6, rtol=0)
        # the per-tile ||k|| bounds written by the layer kernel's epilogue == those of the pre-split pass, and both bound the keys
        ntile = H * M // 32
        o_sum, o_norm = H * M * 16, H * M * 16 + 4 * ntile                         # workspace layout: K | V | key sums | tile norms
        kn_img, kn_ref = ws[o_norm:o_norm + ntile], ws_ref[o_norm:o_norm + ntile]
        torch.testing.assert_close(kn_img, kn_ref, atol=0, rtol=1e-6)
        true_max = k.double().norm(dim=-1).view(ntile, 32).max(dim=1).values
        assert bool((kn_ref.double() >= true_max).all()) and bool((kn_ref.double() <= true_max * (1 + 1e-5) + 1e-30).all())
        # ... and the per-tile key sums behind the row-sum lower bound: the same summation tree in both producers
        ks_img, ks_ref = ws[o_sum:o_sum + 4 * ntile].view(ntile, 4), ws_ref[o_sum:o_sum + 4 * ntile].view(ntile, 4)
        torch.testing.assert_close(ks_img, ks_ref, atol=2e-5, rtol=1e-5)           # (k itself differs in the last bits between the layer variants)
        torch.testing.assert_close(ks_ref.double(), k.double().view(ntile, 32, 4).sum(dim=1), atol=1e-4, rtol=1e-5)


def test_layer_kernel_range_screen_falls_back_to_bf16x3(G, golden, monkeypatch):
    """The f16 hi + lo layer kernel carries activations as 16 a in f16: |a| >= 4094 overflows.  A hidden unit pushed to ~6000 (an
    outlier MLP activation of a trained checkpoint) must not produce garbage: the kernel raises its range flag, the caller reruns on
    the bf16x3 kernel (f32 range), and logits / sampled tokens are those of the CPU oracle resp. of a forced bf16x3 run."""
    from oracle import d3pm as od
    sd, a, cfg = golden("d3pm_L64")
    sd = {k_: v_.clone() for k_, v_ in sd.items()}
    sd["transformer.blocks.1.mlp.0.bias"][7] = 6000.0
    dm = build_d3pm(G, sd, cfg)
    tr = dm.transformer
    xt, cond, t = dev(a["step_xt"]), dev(a["step_cond"]), dev(a["step_t"])
    logits = tr(xt, cond, t)
    assert getattr(tr, "range_demotions", 0) == 1 and "w2_x3" in tr.packed()["layers"][0] and "lay_h2" not in tr.packed()["layers"][0]
    assert bool(torch.isfinite(logits).all())
    with torch.no_grad():
        want = od.denoiser(torch.from_numpy(a["step_xt"]), torch.from_numpy(a["step_cond"]), torch.from_numpy(a["step_t"]), sd)
    torch.testing.assert_close(logits.cpu(), want, atol=LOGIT_TOL, rtol=1e-5)
    # the sampler: flag read once after the captured loop, whole call repeated on the bf16x3 kernel with the same noise stream
    B = cfg["B"]
    dm2 = build_d3pm(G, sd, cfg)
    dm2.set_noise(cfg["noise_seed"])
    tok = dm2.sample(["a"] * B, None, cond, torch.zeros_like(cond), filter_ratio=0)["content_token"].cpu()
    assert dm2.transformer.range_demotions == 1 and dm2.noise_stream == cfg["T"]
    monkeypatch.setenv("GSDD_LAYER", "x3p")
    dm3 = build_d3pm(G, sd, cfg)
    dm3.set_noise(cfg["noise_seed"])
    tok3 = dm3.sample(["a"] * B, None, cond, torch.zeros_like(cond), filter_ratio=0)["content_token"].cpu()
    assert getattr(dm3.transformer, "range_demotions", 0) == 0
    assert torch.equal(tok, tok3) and int(tok.max()) < cfg["K"]
    monkeypatch.delenv("GSDD_LAYER")
    # an unmodified model never trips the screen
    dm4 = build_d3pm(G, golden("d3pm_L64")[0], cfg)
    dm4.transformer(xt, cond, t)
    assert getattr(dm4.transformer, "range_demotions", 0) == 0 and "lay_h2" in dm4.transformer.packed()["layers"][0]


@pytest.mark.parametrize("L,spatial", [(48, [8, 8]), (96, [16, 8])])
def test_denoiser_ragged_lengths_match_oracle(G, L, spatial):
    """L % 32 != 0 (VALU attention, f32 q|k|v rows, generic block-0 GEMM) and L % 32 == 0 with the image path: both against the
    CPU oracle's denoiser on the same random weights."""
    from oracle import d3pm as od
    torch.manual_seed(L)
    K, B = 32, 3
    d = G.DalleMaskImageEmbedding(num_embed=K, spatial_size=spatial, embed_dim=64)
    tr = G.Text2ImageTransformer(dalle=d, n_layer=3, n_embd=64, n_head=16, content_seq_len=L, block_activate="GELU2",
                                 content_spatial_size=spatial, diffusion_step=10)
    with torch.no_grad():
        for p_ in tr.parameters():
            p_.mul_(6.0)
    sd = {"transformer." + k: v.detach().clone() for k, v in tr.state_dict().items()}
    tok = torch.randint(0, K + 1, (B, L))
    cond = torch.randn(B, 1, 512)
    t = torch.tensor([0, 4, 9])
    want = od.denoiser(tok, cond, t, sd)
    got = tr.cuda()(tok.cuda(), cond.cuda(), t.cuda())
    torch.testing.assert_close(got.cpu(), want, atol=LOGIT_TOL, rtol=0)


def test_layer_kernel_variant_without_its_images_fails_loudly(G, monkeypatch):
    """Asking for an image kernel (variant 'h2' / 'x3p', or GSDD_LAYER in the environment) on a call that carries no images of that
    kind is an error, not a silent switch; so is a call with no images at all (the split-on-the-fly kernels are gone), and an unknown
    variant name never reaches the library."""
    torch.manual_seed(3)
    d = G.DalleMaskImageEmbedding(num_embed=33, spatial_size=[8, 8], embed_dim=64)
    tr = G.Text2ImageTransformer(dalle=d, n_layer=1, n_embd=64, n_head=16, content_seq_len=64, block_activate="GELU2",
                                 content_spatial_size=[8, 8], diffusion_step=10).cuda()
    lay = dict(tr.packed()["layers"][0])
    x = torch.randn(64, 64, device="cuda")
    y = torch.randn(64, 64, device="cuda")
    for forced in ("h2", "x3p"):
        with pytest.raises(G.GsddError):
            G.ops.d3pm_layer(y, x.clone(), 64, lay, variant=forced)
        monkeypatch.setenv("GSDD_LAYER", forced)
        with pytest.raises(G.GsddError):
            G.ops.d3pm_layer(y, x.clone(), 64, lay)
        monkeypatch.delenv("GSDD_LAYER")
    with pytest.raises(G.GsddError):
        G.ops.d3pm_layer(y, x.clone(), 64, lay)   # no images at all
    with pytest.raises(G.GsddError):
        G.ops.d3pm_layer(y, x.clone(), 64, lay, variant="f32")
    h2 = dict(lay)
    h2["lay_h2"], _ = G.ops.d3pm_layer_pack_h2(lay["w1"], lay["w2"], lay["wproj"], lay["wqkv"])
    with pytest.raises(G.GsddError):
        G.ops.d3pm_layer(y, x.clone(), 64, h2, variant="x3p")     # only the f16 images were given
    G.ops.d3pm_layer(y, x.clone(), 64, h2)


def test_layer_kernel_weight_range_guard(G):
    """The default fused-layer kernel holds weights as f16 images of 2^8 w, so a weight of 255 or more must not take it: such a model is
    routed to the bf16x3 kernel (f32 range) and still matches the oracle."""
    from oracle import d3pm as od
    torch.manual_seed(5)
    K, B, L = 32, 2, 64
    d = G.DalleMaskImageEmbedding(num_embed=K, spatial_size=[8, 8], embed_dim=64)
    tr = G.Text2ImageTransformer(dalle=d, n_layer=2, n_embd=64, n_head=16, content_seq_len=L, block_activate="GELU2",
                                 content_spatial_size=[8, 8], diffusion_step=10)
    with torch.no_grad():
        tr.blocks[0].mlp[0].weight[3, 5] = 300.0          # one hidden unit with an absurd input weight
        tr.blocks[0].mlp[2].weight[:, 3] = 0.0            # ... which feeds nothing, so the output stays moderate
    sd = {"transformer." + k: v.detach().clone() for k, v in tr.state_dict().items()}
    tok = torch.randint(0, K + 1, (B, L))
    cond = torch.randn(B, 1, 512)
    t = torch.tensor([1, 7])
    want = od.denoiser(tok, cond, t, sd)
    got = tr.cuda()(tok.cuda(), cond.cuda(), t.cuda())
    lay = tr.packed()["layers"][0]
    assert "w2_x3" in lay and "lay_h2" not in lay
    torch.testing.assert_close(got.cpu(), want, atol=LOGIT_TOL, rtol=0)


# ----------------------------------------------------------------------------- the attention kernel's arithmetic modes
def test_attention_arithmetic_modes(G, golden, monkeypatch):
    """GSDD_ATTN_P: the default (adaptive lo half for L >= 2048, hi + lo everywhere below) and the most exact mode (22) meet the bars --
    denoiser logits of the reference fixture within 1e-4, attention within 2e-5 of fp64 on flat, trained-like and peaky rows at
    L = 1024 and L = 4096; the measured errors of every mode (incl. a8 forced at L = 1024, a12, and the hi-only mode 11, which does
    not meet the bars and is not a default) go to the report."""
    sd, a, cfg = golden("d3pm_L64")
    H = 16
    rec = {}
    for mode in ("default", "22", "a8", "a12", "11"):
        if mode == "default":
            monkeypatch.delenv("GSDD_ATTN_P", raising=False)
        else:
            monkeypatch.setenv("GSDD_ATTN_P", mode)
        dm = build_d3pm(G, sd, cfg)
        logits = dm.transformer(dev(a["step_xt"]), dev(a["step_cond"]), dev(a["step_t"])).cpu()
        rec[f"fixture_logits_err[{mode}]"] = (logits - torch.from_numpy(a["step_logits"])).abs().max().item()
        for L in (1024, 4096):
            g = torch.Generator().manual_seed(3)
            errs = []
            for scale in (0.05, 1.0, 3.0):
                q, k = (torch.randn(1, H, L, 4, generator=g) * scale for _ in range(2))
                v = torch.randn(1, H, L, 4, generator=g)
                qd, kd, vd = q.double().cuda(), k.double().cuda(), v.double().cuda()
                want = (torch.softmax(qd @ kd.transpose(-1, -2) * 0.5, dim=-1) @ vd).permute(0, 2, 1, 3).reshape(L, H * 4)
                hm = lambda z: z.permute(1, 0, 2, 3).reshape(H, L, 4).contiguous().cuda()
                out = torch.empty((L, H * 4), device="cuda")
                G.ops.d3pm_attention(hm(q), hm(k), hm(v), 1, L, H, out, ws=G.ops.d3pm_attention_workspace(1, L, H, "cuda"))
                errs.append((out.double() - want).abs().max().item())
            rec[f"attention_err_flat_trained_peaky[{mode}, L={L}]"] = errs
    monkeypatch.delenv("GSDD_ATTN_P", raising=False)
    parity_report("attention_arithmetic_modes", rec)
    for mode in ("default", "22", "a12"):
        assert rec[f"fixture_logits_err[{mode}]"] < LOGIT_TOL, (mode, rec)
        for L in (1024, 4096):
            assert max(rec[f"attention_err_flat_trained_peaky[{mode}, L={L}]"]) < 2e-5, (mode, L, rec)
    assert max(rec["attention_err_flat_trained_peaky[a8, L=4096]"]) < 2e-5, rec
    assert rec["attention_err_flat_trained_peaky[default, L=4096]"] == rec["attention_err_flat_trained_peaky[a8, L=4096]"]
    assert rec["attention_err_flat_trained_peaky[default, L=1024]"] == rec["attention_err_flat_trained_peaky[22, L=1024]"]
    assert rec["fixture_logits_err[22]"] <= rec["fixture_logits_err[11]"]


def test_two_streams_run_two_attention_modes_concurrently(G):
    """The arithmetic mode is an argument of the call, not process state: two HIP streams run the attention kernel in two modes at
    the same time (hi + lo everywhere on one, hi only on the other, interleaved launches) and each reproduces, bit for bit, what
    its mode gives when it runs alone."""
    H, B, L = 16, 2, 2048
    g = torch.Generator().manual_seed(21)
    q, k = (torch.randn(H, B * L, 4, generator=g).cuda() * 1.5 for _ in range(2))
    v = torch.randn(H, B * L, 4, generator=g).cuda()
    modes = ("22", "11")
    alone = {}
    for m in modes:
        out = torch.empty((B * L, H * 4), device="cuda")
        G.ops.d3pm_attention(q, k, v, B, L, H, out, ws=G.ops.d3pm_attention_workspace(B, L, H, "cuda"), mode=m)
        alone[m] = out.clone()
    assert not torch.equal(alone["22"], alone["11"])                     # the modes do differ on these rows
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in modes]
    outs = [[torch.empty((B * L, H * 4), device="cuda") for _ in range(4)] for _ in modes]
    wss = [G.ops.d3pm_attention_workspace(B, L, H, "cuda") for _ in modes]
    for rep in range(4):
        for i, m in enumerate(modes):
            G.ops.d3pm_attention(q, k, v, B, L, H, outs[i][rep], ws=wss[i], mode=m, stream=streams[i])
    for s_ in streams:
        s_.synchronize()
    for i, m in enumerate(modes):
        for rep in range(4):
            assert torch.equal(outs[i][rep], alone[m]), (m, rep)


@pytest.mark.parametrize("filter_ratio", [0.3, 0.05])
def test_sample_from_partially_noised_start(G, golden, filter_ratio):
    """sample(content_token=..., filter_ratio > 0) (diffusion_transformer.py:590-592, :626-634): q_sample of the given tokens at
    t = int(T * filter_ratio) - 1, then that many reverse steps.  The reference's own loop for this branch raises TypeError (it passes
    p_sample four of its six positional parameters), so the pin is the oracle's restatement of what the branch is written to do, built
    from pieces that are themselves pinned to the reference (q_pred, Gumbel arg-max, p_sample_step): tokens bit-exact, eager and
    captured, and the noise stream advances by the draws spent (one q_sample + the steps)."""
    from oracle import d3pm as od
    sd, a, cfg = golden("d3pm_L64")
    dm = build_d3pm(G, sd, cfg)
    B, L, K, T = cfg["B"], cfg["L"], cfg["K"], cfg["T"]
    g = torch.Generator().manual_seed(31)
    x0 = torch.randint(0, K, (B, L), generator=g)
    cond = torch.from_numpy(a["step_cond"])
    with torch.no_grad():
        want = od.sample_from(x0, filter_ratio, cond, torch.zeros_like(cond), sd, cfg["guidance"], cfg["noise_seed"], stream0=7)
    steps = int(T * filter_ratio)
    for use_graph in (False, True):
        dm.set_noise(cfg["noise_seed"], stream=7)
        got = dm.sample(["a"] * B, None, cond.cuda(), torch.zeros_like(cond).cuda(), content_token=x0.cuda(), filter_ratio=filter_ratio,
                        use_graph=use_graph)["content_token"].cpu()
        assert torch.equal(got, want), f"use_graph={use_graph}: {(got != want).sum().item()} tokens differ"
        assert dm.noise_stream == 7 + 1 + steps
    # (the last step, t = 0, draws from the model's own x_0 prediction -- the posterior's t - 1 wrap -- so with random weights few of
    # the given tokens survive: that is the reference's arithmetic, not a property to assert.)  No tokens given: an error
    with pytest.raises(G.GsddError):
        dm.sample(["a"] * B, None, cond.cuda(), torch.zeros_like(cond).cuda(), content_token=None, filter_ratio=filter_ratio)
