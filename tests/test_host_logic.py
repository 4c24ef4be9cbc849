"""CPU tests of the host side: config composition / instantiation through the reference's `_target_` paths,
weight repacking, C-ABI symbol export, state_dict key compatibility."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import gsdd_amd
    if "GSDD_LIB_PATH" not in os.environ:                 # incremental: a no-op when the library is newer than its sources and the header
        subprocess.check_call(["bash", os.path.join(REPO, "build.sh")], cwd=REPO, stdout=subprocess.DEVNULL)
    header = open(os.path.join(REPO, "include", "gsdd.h")).read()
    declared = set(re.findall(r"\b(gsdd_[a-z0-9_]+)\s*\(", header))
    assert declared == set(gsdd_amd.EXPORTS), declared ^ set(gsdd_amd.EXPORTS)
    L = ctypes.CDLL(gsdd_amd.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert gsdd_amd.lib().gsdd_version() >= 101          # (lib() also checks the descriptor sizes against gsdd_abi_sizeof)


def test_cpu_inputs_fail_loudly_without_fallback():
    import gsdd_amd
    m = gsdd_amd.VQVAE(None, 8, 32, 16, 1, [1, 4, 4], 4, 16).eval()
    with pytest.raises(gsdd_amd.GsddError):
        m.encode(torch.randn(1, 3, 4, 16, 16))
    with pytest.raises(gsdd_amd.GsddError):
        m.decode(torch.zeros(1, 4, 4, 4, dtype=torch.long))


SMALL = ["batch_size=2", "datamodule.sequence_length=4", "datamodule.resolution=32"]


def test_compose_and_instantiate_reference_targets(monkeypatch):
    """configs/ mirrors the reference's tree (same group files, same defaults lists): `model=discrete_diffusion` composes through
    model/motionencoder/{diffusion_transformer,transformer_utils,dalle_mask_image_embedding}.yaml, model/textencoder/*, model/
    evaluator.yaml, callbacks/, paths/, hydra/ and instantiates through the reference's `_target_` strings."""
    import src  # noqa: F401
    from gsdd_amd.hydra_lite import compose, instantiate
    monkeypatch.setenv("PROJECT_ROOT", REPO)
    cfg = compose(os.path.join(REPO, "configs"), "eval.yaml", SMALL + ["model.generator.diffusion_model.transformer.n_layer=2"])
    assert cfg.model._target_ == "src.models.multistage_text_motion_model.MultistageTextMotionModel"
    assert cfg.model.autoencoder.sequence_length == 4 and cfg.datamodule.batch_size == 2
    assert cfg.model.generator.diffusion_model._target_ == "src.models.motionencoder.diffusion_transformer.DiffusionTransformer"
    assert cfg.model.generator.diffusion_model.transformer.dalle.num_embed == 4096
    assert cfg.model.generator.textencoder._target_ == "src.models.text_models.clip_text_embedding.CLIPTextEmbedding"
    assert cfg.model.evaluator._target_ == "src.utils.evaluator.Evaluator" and cfg.model.evaluator.checkpoint_paths == "__None__"
    assert cfg.callbacks.model_checkpoint.monitor == "total/val" and cfg.callbacks.model_checkpoint.save_last is True
    assert cfg.callbacks.model_checkpoint.dirpath.startswith(REPO) and cfg.callbacks.model_checkpoint.dirpath.endswith("/checkpoints")
    assert cfg.trainer.default_root_dir == cfg.paths.output_dir and "/eval/runs/" in cfg.paths.output_dir
    assert "hydra" not in cfg
    model = instantiate(cfg.model, _recursive_=False)
    assert type(model.autoencoder).__name__ == "VQVAE" and model.autoencoder.latent_shape == (4, 4, 4)
    assert type(model.generator).__name__ == "DiscreteDiffusion" and model.generator.zero_text_emb is True
    tr = model.generator.diffusion_model.transformer
    assert len(tr.blocks) == 2 and tr.content_emb.num_embed == 4097
    assert tuple(tr.blocks[0].ln1.emb.weight.shape) == (1000, 64)          # the reference leaves the transformer's diffusion_step at 1000
    opts = model.configure_optimizers()
    assert len(opts) == 2 and opts[0].defaults["lr"] == 1e-4 and opts[1].defaults["lr"] == 1e-6 and opts[0].defaults["betas"] == (0.5, 0.999)
    assert not model.train().autoencoder.training and model.generator.training
    cfg1 = compose(os.path.join(REPO, "configs"), "train.yaml", ["model=videogpt_vq_vae"])
    m1 = instantiate(cfg1.model, _recursive_=False)
    o1, sched = m1.configure_optimizers()
    assert m1.generator.n_codes == 4096 and len(o1) == 1 and sched == [] and o1[0].defaults["lr"] == 4e-4
    dm = instantiate(compose(os.path.join(REPO, "configs"), "train.yaml", ["datamodule=msrvtt"] + SMALL).datamodule)
    assert type(dm).__name__ == "MSRVTTDataModule" and dm.sequence_length == 4
    with pytest.warns(UserWarning, match="no checkpoint"):             # the I3D extractor exists now; without weights its FVD means nothing
        ev = instantiate(cfg.model.evaluator, device="cpu", _recursive_=False)
    assert type(ev.videoencoder).__name__ == "InceptionI3d" and not ev.videoencoder.training
    assert len(ev.videoencoder.state_dict()) == 344


@pytest.mark.skipif(not os.path.isdir("/root/reference/configs"), reason="the reference tree only exists in the build container")
def test_reference_config_tree_composes_and_instantiates_unchanged(monkeypatch):
    """The reference's OWN configs/ (train.yaml + model=... as its job files pass it, vqvae.job:15, ucf-ddiff-train.job:15) through
    hydra_lite and this build's `src.` modules: paths/hydra/now resolvers, callbacks' sibling defaults, nested `@package`
    placement.  Only the evaluator (I3D weights) and the checkpoint path (a file on the authors' cluster) are switched off."""
    import src  # noqa: F401
    from gsdd_amd.hydra_lite import compose, instantiate
    monkeypatch.setenv("PROJECT_ROOT", REPO)
    ref = "/root/reference/configs"
    cfg = compose(ref, "train.yaml", ["model=discrete_diffusion", "model.do_evaluation=false",
                                      "model.checkpoint_paths.autoencoder=__None__"])
    assert cfg.model.autoencoder.n_codes == 2048 and cfg.model.autoencoder.sequence_length == 4
    assert cfg.model.generator.diffusion_model.diffusion_step == 50 and cfg.model.devices == [0]
    assert cfg.callbacks.model_checkpoint.filename == "epoch_{epoch:03d}" and cfg.paths.datasets == "/home1/chemburk"
    model = instantiate(cfg.model, _recursive_=False)
    dmod = model.generator.diffusion_model
    assert model.autoencoder.latent_shape == (2, 16, 16) and dmod.num_timesteps == 50 and dmod.num_classes == 2049
    assert len(dmod.transformer.blocks) == 19 and dmod.guidance_scale == 2
    # a reference-shaped stage-2 state_dict (dead attn2.mask buffers, 1000-row AdaLayerNorm tables) loads into it
    from gsdd_amd.checkpoint import load_reference_checkpoint
    sd = {"generator." + k: v.clone() for k, v in model.generator.state_dict().items()}
    sd["generator.diffusion_model.transformer.blocks.3.attn2.mask"] = torch.ones(1, 1, 4, 4)
    assert load_reference_checkpoint(model.generator, {"state_dict": sd}) == ["diffusion_model.transformer.blocks.3.attn2.mask"]
    cfg1 = compose(ref, "train.yaml", ["model=videogpt_vq_vae", "model.do_evaluation=false"])
    m1 = instantiate(cfg1.model, _recursive_=False)
    assert m1.generator.downsample == [1, 16, 16] and m1.lr_args["gen_lr"] == 4e-4
    assert compose(ref, "train.yaml", ["model=videogpt_vq_vae", "datamodule=msrvtt"]).datamodule._target_.endswith("MSRVTTDataModule")


def test_compute_losses_accumulator_and_log_names():
    """src/models/metrics/loss.py:29-60: update() returns the weighted sum with its graph, compute() the means, loss2logname."""
    from src.models.metrics.loss import ComputeLosses
    cl = ComputeLosses(loss_dict={"l_dummy": 2.0})
    a = torch.tensor(3.0, requires_grad=True)
    t1 = cl.update({"losses": a})                                         # stage 2: a scalar
    t2 = cl.update({"losses": {"commitment_loss": torch.tensor(1.0), "recon_loss": torch.tensor(4.0)}})   # stage 1: a dict
    assert t1.requires_grad and float(t1) == 6.0 and float(t2) == 10.0
    out = cl.compute()
    assert float(out["l_dummy"]) == 4.0 and float(out["total"]) == 8.0
    assert cl.loss2logname("l_dummy", "val") == "l/dummy/val" and cl.loss2logname("total", "train") == "total/train"
    cl.reset()
    assert cl.count == 0
    with pytest.raises(KeyError):
        ComputeLosses(loss_dict={"l_codebook": 1.0})


def test_stage1_export_survives_the_reference_ten_character_strip(tmp_path):
    """multistage_text_motion_model.py:113-122 strips param_key[10:] from every key: a VQ-VAE exported as
    lightning_state(generator=vq) reads back bare; the file also loads through load_reference_checkpoint."""
    import gsdd_amd
    from gsdd_amd.checkpoint import lightning_state, load_reference_checkpoint
    torch.manual_seed(1)
    vq = gsdd_amd.VQVAE(None, 8, 32, 16, 1, [1, 4, 4], 4, 16)
    path = str(tmp_path / "stage1.ckpt")
    torch.save(lightning_state(generator=vq), path)
    state = torch.load(path, weights_only=True)["state_dict"]
    stripped = {k[10:]: v for k, v in state.items()}                       # the reference's loop, verbatim in effect
    assert set(stripped) == set(vq.state_dict())
    dst = gsdd_amd.VQVAE(None, 8, 32, 16, 1, [1, 4, 4], 4, 16)
    dst.load_state_dict(stripped)
    dst2 = gsdd_amd.VQVAE(None, 8, 32, 16, 1, [1, 4, 4], 4, 16)
    load_reference_checkpoint(dst2, path)
    for k, v in vq.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v) and torch.equal(dst2.state_dict()[k], v)


def test_state_dict_keys_match_reference(golden):
    import gsdd_amd
    sd, _, cfg = golden("vqvae_ds188")
    m = gsdd_amd.VQVAE(None, cfg["embedding_dim"], cfg["n_codes"], cfg["n_hiddens"], cfg["n_res_layers"],
                       cfg["downsample"], cfg["sequence_length"], cfg["resolution"])
    assert set(m.state_dict().keys()) == set(sd.keys())
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k


def test_convT_phase_tables_cover_every_tap_once():
    from gsdd_amd.vqvae import convT_phases
    for stride in [(1, 2, 2), (2, 2, 2), (1, 1, 1)]:
        pf = tuple((4 - s) // 2 + (4 - s) % 2 for s in stride)
        seen = []
        for (_, ks, offs) in convT_phases((4, 4, 4), stride, pf):
            assert len(ks) == len(offs) == 64 // (stride[0] * stride[1] * stride[2])
            seen += ks
        assert sorted(seen) == sorted((a, b, c) for a in range(4) for b in range(4) for c in range(4))


def test_weight_packing_matches_conv_semantics():
    """pack_conv_weight / conv_taps / pack_convT_weight reproduce conv3d / conv_transpose3d when applied naively."""
    from gsdd_amd.vqvae import conv_taps, convT_phases, pack_conv_weight, pack_convT_weight
    from oracle import vqvae as ov
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 3, 6, 6, generator=g)
    w = torch.randn(5, 4, 4, 4, 4, generator=g)
    stride, pf = (1, 2, 2), (2, 1, 1)
    want = ov.same_pad_conv3d(x, w, None, stride)
    wp, taps = pack_conv_weight(w), conv_taps((4, 4, 4), stride, pf)
    got = torch.zeros_like(want)
    xp = x[0].permute(1, 2, 3, 0)
    for ti, (dt, dh, dw) in enumerate(taps):
        for to in range(3):
            for ho in range(3):
                for wo in range(3):
                    t, h, ww = to * stride[0] + dt, ho * stride[1] + dh, wo * stride[2] + dw
                    if 0 <= t < 3 and 0 <= h < 6 and 0 <= ww < 6:
                        got[0, :, to, ho, wo] += wp[ti] @ xp[t, h, ww]
    torch.testing.assert_close(got, want, atol=1e-4, rtol=1e-4)
    wt = torch.randn(4, 5, 4, 4, 4, generator=g)
    want = ov.same_pad_convT3d(x, wt, None, stride)
    got = torch.zeros_like(want)
    for (ph, ks, offs) in convT_phases((4, 4, 4), stride, pf):
        wp = pack_convT_weight(wt, ks)
        for ti, (dt, dh, dw) in enumerate(offs):
            for to in range(3):
                for ho in range(6):
                    for wo in range(6):
                        t, h, ww = to + dt, ho + dh, wo + dw
                        if 0 <= t < 3 and 0 <= h < 6 and 0 <= ww < 6:
                            got[0, :, to * stride[0] + ph[0], ho * stride[1] + ph[1], wo * stride[2] + ph[2]] += wp[ti] @ xp[t, h, ww]
    torch.testing.assert_close(got, want, atol=1e-4, rtol=1e-4)


# ----------------------------------------------------------------------------- checkpoint interop
def _tiny_vq():
    import gsdd_amd
    torch.manual_seed(3)
    return gsdd_amd.VQVAE(None, 8, 32, 16, 1, [1, 8, 8], 4, 32)


def test_checkpoint_lightning_prefix_roundtrip(tmp_path):
    """Stage-1 Lightning checkpoints hold "generator.<key>"; the reference strips 10 characters
    (multistage_text_motion_model.py:113-122).  Stage-2 checkpoints hold "autoencoder.<key>" next to the generator's."""
    from gsdd_amd.checkpoint import lightning_state, load_reference_checkpoint
    src, dst = _tiny_vq(), _tiny_vq()
    with torch.no_grad():
        for prm in src.parameters():
            prm.add_(0.25)
    assert dst.codebook._need_init
    path = tmp_path / "stage1.ckpt"
    torch.save(lightning_state(generator=src), path)
    assert load_reference_checkpoint(dst, str(path)) == []
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    assert not dst.codebook._need_init
    # stage-2 layout: the autoencoder sits under "autoencoder.", unrelated generator keys next to it
    dst2 = _tiny_vq()
    st = lightning_state(autoencoder=src)
    st["state_dict"]["generator.diffusion_model.transformer.blocks.0.attn2.mask"] = torch.zeros(1, 1, 4, 4)
    load_reference_checkpoint(dst2, st)
    assert torch.equal(dst2.pre_vq_conv.conv.weight, src.pre_vq_conv.conv.weight)
    # a bare state_dict loads too
    dst3 = _tiny_vq()
    load_reference_checkpoint(dst3, src.state_dict())
    assert torch.equal(dst3.decoder.convts[0].convt.weight, src.decoder.convts[0].convt.weight)


def test_checkpoint_mismatch_is_loud():
    from gsdd_amd.checkpoint import load_reference_checkpoint
    src, dst = _tiny_vq(), _tiny_vq()
    sd = dict(src.state_dict())
    sd.pop("encoder.conv_last.conv.bias")
    with pytest.raises(KeyError):
        load_reference_checkpoint(dst, sd)
    sd = dict(src.state_dict())
    sd["encoder.conv_last.conv.bias"] = torch.zeros(3)
    with pytest.raises(ValueError):
        load_reference_checkpoint(dst, sd)


def test_checkpoint_drops_only_documented_dead_keys():
    import gsdd_amd
    from gsdd_amd.checkpoint import load_reference_checkpoint
    torch.manual_seed(0)

    def build():
        d = gsdd_amd.DalleMaskImageEmbedding(num_embed=16, spatial_size=[4, 4], embed_dim=64)
        tr = gsdd_amd.Text2ImageTransformer(dalle=d, n_layer=1, n_embd=64, n_head=16, content_seq_len=16, block_activate="GELU2",
                                            content_spatial_size=[4, 4], condition_dim=512, diffusion_step=10)
        return gsdd_amd.DiffusionTransformer(transformer=tr, diffusion_step=10, alpha_init_type="alpha1", guidance_scale=2,
                                             content_seq_len=16)
    src, dst = build(), build()
    with torch.no_grad():
        src.transformer.to_logits[1].bias.add_(1.0)
    sd = {"generator.diffusion_model." + k: v for k, v in src.state_dict().items()}
    sd["generator.diffusion_model.transformer.blocks.0.attn2.mask"] = torch.tril(torch.ones(16, 16)).view(1, 1, 16, 16)
    sd["generator.diffusion_model.zero_vector"] = torch.zeros(1)
    sd["generator.textencoder.clip_model.positional_embedding"] = torch.zeros(77, 512)

    class Gen(torch.nn.Module):            # the generator glue owns `diffusion_model` (discrete_diffusion.py:11-12)
        def __init__(self, dm):
            super().__init__()
            self.diffusion_model = dm
    dropped = load_reference_checkpoint(Gen(dst), {"state_dict": sd})
    assert sorted(dropped) == ["diffusion_model.transformer.blocks.0.attn2.mask", "diffusion_model.zero_vector",
                               "textencoder.clip_model.positional_embedding"]
    assert torch.equal(dst.transformer.to_logits[1].bias, src.transformer.to_logits[1].bias)
    sd["generator.diffusion_model.transformer.blocks.0.attn1.bogus"] = torch.zeros(1)
    with pytest.raises(KeyError):
        load_reference_checkpoint(Gen(build()), {"state_dict": sd})


def test_clip_folder_dataset_enumerates_like_the_reference(tmp_path):
    """classes = sorted parent directories, windows of `sequence_length` frames every 100 frames (ucf101_dataset.py:57-66)."""
    import numpy as np
    from src.datamodules.clip_folder_datamodule import ClipFolderDataset
    for cls, n in (("Swing", 130), ("Archery", 20), ("Biking", 7)):
        d = tmp_path / "train" / cls
        d.mkdir(parents=True)
        np.save(d / "v0.npy", np.zeros((n, 6, 8, 3), dtype=np.uint8))
    ds = ClipFolderDataset(str(tmp_path), sequence_length=16, split="train", resolution=4)
    assert ds.classes == ["Archery", "Biking", "Swing"] and ds.n_classes == 3
    got = [(os.path.basename(os.path.dirname(f)), s) for f, s in ds.clips]
    assert got == [("Archery", 0), ("Swing", 0), ("Swing", 100)]            # Biking is shorter than one window
    item = ds[2]
    assert item["label"] == 2 and item["text"] == "Swing" and tuple(item["frames"].shape) == (16, 6, 8, 3)


# ----------------------------------------------------------------------------- sub-pixel phase tables vs brute force (1-D, separable)
def _same_pad(k, s):
    total = k - s
    return total // 2 + total % 2          # pad_front of SamePadConv3d / SamePadConvTranspose3d (videogpt_vq_vae.py:295-300, 318-323)


@pytest.mark.parametrize("k,s", [(4, 2), (4, 1), (3, 1), (1, 1), (5, 2), (6, 3), (2, 2)])
def test_conv_dgrad_phases_are_the_adjoint_of_the_strided_conv(k, s):
    """conv_dgrad_phases (VQ-VAE trainer): dIn computed phase by phase over the input grid == the brute-force adjoint of
    out[o] = sum_kk w[kk] in[o*s + kk - pf] (zero padding)."""
    import numpy as np
    from gsdd_amd.vqvae_trainer import conv_dgrad_phases
    rng = np.random.default_rng(k * 10 + s)
    n_out = 5
    n_in = n_out * s
    pf = _same_pad(k, s)
    w, dout = rng.standard_normal(k), rng.standard_normal(n_out)
    want = np.zeros(n_in)
    for o in range(n_out):
        for kk in range(k):
            i = o * s + kk - pf
            if 0 <= i < n_in:
                want[i] += w[kk] * dout[o]
    got = np.zeros(n_in)
    for (phase, ks, offs) in conv_dgrad_phases((k, 1, 1), (s, 1, 1), (pf, 0, 0)):
        p = phase[0]
        for ic in range(n_in // s):                      # coarse input index: fine index ic*s + p
            for (kk, _, _), (off, _, _) in zip(ks, offs):
                o = ic + off
                if 0 <= o < n_out:
                    got[ic * s + p] += w[kk] * dout[o]
    np.testing.assert_allclose(got, want, atol=1e-12)


@pytest.mark.parametrize("k,s", [(4, 2), (4, 1), (3, 1), (5, 3), (1, 1)])     # the reference layer is size-preserving only for s == 1 or k == s + 2
def test_convT_phases_reproduce_conv_transpose(k, s):
    """convT_phases: out[o'*s + p] = sum_taps w[kk] in[o' + off] == F.conv_transpose1d on the front/back padded input with
    padding = k - 1 (SamePadConvTranspose3d, videogpt_vq_vae.py:312-332)."""
    import numpy as np
    import torch.nn.functional as F
    from gsdd_amd.vqvae import convT_phases
    rng = np.random.default_rng(k * 7 + s)
    n_in = 6
    pf = _same_pad(k, s)
    pb = (k - s) // 2
    w, x = rng.standard_normal(k), rng.standard_normal(n_in)
    xp = F.pad(torch.from_numpy(x).view(1, 1, -1), (pf, pb))
    want = F.conv_transpose1d(xp, torch.from_numpy(w).view(1, 1, -1), stride=s, padding=k - 1).view(-1).numpy()
    assert want.shape[0] == n_in * s
    got = np.zeros(n_in * s)
    for (phase, ks, offs) in convT_phases((k, 1, 1), (s, 1, 1), (pf, 0, 0)):
        p = phase[0]
        for oc in range(n_in):
            for (kk, _, _), (off, _, _) in zip(ks, offs):
                i = oc + off
                if 0 <= i < n_in:
                    got[oc * s + p] += w[kk] * x[i]
    np.testing.assert_allclose(got, want, atol=1e-12)


def test_clip_folder_datamodule_shards_whole_rounds(tmp_path, monkeypatch):
    """set_shard(rank, world): every rank reads only its own batches (b = rank, rank + world, ...), all ranks take the same number of
    steps (a trailing round that cannot serve every rank is dropped), and together they cover the single-process order."""
    import numpy as np
    import src  # noqa: F401
    import gsdd_amd.data
    from src.datamodules.clip_folder_datamodule import ClipFolderDataModule
    rng = np.random.default_rng(0)
    for c in ("a", "b"):
        os.makedirs(tmp_path / "train" / c)
        for i in range(5):
            clip = rng.integers(0, 255, (4, 8, 8, 3), dtype=np.uint8)
            clip[0, 0, 0, 0] = len(os.listdir(tmp_path / "train" / c)) + (100 if c == "b" else 0)      # a recognisable first byte
            np.save(tmp_path / "train" / c / f"clip{i}.npy", clip)
    reads = []
    monkeypatch.setattr(gsdd_amd.data, "preprocess", lambda frames, res: (reads.append(int(frames[0, 0, 0, 0])), frames.float().permute(3, 0, 1, 2))[1])

    def batches(rank, world):
        dm = ClipFolderDataModule(str(tmp_path), sequence_length=4, resolution=8, batch_size=2, device="cpu", shuffle_seed=3)
        dm.set_shard(rank, world)
        reads.clear()
        out = [[int(v[0, 0, 0, 0]) for v in b["video"]] for b in dm.train_dataloader()]
        return out, list(reads)
    single, _ = batches(0, 1)                        # 10 clips, batch 2 -> 5 batches
    assert len(single) == 5
    for world in (2, 3):
        per_rank = [batches(r, world) for r in range(world)]
        rounds = 5 // world
        assert all(len(b) == rounds for b, _ in per_rank)
        for r, (b, read) in enumerate(per_rank):
            assert b == [single[i * world + r] for i in range(rounds)]
            assert len(read) == 2 * rounds            # only its own clips were read and preprocessed
    # an odd clip count: the single process keeps the short trailing batch (as the reference's loader does); data-parallel ranks only
    # ever see full batches (a short one would shift that rank's noise rows onto another rank's and unweight the gradient mean)
    np.save(tmp_path / "train" / "b" / "clip5.npy", rng.integers(0, 255, (4, 8, 8, 3), dtype=np.uint8))
    single, _ = batches(0, 1)
    assert [len(b) for b in single] == [2, 2, 2, 2, 2, 1]
    for world in (2, 3):
        for r in range(world):
            b, _ = batches(r, world)
            assert len(b) == 5 // world and all(len(x) == 2 for x in b)
    # the runner takes a loader's word for being sharded only from the loader itself
    from src.tasks.runner import Trainer
    dm = ClipFolderDataModule(str(tmp_path), sequence_length=4, resolution=8, batch_size=2, device="cpu", shuffle_seed=3)
    dm.set_shard(1, 2)
    assert getattr(dm.train_dataloader(), "gsdd_sharded", False)
    tr = Trainer.__new__(Trainer)
    tr._dm_sharded, tr.limit_batches, tr.rank, tr.world = True, None, 1, 2
    foreign = [10, 11, 12, 13, 14]                     # not the datamodule's: dealt by the runner, whole rounds only
    assert list(tr._batches(foreign)) == [(0, 11), (1, 13)]
    assert [i for i, _ in tr._batches(dm.train_dataloader())] == [0, 1]


def test_clip_text_provider_runs_a_supplied_local_tower(tmp_path):
    """`CLIPTextEmbedding(weights=<local directory>)`: the reference's recipe (clip_text_embedding.py:56-64: tokens truncated to
    start + 20 + end, ids zero-padded to 77, the projected feature at the end-of-text token) on a tower loaded from disk -- here a
    small randomly initialised one of the CLIP text architecture with a byte-level vocabulary, since neither ViT-B/32 nor its BPE
    table exist offline (parity with `clip.encode_text` itself is therefore unpinned).  Without `weights`: the hash stand-in."""
    transformers = pytest.importorskip("transformers")
    import src  # noqa: F401
    from src.models.text_models.clip_text_embedding import CLIPTextEmbedding
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    chars = [chr(c) for c in cs]
    vocab = {c: i for i, c in enumerate(chars)}
    vocab.update({c + "</w>": 256 + i for i, c in enumerate(chars)})
    vocab["<|startoftext|>"], vocab["<|endoftext|>"] = 512, 513
    transformers.CLIPTokenizer(vocab=vocab, merges=[]).save_pretrained(tmp_path)
    cfg = transformers.CLIPTextConfig(vocab_size=514, hidden_size=32, intermediate_size=64, projection_dim=16, num_hidden_layers=2,
                                      num_attention_heads=2, max_position_embeddings=77, bos_token_id=512, eos_token_id=513, pad_token_id=0)
    torch.manual_seed(0)
    tower = transformers.CLIPTextModelWithProjection(cfg)
    tower.save_pretrained(tmp_path)
    p = CLIPTextEmbedding(clip_dim=16, weights=str(tmp_path))
    assert not any(q.requires_grad for q in p.clip_model.parameters()) and not p.train().clip_model.training
    texts = ["a dog runs", "x" * 40, ""]
    ids = p.tokenize(texts)
    assert tuple(ids.shape) == (3, 77) and ids[0, 0] == 512 and int((ids[1] != 0).sum()) == 22 and ids[1, 21] == 513
    assert ids[2, :3].tolist() == [512, 513, 0]
    out = p(texts)
    with torch.no_grad():
        want = tower.eval()(input_ids=ids).text_embeds
    assert tuple(out.shape) == (3, 16) and out.dtype == torch.float32
    torch.testing.assert_close(out, want, atol=1e-6, rtol=1e-6)
    with pytest.raises(FileNotFoundError):
        CLIPTextEmbedding(clip_dim=16, weights=str(tmp_path / "missing"))
    with pytest.raises(ValueError):
        CLIPTextEmbedding(clip_dim=512, weights=str(tmp_path))
    h = CLIPTextEmbedding(clip_dim=512)(["a", "b", "a"])
    assert tuple(h.shape) == (3, 512) and torch.equal(h[0], h[2]) and not torch.equal(h[0], h[1])


def test_lazy_outputs_compute_on_first_access_only():
    """gsdd_amd.d3pm.LazyOutputs (the generator glue's output mapping): deferred values are computed once, when read; `losses`-only
    readers leave them pending; copying the mapping computes them; the mapping takes new keys like the dict the reference returns."""
    from gsdd_amd.d3pm import Deferred, LazyOutputs
    calls = []

    def make(tag):
        def fn():
            calls.append(tag)
            assert not torch.is_grad_enabled()
            return torch.full((2,), float(len(calls)))
        return Deferred(fn)

    out = LazyOutputs(pred_data=make("pred"), gt_data=torch.zeros(2), losses=torch.ones(()), test=make("test"))
    assert set(out) == {"pred_data", "gt_data", "losses", "test"} and len(out) == 4 and "test" in out
    assert out.pending() == ["pred_data", "test"] and calls == []
    assert float(out["losses"]) == 1.0 and out.get("missing") is None and calls == []
    first = out["test"]
    assert calls == ["test"] and out["test"] is first and out.pending() == ["pred_data"]
    out["length"] = [16, 16]
    copied = dict(out)
    assert calls == ["test", "pred"] and out.pending() == [] and copied["length"] == [16, 16] and copied["pred_data"] is out["pred_data"]
    target = {}
    target.update(LazyOutputs(a=make("a"), b=3))
    assert torch.is_tensor(target["a"]) and target["b"] == 3 and calls[-1] == "a"


def test_bench_reads_attention_traffic_of_its_own_grid_from_the_committed_profile():
    """bench.py's `roofline.traffic` is read from profiles/r*_pmc_traffic.csv for the kernel AND the bench shape's grid (8192
    workgroups at 2B = 32 rows, L = 4096, 16 heads); the training forward's launches of the same kernel (grid 4096) must not be mixed in."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full, src = bench.profile_traffic("d3pm_attention_v4_kernel<384, 8>", 8192)
    half, _ = bench.profile_traffic("d3pm_attention_v4_kernel<384, 8>", 4096)
    assert src is not None and src.endswith("_pmc_traffic.csv")
    assert 1.9e8 < full < 2.2e8 and 0.9e8 < half < 1.2e8
    assert bench.profile_traffic("no_such_kernel", 1) == (None, None)


def test_library_reads_no_environment_and_variant_arguments_are_bound(monkeypatch):
    """The C ABI takes arithmetic mode / kernel variant as arguments (include/gsdd.h): no getenv anywhere under csrc/, the ctypes
    prototypes carry the extra int, and the Python wrappers translate the debugging environment variables per call -- explicit
    values only (a typo is an error, not a silent default)."""
    import glob
    import gsdd_amd
    from gsdd_amd import _lib as abi, ops
    for f in glob.glob(os.path.join(REPO, "gif-synthesis-with-discrete-diffusion_amd", "csrc", "*")):
        assert "getenv" not in open(f).read(), f
    header = open(os.path.join(REPO, "include", "gsdd.h")).read()
    for name, value in (("GSDD_ATTN_AUTO", abi.ATTN_AUTO), ("GSDD_ATTN_P22", abi.ATTN_P22), ("GSDD_ATTN_P11", abi.ATTN_P11),
                        ("GSDD_ATTN_A8", abi.ATTN_A8), ("GSDD_ATTN_A12", abi.ATTN_A12), ("GSDD_ATTN_F32PV", abi.ATTN_F32PV),
                        ("GSDD_ATTN_KC256", abi.ATTN_KC256), ("GSDD_LAYER_AUTO", abi.LAYER_AUTO), ("GSDD_LAYER_X3P", abi.LAYER_X3P),
                        ("GSDD_LAYER_H2", abi.LAYER_H2), ("GSDD_GEMM_EXACT_F32", abi.GEMM_EXACT_F32), ("GSDD_AXIAL_VALU", abi.AXIAL_VALU),
                        ("GSDD_ATTN_BWD_VALU", abi.ATTN_BWD_VALU), ("GSDD_ATTN_BWD_SPLIT", abi.ATTN_BWD_SPLIT),
                        ("GSDD_ATTN_BWD_NW8", abi.ATTN_BWD_NW8), ("GSDD_ATTN_BWD_DBG2", abi.ATTN_BWD_DBG2)):
        m = re.search(r"#define\s+" + name + r"\s+(\d+)", header)
        assert m and int(m.group(1)) == value, name
    L = gsdd_amd.lib()
    assert len(L.gsdd_d3pm_attention.argtypes) == 12 and len(L.gsdd_d3pm_attention_train.argtypes) == 12
    assert len(L.gsdd_d3pm_attention_bwd.argtypes) == 15 and len(L.gsdd_axial_attention.argtypes) == 10
    assert [n for n, _ in abi.GemmDesc._fields_][-1] == "flags" and "variant" in dict(abi.LayerDesc._fields_)
    for var in ("GSDD_ATTN_P", "GSDD_ATTN_V3", "GSDD_ATTN_KC", "GSDD_ATTN_TRAIN_P", "GSDD_LAYER", "GSDD_GEMM_F32", "GSDD_AXIAL_VALU",
                "GSDD_ATTN_BWD"):
        monkeypatch.delenv(var, raising=False)
    assert ops.attn_mode() == abi.ATTN_AUTO and ops.attn_mode("11") == abi.ATTN_P11 and ops.attn_mode(abi.ATTN_A12) == abi.ATTN_A12
    monkeypatch.setenv("GSDD_ATTN_P", "a8")
    assert ops.attn_mode() == abi.ATTN_A8 and ops.attn_mode("22") == abi.ATTN_P22          # the keyword wins over the environment
    monkeypatch.setenv("GSDD_ATTN_P", "a9")
    with pytest.raises(gsdd_amd.GsddError):
        ops.attn_mode()
    monkeypatch.setenv("GSDD_ATTN_TRAIN_P", "22")
    assert ops.attn_train_mode() == abi.ATTN_P22
    for bad in ("a8x", "11", "0"):                       # (the advisor's finding: 'a8' used to parse through atoi to 0)
        monkeypatch.setenv("GSDD_ATTN_TRAIN_P", bad)
        with pytest.raises(gsdd_amd.GsddError):
            ops.attn_train_mode()
    monkeypatch.setenv("GSDD_ATTN_TRAIN_P", "a8")
    assert ops.attn_train_mode() == abi.ATTN_A8
    monkeypatch.setenv("GSDD_LAYER", "f32")
    with pytest.raises(gsdd_amd.GsddError):
        ops.layer_variant()
    monkeypatch.setenv("GSDD_GEMM_F32", "1")
    assert ops.gemm_flags() == abi.GEMM_EXACT_F32 and ops.gemm_flags(False) == 0


def test_host_side_of_the_c_abi_under_address_and_ub_sanitizers():
    """tools/host_asan: the C-ABI wrappers of csrc/*.hip compiled host-only with -fsanitize=address,undefined against a stub HIP runtime
    (no kernel runs).  The driver calls every family of entry points with valid descriptors at the workload's sizes, with each workspace
    contract violated by one byte (-> GSDD_E_ARG and no launch), with bad pointers / sizes / variant values, and through a failing
    hipFuncSetAttribute (reported, then retried).  A sanitizer report or a wrong return code fails the run."""
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run(["bash", os.path.join(REPO, "tools", "host_asan", "build.sh"), "run"], cwd=REPO, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "0 failed" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stdout[-2000:] + r.stderr[-2000:]


def test_bench_reads_occupancy_counters_of_its_own_grid_from_the_committed_profile():
    """bench.py's `roofline.counters` / `extra.roofline_families[*].counters` are read from profiles/r*_pmc_sq_counters.csv (a --pmc pass
    over bench.py itself) for the kernel, the weight regime and the bench shape's grid; older files in the per-variant layout of rounds
    1-3 are skipped; an unknown kernel gives None."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    flat = bench.profile_counters("d3pm_attention_v4_kernel<384, 8>", 8192, "flat")
    trained = bench.profile_counters("d3pm_attention_v4_kernel<384, 8>", 8192, "trained_like")
    assert flat is not None and flat["source"].endswith("_pmc_sq_counters.csv") and flat["grid_workgroups"] == 8192
    # the attention kernel is bound by vector issue, not by the matrix pipe; trained-like rows spend more instructions per score
    assert 0.7 < flat["valu_issue_busy"] < 1.0 and 0.2 < flat["mfma_busy"] < 0.5
    assert trained["valu_insts_per_dispatch"] > flat["valu_insts_per_dispatch"]
    step = bench.profile_counters("d3pm_step_kernel", 16384, "flat")
    assert step["mfma_busy"] == 0.0 and step["valu_issue_busy"] > 0.9
    assert bench.profile_counters("d3pm_logits_kernel", 256, "flat")["mfma_busy"] > 0.6
    assert bench.profile_counters("no_such_kernel", None, "flat") is None
    assert bench.profile_counters("d3pm_attention_v4_kernel<384, 8>", 12345, "flat") is None
