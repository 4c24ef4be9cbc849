"""Clip preprocessing kernel (gsdd_preprocess_clip) against the reference function's outputs (tests/golden/preprocess.npz) and
the CPU oracle; full-size properties (UCF101 frames are 240x320)."""
import os

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


def test_preprocess_matches_reference_outputs(G):
    from tests.conftest import GOLDEN
    from gsdd_amd.data import preprocess
    z = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    for i in range(5):
        r, sl = (int(v) for v in z[f"cfg{i}"])
        out = preprocess(torch.from_numpy(z[f"in{i}"]).cuda(), r, None if sl < 0 else sl)
        np.testing.assert_allclose(out.cpu().numpy(), z[f"out{i}"], atol=1e-6, rtol=0)      # fp32, 1-3 ulp of association


def test_preprocess_full_size_batch_and_properties(G):
    from gsdd_amd.data import preprocess
    from oracle import preprocess as op
    rng = np.random.default_rng(0)
    clips = rng.integers(0, 256, size=(3, 16, 240, 320, 3), dtype=np.uint8)             # UCF101 frame size
    out = preprocess(torch.from_numpy(clips).cuda(), 128, 16)
    assert tuple(out.shape) == (3, 3, 16, 128, 128)
    want = op.preprocess(clips[1], 128, 16)
    np.testing.assert_allclose(out[1].cpu().numpy(), want, atol=1e-6, rtol=0)
    # a constant frame stays constant (bilinear weights sum to one), and no-resize + full crop is the plain normalisation
    flat = np.full((1, 2, 64, 64, 3), 200, dtype=np.uint8)
    o = preprocess(torch.from_numpy(flat).cuda(), 64).cpu().numpy()
    mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
    np.testing.assert_allclose(o[0], np.broadcast_to(((np.float32(200) / np.float32(255) - mean) / std)[:, None, None, None], o[0].shape),
                               atol=1e-6)
    sq = rng.integers(0, 256, size=(2, 32, 32, 3), dtype=np.uint8)
    o = preprocess(torch.from_numpy(sq).cuda(), 32).cpu().numpy()
    np.testing.assert_allclose(o, ((sq.astype(np.float32) / np.float32(255) - mean) / std).transpose(3, 0, 1, 2), atol=1e-6)


def test_preprocess_rejects_bad_input(G):
    from gsdd_amd.data import preprocess
    with pytest.raises(G.GsddError):
        preprocess(torch.zeros((2, 8, 8, 3), dtype=torch.uint8), 8)                       # CPU tensor: no fallback
    with pytest.raises(G.GsddError):
        preprocess(torch.zeros((2, 8, 8, 3), dtype=torch.float32, device="cuda"), 8)
    with pytest.raises(AssertionError):
        preprocess(torch.zeros((2, 8, 8, 3), dtype=torch.uint8, device="cuda"), 8, sequence_length=3)


def test_clip_folder_datamodule_batches(G, tmp_path):
    from oracle import preprocess as op
    from src.datamodules.clip_folder_datamodule import ClipFolderDataModule
    rng = np.random.default_rng(1)
    clips = {}
    for cls in ("Swing", "Archery"):
        d = tmp_path / "test" / cls
        d.mkdir(parents=True)
        clips[cls] = rng.integers(0, 256, size=(5, 30, 44, 3), dtype=np.uint8)
        np.save(d / "a.npy", clips[cls])
    dm = ClipFolderDataModule(str(tmp_path), sequence_length=4, resolution=16, batch_size=2)
    batches = list(dm.test_dataloader())
    assert len(batches) == 1
    b = batches[0]
    assert tuple(b["video"].shape) == (2, 3, 4, 16, 16) and b["text"] == ["Archery", "Swing"] and b["label"].tolist() == [0, 1]
    assert b["length"] == [3, 3] and b["orig_length"] == [4, 4]
    np.testing.assert_allclose(b["video"][1].cpu().numpy(), op.preprocess(clips["Swing"][:4], 16), atol=1e-6, rtol=0)
