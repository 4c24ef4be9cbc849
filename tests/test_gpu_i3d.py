"""I3D feature extractor (FVD evaluator) on the HIP conv path against reference-generated fixtures (tests/golden/i3d.npz, made by
tests/golden/make_golden_i3d.py from the reference's InceptionI3d on seeded weights) and end to end through the evaluator."""
import numpy as np
import pytest
import torch

from tests.conftest import parity_report
from tests.test_oracle_golden import i3d_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gsdd_amd
    assert torch.cuda.is_available()
    gsdd_amd.lib()
    return gsdd_amd


@pytest.fixture(scope="module")
def model(G):
    z, sd = i3d_fixture()
    m = G.InceptionI3d()
    m.load_state_dict(sd)
    return z, sd, m.cuda().eval()


def test_i3d_matches_reference(model):
    """Logits (time-averaged) and pooled features of a 16x224x224 clip and of two 32-frame clips, and a strided sample of every end
    point, within 1e-4 of the tensor's scale of the reference module's outputs."""
    z, sd, m = model
    rec = {}
    for tag in ("a", "b"):
        B, T, seed = z[f"x_{tag}"].tolist()
        x = torch.randn(B, 3, T, 224, 224, generator=torch.Generator().manual_seed(seed)).cuda()
        eps = {} if tag == "a" else None
        feats = m.extract_features(x, eps).cpu()
        logits = m(x).cpu()
        want_f, want_l = torch.from_numpy(z[f"features_{tag}"]), torch.from_numpy(z[f"logits_{tag}"])
        assert feats.shape == want_f.shape and logits.shape == want_l.shape
        rec[f"features_{tag}"] = ((feats - want_f).abs().max() / want_f.abs().max()).item()
        rec[f"logits_{tag}"] = ((logits - want_l).abs().max() / want_l.abs().max()).item()
        if eps is not None:
            for name, t in eps.items():
                want = torch.from_numpy(z["ep_" + name])
                got = t[:, ::7, ::3, ::5, ::5].cpu()
                assert got.shape == want.shape, (name, got.shape, want.shape)
                rec["ep_" + name] = ((got - want).abs().max() / want.abs().max()).item()
    parity_report("i3d_vs_reference_rel_err", rec)
    bad = {k: v for k, v in rec.items() if not v < 1e-4}
    assert not bad, bad


def test_i3d_pool_semantics(G):
    """gsdd_pool3d against torch: the 'same' max pool pads with zeros that take part in the maximum (an all-negative border window
    gives 0), channel-slice output addressing, and the unpadded mean pool."""
    import torch.nn.functional as F
    from oracle import i3d as oi
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 24, 5, 9, 11, generator=g) - 1.0                      # mostly negative: the zero padding shows
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, 24).contiguous().cuda()
    for k, s in (((1, 3, 3), (1, 2, 2)), ((3, 3, 3), (2, 2, 2)), ((3, 3, 3), (1, 1, 1)), ((2, 2, 2), (2, 2, 2))):
        want = oi.max_pool_same(x, k, s)
        pads = tuple(oi.same_pad(sz, kk, ss)[0] for sz, kk, ss in zip(x.shape[2:], k, s))
        grid = tuple(want.shape[2:])
        out = torch.full((2 * grid[0] * grid[1] * grid[2], 32), 7.0, device="cuda")          # a wider row: write channels 4..27
        G.ops.pool3d(rows, (2, 5, 9, 11), 24, k, s, pads, grid, out.view(-1)[4:], mode="max", out_pitch=32)
        got = out.view(2, *grid, 32)
        assert torch.equal(got[..., 4:28].permute(0, 4, 1, 2, 3).cpu(), want), (k, s)
        assert bool((got[..., :4] == 7.0).all()) and bool((got[..., 28:] == 7.0).all())
    want = F.avg_pool3d(x, (2, 7, 7), (1, 1, 1))
    grid = tuple(want.shape[2:])
    out = torch.empty((2 * grid[0] * grid[1] * grid[2], 24), device="cuda")
    G.ops.pool3d(rows, (2, 5, 9, 11), 24, (2, 7, 7), (1, 1, 1), (0, 0, 0), grid, out, mode="mean")
    torch.testing.assert_close(out.view(2, *grid, 24).permute(0, 4, 1, 2, 3).cpu(), want, atol=1e-6, rtol=1e-6)


def test_evaluator_runs_i3d_end_to_end(model, tmp_path):
    """configs/model/evaluator.yaml's object graph with a supplied I3D state_dict: push_vals on (generated, real) 16x128x128 clips
    -> resize to 224 on the preprocessing kernel -> I3D logits on the HIP path -> Frechet distance; the same statistic from the
    CPU oracle's I3D on the same prepared clips agrees."""
    from oracle import i3d as oi
    from src.utils.evaluator import Evaluator
    from gsdd_amd.metrics import frechet_distance
    z, sd, _ = model
    ck = tmp_path / "i3d.pt"
    torch.save(sd, ck)
    ev = Evaluator("cuda", {"_target_": "src.models.motionencoder.pytorch_i3d.InceptionI3d"}, str(ck))
    assert not ev.videoencoder.training
    g = torch.Generator().manual_seed(4)
    feats_gen, feats_gt = [], []
    for i in range(2):
        real = torch.randn(3, 3, 16, 128, 128, generator=g)
        fake = 0.8 * torch.randn(3, 3, 16, 128, 128, generator=g) + 0.1
        ev.push_vals({"video": real.cuda()}, i, fake.cuda())
        with torch.no_grad():
            feats_gen.append(oi.forward(ev._prepare(fake.cuda()).cpu(), sd))
            feats_gt.append(oi.forward(ev._prepare(real.cuda()).cpu(), sd))
    got = ev.evaluate_metrics()["fvd"]
    want = float(frechet_distance(torch.cat(feats_gen), torch.cat(feats_gt)))
    parity_report("evaluator_i3d_fvd", {"fvd_hip": got, "fvd_oracle_features": want})
    assert np.isfinite(got) and abs(got - want) <= 1e-2 * abs(want), (got, want)     # (6 samples in 400 dimensions: sqrt of rank-deficient covariances)
