"""CPU restatement (torch-CPU ops) of the reference's I3D feature extractor -- TEST INFRASTRUCTURE, not product code: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Reference: src/models/motionencoder/pytorch_i3d.py (InceptionI3d.forward :309-320, extract_features :323-327; Unit3D.forward
:72-98; MaxPool3dSamePadding.forward :15-34; InceptionModule.forward :127-132).  A functional forward over a state_dict with the
reference's key names, eval mode (BatchNorm running statistics, dropout = identity).  Pinned against outputs of the reference
module itself (tests/golden/make_golden_i3d.py -> tests/golden/i3d.npz, checked in tests/test_oracle_golden.py).
"""
import torch
import torch.nn.functional as F

# (name, kind, arguments) in VALID_ENDPOINTS order (pytorch_i3d.py:150-169, :197-293)
ENDPOINTS = (
    ("Conv3d_1a_7x7", "unit", dict(cin=3, cout=64, k=(7, 7, 7), s=(2, 2, 2))),
    ("MaxPool3d_2a_3x3", "pool", dict(k=(1, 3, 3), s=(1, 2, 2))),
    ("Conv3d_2b_1x1", "unit", dict(cin=64, cout=64, k=(1, 1, 1), s=(1, 1, 1))),
    ("Conv3d_2c_3x3", "unit", dict(cin=64, cout=192, k=(3, 3, 3), s=(1, 1, 1))),
    ("MaxPool3d_3a_3x3", "pool", dict(k=(1, 3, 3), s=(1, 2, 2))),
    ("Mixed_3b", "mixed", dict(cin=192, out=(64, 96, 128, 16, 32, 32))),
    ("Mixed_3c", "mixed", dict(cin=256, out=(128, 128, 192, 32, 96, 64))),
    ("MaxPool3d_4a_3x3", "pool", dict(k=(3, 3, 3), s=(2, 2, 2))),
    ("Mixed_4b", "mixed", dict(cin=480, out=(192, 96, 208, 16, 48, 64))),
    ("Mixed_4c", "mixed", dict(cin=512, out=(160, 112, 224, 24, 64, 64))),
    ("Mixed_4d", "mixed", dict(cin=512, out=(128, 128, 256, 24, 64, 64))),
    ("Mixed_4e", "mixed", dict(cin=512, out=(112, 144, 288, 32, 64, 64))),
    ("Mixed_4f", "mixed", dict(cin=528, out=(256, 160, 320, 32, 128, 128))),
    ("MaxPool3d_5a_2x2", "pool", dict(k=(2, 2, 2), s=(2, 2, 2))),
    ("Mixed_5b", "mixed", dict(cin=832, out=(256, 160, 320, 32, 128, 128))),
    ("Mixed_5c", "mixed", dict(cin=832, out=(384, 192, 384, 48, 128, 128))),
)
BN_EPS = 1e-5                     # pytorch_i3d.py:66


def same_pad(size, k, s):
    """compute_pad (:9-13, :68-72) + the front/back split of forward (:26-31, :86-91) -> (front, back)."""
    pad = max(k - s, 0) if size % s == 0 else max(k - (size % s), 0)
    return pad // 2, pad - pad // 2


def _pad(x, k, s):
    t, h, w = x.shape[2:]
    (tf, tb), (hf, hb), (wf, wb) = same_pad(t, k[0], s[0]), same_pad(h, k[1], s[1]), same_pad(w, k[2], s[2])
    return F.pad(x, (wf, wb, hf, hb, tf, tb))


def unit3d(x, sd, p, k, s, bn=True, relu=True):
    """Unit3D.forward (:72-98): dynamic 'same' zero padding, conv, BatchNorm (eval), ReLU."""
    x = F.conv3d(_pad(x, k, s), sd[p + "conv3d.weight"], sd.get(p + "conv3d.bias"), stride=s)
    if bn:
        x = F.batch_norm(x, sd[p + "bn.running_mean"], sd[p + "bn.running_var"], sd[p + "bn.weight"], sd[p + "bn.bias"], False, 0.0,
                         BN_EPS)
    return F.relu(x) if relu else x


def max_pool_same(x, k, s):
    """MaxPool3dSamePadding.forward (:15-34): zero padding, then an unpadded max pool (the zeros take part in the maximum)."""
    return F.max_pool3d(_pad(x, k, s), k, s)


def mixed(x, sd, p):
    """InceptionModule.forward (:127-132)."""
    one, three = (1, 1, 1), (3, 3, 3)
    b0 = unit3d(x, sd, p + "b0.", one, one)
    b1 = unit3d(unit3d(x, sd, p + "b1a.", one, one), sd, p + "b1b.", three, one)
    b2 = unit3d(unit3d(x, sd, p + "b2a.", one, one), sd, p + "b2b.", three, one)
    b3 = unit3d(max_pool_same(x, three, one), sd, p + "b3b.", one, one)
    return torch.cat([b0, b1, b2, b3], dim=1)


def extract_features(x, sd, endpoints=None):
    """InceptionI3d.extract_features (:323-327) -> (B,1024,T',1,1).  `endpoints`: optional dict filled with every end point's output."""
    for name, kind, a in ENDPOINTS:
        if kind == "unit":
            x = unit3d(x, sd, name + ".", a["k"], a["s"])
        elif kind == "pool":
            x = max_pool_same(x, a["k"], a["s"])
        else:
            x = mixed(x, sd, name + ".")
        if endpoints is not None:
            endpoints[name] = x
    return F.avg_pool3d(x, (2, 7, 7), (1, 1, 1))                    # :296


def forward(x, sd):
    """InceptionI3d.forward (:309-320) in eval mode -> (B, num_classes): logits averaged over time."""
    x = unit3d(extract_features(x, sd), sd, "logits.", (1, 1, 1), (1, 1, 1), bn=False, relu=False)
    return x.squeeze(3).squeeze(3).mean(dim=2)


def seeded_state_dict(keys_shapes, seed):
    """The weight recipe shared by the fixture generator (applied to the reference module's state_dict) and the tests (applied to
    the product module's): He-scaled conv weights so that 20 layers keep O(1) activations, non-trivial BatchNorm statistics.
    keys_shapes: ordered (key, shape) pairs; the draw order is the key order, so both sides must enumerate the same keys."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in keys_shapes:
        shape = tuple(shape)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros(shape, dtype=torch.long)
        elif k.endswith("conv3d.weight"):
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            sd[k] = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif k.endswith("bn.weight") or k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(shape, generator=g)
        else:                                                           # bn.bias, running_mean, conv3d.bias
            sd[k] = 0.1 * torch.randn(shape, generator=g)
    return sd
