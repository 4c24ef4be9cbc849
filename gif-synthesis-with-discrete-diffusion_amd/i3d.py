"""Inception-3D feature extractor of the FVD evaluator on the HIP conv path.

Drop-in for src/models/motionencoder/pytorch_i3d.py (InceptionI3d :135-327, Unit3D :37-98, MaxPool3dSamePadding :7-34,
InceptionModule :102-132): same constructor, `forward` (logits averaged over time), `extract_features`, `replace_logits`, and the
reference's `state_dict` key names and shapes (`Conv3d_1a_7x7.conv3d.weight`, `Mixed_4b.b1a.bn.running_mean`, `logits.conv3d.bias`
...), so a checkpoint made for the reference module loads unchanged.  As in vqvae.py, the nn.Module tree only OWNS parameters; the
forward runs channels-last on the C ABI:

  Unit3D (TF-"same" zero padding + Conv3d + BatchNorm(eval) + ReLU) -> one gsdd_gemm implicit GEMM (asymmetric padding = tap offsets
                                                                      with bounds, BatchNorm folded into the epilogue scale/shift)
  first 7x7x7 / stride 2 conv on 3 channels                         -> kw merged into the contraction over W-padded NDHWC4 rows
  MaxPool3dSamePadding, AvgPool3d                                   -> gsdd_pool3d (padding counts as 0 in the maximum, as F.pad does)
  InceptionModule's torch.cat                                       -> every branch writes its channel slice of the output rows

Eval mode only (the FVD evaluator's use; the reference never calls .eval() on it -- src/utils/evaluator.py:14-29 -- so its
published numbers ran BatchNorm on batch statistics with dropout active; src/utils/evaluator.py here calls .eval()).
Pretrained Kinetics weights are not obtainable offline: parity is pinned on seeded random weights (tests/golden/i3d.npz).
"""
import torch
import torch.nn as nn

from . import ops
from ._lib import GsddError
from .vqvae import conv_taps, fold_bn, pack_conv0_weight, pack_conv_weight


def same_pad(size, k, s):
    """compute_pad + front/back split (pytorch_i3d.py:9-13, 26-31) -> (front, back)"""
    pad = max(k - s, 0) if size % s == 0 else max(k - (size % s), 0)
    return pad // 2, pad - pad // 2


class MaxPool3dSamePadding(nn.Module):
    def __init__(self, kernel_size, stride, padding=0):
        super().__init__()
        self.kernel_size, self.stride = tuple(kernel_size), tuple(stride)


class Unit3D(nn.Module):
    """Parameter holder for pytorch_i3d.py:37-98."""

    def __init__(self, in_channels, output_channels, kernel_shape=(1, 1, 1), stride=(1, 1, 1), padding=0, activation_fn="relu",
                 use_batch_norm=True, use_bias=False, name="unit_3d"):
        super().__init__()
        self._kernel_shape, self._stride = tuple(kernel_shape), tuple(stride)
        self._use_batch_norm, self._relu, self.name = use_batch_norm, activation_fn is not None, name
        self.conv3d = nn.Conv3d(in_channels, output_channels, self._kernel_shape, stride=self._stride, padding=0, bias=use_bias)
        if use_batch_norm:
            self.bn = nn.BatchNorm3d(output_channels, eps=1e-5, momentum=0.001)


class InceptionModule(nn.Module):
    def __init__(self, in_channels, out_channels, name):
        super().__init__()
        o = out_channels
        self.b0 = Unit3D(in_channels, o[0], name=name + "/Branch_0/Conv3d_0a_1x1")
        self.b1a = Unit3D(in_channels, o[1], name=name + "/Branch_1/Conv3d_0a_1x1")
        self.b1b = Unit3D(o[1], o[2], kernel_shape=(3, 3, 3), name=name + "/Branch_1/Conv3d_0b_3x3")
        self.b2a = Unit3D(in_channels, o[3], name=name + "/Branch_2/Conv3d_0a_1x1")
        self.b2b = Unit3D(o[3], o[4], kernel_shape=(3, 3, 3), name=name + "/Branch_2/Conv3d_0b_3x3")
        self.b3a = MaxPool3dSamePadding((3, 3, 3), (1, 1, 1))
        self.b3b = Unit3D(in_channels, o[5], name=name + "/Branch_3/Conv3d_0b_1x1")
        self.out_channels = tuple(o)
        self.name = name


class InceptionI3d(nn.Module):
    VALID_ENDPOINTS = ("Conv3d_1a_7x7", "MaxPool3d_2a_3x3", "Conv3d_2b_1x1", "Conv3d_2c_3x3", "MaxPool3d_3a_3x3", "Mixed_3b",
                       "Mixed_3c", "MaxPool3d_4a_3x3", "Mixed_4b", "Mixed_4c", "Mixed_4d", "Mixed_4e", "Mixed_4f", "MaxPool3d_5a_2x2",
                       "Mixed_5b", "Mixed_5c", "Logits", "Predictions")

    def __init__(self, num_classes=400, spatial_squeeze=True, final_endpoint="Logits", name="inception_i3d", in_channels=3,
                 dropout_keep_prob=0.5):
        if final_endpoint not in self.VALID_ENDPOINTS:
            raise ValueError("Unknown final endpoint %s" % final_endpoint)
        if final_endpoint != "Logits":
            raise NotImplementedError("only final_endpoint='Logits' (the evaluator's configuration) is built")
        super().__init__()
        self._num_classes, self._spatial_squeeze, self._final_endpoint = num_classes, spatial_squeeze, final_endpoint
        self.in_channels = in_channels
        mods = [("Conv3d_1a_7x7", Unit3D(in_channels, 64, (7, 7, 7), (2, 2, 2), (3, 3, 3), name=name + "Conv3d_1a_7x7")),
                ("MaxPool3d_2a_3x3", MaxPool3dSamePadding((1, 3, 3), (1, 2, 2))),
                ("Conv3d_2b_1x1", Unit3D(64, 64, name=name + "Conv3d_2b_1x1")),
                ("Conv3d_2c_3x3", Unit3D(64, 192, (3, 3, 3), padding=1, name=name + "Conv3d_2c_3x3")),
                ("MaxPool3d_3a_3x3", MaxPool3dSamePadding((1, 3, 3), (1, 2, 2))),
                ("Mixed_3b", InceptionModule(192, [64, 96, 128, 16, 32, 32], name + "Mixed_3b")),
                ("Mixed_3c", InceptionModule(256, [128, 128, 192, 32, 96, 64], name + "Mixed_3c")),
                ("MaxPool3d_4a_3x3", MaxPool3dSamePadding((3, 3, 3), (2, 2, 2))),
                ("Mixed_4b", InceptionModule(480, [192, 96, 208, 16, 48, 64], name + "Mixed_4b")),
                ("Mixed_4c", InceptionModule(512, [160, 112, 224, 24, 64, 64], name + "Mixed_4c")),
                ("Mixed_4d", InceptionModule(512, [128, 128, 256, 24, 64, 64], name + "Mixed_4d")),
                ("Mixed_4e", InceptionModule(512, [112, 144, 288, 32, 64, 64], name + "Mixed_4e")),
                ("Mixed_4f", InceptionModule(528, [256, 160, 320, 32, 128, 128], name + "Mixed_4f")),
                ("MaxPool3d_5a_2x2", MaxPool3dSamePadding((2, 2, 2), (2, 2, 2))),
                ("Mixed_5b", InceptionModule(832, [256, 160, 320, 32, 128, 128], name + "Mixed_5b")),
                ("Mixed_5c", InceptionModule(832, [384, 192, 384, 48, 128, 128], name + "Mixed_5c"))]
        self.end_points = dict(mods)
        self.avg_pool = nn.AvgPool3d(kernel_size=[2, 7, 7], stride=(1, 1, 1))
        self.dropout = nn.Dropout(dropout_keep_prob)
        self.logits = Unit3D(1024, num_classes, activation_fn=None, use_batch_norm=False, use_bias=True, name="logits")
        for k, m in mods:                                     # build() (:303-305): registered after `logits`, like the reference
            self.add_module(k, m)
        self._packed, self._packed_key = None, None

    def replace_logits(self, num_classes):
        self._num_classes = num_classes
        self.logits = Unit3D(1024, num_classes, activation_fn=None, use_batch_norm=False, use_bias=True, name="logits")

    # ------------------------------------------------------------------ packed weights (BatchNorm folded, tap tables)
    def _state_key(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def _pack_unit(self, u, dev, first=False):
        w = u.conv3d.weight
        d = dict(k=u._kernel_shape, s=u._stride, relu=u._relu, cout=w.shape[0], first=first)
        d["w"] = pack_conv0_weight(w) if first else pack_conv_weight(w)
        if u._use_batch_norm:
            d["scale"], d["shift"] = fold_bn(u.bn)
        else:
            d["scale"], d["shift"] = None, (u.conv3d.bias.contiguous() if u.conv3d.bias is not None else None)
        return d

    def packed(self):
        key = self._state_key()
        if self._packed is None or key != self._packed_key:
            dev = self.logits.conv3d.weight.device
            with torch.no_grad():
                p = {}
                for name, m in self.end_points.items():
                    if isinstance(m, Unit3D):
                        p[name] = self._pack_unit(m, dev, first=(name == "Conv3d_1a_7x7" and self.in_channels <= 4))
                    elif isinstance(m, InceptionModule):
                        p[name] = {b: self._pack_unit(getattr(m, b), dev) for b in ("b0", "b1a", "b1b", "b2a", "b2b", "b3b")}
                p["logits"] = self._pack_unit(self.logits, dev)
            self._packed, self._packed_key, self._taps = p, key, {}
        return self._packed

    def _tap_table(self, k, pads, dev):
        key = (k, pads)
        if key not in self._taps:
            self._taps[key] = ops.taps_tensor(conv_taps(k, (1, 1, 1), pads), dev)        # offsets k - pad_front (stride lives in the GEMM)
        return self._taps[key]

    # ------------------------------------------------------------------ HIP pipeline, channels-last rows [N*T*H*W][C]
    def _unit(self, h, dims, u, out=None, out_pitch=None):
        """One Unit3D on rows h (or, for the first conv, on the NCDHW clip): -> (out rows, out dims)."""
        N, T, H, W = dims
        k, s = u["k"], u["s"]
        pads = tuple(same_pad(sz, kk, ss) for sz, kk, ss in zip((T, H, W), k, s))
        To, Ho, Wo = -(-T // s[0]), -(-H // s[1]), -(-W // s[2])
        dev = h.device
        if out is None:
            out = torch.empty((N * To * Ho * Wo, u["cout"]), dtype=torch.float32, device=dev)
        act = ops.ACT_RELU if u["relu"] else ops.ACT_NONE
        if u["first"]:
            # kw merged into the contraction: rows of 4 channels over a W axis padded so that every 7-pixel run stays inside its row
            pf, pb = pads[2]
            padw = max(pf, (Wo - 1) * s[2] + k[2] - pf - W)            # front >= pf; back covers the last window
            padw = max(padw, pb)
            xr = ops.ncdhw_to_rows(h, 4, padw)                         # (N,T,H,W+2 padw,4)
            taps = ops.taps_tensor([(a - pads[0][0], b - pads[1][0], padw - pf) for a in range(k[0]) for b in range(k[1])], dev)
            ops.gemm(xr, u["w"], out, in_dims=(N, T, H, W + 2 * padw), out_grid=(To, Ho, Wo), stride=s, taps=taps,
                     ntaps=u["w"].shape[0], cin=u["w"].shape[2], in_pitch=4, epi_scale=u["scale"], epi_shift=u["shift"], act=act,
                     out_pitch=out_pitch, cout=u["cout"])
        else:
            ntaps = k[0] * k[1] * k[2]
            taps = self._tap_table(k, tuple(p[0] for p in pads), dev) if ntaps > 1 else None
            ops.gemm(h, u["w"], out, in_dims=dims, out_grid=(To, Ho, Wo), stride=s, taps=taps, ntaps=ntaps, epi_scale=u["scale"],
                     epi_shift=u["shift"], act=act, out_pitch=out_pitch, cout=u["cout"])
        return out, (N, To, Ho, Wo)

    def _pool(self, h, dims, C_, k, s, mode="max", same=True, out=None, out_pitch=None):
        N, T, H, W = dims
        if same:
            pads = tuple(same_pad(sz, kk, ss)[0] for sz, kk, ss in zip((T, H, W), k, s))
            grid = (-(-T // s[0]), -(-H // s[1]), -(-W // s[2]))
        else:
            pads = (0, 0, 0)
            grid = tuple((sz - kk) // ss + 1 for sz, kk, ss in zip((T, H, W), k, s))
            if min(grid) < 1:
                raise GsddError(f"I3D: a {T}x{H}x{W} feature map is smaller than the {k} pooling window (clips need >= 16 frames "
                                "of 224x224, pytorch_i3d.py:296)")
        if out is None:
            out = torch.empty((N * grid[0] * grid[1] * grid[2], C_), dtype=torch.float32, device=h.device)
        ops.pool3d(h, dims, C_, k, s, pads, grid, out, mode=mode, out_pitch=out_pitch)
        return out, (N,) + grid

    def _mixed(self, h, dims, cin, m, pk):
        """InceptionModule.forward (:127-132); the four branches write their channel slices of one output row."""
        o = m.out_channels
        ctot = o[0] + o[2] + o[4] + o[5]
        M = h.shape[0]
        out = torch.empty((M, ctot), dtype=torch.float32, device=h.device)
        flat = out.view(-1)
        self._unit(h, dims, pk["b0"], out=flat, out_pitch=ctot)
        a, _ = self._unit(h, dims, pk["b1a"])
        self._unit(a, dims, pk["b1b"], out=flat[o[0]:], out_pitch=ctot)
        a, _ = self._unit(h, dims, pk["b2a"])
        self._unit(a, dims, pk["b2b"], out=flat[o[0] + o[2]:], out_pitch=ctot)
        a, _ = self._pool(h, dims, cin, (3, 3, 3), (1, 1, 1))
        self._unit(a, dims, pk["b3b"], out=flat[o[0] + o[2] + o[4]:], out_pitch=ctot)
        return out, ctot

    def _features_rows(self, x, endpoints=None):
        if not x.is_cuda:
            raise GsddError("InceptionI3d runs on the HIP path only: move the module and the clips to a ROCm device")
        if self.training:
            raise GsddError("InceptionI3d is built for eval mode (BatchNorm running statistics, no dropout): call .eval()")
        p = self.packed()
        x = x.contiguous().float()
        N, _, T, H, W = x.shape
        h, dims, C_ = x, (N, T, H, W), self.in_channels
        for name, m in self.end_points.items():
            if isinstance(m, Unit3D):
                if not p[name]["first"] and h.dim() == 5:               # in_channels > 4: plain channels-last rows
                    h = h.permute(0, 2, 3, 4, 1).reshape(-1, C_).contiguous()
                h, dims = self._unit(h, dims, p[name])
                C_ = p[name]["cout"]
            elif isinstance(m, MaxPool3dSamePadding):
                h, dims = self._pool(h, dims, C_, m.kernel_size, m.stride)
            else:
                h, C_ = self._mixed(h, dims, C_, m, p[name])
            if endpoints is not None:
                endpoints[name] = h.view(*dims, C_).permute(0, 4, 1, 2, 3)
        h, dims = self._pool(h, dims, C_, (2, 7, 7), (1, 1, 1), mode="mean", same=False)
        return h, dims, C_

    @torch.no_grad()
    def extract_features(self, x, endpoints=None):
        """(B,3,T,H,W) -> (B,1024,T',1,1)  (pytorch_i3d.py:323-327)"""
        h, dims, C_ = self._features_rows(x, endpoints)
        return h.view(*dims, C_).permute(0, 4, 1, 2, 3).contiguous()

    @torch.no_grad()
    def forward(self, x):
        """(B,3,T,H,W) -> (B,num_classes): the logits averaged over time (pytorch_i3d.py:309-320)."""
        h, dims, C_ = self._features_rows(x)
        lg, dims = self._unit(h, dims, self.packed()["logits"])
        lg = lg.view(dims[0], dims[1], dims[2], dims[3], -1)
        if self._spatial_squeeze:
            lg = lg.squeeze(3).squeeze(2)                                # H', W' == 1 -> (B,T',classes)
        return lg.mean(dim=1)
