"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's clip preprocessing.

Reference: src/datamodules/datasets/ucf101_dataset.py:105-140 (`preprocess`).  Pinned by tests/golden/preprocess.npz, whose
outputs were produced by the reference function itself (tests/golden/make_golden_preprocess.py).

Written as explicit index arithmetic (not F.interpolate) so that the HIP kernel can be checked against the same formula:
PyTorch's bilinear upsampling with align_corners=False and an explicit output size uses scale = in/out (fp32),
src = max(fma(scale, dst + 0.5, -0.5), 0), i0 = floor(src), i1 = min(i0 + 1, in - 1), lambda = src - i0."""
import math

import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def resize_geometry(h, w, resolution):
    """(:121-127) scale the shorter side to `resolution` (ceil on the other), then the centre-crop offsets (:131-134)."""
    scale = resolution / min(h, w)
    th, tw = (resolution, math.ceil(w * scale)) if h < w else (math.ceil(h * scale), resolution)
    return th, tw, (th - resolution) // 2, (tw - resolution) // 2


def _axis(in_size, out_size, start, n):
    scale = np.float32(in_size) / np.float32(out_size)
    dst = np.arange(start, start + n, dtype=np.float32)
    # one rounding (the reference's torch build contracts scale*(dst+0.5)-0.5 into an FMA): double product of two floats is exact
    src = (scale.astype(np.float64) * (dst + np.float32(0.5)).astype(np.float64) - 0.5).astype(np.float32)
    src = np.maximum(src, np.float32(0))
    i0 = np.floor(src).astype(np.int64)
    i0 = np.minimum(i0, in_size - 1)
    i1 = np.minimum(i0 + 1, in_size - 1)
    lam = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, lam


def preprocess(video, resolution, sequence_length=None):
    """video: (T, H, W, 3) uint8 -> (3, T', resolution, resolution) float32."""
    v = np.asarray(video)
    if sequence_length is not None:
        assert sequence_length <= v.shape[0]
        v = v[:sequence_length]
    t, h, w, _ = v.shape
    x = (v.astype(np.float32) / np.float32(255.0) - MEAN) / STD                     # (:107-111)
    th, tw, hs, ws = resize_geometry(h, w, resolution)
    y0, y1, ly = _axis(h, th, hs, resolution)
    x0, x1, lx = _axis(w, tw, ws, resolution)
    ly = ly[None, :, None, None]
    lx = lx[None, None, :, None]
    one = np.float32(1.0)
    top = (one - lx) * x[:, y0][:, :, x0] + lx * x[:, y0][:, :, x1]
    bot = (one - lx) * x[:, y1][:, :, x0] + lx * x[:, y1][:, :, x1]
    out = (one - ly) * top + ly * bot                                               # (T, R, R, 3)
    return np.ascontiguousarray(out.transpose(3, 0, 1, 2)).astype(np.float32)       # CTHW (:135)
