"""Clip preprocessing on the HIP path: same name, arguments and result as the reference's
`preprocess` (src/datamodules/datasets/ucf101_dataset.py:105-140), for uint8 clips resident on the GPU."""
import math

import torch

from ._lib import GsddError, check, lib, ptr, stream_ptr


def resize_geometry(h, w, resolution):
    """Target size of the shorter-side resize (:121-127) and the centre-crop offsets (:131-134)."""
    scale = resolution / min(h, w)
    th, tw = (resolution, math.ceil(w * scale)) if h < w else (math.ceil(h * scale), resolution)
    return th, tw, (th - resolution) // 2, (tw - resolution) // 2


def preprocess(video, resolution, sequence_length=None, stream=None):
    """video: (T, H, W, 3) or (N, T, H, W, 3) uint8 on a ROCm device -> (3, T', R, R) / (N, 3, T', R, R) float32."""
    if not video.is_cuda:
        raise GsddError("preprocess runs on the HIP path only: move the uint8 clip to a ROCm device")
    if video.dtype != torch.uint8 or video.shape[-1] != 3 or video.dim() not in (4, 5):
        raise GsddError("expected a uint8 clip laid out (T, H, W, 3) or (N, T, H, W, 3)")
    batched = video.dim() == 5
    v = video.contiguous() if batched else video.contiguous().unsqueeze(0)
    N, T, H, W, _ = v.shape
    t_out = T if sequence_length is None else sequence_length
    if t_out > T:
        raise AssertionError("sequence_length <= t")                      # the reference's assert (:117)
    th, tw, hs, ws = resize_geometry(H, W, resolution)
    out = torch.empty((N, 3, t_out, resolution, resolution), dtype=torch.float32, device=v.device)
    check(lib().gsdd_preprocess_clip(ptr(v), N, T, H, W, t_out, th, tw, hs, ws, resolution, ptr(out), stream_ptr(stream)))
    return out if batched else out[0]
