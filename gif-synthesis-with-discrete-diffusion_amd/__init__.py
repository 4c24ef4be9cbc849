"""MI355X-native video-token generation path (VQ-VAE encode/quantise/decode + D3PM reverse diffusion).

The directory name carries a hyphen, so import it through the root-level ``gsdd_amd`` shim:

    import gsdd_amd
    vq = gsdd_amd.VQVAE(...).cuda().eval()
"""
from . import ops  # noqa: F401
from ._lib import GsddError, LIB_PATH, EXPORTS, lib  # noqa: F401
from .d3pm import (DalleMaskImageEmbedding, DiffusionTransformer, DiscreteDiffusion,  # noqa: F401
                   Text2ImageTransformer)
from .i3d import InceptionI3d  # noqa: F401
from .vqvae import VQVAE  # noqa: F401
