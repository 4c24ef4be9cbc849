"""`src.models.motionencoder.diffusion_transformer.DiffusionTransformer` -> HIP-backed drop-in."""
from gsdd_amd.d3pm import DiffusionTransformer, alpha_schedule  # noqa: F401
