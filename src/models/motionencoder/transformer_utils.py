"""`src.models.motionencoder.transformer_utils.Text2ImageTransformer` -> HIP-backed drop-in."""
from gsdd_amd.d3pm import Text2ImageTransformer  # noqa: F401
