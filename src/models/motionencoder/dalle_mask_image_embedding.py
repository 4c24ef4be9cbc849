"""`src.models.motionencoder.dalle_mask_image_embedding.DalleMaskImageEmbedding` -> HIP-backed drop-in."""
from gsdd_amd.d3pm import DalleMaskImageEmbedding  # noqa: F401
