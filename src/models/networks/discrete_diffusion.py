"""`src.models.networks.discrete_diffusion.DiscreteDiffusion` -> HIP-backed drop-in."""
from gsdd_amd.d3pm import DiscreteDiffusion  # noqa: F401
