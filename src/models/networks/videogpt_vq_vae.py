"""`src.models.networks.videogpt_vq_vae.VQVAE` -> HIP-backed drop-in (reference file of the same name)."""
from gsdd_amd.vqvae import VQVAE  # noqa: F401
