"""reference graphs/stylegan_v2_real/networks.py — frozen, HIP-backed Generator / Discriminator."""
from latent2im_amd.discriminator import Discriminator  # noqa: F401
from latent2im_amd.generator import Generator  # noqa: F401
