"""reference graphs/pggan/pggan_256.py:11-51 — netG / netD holder."""
from latent2im_amd.pggan import PGGAN  # noqa: F401
