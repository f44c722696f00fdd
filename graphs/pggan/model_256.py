"""reference graphs/pggan/model_256.py:188-259 — the frozen PGGAN-256 generator on the HIP kernels."""
from latent2im_amd.pggan import Generator  # noqa: F401
