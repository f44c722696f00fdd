"""reference graphs/pggan/constants.py — the same module object as latent2im_amd.constants (BATCH_SIZE / DIM_Z / NUM_CHANNELS are shared
with the StyleGAN2 graph; the PGGAN resolution is ``PG_RESOLUTION``)."""
import sys

import latent2im_amd.constants as _c

sys.modules[__name__] = _c
