"""reference graphs/stylegan_v2_real/constants.py — the same module object as latent2im_amd.constants so that
``constants.BATCH_SIZE = n`` set by a driver is seen by the graph."""
import sys

import latent2im_amd.constants as _c

sys.modules[__name__] = _c
