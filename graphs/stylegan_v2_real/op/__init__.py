"""reference graphs/stylegan_v2_real/op/__init__.py:1-2."""
from latent2im_amd.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d  # noqa: F401
