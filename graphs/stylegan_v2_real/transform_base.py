"""reference graphs/stylegan_v2_real/transform_base.py — walk modules and the TransformGraph; pickled walk checkpoints
(``model_w_<epoch>_walk_module.ckpt``) name this module path."""
from latent2im_amd.graph import (ContentLoss, PixelTransform, StyleGAN, TransformGraph,  # noqa: F401
                                 WalkLinearMultiW)
