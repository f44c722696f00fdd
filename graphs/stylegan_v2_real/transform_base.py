"""reference graphs/stylegan_v2_real/transform_base.py — walk modules and the TransformGraph; pickled walk checkpoints
(``model_w_<epoch>_walk_module.ckpt``) name this module path, so every class whose ``__module__`` points here
(latent2im_amd/graph.py) must be importable from here."""
from latent2im_amd.graph import (ContentLoss, PixelTransform, StyleGAN, TransformGraph,  # noqa: F401
                                 WalkLinearMultiW, WalkMlpMultiW, WalkNonLinearW)
