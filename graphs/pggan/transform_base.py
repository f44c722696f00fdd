"""reference graphs/pggan/transform_base.py — WalkLinearZ_free (:86-102) and the PGGAN TransformGraph (:211-640); walk checkpoints pickle
this module path (latent2im_amd/pggan.py sets ``WalkLinearZ_free.__module__``)."""
from latent2im_amd.pggan import PGGAN, ContentLoss, PixelTransform, TransformGraph, WalkLinearZ_free  # noqa: F401
