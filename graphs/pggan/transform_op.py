"""reference graphs/pggan/transform_op.py:65-77 — the alpha samplers the PGGAN graphs mix in."""
from latent2im_amd.graph import FaceTransform, SceneTransform  # noqa: F401
