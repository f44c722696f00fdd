"""reference graphs/stylegan_v2_real/transform_op.py:65-77 (face / scene alpha samplers)."""
from latent2im_amd.graph import FaceTransform, SceneTransform  # noqa: F401
