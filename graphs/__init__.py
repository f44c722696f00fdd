"""``graphs`` plugin package with the reference's lookup entry point (reference graphs/__init__.py:3-22)."""
from latent2im_amd.graph import find_model_using_name  # noqa: F401
