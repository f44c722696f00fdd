"""reference options/vis_options.py."""
from latent2im_amd.vis import VisOptions  # noqa: F401
