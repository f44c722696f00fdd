"""reference graphs/stylegan_v2_real/graph_util.py:5-19."""
from latent2im_amd.hostutil import graph_input, z_sample  # noqa: F401
