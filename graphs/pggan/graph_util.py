"""reference graphs/pggan/graph_util.py:5-14."""
from latent2im_amd.hostutil import graph_input, z_sample  # noqa: F401
