"""reference utils/util.py:5-121."""
from latent2im_amd.hostutil import batch_input, set_graph_kwargs  # noqa: F401
