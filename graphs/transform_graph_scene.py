"""reference graphs/transform_graph_scene.py:5-125."""
from latent2im_amd.graph import SceneGraph, faceGraph, get_transform_graphs  # noqa: F401
