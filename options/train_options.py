"""reference options/train_options.py."""
from latent2im_amd.options import TrainOptions  # noqa: F401
