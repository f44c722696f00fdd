"""Drop-in for the reference's ``train_multi_attr.py`` (several attributes at once, clamped targets, 3 epochs)."""
from latent2im_amd.trainer import main

if __name__ == '__main__':
    main(multi_attr=True)
