"""reference scene_regressor_256.py — training the 40-attribute ResNet-50 regressor (MI355X-native: latent2im_amd/regressor_train.py)."""
from latent2im_amd.regressor_train import (CustomDataset, TrainableResNet50, load_ckpt, load_labelfile, main, make_optimizer,  # noqa: F401
                                           save_ckpt, train_step)

if __name__ == '__main__':
    main()
