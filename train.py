"""Drop-in for the reference's ``train.py`` (single-attribute walk training), same option surface:

python train.py --model stylegan_v2_real --transform face --num_samples 20000 --learning_rate 1e-4 --latent w \
        --walk_type linear --loss l2 --gpu 0 --attrList Smiling --attrPath ./dataset/attributes_celeba.txt \
        --models_dir ./models_celeba --overwrite_config [--resolution 1024 --batch_size 8]

Multi-GPU (one process per MI355X, RCCL): python -m torch.distributed.run --nproc-per-node 8 train.py ...
"""
from latent2im_amd.trainer import main

if __name__ == '__main__':
    main(multi_attr=False)
