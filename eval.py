"""Drop-in for the attribute-preservation half of the reference's ``eval.py``:

python eval.py models_celeba/stylegan_v2_real_face_linear_lr0.0001_l2_w/opt.yml --gpu 0 --noise_seed 0 --num_samples 10 \
    --num_panels 10 --attrPath ./dataset/attributes_celeba.txt --target_attrList Smiling \
    --save_path_w ./models_celeba/.../model_w_10_final_walk_module.ckpt
"""
from latent2im_amd.evaluate import main

if __name__ == '__main__':
    main()
