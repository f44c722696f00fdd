import sys; sys.path.insert(0, '.')
from latent2im_amd import constants
constants.ALLOW_SYNTHETIC_WEIGHTS = True
from graphs import find_model_using_name
from latent2im_amd import pggan, synth
g = find_model_using_name('pggan', 'face')(lr=1e-4, walk_type='linear', loss='l2', trainEmbed=False, attrList=['Smiling'],
                                           attrTable={'Smiling': 31}, layers=None, pgan_opts=None)
loss, x0, x1, a0, target = pggan.walk_training_step(g, synth.z_sample(4, seed=0), [[0.3]] * 4, no_gan_loss=True)
print(float(loss.detach()), tuple(x0.shape), g.weight_sources)
