"""Module-level constants of the StyleGAN2 graph (reference graphs/stylegan_v2_real/constants.py:1-18), read by the
drivers as ``constants.BATCH_SIZE`` (train.py:36).  The reference hard-codes 256^2 / batch 4; here ``--resolution`` and
``--batch_size`` overwrite them before the graph is built (BASELINE configs 3-5 run 1024^2)."""
BATCH_SIZE = 4
DIM_Z = 512
resolution = 256
useGPU = True
NUM_CHANNELS = 3

# checkpoint paths (reference placeholders '/path/...').  A configured path that does not exist is an ERROR (the reference
# crashes in torch.load, transform_base.py:524-547) unless synthetic weights are requested explicitly: --synthetic_weights
# on the drivers, L2I_SYNTHETIC_WEIGHTS=1 in the environment, or ALLOW_SYNTHETIC_WEIGHTS = True (bench.py, tests, smoke()).
import os as _os
ALLOW_SYNTHETIC_WEIGHTS = _os.environ.get('L2I_SYNTHETIC_WEIGHTS', '') == '1'
reg_json = None
reg_path = '/path/003_dict.model'
g_path = '/path/550000.pt'
vgg_path = ''
# BASELINE config 1 (graphs/pggan): the in-repo PGGAN-256 generator checkpoint ({'G': state_dict}, transform_base.py:578-590) and its resolution
pg_path = '/path/280000_dict.model'
PG_RESOLUTION = 256

SYNTH_SEED_G, SYNTH_SEED_D, SYNTH_SEED_R, SYNTH_SEED_V = 100, 200, 300, 400
# per-layer NoiseInjection weights of the synthetic generator: 0 like a fresh reference Generator (networks.py:279; parity tests), > 0
# like a trained one (bench.py: the per-layer N(0,1) draw of networks.py:281-286 is then inside the timed region)
SYNTH_NOISE_STRENGTH = 0.0

# run the discriminator / VGG / regressor loss branches on separate HIP streams (see graph.TransformGraph.get_w_loss)
CONCURRENT_LOSS_BRANCHES = True
# [r6] start the VGG prefix of the ORIGINAL image on the content branch's stream as soon as that image exists (graph.prefetch_content_taps); L2I_PREFETCH_TAPS=1 / 0: force on / off (A/B)
PREFETCH_CONTENT_TAPS = {'1': True, '0': False}.get(_os.environ.get('L2I_PREFETCH_TAPS', ''))      # None: on for the 16-bit path only (measured: graph.prefetch_content_taps)

# transform_base.py:290 hard-codes ``is_mlp = False`` ("TODO: Hard code"); True builds WalkMlpMultiW instead of WalkLinearMultiW
WALK_IS_MLP = False

# fp16 elements (--precision f16): clean steps before the dynamic loss-scale factor doubles again (torch.cuda.amp.GradScaler's growth_interval)
LOSS_SCALE_GROWTH_INTERVAL = 2000

# [r6] regressor head + BCE (+ their backward) as one launch each way (csrc/l2i_loss.hip); 0: the torch ops (A/B)
FUSED_REG_LOSS = _os.environ.get('L2I_FUSED_REG_LOSS', '1') != '0'
