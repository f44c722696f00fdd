"""Inference driver: load a trained walk and sweep alpha over ``num_panels`` values (reference vis_w.py:21-118,
options/vis_options.py).  Forward-only reuse of the training kernels."""
import argparse
import os

import yaml

from . import constants, dist, hostutil


class VisOptions:
    def __init__(self):
        self.initialized = False
        self.parser = argparse.ArgumentParser('Visualization Parser')

    def initialize(self):
        p = self.parser
        p.add_argument('config_file', type=argparse.FileType(mode='r'), help='configuration yml file (opt.yml of the training run)')
        p.add_argument('--save_path_w', type=str)
        p.add_argument('--save_path_gan', type=str)
        p.add_argument('--gpu', default='', type=str)
        p.add_argument('--noise_seed', type=int, default=0, help='noise seed for z samples')
        p.add_argument('--output_dir', help='where to save output; overrides output_dir of the config file')
        p.add_argument('--attrList', type=str)
        p.add_argument('--attrPath', type=str, default='')
        self.initialized = True
        return p

    def parse(self, argv=None):
        if not self.initialized:
            self.initialize()
        opt = self.parser.parse_args(argv)
        data = yaml.load(opt.config_file, Loader=yaml.FullLoader)
        for k, v in data.items():
            if isinstance(v, dict):
                data[k] = argparse.Namespace(**v)
        self.opt, self.data = opt, argparse.Namespace(**data)
        return self.opt, self.data


def main(argv=None):
    from . import graph as graph_mod
    v = VisOptions()
    v.initialize()
    v.parser.add_argument('--num_samples', type=int, default=10)
    v.parser.add_argument('--num_panels', type=int, default=7)
    v.parser.add_argument('--max_alpha', type=float, default=1)
    v.parser.add_argument('--min_alpha', type=float, default=0)
    v.parser.add_argument('--layers', type=str, default=None)
    v.parser.add_argument('--trainEmbed', action='store_true')
    v.parser.add_argument('--updateGAN', action='store_true')
    opt, conf = v.parse(argv)
    dist.select_gpu(opt.gpu)                         # before the first torch.cuda call (vis_w.py / eval.py set CUDA_VISIBLE_DEVICES)
    dist.init_from_env()
    if getattr(conf, 'synthetic_weights', False):    # the training run's explicit choice travels in its opt.yml
        constants.ALLOW_SYNTHETIC_WEIGHTS = True
    if getattr(conf, 'resolution', None):
        constants.resolution = conf.resolution
    if getattr(conf, 'batch_size', None):
        constants.BATCH_SIZE = conf.batch_size
    output_dir = opt.output_dir if opt.output_dir else os.path.join(conf.output_dir, 'images')
    os.makedirs(output_dir, exist_ok=True)
    g = graph_mod.find_model_using_name(conf.model, conf.transform)(**hostutil.set_graph_kwargs(conf))
    g.load_multi_models(opt.save_path_w, None, trainEmbed=opt.trainEmbed, updateGAN=opt.updateGAN)
    graph_inputs = hostutil.graph_input(g, opt.num_samples, seed=opt.noise_seed)
    epochs = opt.save_path_w.split('/')[-1].split('_')[2]
    filename = os.path.join(output_dir, 'w_{}_seed{}'.format(epochs, opt.noise_seed))
    name = conf.attrList.strip().split(',')[0]
    layers = None if opt.layers in (None, 'None') else [int(i) for i in opt.layers.split(',')]
    bs = constants.BATCH_SIZE
    written = []
    for batch_start in range(0, opt.num_samples, bs):
        s = slice(batch_start, min(opt.num_samples, batch_start + bs))
        batch = hostutil.batch_input(graph_inputs, s)
        new_filename = filename + '_{}_max{}_min{}'.format(name, opt.max_alpha, opt.min_alpha)
        ag, at = g.vis_image_batch(batch, new_filename, s.start, num_panels=opt.num_panels, max_alpha=opt.max_alpha,
                                   min_alpha=opt.min_alpha, wgt=True)
        written += g.vis_multi_image_batch_alphas(batch, new_filename, alphas_to_graph=ag, alphas_to_target=at, layers=layers,
                                                  batch_start=s.start, name=name, wgt=False, wmask=False, trainEmbed=opt.trainEmbed,
                                                  computeL2=False, given_w=None)
    with open(os.path.join(output_dir, 'index.html'), 'w') as f:      # utils/html.make_html
        f.write('<html><body>' + ''.join('<p>%s<br><img src="%s"></p>' % (os.path.basename(w), os.path.basename(w)) for w in written)
                + '</body></html>')
    return written
