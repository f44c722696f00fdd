"""Option surface of train.py / train_multi_attr.py (reference options/train_options.py:15-209): same flags, same
nested namespaces (opt.nn / .color / .biggan / .stylegan / .pggan), same YAML merge rule (command line beats config
file), same output-directory naming.  Works on Python >= 3.10 (the reference relies on the argparse group title
'optional arguments', renamed 'options' in 3.10, and crashes there).  New flags are additive."""
import argparse
import os
import sys
from collections import OrderedDict

import yaml


class TrainOptions:
    def __init__(self):
        self.initialized = False
        self.parser = argparse.ArgumentParser('Training Parser')

    def initialize(self):
        p = self.parser
        p.add_argument('--config_file', type=argparse.FileType(mode='r'), help='configuration yml file')
        p.add_argument('--overwrite_config', action='store_true', help='overwrite config files if they exist')
        p.add_argument('--model', default='biggan', help='pretrained model to use, e.g. stylegan_v2_real')
        p.add_argument('--transform', default='zoom', help='transform operation, e.g. face, scene')
        p.add_argument('--num_samples', type=int, default=20000, help='number of latent z samples')
        p.add_argument('--loss', type=str, default='l2', choices=['l2', 'lpips'], help='loss to use for training')
        p.add_argument('--learning_rate', type=float, default=0.0001, help='learning rate for training')
        p.add_argument('--walk_type', type=str, default='NNz', choices=['NNz', 'linear'], help='type of latent walk')
        p.add_argument('--models_dir', type=str, default='./models', help='output directory for saved checkpoints')
        p.add_argument('--model_save_freq', type=int, default=400, help='dump sample grids after this many batches')
        p.add_argument('--name', type=str, help='experiment name, saved within models_dir')
        p.add_argument('--suffix', type=str, help='suffix for experiment name')
        p.add_argument('--prefix', type=str, help='prefix for experiment name')
        p.add_argument('--gpu', default='', type=str, help='GPUs to use')
        p.add_argument('--trainEmbed', action='store_true')
        p.add_argument('--updateGAN', action='store_true')
        p.add_argument('--attrList', type=str)
        p.add_argument('--attrPath', type=str, default='')
        p.add_argument('--layers', type=str)
        p.add_argument('--no_content_loss', action='store_true')
        p.add_argument('--no_gan_loss', action='store_true')
        # --- additive flags of this build ---
        p.add_argument('--resolution', type=int, default=None, help='generator resolution (reference: fixed 256)')
        p.add_argument('--batch_size', type=int, default=None, help='global batch size (reference: constants.BATCH_SIZE = 4)')
        p.add_argument('--n_epoch', type=int, default=None, help='epochs (reference: 10, multi-attr 3)')
        p.add_argument('--max_iters', type=int, default=None, help='stop each epoch after this many iterations')
        p.add_argument('--seed', type=int, default=None, help='seed numpy global RNG (walk init, alpha draws)')
        p.add_argument('--no_log_sync', action='store_true', help='do not read the loss back every step (train.py:110)')
        p.add_argument('--hip_graph', action='store_true',
                       help='record forward+backward of the step once and replay it from one hipGraph per iteration (static batch size)')
        p.add_argument('--precision', type=str, default=None, choices=['f32', 'bf16x3', 'bf16', 'f16'],
                       help='matrix path of the frozen networks: exact fp32 MFMA (default), the 3-term bf16 split on fp32 tensors (fp32 accuracy), '
                            'or the 16-bit path (16-bit feature maps in HBM, one 16-bit MFMA per MAC, fp32 accumulation; BASELINE config 5): '
                            'f16 = IEEE fp16 elements with static power-of-two gradient scales (gradient within 3 %% of float64), bf16 = bfloat16 elements')
        p.add_argument('--no_gc_freeze', action='store_true', help='do not move the host objects to the garbage collector\'s permanent generation after the first step')
        p.add_argument('--synthetic_weights', action='store_true',
                       help='run on seeded random-init G / regressor / VGG when the checkpoint paths of constants.py do not exist '
                            '(default: a missing checkpoint is an error, as in the reference)')
        g = p.add_argument_group('nn', 'parameters used to specify NN walk')
        g.add_argument('--eps', type=float, help='step size of each NN block')
        g.add_argument('--num_steps', type=int, help='number of NN blocks')
        g = p.add_argument_group('color', 'parameters used for color walk')
        g.add_argument('--channel', type=int)
        g = p.add_argument_group('biggan', 'parameters used for biggan walk')
        g.add_argument('--category', type=int)
        g = p.add_argument_group('stylegan', 'parameters used for stylegan walk')
        g.add_argument('--dataset', default='scene')
        g.add_argument('--latent', default='w', help='which latent space to use; z or w')
        g.add_argument('--truncation_psi', default=1.0)
        g = p.add_argument_group('pggan', 'parameters used for pggan walk')
        g.add_argument('--dset', default='celebahq')
        self.initialized = True
        return p

    @staticmethod
    def _flatten(data):
        out = {}
        for k, v in data.items():
            if isinstance(v, dict):
                out.update(TrainOptions._flatten(v))
            else:
                out[k] = v
        return out

    def parse(self, print_opt=True, argv=None):
        if not self.initialized:
            self.initialize()
        argv = sys.argv[1:] if argv is None else list(argv)
        opt = self.parser.parse_args(argv)
        data = self._flatten(yaml.load(opt.config_file, Loader=yaml.FullLoader)) if opt.config_file else {}
        option_strings = {}
        for grp in self.parser._action_groups:
            for action in grp._group_actions:
                for s in action.option_strings:
                    option_strings[s] = action.dest
        specified = set(option_strings[a.split('=')[0]] for a in argv if a.split('=')[0] in option_strings)
        args = {}
        top_level = {self.parser._positionals.title, self.parser._optionals.title}
        for grp in self.parser._action_groups:
            d = {a.dest: (data[a.dest] if (a.dest in data and a.dest not in specified) else getattr(opt, a.dest, None))
                 for a in grp._group_actions}
            if grp.title in top_level:
                args.update(d)
            else:
                args[grp.title] = argparse.Namespace(**d)
        opt = argparse.Namespace(**args)
        delattr(opt, 'config_file')
        if opt.name:
            output_dir = opt.name
        else:
            output_dir = '_'.join([opt.model, opt.transform, opt.walk_type, 'lr' + str(opt.learning_rate), opt.loss])
            if opt.model == 'biggan':
                if opt.biggan.category:
                    output_dir += '_cat{}'.format(opt.biggan.category)
            elif 'stylegan' in opt.model:
                output_dir += '_{}'.format(opt.stylegan.latent)
            if opt.transform.startswith('color') and opt.color.channel is not None:
                output_dir += '_chn{}'.format(opt.color.channel)
        if opt.suffix:
            output_dir += opt.suffix
        if opt.prefix:
            output_dir = opt.prefix + output_dir
        opt.output_dir = os.path.join(opt.models_dir, output_dir)
        if print_opt:
            self.print_options(opt)
        self.opt = opt
        return opt

    def print_options(self, opt):
        lines, dump, groups = ['----------------- Options ---------------'], OrderedDict(), []
        for k, v in sorted(vars(opt).items()):
            if isinstance(v, argparse.Namespace):
                groups.append((k, v))
                continue
            default = self.parser.get_default(k)
            lines.append('{:>25}: {:<30}{}'.format(str(k), str(v), '' if v == default else '\t[default: %s]' % str(default)))
            dump[k] = v
        for k, v in groups:
            lines.append('{} '.format(k).ljust(20, '-'))
            dump[k] = OrderedDict()
            for k1, v1 in sorted(vars(v).items()):
                default = self.parser.get_default(k1)
                lines.append('{:>25}: {:<30}{}'.format(str(k1), str(v1), '' if v1 == default else '\t[default: %s]' % str(default)))
                dump[k][k1] = v1
        lines.append('----------------- End -------------------')
        message = '\n'.join(lines)
        print(message)
        expr_dir = getattr(opt, 'output_dir', './')
        os.makedirs(expr_dir, exist_ok=True)
        if not opt.overwrite_config:
            assert not os.path.isfile(os.path.join(expr_dir, 'opt.txt')), 'config file exists, use --overwrite_config'
            assert not os.path.isfile(os.path.join(expr_dir, 'opt.yml')), 'config file exists, use --overwrite_config'
        with open(os.path.join(expr_dir, 'opt.txt'), 'wt') as f:
            f.write(message + '\n')
        with open(os.path.join(expr_dir, 'opt.yml'), 'wt') as f:
            dump['overwrite_config'] = False
            yaml.dump(_plain(dump), f, default_flow_style=False)


def _plain(d):
    return {k: (_plain(v) if isinstance(v, dict) else v) for k, v in d.items()}
