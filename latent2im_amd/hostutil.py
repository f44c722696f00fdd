"""Host-side helpers of the training drivers: batching, attribute tables, latent sampling.
Reference: utils/util.py:5-121, graphs/stylegan_v2_real/graph_util.py:5-19."""
from collections import OrderedDict

import numpy as np

from . import constants

SCENE_DEFAULT_TABLE = (('daylight', 1), ('night', 2), ('sunrisesunset', 3), ('sunny', 5), ('clouds', 6), ('fog', 7),
                       ('snow', 9), ('warm', 10), ('cold', 11), ('beautiful', 13), ('flowers', 14), ('spring', 15),
                       ('summer', 16), ('autumn', 17), ('winter', 18), ('colorful', 20), ('dark', 24), ('bright', 25),
                       ('rain', 29), ('boring', 37), ('lush', 39))


def batch_input(graph_inputs, s):
    """Slice every ndarray value of ``graph_inputs`` with ``s`` (utils/util.py:5-16)."""
    return {k: (v[s] if isinstance(v, np.ndarray) else v) for k, v in graph_inputs.items()}


def _read_attr_file(path):
    names, table = [], OrderedDict()
    with open(path, 'r') as f:
        for i, line in enumerate(f.readlines()):
            if line.strip():
                names.append(line.strip())
                table[line.strip()] = i                      # 0-based LINE index (blank lines still count)
    assert len(names) == 40, ' len(attrList) should be 40'
    return names, table


def set_graph_kwargs(opt):
    """Constructor kwargs for the graph (utils/util.py:19-121): attrList / attrTable from ``--attrPath``
    (name -> line index; e.g. Smiling -> 31 for CelebA), walk / loss options, ``--layers``."""
    kw = dict(lr=opt.learning_rate, walk_type=opt.walk_type, loss=opt.loss)
    kw['trainEmbed'] = opt.trainEmbed
    if opt.transform == 'face':
        if opt.attrPath:
            names, table = _read_attr_file(opt.attrPath)
        else:
            table = OrderedDict(SCENE_DEFAULT_TABLE)
            names = list(table.keys())
        kw['attrList'] = names if not opt.attrList else opt.attrList.split(',')
        kw['attrTable'] = table
    elif opt.transform == 'scene':
        names, table = _read_attr_file(opt.attrPath)
        kw['attrList'] = names if not opt.attrList else opt.attrList.split(',')
        kw['attrTable'] = table
    else:
        raise NotImplementedError('transform %r: only the face / scene graphs are on the walk-training path' % opt.transform)
    try:
        kw['layers'] = [int(x) for x in opt.layers.split(',')]      # reference keeps strings (a latent bug)
    except AttributeError:
        kw['layers'] = None
    if opt.walk_type.startswith('NN'):
        if opt.nn.eps:
            kw['eps'] = opt.nn.eps
        if opt.nn.num_steps:
            kw['N_f'] = opt.nn.num_steps
    if 'stylegan' in opt.model:
        kw['stylegan_opts'] = opt.stylegan
    if opt.model == 'pggan':
        kw['pgan_opts'] = opt.pggan
    return kw


def z_sample(batch_size, seed=0, dim_z=constants.DIM_Z):
    """graph_util.z_sample: RandomState(seed).randn(batch_size, dim_z) (float64)."""
    return np.random.RandomState(seed).randn(batch_size, dim_z)


def graph_input(graph, num_samples, seed=0, **kwargs):
    return {'z': z_sample(num_samples, seed, graph.dim_z)}
