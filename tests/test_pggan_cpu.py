"""Host logic of the product-side PGGAN graph (BASELINE config 1) that needs no GPU: plugin discovery, module paths of the pickled walk,
the z walk against the reference fixture, constructor kwargs, and the loud failure without a GPU."""
import io
import types

import numpy as np
import pytest
import torch

from latent2im_amd import hostutil, specs, synth
from latent2im_amd import pggan as pg

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def test_plugin_lookup_and_module_shims():
    from graphs import find_model_using_name
    from graphs.pggan import constants, graph_util, model_256, pggan_256, transform_base, transform_op     # noqa: F401
    assert find_model_using_name('pggan', 'face') is pg.faceGraph
    assert find_model_using_name('pggan', 'scene') is pg.SceneGraph
    assert transform_base.WalkLinearZ_free is pg.WalkLinearZ_free and model_256.Generator is pg.Generator
    assert graph_util.z_sample(3, seed=1).shape == (3, 512)
    with pytest.raises(ImportError):
        from latent2im_amd import graph
        graph.get_transform_graphs('biggan')


def test_walk_linear_z_free_matches_reference_fixture_and_pickles(golden):
    g = golden('pggan')
    np.random.seed(5)                                           # make_golden.py::gen_pggan draws the walk after np.random.seed(5)
    walk = pg.WalkLinearZ_free(512, 6, 1, ['Smiling'])
    np.testing.assert_array_equal(walk.w.detach().numpy(), g['walk_w0'])
    z = T(synth.z_sample(4, seed=0)).float()
    z1 = walk(z, T(g['eps']))
    np.testing.assert_allclose(z1.detach().numpy(), g['z1'], rtol=1e-6, atol=1e-7)
    buf = io.BytesIO()
    torch.save(walk, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    assert type(back).__module__ == 'graphs.pggan.transform_base' and torch.equal(back.w, walk.w)


def test_pggan_kwargs_spec_and_no_cpu_path():
    opt = types.SimpleNamespace(learning_rate=1e-4, walk_type='linear', loss='l2', trainEmbed=False, transform='face', attrPath='dataset/attributes_celeba.txt',
                                attrList='Smiling', layers=None, model='pggan', pggan=types.SimpleNamespace(dset='celebahq'), nn=types.SimpleNamespace(eps=None, num_steps=None))
    kw = hostutil.set_graph_kwargs(opt)
    assert kw['pgan_opts'].dset == 'celebahq' and 'stylegan_opts' not in kw and kw['attrTable']['Smiling'] == 31
    spec = specs.pggan_generator_spec()
    st = synth.pggan_generator_state(seed=1)
    assert list(st) == list(spec) and all(st[k].shape == tuple(v) for k, v in spec.items())
    assert spec['progression.0.conv.0.conv.weight_orig'] == (512, 512, 4, 4) and spec['to_rgb.5.weight'] == (3, 128, 1, 1)
    import json, os
    layout = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pggan_layout.json')))
    assert [[k, list(v)] for k, v in spec.items()] == layout             # == model_256.Generator(511, 1).state_dict() of the reference
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            pg.faceGraph(lr=1e-3, walk_type='linear', loss='l2', trainEmbed=False, attrList=['Smiling'], attrTable={'Smiling': 31}, layers=None)
