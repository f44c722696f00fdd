"""Host-side logic of the drop-in surface against the fixtures captured from the reference (CPU, no GPU)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'host.json')))


def _opt(argv):
    from options.train_options import TrainOptions
    return TrainOptions().parse(print_opt=False, argv=argv)


README_ARGV = ['--model', 'stylegan_v2_real', '--transform', 'face', '--num_samples', '20000', '--learning_rate', '1e-4',
               '--latent', 'w', '--walk_type', 'linear', '--loss', 'l2', '--gpu', '3', '--attrList', 'Smiling',
               '--attrPath', './dataset/attributes_celeba.txt', '--models_dir', './models_celeba', '--overwrite_config']


def test_train_options_match_reference_namespace():
    """README command line -> same nested namespace and output_dir as the reference parser (options/train_options.py)."""
    opt = _opt(README_ARGV)
    ref = HOST['readme_opt']

    def ns(o):
        return {k: (ns(v) if hasattr(v, '__dict__') else v) for k, v in vars(o).items()}
    mine = ns(opt)
    for k, v in ref.items():
        if k == 'help':
            continue
        assert str(mine[k]) == str(v) if not isinstance(v, dict) else mine[k] == v, (k, mine[k], v)
    assert opt.output_dir == './models_celeba/stylegan_v2_real_face_linear_lr0.0001_l2_w'
    extra = set(mine) - set(ref)
    assert extra <= {'resolution', 'batch_size', 'n_epoch', 'max_iters', 'seed', 'no_log_sync', 'synthetic_weights', 'hip_graph', 'precision', 'no_gc_freeze'}      # additive flags only


def test_yaml_config_and_cli_precedence(tmp_path):
    cfg = tmp_path / 'c.yml'
    cfg.write_text('model: stylegan_v2_real\ntransform: scene\nlearning_rate: 0.01\nstylegan:\n  latent: z\n')
    opt = _opt(['--config_file', str(cfg), '--learning_rate', '0.5'])
    assert opt.transform == 'scene' and opt.learning_rate == 0.5 and opt.stylegan.latent == 'z'


def test_refuses_to_overwrite_config(tmp_path):
    from options.train_options import TrainOptions
    argv = ['--model', 'stylegan_v2_real', '--transform', 'face', '--models_dir', str(tmp_path)]
    TrainOptions().parse(print_opt=True, argv=argv)
    assert os.path.isfile(os.path.join(str(tmp_path), 'stylegan_v2_real_face_NNz_lr0.0001_l2_w', 'opt.yml'))
    with pytest.raises(AssertionError):
        TrainOptions().parse(print_opt=True, argv=argv)
    TrainOptions().parse(print_opt=True, argv=argv + ['--overwrite_config'])


def test_set_graph_kwargs_attr_tables():
    from utils import util
    opt = _opt(README_ARGV[:-7] + ['--attrList', 'Smiling,Young', '--attrPath', os.path.join(ROOT, 'dataset', 'attributes_celeba.txt')])
    kw = util.set_graph_kwargs(opt)
    assert kw['attrList'] == HOST['face.attrList'] == ['Smiling', 'Young']
    assert [[k, v] for k, v in kw['attrTable'].items()] == HOST['face.attrTable']
    assert kw['attrTable']['Smiling'] == 31
    assert sorted(kw) == HOST['face.keys']
    opt = _opt(['--model', 'stylegan_v2_real', '--transform', 'scene', '--walk_type', 'linear',
                '--attrPath', os.path.join(ROOT, 'dataset', 'attributes_scene.txt')])
    kw = util.set_graph_kwargs(opt)
    assert kw['attrList'] == HOST['scene.attrList'] and [[k, v] for k, v in kw['attrTable'].items()] == HOST['scene.attrTable']
    opt = _opt(README_ARGV + ['--layers', '0,1,5'])
    assert util.set_graph_kwargs(opt)['layers'] == [0, 1, 5]


def test_z_sample_and_alpha_samplers(golden):
    from graphs.stylegan_v2_real import graph_util
    from graphs.stylegan_v2_real.transform_op import FaceTransform, SceneTransform
    g = golden('host')
    np.testing.assert_array_equal(graph_util.z_sample(3, seed=0)[:, :8], g['z_seed0'])
    np.testing.assert_array_equal(graph_util.z_sample(2, seed=3)[:, :8], g['z_seed3'])
    ft = FaceTransform()
    ft.Nsliders = 1
    np.random.seed(1234)
    s, a, idx = ft.get_train_alpha(np.zeros((4, 512)), N_attr=3)
    np.testing.assert_array_equal(s, g['face_alpha_slider'])
    np.testing.assert_array_equal(a, g['face_alpha_val'])
    assert idx is None
    st = SceneTransform()
    st.Nsliders = 1
    np.random.seed(1234)
    s, a, _ = st.get_train_alpha(np.zeros((4, 512)), N_attr=3)
    np.testing.assert_array_equal(s, g['scene_alpha_slider'])
    np.testing.assert_array_equal(ft.scale_test_alpha_for_graph(0.25, np.zeros((4, 512))), g['face_test_slider'])


def test_plugin_lookup_and_pickle_path():
    import graphs
    import pickle
    import torch
    cls = graphs.find_model_using_name('stylegan_v2_real', 'face')
    assert cls.__name__ == 'faceGraph'
    assert graphs.find_model_using_name('stylegan_v2_real', 'scene').__name__ == 'SceneGraph'
    with pytest.raises(SystemExit):
        graphs.find_model_using_name('stylegan_v2_real', 'zoom')
    from graphs.stylegan_v2_real.transform_base import WalkLinearMultiW
    np.random.seed(0)
    w = WalkLinearMultiW(512, 6, 1, ['Smiling'])
    assert tuple(w.w.shape) == (1, 14, 512)
    blob = pickle.dumps(w)
    assert b'graphs.stylegan_v2_real.transform_base' in blob       # what the reference's vis_w.py unpickles
    w2 = pickle.loads(blob)
    assert torch.equal(w2.w, w.w)
    # forward = w_i + alpha @ W[:, i, :] for the selected layers only
    ws = [torch.zeros(2, 512)] * 14
    out = w(ws, torch.tensor([[0.5], [2.0]]), layers=[0, 3])
    assert torch.allclose(out[3], torch.tensor([[0.5], [2.0]]) @ w.w[:, 3, :]) and torch.equal(out[1], ws[1])


def test_constants_module_is_shared():
    import graphs.stylegan_v2_real.constants as c
    import latent2im_amd.constants as c2
    assert c is c2 and c.BATCH_SIZE == 4 and c.DIM_Z == 512


def test_c_abi_library_exports_every_declared_symbol():
    """include/l2i.h <-> libl2i_hip.so <-> ctypes table agree; loading needs no GPU."""
    from latent2im_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'l2i.h')).read()
    declared = set(re.findall(r'\b(l2i_[a-z0-9_]+)\s*\(', hdr)) - {'l2i_conv_params'}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert os.path.isfile(_lib.LIB_PATH), 'run `python __graft_entry__.py build` first'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.l2i_abi_version.restype = ctypes.c_int
    assert lib.l2i_abi_version() == _lib.ABI_VERSION == int(re.search(r'#define L2I_ABI_VERSION (\d+)', hdr).group(1))
    lib.l2i_sizeof_conv_params.restype = ctypes.c_int
    assert lib.l2i_sizeof_conv_params() == ctypes.sizeof(_lib.ConvParams)
    # the struct mirror has the same size as the C struct (pointer/int/float layout rules are identical)
    assert ctypes.sizeof(_lib.ConvParams) % 8 == 0


def test_product_never_imports_the_oracle_or_the_reference():
    """The oracle is a checker: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it."""
    offenders = []
    for base, _, files in os.walk(os.path.join(ROOT, 'latent2im_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(base, f)).read()
                rel = os.path.relpath(os.path.join(base, f), ROOT)
                if re.search(r'^\s*(from|import)\s+oracle\b', src, re.M) and rel != os.path.join('latent2im_amd', 'selfcheck.py'):
                    offenders.append(rel)
                if '/root/reference' in src:
                    offenders.append(rel + ' (reads /root/reference)')
    for top in ('graphs', 'options', 'utils', 'train.py', 'train_multi_attr.py'):
        path = os.path.join(ROOT, top)
        files = [path] if os.path.isfile(path) else [os.path.join(b, f) for b, _, fs in os.walk(path) for f in fs if f.endswith('.py')]
        for f in files:
            if re.search(r'^\s*(from|import)\s+oracle\b', open(f).read(), re.M):
                offenders.append(os.path.relpath(f, ROOT))
    assert not offenders, offenders
    # selfcheck imports it inside smoke() only
    src = open(os.path.join(ROOT, 'latent2im_amd', 'selfcheck.py')).read()
    assert re.findall(r'^(from|import)\s+oracle', src, re.M) == []


def test_ops_fail_loudly_without_gpu_or_library(monkeypatch):
    import torch
    from latent2im_amd import _lib, kernels
    if not torch.cuda.is_available():
        with pytest.raises(_lib.L2IError):
            kernels.fused_bias_act(torch.zeros(4, 4), torch.zeros(4), None, 3, 0, 0.2, 1.0)     # CPU tensor: no fallback
        from latent2im_amd import graph
        import types
        with pytest.raises(RuntimeError):
            graph.faceGraph(lr=1e-4, walk_type='linear', loss='l2', trainEmbed=False, attrList=['Smiling'],
                            attrTable={'Smiling': 31}, layers=None, stylegan_opts=types.SimpleNamespace(latent='w'))
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libl2i_hip.so')
    with pytest.raises(_lib.L2IError):
        _lib.load()


def test_image_conv_weight_pack_follows_the_kernels_contraction_index():
    """[r5] conv.pack_weight_img_h8 (include/l2i.h: l2i_conv_img_h8): element (s, half, co, e) of the 16-bit planes is w[co, c, ky, kx = e] with
    (c, ky) = divmod(2 s + half, K), zero for e >= K and for rows past Cin K — for the three image-side convs of the 16-bit path (1x1, 3x3, 7x7)."""
    import torch
    from latent2im_amd import conv
    for cout, cin, k in ((32, 3, 1), (64, 3, 3), (64, 3, 7), (40, 1, 3), (96, 4, 3)):
        w = torch.randn(cout, cin, k, k)
        for dt in (torch.float16, torch.bfloat16):
            planes = conv.pack_weight_img_h8(w, dtype=dt)
            ns, coutp = (cin * k + 1) // 2, (cout + 31) // 32 * 32
            assert tuple(planes.shape) == (ns, 1, 2, coutp, 8) and planes.dtype == torch.int16
            got = planes.view(dt).float()
            want = torch.zeros(ns, 1, 2, coutp, 8)
            for p in range(cin * k):
                c, ky = divmod(p, k)
                want[p // 2, 0, p % 2, :cout, :k] = w[:, c, ky, :].to(dt).float()
            assert torch.equal(got, want)


def test_modulate_plan_table_rows():
    """[r5] kernels16.ModulatePlan: the device-side table of l2i_modulate_planes_multi_h8 (eight int64 per layer: weight / scale / output offsets, slots
    per sample, taps, CoutP, Cs, first block) is consistent with the per-layer views it hands out."""
    import torch
    from latent2im_amd import conv, kernels16 as K16
    shapes = [(64, 32, 3), (32, 96, 3), (128, 64, 1)]                              # cin, cout, k
    w32s = [conv.pack_weight_h8_f32(torch.randn(co, ci, k, k)) for ci, co, k in shapes]
    offs = [0, 64, 96]
    plan = K16.ModulatePlan(w32s, offs, 'cpu')
    B = 3
    table, slots, nblocks = plan._table(B)
    rows = table.tolist()
    assert len(rows) == 3 and rows[0][7] == 0 and all(rows[i][7] < rows[i + 1][7] for i in range(2)) and nblocks > rows[2][7]
    out_off = 0
    for (ci, co, k), w32, off, r in zip(shapes, w32s, offs, rows):
        sps = w32.numel() // 8
        assert r[:7] == [plan.w_off[rows.index(r)], off * B, out_off, sps, k * k, (co + 31) // 32 * 32, (ci + 31) // 32 * 32]
        out_off += sps * B
    assert slots == out_off
