"""SURVEY 8(f) rows on the CPU: the oracle restatement of apply_alpha / eval buckets / non-linear walks against the
fixture captured from the reference's own methods (tests/golden/next.npz), and the product's host-side pieces (bucket
selection, attribute-preservation metric, walk modules — plain torch modules, they run anywhere) against both."""
import os

import numpy as np
import pytest
import torch

from latent2im_amd import evaluate, graph, synth
from oracle import evalpath, step as ostep
from tests import emu

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def _nets(g):
    R = synth.resnet50_state(seed=300)
    R['fc.weight'] = R['fc.weight'] * float(g['fc_scale'])
    return ostep.to_torch(synth.generator_state(64, seed=100)), ostep.to_torch(R)


def test_oracle_apply_alpha_and_buckets_match_reference(golden):
    g = golden('next')
    PG, PR = _nets(g)
    walk = T(synth.walk_init(2, 10, seed=9)) * float(g['walk_scale'])
    zs = T(synth.z_sample(4, seed=11)).float()
    idx = int(g['index_'])
    assert idx == 31
    x1, a0, x0 = evalpath.apply_alpha(PG, PR, walk, zs, g['alphas'][3], [31, 39], index_=idx)
    np.testing.assert_allclose(a0.numpy(), g['apply.a0'], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(x0.numpy().sum(3), g['apply.x0_rowsum'], rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(x1.numpy().sum(3), g['apply.x1_rowsum'], rtol=1e-3, atol=2e-3)
    ma, ao, im, og = evalpath.compute_multi_attr(PG, PR, walk, zs, list(g['alphas']), [31, 39], idx)
    assert [len(m) for m in ma] == [6, 5, 6]                       # 3 of the 20 (alpha, sample) pairs moved by more than 1
    for k in range(3):
        np.testing.assert_allclose(np.asarray(ma[k]), g['bucket%d.multi_attr' % k], rtol=1e-3, atol=2e-3)
        np.testing.assert_allclose(np.asarray(ao[k]), g['bucket%d.attri_org' % k], rtol=1e-3, atol=2e-3)
        # uint8 images: the sum over 3*64*64 truncated pixels may move by a few counts between BLAS builds
        s = np.asarray([int(i.astype(np.int64).sum()) for i in im[k]])
        assert np.all(np.abs(s - g['bucket%d.img_sum' % k]) <= 64)
    # metric of eval.py on the reference's own buckets
    ref_m = [[r for r in g['bucket%d.multi_attr' % k]] for k in range(3)]
    ref_o = [[r for r in g['bucket%d.attri_org' % k]] for k in range(3)]
    want = evalpath.attribute_preservation(ref_m, ref_o, idx)
    _, got = evaluate.attribute_preservation(ref_m, ref_o, idx)
    np.testing.assert_allclose(got, want, rtol=1e-6)
    assert len(got) == 3 and all(v > 0 for v in got)


def test_bucket_selection_is_exact(golden):
    g = golden('next')
    for k in range(3):
        p, o = g['bucket%d.multi_attr' % k][:, 31], g['bucket%d.attri_org' % k][:, 31]
        assert np.array_equal(graph.attribute_change_bucket(p, o), np.full(len(p), k))
        assert np.array_equal(evalpath.change_bucket(p, o), np.full(len(p), k))
    # edges: thresholds are inclusive, anything above 1 is dropped, empty input
    org = np.zeros(7, dtype=np.float32)
    pred = np.asarray([0.0, 0.3, 0.30001, 0.6, 0.60001, 1.0, 1.00001], dtype=np.float32)
    assert list(graph.attribute_change_bucket(pred, org)) == [0, 0, 1, 1, 2, 2, 3]
    assert list(graph.attribute_change_bucket(-pred, org)) == [0, 0, 1, 1, 2, 2, 3]
    assert graph.attribute_change_bucket(np.zeros(0), np.zeros(0)).shape == (0,)
    rs = np.random.RandomState(0)
    p, o = rs.uniform(-1, 2, 1000).astype(np.float32), rs.uniform(0, 1, 1000).astype(np.float32)
    assert np.array_equal(graph.attribute_change_bucket(p, o), evalpath.change_bucket(p, o))
    # empty buckets are skipped by the metric (eval.py:222-223)
    res, avg = evaluate.attribute_preservation([[], [np.ones(40)], []], [[], [np.zeros(40)], []], 31)
    assert avg == [1.0] and res == [39.0]


def test_nonlinear_walks_match_reference(golden):
    g = golden('next')
    ws = [T(synth.z_sample(4, seed=21 + i)).float() for i in range(3)]
    al = T(np.asarray([[0.3], [-0.7], [1.2], [0.0]], dtype=np.float32))
    mlp = graph.WalkMlpMultiW(512, 6, 1, ['Smiling'])
    P = {k: T(v) for k, v in emu.seeded_state(mlp, 31).items()}
    want = g['mlp.out']
    np.testing.assert_allclose(torch.stack(mlp(ws, al)).detach().numpy(), want, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(torch.stack(evalpath.walk_mlp_multi_w(ws, al, P)).numpy(), want, rtol=1e-4, atol=1e-5)
    same = [ws[0]] * 4                                                 # the get_w case: one tensor repeated
    np.testing.assert_allclose(torch.stack(mlp(same, al)).detach().numpy(),
                               torch.stack(evalpath.walk_mlp_multi_w(same, al, P)).numpy(), rtol=1e-5, atol=1e-6)
    assert str(g['mlp.layers_error']) == 'TypeError'
    with pytest.raises(TypeError):
        mlp(ws, al, layers=[0])
    nl = graph.WalkNonLinearW(512, 6, 1, ['Smiling'])
    P = {k: T(v) for k, v in emu.seeded_state(nl, 32).items()}
    np.testing.assert_allclose(torch.stack(nl(ws, None, al, None)).detach().numpy(), g['nonlinear.out'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(torch.stack(evalpath.walk_nonlinear_w(ws, al, P)).numpy(), g['nonlinear.out'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(torch.stack(nl(ws, None, al, None, layers=[1])).detach().numpy(), g['nonlinear.out_layers1'],
                               rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(torch.stack(evalpath.walk_nonlinear_w(ws, al, P, layers=[1])).numpy(), g['nonlinear.out_layers1'],
                               rtol=1e-4, atol=1e-5)
    # the reference's graph calls every walk as walk(ws, alpha=, layers=): TypeError with this module (kept)
    assert str(g['nonlinear.graph_call_error']) == 'TypeError'
    with pytest.raises(TypeError):
        nl(ws, alpha=al, layers=None)
    # pickles resolve under the reference's module path
    assert graph.WalkNonLinearW.__module__ == graph.WalkMlpMultiW.__module__ == 'graphs.stylegan_v2_real.transform_base'


def test_walk_modules_pickle_round_trip(tmp_path):
    """save_multi_models pickles the whole walk module (transform_base.py:492-499): every walk class must resolve under the
    reference's module path, for torch.save and for torch.load in a process that only knows ``graphs.*``."""
    import subprocess
    import sys
    import os
    for cls in (graph.WalkLinearMultiW, graph.WalkMlpMultiW, graph.WalkNonLinearW):
        m = cls(512, 6, 1, ['Smiling'])
        path = str(tmp_path / (cls.__name__ + '.ckpt'))
        torch.save(m, path)
        back = torch.load(path, map_location='cpu', weights_only=False)
        assert type(back) is cls and type(back).__module__ == 'graphs.stylegan_v2_real.transform_base'
        for (k, a), (_, b) in zip(m.state_dict().items(), back.state_dict().items()):
            assert torch.equal(a, b), k
    # a fresh interpreter that never imported latent2im_amd.graph by name (the reference's vis_w.py does exactly this)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import torch, sys; sys.path.insert(0, %r); "
            "[print(type(torch.load(p, map_location='cpu', weights_only=False)).__name__) for p in sys.argv[1:]]" % root)
    out = subprocess.run([sys.executable, '-c', code] + [str(tmp_path / (n + '.ckpt')) for n in ('WalkLinearMultiW', 'WalkMlpMultiW', 'WalkNonLinearW')],
                         capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ['WalkLinearMultiW', 'WalkMlpMultiW', 'WalkNonLinearW']


def test_missing_checkpoint_is_an_error_unless_synthetic_weights_are_requested(tmp_path):
    from latent2im_amd import constants
    saved = constants.ALLOW_SYNTHETIC_WEIGHTS
    try:
        constants.ALLOW_SYNTHETIC_WEIGHTS = False
        with pytest.raises(FileNotFoundError, match='synthetic_weights'):
            graph._checkpoint_or_synthetic('generator', '/path/550000.pt')          # the reference's placeholder default
        with pytest.raises(FileNotFoundError):
            graph._checkpoint_or_synthetic('VGG-19', '')
        real = tmp_path / 'g.pt'
        real.write_bytes(b'x')
        assert graph._checkpoint_or_synthetic('generator', str(real)) is True
        constants.ALLOW_SYNTHETIC_WEIGHTS = True
        assert graph._checkpoint_or_synthetic('generator', '/path/550000.pt') is False
        assert graph._checkpoint_or_synthetic('generator', str(real)) is True
    finally:
        constants.ALLOW_SYNTHETIC_WEIGHTS = saved


def test_gpu_flag_is_applied_before_the_runtime_starts(monkeypatch):
    """--gpu (train.py:150): exported before the first torch.cuda call; ignored under torchrun and when the launcher pinned devices."""
    from latent2im_amd import dist
    for k in ('CUDA_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'WORLD_SIZE'):
        monkeypatch.delenv(k, raising=False)
    assert not torch.cuda.is_initialized()
    dist.select_gpu('')
    assert 'CUDA_VISIBLE_DEVICES' not in os.environ
    dist.select_gpu('3')
    assert os.environ['CUDA_VISIBLE_DEVICES'] == '3' and os.environ['HIP_VISIBLE_DEVICES'] == '3'
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '5')
    dist.select_gpu('1')
    assert os.environ['HIP_VISIBLE_DEVICES'] == '5'
    monkeypatch.delenv('HIP_VISIBLE_DEVICES')
    monkeypatch.delenv('CUDA_VISIBLE_DEVICES')
    monkeypatch.setenv('WORLD_SIZE', '2')
    dist.select_gpu('1')
    assert 'CUDA_VISIBLE_DEVICES' not in os.environ
    # trainer.main parses --gpu and calls select_gpu ahead of init_from_env
    import inspect
    from latent2im_amd import trainer, vis, evaluate
    for mod in (trainer, vis, evaluate):
        src = inspect.getsource(mod.main)
        assert 0 < src.index('dist.select_gpu(') < src.index('dist.init_from_env()'), mod.__name__


def test_regressor_training_label_file_and_dataset(tmp_path):
    """scene_regressor_256.py:27-74 on the host: annotations.tsv (name <TAB> 40 x 'value,confidence') through load_labelfile, images
    listed in a split file through CustomDataset — Resize(256) like torchvision (short side 256, long side int(256 * long / short),
    truncated), CenterCrop(256), ToTensor, Normalize(0.5, 0.5)."""
    from PIL import Image
    from latent2im_amd import regressor_train as RT
    rs = np.random.RandomState(0)
    names = ['00000001/1.jpg', '00000001/2.jpg', '00000002/7.jpg']
    vals = rs.rand(3, 40)
    with open(tmp_path / 'annotations.tsv', 'w') as f:
        for n, v in zip(names, vals):
            f.write(n + '\t' + '\t'.join('%.4f,%.2f' % (x, 0.9) for x in v) + '\n')
    labels = RT.load_labelfile(str(tmp_path / 'annotations.tsv'))
    assert set(labels) == set(names)
    np.testing.assert_allclose(labels[names[2]], np.round(vals[2], 4), atol=1e-12)
    sizes = {names[0]: (300, 451), names[1]: (517, 260), names[2]: (256, 256)}          # (w, h)
    for n, (w, h) in sizes.items():
        os.makedirs(tmp_path / 'imgs' / n.split('/')[0], exist_ok=True)
        Image.fromarray(rs.randint(0, 255, (h, w, 3), dtype=np.uint8)).save(tmp_path / 'imgs' / n, quality=95)
    (tmp_path / 'training.txt').write_text(names[0] + '\n' + names[2] + '\n')
    ds = RT.CustomDataset(str(tmp_path / 'imgs'), labels, str(tmp_path / 'training.txt'))
    assert len(ds) == 2
    x, y = ds[0]
    assert tuple(x.shape) == (3, 256, 256) and x.dtype == torch.float32 and float(x.min()) >= -1 and float(x.max()) <= 1
    np.testing.assert_allclose(y.numpy(), labels[names[0]].astype(np.float32))
    # the resize rule (long side truncated): 300 x 451 -> 256 x int(256 * 451 / 300) = 384 (round() would give 385)
    seen = []
    orig = Image.Image.resize
    try:
        Image.Image.resize = lambda self, size, *a, **k: (seen.append(tuple(size)), orig(self, size, *a, **k))[1]
        ds[0]
        ds2 = RT.CustomDataset(str(tmp_path / 'imgs'), labels, str(tmp_path / 'all.txt')) if (tmp_path / 'all.txt').write_text('\n'.join(names)) else None
        ds2[1]
    finally:
        Image.Image.resize = orig
    assert seen == [(256, int(256 * 451 / 300)), (int(256 * 517 / 260), 256)] and seen[0][1] == 384
    # TrainableResNet50.load_state_dict keeps num_batches_tracked (checked on the GPU in test_regressor_training_step_vs_oracle)

