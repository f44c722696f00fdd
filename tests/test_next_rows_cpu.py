"""SURVEY 8(f) rows on the CPU: the oracle restatement of apply_alpha / eval buckets / non-linear walks against the
fixture captured from the reference's own methods (tests/golden/next.npz), and the product's host-side pieces (bucket
selection, attribute-preservation metric, walk modules — plain torch modules, they run anywhere) against both."""
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
