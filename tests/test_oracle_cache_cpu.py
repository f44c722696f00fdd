"""tests/golden/oracle_1024.npz (the CPU oracle's outputs at the BASELINE shapes, written by tests/golden/make_oracle_cache.py) is data the GPU tests
trust instead of a 128 s oracle evaluation each.  Kept honest here: the 256^2 entry is re-derived from a LIVE oracle evaluation through the same
`oracle_cache.evaluate` / `summarize` the script uses and must match the file; the 1024^2 entries must be present, complete and self-consistent
(the total is the weighted sum of the stored terms, the clamp pair satisfies target = clamp(alpha_org + delta), the probe pixels lie inside the
stored row sums' range)."""
import os

import numpy as np
import torch

from tests import oracle_cache
from tests.conftest import GOLDEN


def test_cached_256_entry_equals_a_live_oracle_evaluation():
    c = oracle_cache.load(GOLDEN, 'c256')
    o, po = oracle_cache.evaluate('c256')
    live = oracle_cache.summarize(o, po)
    assert set(live) == set(c.d)
    for k, v in live.items():
        # the same float32 oracle on another host / thread count: summation order inside torch's CPU convolutions may differ in the last bits
        scale = float(np.abs(c.d[k]).max()) + 1e-30
        np.testing.assert_allclose(v, c.d[k], rtol=2e-4, atol=2e-5 * scale, err_msg=k)


def test_cached_entries_carry_the_fingerprint_of_the_current_oracle():
    """[r6] Every committed entry was written by THIS oracle on THIS case spec: `<name>.meta_sha` = sha256 over the oracle sources `evaluate` runs, the
    synthetic-weight generator, the case's spec and alpha values (oracle_cache.case_fingerprint).  A change to any of them fails here until
    `python tests/golden/make_oracle_cache.py <case>` has re-evaluated the entry — the GPU tests never compare against stale constants silently."""
    z = np.load(os.path.join(GOLDEN, 'oracle_1024.npz'), allow_pickle=False)
    for name in oracle_cache.CASES:
        c = oracle_cache.Cached(z, name)
        assert c.sha == oracle_cache.case_fingerprint(name), 'cached oracle case %r is stale: re-run tests/golden/make_oracle_cache.py %s' % (name, name)


def test_cached_float64_scene_entry_equals_a_live_oracle_evaluation():
    """[r6] The second honesty entry: `s64` — SceneGraph attribute indices, clamp flow, FLOAT64 evaluation plus the float32 gradient beside it — the code
    paths the float32 face-attribute entry above does not walk, re-derived live (64^2, batch 4: seconds)."""
    c = oracle_cache.load(GOLDEN, 's64')
    live = oracle_cache.evaluate_and_summarize('s64')
    assert set(live) == set(c.d)
    for k, v in live.items():
        scale = float(np.abs(c.d[k]).max()) + 1e-30
        tol = (2e-4, 2e-5) if k in ('grad32', 'x0.probes', 'x1.probes') else (1e-7, 1e-9)        # float32 quantities / float64 quantities
        np.testing.assert_allclose(v, c.d[k], rtol=tol[0], atol=tol[1] * scale, err_msg=k)


def test_cached_1024_entries_are_complete_and_self_consistent():
    z = np.load(os.path.join(GOLDEN, 'oracle_1024.npz'), allow_pickle=False)
    for name, case in oracle_cache.CASES.items():
        c = oracle_cache.Cached(z, name)
        B, C, S = case['batch'], len(case['attrs']), case['size']
        n_latent = 2 * int(np.log2(S)) - 2
        assert c.d['alpha_org'].shape == (B, C) and c.d['eps'].shape == (B, C) and c.d['target'].shape == (B, C) and c.d['po'].shape == (B, C)
        assert c.d['grad'].shape == (C, n_latent, 512) and np.isfinite(c.d['grad']).all() and np.abs(c.d['grad']).max() > 0
        for k in ('x0', 'x1'):
            assert c.d[k + '.rows'].shape == (B, 3, S) and c.d[k + '.cols'].shape == (B, 3, S) and c.d[k + '.probes'].shape == (B, 3, oracle_cache.NPROBE)
            np.testing.assert_allclose(c.d[k + '.rows'].sum(2), c.d[k + '.cols'].sum(2), rtol=1e-9, atol=1e-6)
            assert np.abs(c.d[k + '.probes']).max() <= float(c.d[k + '.absmax'])
        # transform_base.py:475-486: loss = 10 reg + 0.05 cont + 0.05 gan
        np.testing.assert_allclose(float(c.d['loss']), 10 * float(c.d['reg']) + 0.05 * float(c.d['cont']) + 0.05 * float(c.d['gan']), rtol=1e-6)
        alpha = case['alpha']()
        if case['clamp']:                                  # graphs/pggan/transform_base.py:358-364
            np.testing.assert_allclose(c.d['target'], np.clip(c.d['alpha_org'] + alpha, 0, 1), atol=1e-6)
            np.testing.assert_allclose(c.d['eps'], c.d['target'] - c.d['alpha_org'], atol=1e-6)
        else:                                              # transform_base.py:405-408
            np.testing.assert_allclose(c.d['eps'], alpha - c.d['alpha_org'], atol=1e-6)
        t, p = torch.from_numpy(c.d['target']), torch.from_numpy(c.d['po'])
        bce = float(-(t * p.clamp(min=1e-12).log() + (1 - t) * (1 - p).clamp(min=1e-12).log()).mean())
        np.testing.assert_allclose(bce, float(c.d['reg']), rtol=1e-4)      # po really are the oracle regressor's outputs on the oracle's x1
