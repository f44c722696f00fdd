"""Build-container script: evaluates the CPU oracle (oracle.step.train_step_bounded, float32) for the cached cases of tests/oracle_cache.py and writes
their compact summaries to tests/golden/oracle_1024.npz.

    python tests/golden/make_oracle_cache.py [case ...]        # default: every case; existing entries of other cases are kept
    python tests/golden/make_oracle_cache.py --stamp           # write the fingerprints of entries that have none (no evaluation)

The 1024^2 batch-8 cases take minutes of CPU each and tens of GB of host memory (one minibatch-stddev subgroup of four samples at a time).  The GPU
tests test_config3_whole_step_1024_batch8_vs_oracle / test_config4_* read the file instead of re-evaluating the oracle on the GPU box's 16-CPU pod
(128 s each); tests/test_oracle_cache_cpu.py re-derives the 256^2 entry from a live evaluation to keep the cache honest."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle                                             # noqa: E402
from tests import oracle_cache                            # noqa: E402

if __name__ == '__main__':
    torch.set_num_threads(min(32, oracle.host_cpus()))
    path = os.path.join(ROOT, 'tests', 'golden', 'oracle_1024.npz')
    data = dict(np.load(path, allow_pickle=False)) if os.path.isfile(path) else {}
    if sys.argv[1:] == ['--stamp']:
        for name in oracle_cache.CASES:
            if '%s.loss' % name in data and '%s.meta_sha' % name not in data:
                data['%s.meta_sha' % name] = np.asarray(oracle_cache.case_fingerprint(name))
                print('stamped', name)
        np.savez_compressed(path, **data)
        sys.exit(0)
    for name in (sys.argv[1:] or list(oracle_cache.CASES)):
        t0 = time.time()
        summ = oracle_cache.evaluate_and_summarize(name)
        data = {k: v for k, v in data.items() if not k.startswith(name + '.')}
        for k, v in summ.items():
            data['%s.%s' % (name, k)] = v
        data['%s.meta_sha' % name] = np.asarray(oracle_cache.case_fingerprint(name))
        print('%s: %.0f s' % (name, time.time() - t0), flush=True)
        np.savez_compressed(path, **data)
    print('%s: %d arrays, %.2f MB' % (path, len(data), os.path.getsize(path) / 1e6))
