import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    # the CPU oracle: as many torch threads as the host really gives this process (the GPU box shows 256 hardware threads to a pod with a cgroup
    # quota of 16 CPUs: 16 threads 5.3 s, 32 threads 7.1 s, 64 threads 10 s for the same oracle step, tools/probes/oracle_threads.py)
    import torch
    import oracle
    torch.set_num_threads(min(32, oracle.host_cpus()))
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: takes more than a few seconds on CPU')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load
