"""CPU oracle for the Latent2im walk-training hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch-on-CPU restatement of
the reference's algorithm for the path named in BASELINE.json (sample z -> W+ ->
linear walk -> StyleGAN2 synthesis x2 -> ResNet-50 / VGG-19 / discriminator
losses -> gradient into the walk -> Adam).  Each function cites the reference
``file:line`` it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``latent2im_amd`` never does (tests/test_host_logic.py::test_product_never_imports_the_oracle_or_the_reference
enforces that).

Pinning status
--------------
* First-party arithmetic (generator, discriminator, fused_bias_act, upfirdn2d,
  walk, losses, optimizer step, PGGAN-256 generator): PINNED against the
  reference's own Python imported in the build container through import shims
  (tests/golden/make_golden.py; outputs committed under tests/golden/*.npz and
  re-checked by tests/test_oracle_golden.py, tests/test_oracle_pggan.py and
  tests/test_next_rows_cpu.py).
* Third-party arithmetic (torchvision ResNet-50 v0.5.0 and VGG-19 ``features``):
  source is neither vendored in the reference nor installed in this image, and
  the reference holds no test or golden vector at that boundary => PARITY
  UNPINNED for those two networks; they are restated from the published
  architecture on top of torch's own conv / batch-norm
  primitives, and the product is checked against this restatement.
"""


def host_cpus():
    """CPUs this process may actually use: the smaller of the visible CPUs, the scheduler affinity and the cgroup CPU quota (``cpu.max``: the GPU
    box shows 256 hardware threads to a pod whose quota is 16 CPUs — more runnable threads than that only get throttled, which is why torch's
    CPU convolutions "collapse" there at high thread counts and why neither threads nor worker processes speed the oracle up)."""
    import math
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, math.ceil(q / per)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n
