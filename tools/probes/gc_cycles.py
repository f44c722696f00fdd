"""python tools/probes/gc_cycles.py: which reference cycles does one training step leave behind?  (A cycle that holds GPU tensors keeps their
memory until the generational collector runs, and a generation-2 pass over the networks' Python objects takes ~50-130 ms of host time: the one
slow step per bench run.)  Runs three 64^2 steps with the collector off, then collects with DEBUG_SAVEALL and prints the garbage by type."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gc
import collections
import time
import numpy as np
import torch
from latent2im_amd import selfcheck, synth
res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
g = selfcheck.build_graph(res, ['Smiling'], 4)
zs = synth.z_sample(4, seed=0)
for _ in range(2):
    selfcheck.run_step(g, zs, np.ones((4, 1)) * 0.2)
torch.cuda.synchronize()
gc.collect()
print('tracked objects after build + 2 steps:', len(gc.get_objects()))
t0 = time.perf_counter(); gc.collect(); print('full collection with everything live: %.1f ms' % ((time.perf_counter() - t0) * 1e3))
gc.disable()
for _ in range(3):
    r = selfcheck.run_step(g, zs, np.ones((4, 1)) * 0.2)
del r
torch.cuda.synchronize()
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
print('unreachable objects after 3 steps:', n)
c = collections.Counter(type(o).__name__ for o in gc.garbage)
print(c.most_common(25))
tens = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
print('tensors in cycles:', len(tens), 'bytes', sum(t.numel() * t.element_size() for t in tens))
for o in gc.garbage[:400]:
    if type(o).__name__ in ('function', 'cell', 'tuple', 'dict') and not isinstance(o, torch.Tensor):
        s = repr(o)[:160]
        if 'latent2im' in s or 'lambda' in s or 'backward' in s:
            print(type(o).__name__, s)
gc.set_debug(0)
gc.garbage.clear()
gc.freeze()
t0 = time.perf_counter(); gc.collect(); print('full collection after gc.freeze(): %.2f ms' % ((time.perf_counter() - t0) * 1e3))
