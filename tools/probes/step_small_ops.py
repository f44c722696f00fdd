"""[r6] Which torch ops of a training step launch the ~200 tiny (< 12 us) kernels, and from where?  One eager step of the c5 (or c3) workload under torch.profiler,
ops grouped by call site (three stack frames), those with short device time listed by count.   usage: python tools/probes/step_small_ops.py [c5|c3]"""
import sys
import torch
sys.path.insert(0, '.')
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c5'
w = bench.Workload(cfg, 'f16' if cfg == 'c5' else 'f32', 1024, 8, None, 1, 6, False)
step = w.stepper(graph=False)
for i in range(3):
    step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(3)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=4):
    dev = getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0))
    self_dev = getattr(e, 'self_device_time_total', getattr(e, 'self_cuda_time_total', 0))
    if self_dev > 0 and self_dev / max(e.count, 1) < 14:
        stack = [s for s in e.stack if 'latent2im_amd' in s or 'bench.py' in s or 'graph' in s][:3]
        rows.append((e.count, self_dev / max(e.count, 1), e.key, str(e.input_shapes)[:60], ' <- '.join(s.split('/')[-1][:60] for s in stack)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('# %s: %d op instances with < 14 us of device time each' % (cfg, tot))
for r in rows[:70]:
    print('%3d x %5.1f us  %-28s %-60s %s' % r)
