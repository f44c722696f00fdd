"""python tools/probes/oracle_threads.py: the CPU oracle's 1024^2 batch-1 training step at several torch thread counts (what is the host really worth?)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import oracle
from latent2im_amd import synth
from oracle import step as ostep
print('cpu_count', os.cpu_count(), 'host_cpus', oracle.host_cpus(), 'cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else None)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nets = dict(G=ostep.to_torch(synth.generator_state(size, seed=100)), D=ostep.to_torch(synth.discriminator_state(size, seed=200)),
            R=ostep.to_torch(synth.resnet50_state(seed=300)), V=ostep.to_torch(synth.vgg19_prefix_state(seed=400)))
n_latent = 2 * int(np.log2(size)) - 2
walk = torch.from_numpy(synth.walk_init(1, n_latent, seed=7))
z = torch.from_numpy(synth.z_sample(1, seed=0)).float()
for th in (8, 16, 24, 32, 64):
    torch.set_num_threads(th)
    ostep.train_step(nets, walk, z, torch.full((1, 1), 0.3), [31])
    t0 = time.time()
    ostep.train_step(nets, walk, z, torch.full((1, 1), 0.3), [31])
    print('threads %3d: %.2f s per %d^2 batch-1 step' % (th, time.time() - t0, size), flush=True)
