"""One rank of the data-parallel check of the PRODUCT graph (tests/test_networks_gpu.py::test_data_parallel_product_graph_matches_single_process):
the real faceGraph for a few optimizeParametersAll steps on this rank's shard of every global batch, exactly the way bench.py and
trainer.train drive it (dist.init_from_env -> broadcast of the walk -> dist.shard -> all-reduce inside optimizeParametersAll).
Run as a script, once per rank (dist.spawn_local) or once without WORLD_SIZE for the single-process answer; rank 0 writes an .npz."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch


def main(out, size, global_batch, steps, no_gan, walk='linear'):
    from latent2im_amd import constants, dist, selfcheck, synth
    rk, world, _ = dist.init_from_env()
    np.random.seed(1234)
    if walk == 'mlp':                                       # transform_base.py:168-204 (is_mlp): several parameter tensors, no ``.w``
        constants.WALK_IS_MLP = True
        torch.manual_seed(100 + rk)                         # every rank starts from a DIFFERENT init: only the broadcast makes them agree
    g = selfcheck.build_graph(size, ['Smiling', 'Young'], global_batch, lr=1e-3)
    dist.broadcast_parameters(g.walk.parameters())          # what trainer.main / bench.py do at start-up
    sl = dist.shard(global_batch)
    zs_all = synth.z_sample(global_batch * steps, seed=3)
    grads, losses = [], []
    for i in range(steps):
        zs = zs_all[i * global_batch:(i + 1) * global_batch][sl]
        alpha = np.ones((zs.shape[0], 2)) * np.random.uniform(0, 1, 2)          # same draw on every rank (same seed)
        r = selfcheck.run_step(g, zs, alpha, no_gan_loss=no_gan)
        grads.append(np.concatenate([p.grad.detach().cpu().numpy().reshape(-1) for p in g.walk.parameters()]))     # AFTER the all-reduce
        t = torch.stack([r['loss'].double()] + [x.double().reshape(()) for x in (r['terms']['reg'], r['terms']['cont'])]
                        + ([] if no_gan else [r['terms']['gan'].double().reshape(())])).cpu()
        if world > 1:                                                           # logging-only mean of the shard losses
            torch.distributed.all_reduce(t)
            t /= world
        losses.append(t.numpy())
    torch.cuda.synchronize()
    if rk == 0:
        np.savez(out, walk=np.concatenate([p.detach().cpu().numpy().reshape(-1) for p in g.walk.parameters()]), grads=np.stack(grads),
                 losses=np.stack(losses), world=world,
                 backend=np.asarray(torch.distributed.get_backend() if dist.is_initialized() else 'none'),
                 precision=np.asarray(__import__('latent2im_amd.conv', fromlist=['conv']).PRECISION))
    dist.barrier()
    dist.shutdown()


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] == '1', sys.argv[6] if len(sys.argv) > 6 else 'linear')
