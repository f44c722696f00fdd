"""RCCL sanity on a 1-GPU box: the collectives the walk-training step uses (all_reduce SUM / MAX, broadcast, barrier) on a
world of one rank, through the same helpers bench.py calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as td
from latent2im_amd import dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
os.environ.setdefault('NCCL_DEBUG_FILE', '/tmp/rccl_debug_%h_%p.log')
if os.environ.get('NCCL_DEBUG', '').upper() == 'VERSION':                 # the box exports it: RCCL then prints a banner to STDOUT at exit
    del os.environ['NCCL_DEBUG']
td.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
p = torch.nn.Parameter(torch.ones(1, 18, 512, device='cuda'))
p.grad = torch.full_like(p, 3.0)
td.all_reduce(p.grad, op=td.ReduceOp.SUM)
td.broadcast(p.data, src=0)
dist.barrier()
print('rccl ok', float(p.grad.mean()), dist.max_over_ranks(1.25, torch.device('cuda', 0)), td.get_backend())
td.destroy_process_group()
