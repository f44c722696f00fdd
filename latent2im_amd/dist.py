"""Data parallelism for the walk-training step: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over
xGMI on ROCm; ``gloo`` for the CPU tests).  The path shards by sample: every rank holds a full replica of the frozen
networks (0.33 GB fp32) and a slice of each batch; the only exchange per step is ONE all-reduce of the walk gradient
([n_attr, n_latent, 512] fp32 = 28..184 KB -> latency-bound, a single small-message collective, nothing to bucket or
overlap: it is the last op before Adam).  The reference has no distributed code at all (SURVEY §2 last rows)."""
import os
import tempfile

import torch
import torch.distributed as td


def is_initialized():
    return td.is_available() and td.is_initialized()


def rank():
    return td.get_rank() if is_initialized() else 0


def world_size():
    return td.get_world_size() if is_initialized() else 1


def select_gpu(gpu):
    """``--gpu`` of the drivers (reference train.py:150 / vis_w.py: CUDA_VISIBLE_DEVICES set before the first CUDA call).  Must run
    BEFORE anything initialises the HIP runtime: then the variable is exported (HIP honours both spellings); if the runtime is
    already up (a caller built a graph earlier in the process) the first listed id becomes the current device instead.  Ignored
    under torchrun (LOCAL_RANK picks the device) and when the launcher already pinned HIP_VISIBLE_DEVICES.  Never re-execs."""
    if not gpu or int(os.environ.get('WORLD_SIZE', '1')) > 1 or 'HIP_VISIBLE_DEVICES' in os.environ:
        return
    gpu = str(gpu)
    if torch.cuda.is_initialized():
        torch.cuda.set_device(int(gpu.split(',')[0]))
    else:
        os.environ['CUDA_VISIBLE_DEVICES'] = gpu
        os.environ['HIP_VISIBLE_DEVICES'] = gpu


def init_from_env(backend=None):
    """Initialise from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rk = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', str(rk)))
    if (world > 1 or os.environ.get('L2I_FORCE_PG')) and not is_initialized():      # L2I_FORCE_PG: build the group for one rank too (rehearsal)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('L2I_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            # one GPU per rank; more ranks than GPUs only with L2I_DIST_BACKEND=gloo (control-flow rehearsal on a single-GPU box)
            torch.cuda.set_device(local if backend == 'nccl' else local % torch.cuda.device_count())
        if backend == 'nccl':
            # RCCL writes its NCCL_DEBUG output (the box exports NCCL_DEBUG=VERSION: a version banner, flushed at exit) to STDOUT,
            # where a driver expects exactly one JSON line: send it to a file instead
            os.environ.setdefault('NCCL_DEBUG_FILE', os.path.join(tempfile.gettempdir(), 'rccl_debug_%h_%p.log'))
            if os.environ.get('NCCL_DEBUG', '').upper() == 'VERSION':      # the version banner ignores NCCL_DEBUG_FILE (measured)
                del os.environ['NCCL_DEBUG']
            td.init_process_group(backend=backend, rank=rk, world_size=world, device_id=torch.device('cuda', local))
        else:
            td.init_process_group(backend=backend, rank=rk, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local if local < torch.cuda.device_count() else 0)
    return rk, world, local


def shard(n, rk=None, world=None):
    """Rows of a global batch of ``n`` samples owned by this rank: every ``world``-th sample starting at ``rank`` (SURVEY 8e allows
    ``r::world`` or contiguous).  Equal shards, so the mean of the shard means is the global mean for every loss term — and the
    STRIDED choice also keeps the discriminator's minibatch-stddev statistics (networks.py:630-638: sample b belongs to subgroup
    b mod (B/4)) identical to a single process on the global batch: with a per-rank batch that is a multiple of 4, two samples of a
    rank share a local subgroup exactly when they share a global one (world * (B_local/4) = B/4), so the GAN term needs no
    cross-rank exchange and data parallelism stays one all-reduce of the walk gradient."""
    rk = rank() if rk is None else rk
    world = world_size() if world is None else world
    if n % world != 0:
        raise ValueError('global batch %d is not divisible by world size %d' % (n, world))
    return slice(rk, n, world)


def spawn_local(n, argv, env=None, timeout=None):
    """Start ``n`` FRESH child processes ``argv`` on this node, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
    MASTER_PORT set; each child calls ``init_from_env``), wait for all of them and return (exit codes, rank 0's stdout).  The other
    ranks' stdout and every stderr go to this process's stderr.  The caller must not have initialised the GPU if it wants to keep
    using this process afterwards for GPU work of its own; nothing here touches the GPU, and no process image is replaced."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen(list(argv), env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    # Poll every child: when one rank dies early (import error, out of memory) the others would wait in RCCL init / the all-reduce for
    # ever, so the first non-zero exit (or the timeout) ends exactly the processes started here.  Rank 0's stdout is drained by a
    # thread so that a chatty rank cannot fill the pipe and stall.
    import threading
    import time
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = None if timeout is None else time.monotonic() + timeout
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes) or (deadline is not None and time.monotonic() > deadline):
            grace = time.monotonic() + 3.0     # let the failing rank's siblings finish / print their own errors first
            while time.monotonic() < grace and any(p.poll() is None for p in procs):
                time.sleep(0.05)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            codes = [p.wait() for p in procs]
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    out0 = b''.join(c for c in chunks if c)
    return codes, out0.decode(errors='replace')


def broadcast_parameters(params, src=0):
    """One source of truth for the trainable state at start-up: every parameter of the walk module from rank ``src`` (the linear walk
    has one tensor ``w``; the MLP / non-linear walks of transform_base.py:168-243 have several and no ``w``)."""
    if world_size() == 1:
        return
    for p in params:
        td.broadcast(p.data, src=src)


def average_gradients(params):
    """All-reduce (mean) the gradients of the trainable parameters — for this path: the walk tensor only.  Without a process group: a
    no-op.  A ONE-rank group (L2I_FORCE_PG=1: the RCCL rehearsal on a single-GPU box) still runs the collective."""
    if not is_initialized():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    if len(grads) == 1:
        flat = grads[0]
        td.all_reduce(flat, op=td.ReduceOp.SUM)
        flat.div_(world_size())
    else:
        flat = torch.cat([g.reshape(-1) for g in grads])
        td.all_reduce(flat, op=td.ReduceOp.SUM)
        flat.div_(world_size())
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()


def barrier():
    if is_initialized():
        td.barrier()


def shutdown():
    if is_initialized():
        td.destroy_process_group()


def max_over_ranks(value, device=None):
    if world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else ('cuda' if td.get_backend() == 'nccl' else 'cpu'))
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def _device_identity():
    """(device index, 'domain:bus:device' PCI address or None) of this rank's current GPU: the address is what tells two ranks on ONE physical GPU
    apart from two ranks on two GPUs when each process sees its own device as index 0 (HIP_VISIBLE_DEVICES per rank)."""
    if not torch.cuda.is_available():
        return None, None
    i = torch.cuda.current_device()
    pr = torch.cuda.get_device_properties(i)
    try:
        pci = '%04x:%02x:%02x' % (int(getattr(pr, 'pci_domain_id', 0)), int(pr.pci_bus_id), int(pr.pci_device_id))
    except (AttributeError, TypeError, ValueError):
        pci = None
    return i, pci


def ranks_seen():
    """[{rank, device, pci, host}] of every rank (all_gather_object; one entry without a process group): self-reporting evidence of which
    GPUs a multi-rank run really used."""
    import socket
    i, pci = _device_identity()
    me = dict(rank=rank(), device=i, pci=pci, host=socket.gethostname())
    if not is_initialized():
        return [me]
    out = [None] * world_size()
    td.all_gather_object(out, me)
    return out


def assert_distinct_devices(seen=None):
    """Start-up check of a multi-rank run over RCCL: the N ranks sit on N DISTINCT physical GPUs (host + PCI address, falling back to the device
    index) — two ranks time-slicing one GPU would still produce a line, with half the throughput and no hint why.  Rehearsals over gloo
    (L2I_DIST_BACKEND=gloo: N ranks on a single-GPU box) are exempt.  Returns the gathered list."""
    seen = ranks_seen() if seen is None else seen
    if is_initialized() and td.get_backend() == 'nccl' and len(seen) > 1:
        ids = [(r['host'], r['pci'] if r.get('pci') else r['device']) for r in seen]
        if len(set(ids)) != len(ids):
            raise RuntimeError('data-parallel run over RCCL with %d ranks on %d distinct GPUs: %s' % (len(ids), len(set(ids)), seen))
    return seen


def gather_floats(value):
    """[value of rank 0, value of rank 1, ...] on every rank."""
    if not is_initialized():
        return [float(value)]
    out = [None] * world_size()
    td.all_gather_object(out, float(value))
    return out


def time_allreduce(tensor, iters=100):
    """Median wall time (microseconds) of ``iters`` all-reduces of a tensor shaped like the walk gradient, each bracketed by a device
    synchronisation — the latency of the step's only collective.  None without a process group."""
    import time
    if not is_initialized():
        return None
    buf = torch.zeros_like(tensor)
    for _ in range(5):
        td.all_reduce(buf)
    torch.cuda.synchronize() if buf.is_cuda else None
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        td.all_reduce(buf)
        if buf.is_cuda:
            torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    return ts[len(ts) // 2]

