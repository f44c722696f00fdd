"""The walk-training step as a hipGraph (BASELINE config 5: "hipGraph-captured step").

Everything between "z and alpha are on the device" and "the walk gradient is complete" is static-shaped GPU work issued from
Python one launch at a time (~300 conv launches + ~500 small kernels per step).  ``CapturedStep`` records that region ONCE with
``torch.cuda.graph`` (HIP stream capture -> hipGraphInstantiate) and replays it with a single ``hipGraphLaunch`` per step:

    z, alpha (static device buffers, refilled by an async copy before each replay)
      -> style MLP -> synthesis (x0) -> regressor -> epsilon (device-side: no ``.item()``, no host read of alpha_org)
      -> walk -> synthesis (x1) -> D / VGG / regressor losses on their three streams (forked from and joined to the captured
         stream) -> backward into ``walk.grad``

The all-reduce of the walk gradient (RCCL) and the Adam update stay OUTSIDE the graph, in the order optimizeParametersAll runs
them: a handful of tiny launches, and the captured region holds no collective, so the same graph serves 1 and N ranks.
Per-layer generator noise (networks.py:281-286) is drawn inside the graph from torch's graph-safe Philox stream (fresh values every
replay).  Reference: train.py:56-110 is the region; the reference itself is eager PyTorch.
"""
import numpy as np
import torch

from . import dist


def freeze_host_objects():
    """Call once after the first training step(s): moves every Python object alive now — the networks' ~270 000 modules, launch plans, packed
    weight handles — into the collector's permanent generation (``gc.freeze``).  A full collection otherwise walks all of them (measured
    73 ms) whenever the young generations overflow, about once per 20 steps, on the thread that launches the step's ~650 kernels and is barely
    ahead of the GPU: one 160-250 ms step among 120 ms ones (the +4 % between a 5-step and a 20-step timing of round 3).  Afterwards a full
    pass only sees objects created since (0.01 ms).  The step itself leaves no reference cycles behind (tools/probes/gc_cycles.py).
    A program that later drops a whole graph and builds another calls ``gc.unfreeze()`` first (frozen objects are still freed by reference
    counting; only a cycle among them would wait for the unfreeze), as bench.py does between its workloads."""
    import gc
    gc.collect()
    gc.freeze()


def forward(g, z, alpha_for_graph, clamp=False, layers=None, content=False):
    """train.py:56-101 on DEVICE inputs: both generator passes, the regressor on the original, epsilon and the walk; no host
    synchronisation.  Returns (feed_dict for optimizeParametersAll, dict of device tensors).  ``content``: the step will take the content
    loss: the VGG taps of the original image start now, beside the regressor and the second generator pass (graph.prefetch_content_taps)."""
    w = g.get_w(z)
    x0 = g.get_logits({'w': w})
    if content:
        g.prefetch_content_taps(x0)
    a0 = g.get_reg_preds(x0)
    if clamp:
        target, eps = g.get_alphas_clamped(a0, alpha_for_graph)
    else:
        target, eps = alpha_for_graph, g.get_alphas(a0, alpha_for_graph)
    w1 = g.get_w_new_tensor(w, eps, layers=layers)
    x1 = g.get_logits({'w': w1})
    return {'w': w1, 'org': x0, 'logit': x1, 'alpha': target}, dict(x0=x0, x1=x1, a0=a0, eps=eps)


def forward_backward(g, z, alpha_for_graph, no_content_loss=False, no_gan_loss=False, clamp=False, layers=None):
    """``forward`` + the loss and backward of optimizeParametersAll (transform_base.py:459-488), without its all-reduce / Adam tail.
    Leaves d loss / d walk in ``walk.grad``."""
    feed, r = forward(g, z, alpha_for_graph, clamp=clamp, layers=layers, content=not no_content_loss)
    g.optimizers.zero_grad()
    loss = g.get_w_loss(feed, no_content_loss, no_gan_loss)
    loss.backward()
    r.update(loss=loss.detach(), terms=g.last_terms)
    return r


class CapturedStep:
    """``step(zs, alpha)`` == selfcheck.run_step(g, zs, alpha, ...) with the forward+backward replayed from one hipGraph."""

    def __init__(self, g, batch, n_attr, no_content_loss=False, no_gan_loss=False, clamp=False, layers=None, warmup=2):
        self.g = g
        dev = g.device
        self.z = torch.zeros(batch, g.dim_z, device=dev)
        self.alpha = torch.zeros(batch, n_attr, device=dev)
        # pinned staging ring: the host runs several replays ahead of the GPU (no per-step sync in bench.py / trainer --no_log_sync), so a
        # slot is rewritten only after the H2D copies that read it have run (one event per slot)
        self._ring = [(torch.zeros(batch, g.dim_z).pin_memory(), torch.zeros(batch, n_attr).pin_memory(), torch.cuda.Event()) for _ in range(4)]
        self._ring_used = [False] * len(self._ring)
        self._slot = 0
        kw = dict(no_content_loss=no_content_loss, no_gan_loss=no_gan_loss, clamp=clamp, layers=layers)
        # warm-up on a side stream (allocator pools, split-K workspaces, lazily packed weights), as stream capture requires
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(warmup, 1)):
                forward_backward(g, self.z, self.alpha, **kw)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g.optimizers.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = forward_backward(g, self.z, self.alpha, **kw)
        self.grads = [p.grad for p in g.walk.parameters()]          # static tensors inside the graph's pool, rewritten by every replay
        # the graph's kernel nodes hold RAW pointers into the split-K workspaces of conv._WS (allocated during warm-up, outside the graph's
        # pool): keep those tensors alive for as long as the graph, even if a later eager launch replaces the dict entry with a larger one
        from . import conv
        self._ws_keep = list(conv._WS.values())
        self.launches = None

    def load(self, zs, alpha):
        """Host -> the graph's static inputs (pinned staging ring, asynchronous on the current stream)."""
        i = self._slot
        self._slot = (i + 1) % len(self._ring)
        stage_z, stage_a, ev = self._ring[i]
        if self._ring_used[i]:
            ev.synchronize()                                  # the copies that last read this slot have completed
        stage_z.copy_(torch.as_tensor(np.asarray(zs), dtype=torch.float32))
        stage_a.copy_(torch.as_tensor(np.asarray(alpha), dtype=torch.float32))
        self.z.copy_(stage_z, non_blocking=True)
        self.alpha.copy_(stage_a, non_blocking=True)
        ev.record()
        self._ring_used[i] = True

    def __call__(self, zs=None, alpha=None, optimize=True):
        if zs is not None:
            self.load(zs, alpha)
        self.graph.replay()
        for p, gr in zip(self.g.walk.parameters(), self.grads):      # Adam reads p.grad: the graph's static gradient tensors
            p.grad = gr
        if optimize:
            dist.average_gradients(self.g.walk.parameters())         # one RCCL all-reduce of <= 184 KB, outside the graph
            self.g.optimizers.step()
        o = dict(self.out)
        o['grad'] = self.grads[0]
        return o
