"""The optimiser tail of the walk step on the fp16 path: a dynamic loss scale and an Adam that skips non-finite steps, both on the device.

The reference ends every step with ``self.optimizers.step()`` (transform_base.py:487-488; torch.optim.Adam(lr, betas=(0.5, 0.99)), :329-331) on the
walk.  Run under autocast — what BASELINE configs[4] ("fp16 MFMA") amounts to for the reference — that call would go through a torch GradScaler:
scale the loss, look for inf / NaN in the gradients, skip the update and halve the scale if there is one, double it after ``growth_interval`` clean
steps.  torch's GradScaler reads its ``found_inf`` back to the host before every optimiser step; the step here never synchronises (train.py:110 is
the only sync of the reference loop, and the hipGraph replay of config 5 has none), so the same semantics are kept in three device words
(csrc/l2i_optim.hip).

``LossScaler``  — per graph: the STATIC per-branch exponents of nets16.loss_scale_for (each loss branch's largest gradient map near 2^5) times ONE
                  dynamic power of two ``scale[0]`` (1.0 at start).  Every branch multiplies the gradient it receives by static * dynamic
                  (a device tensor, so a captured hipGraph sees every update) and the generator's latent gradient by the inverse: walk.grad is the
                  true gradient, or non-finite if anything overflowed on the way.
``GuardedAdam`` — torch.optim.Adam whose step() is one launch of l2i_adam_guarded_f32 per parameter tensor: finite check, update or skip,
                  scale-state update.  Same state layout as torch's (exp_avg, exp_avg_sq, step) so state_dict() round-trips.
"""
import torch

from . import _lib


class LossScaler:
    GROWTH, BACKOFF = 2.0, 0.5

    def __init__(self, log2, device, growth_interval=2000, max_log2=8):
        """``log2``: dict(R, V, D, G) of static exponents.  ``growth_interval``: clean steps before the dynamic factor doubles (torch's default);
        ``max_log2``: it never grows beyond 2^max_log2 (the static exponents already sit eleven octaves under fp16's largest number)."""
        self.log2 = dict(log2)
        self.device = device
        self.growth_interval = int(growth_interval)
        self.max_scale = float(2.0 ** max_log2)
        self.scale = torch.tensor([1.0, 1.0], dtype=torch.float32, device=device)          # [dynamic factor, its inverse]
        self.state = torch.zeros(4, dtype=torch.int32, device=device)                      # L2I_LS_FOUND / TRACKER / SKIPPED / STEPS
        self.dyn, self.inv_dyn = self.scale[0:1], self.scale[1:2]                          # views: what the captured graph multiplies by

    def static(self, key):
        return float(2.0 ** self.log2[key])

    def stats(self):
        """dict(scale, tracker, skipped, steps) — a host read (synchronises): logging and tests only."""
        st = self.state.tolist()
        return dict(scale=float(self.scale[0]), tracker=st[1], skipped=st[2], steps=st[3])


class GuardedAdam(torch.optim.Adam):
    """torch.optim.Adam(params, lr, betas) for float32 CUDA parameters with the update on l2i_adam_guarded_f32: skipped as a whole when any gradient
    of the step is non-finite; ``scaler`` (a LossScaler or None) is advanced by the last launch of the step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, scaler=None):
        super().__init__(params, lr=lr, betas=betas, eps=eps)
        self.scaler = scaler
        self._own_state = None

    def _flags(self, device):
        if self.scaler is not None:
            return self.scaler.state, self.scaler.scale
        if self._own_state is None or self._own_state.device != device:
            self._own_state = torch.zeros(4, dtype=torch.int32, device=device)
        return self._own_state, None

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError('GuardedAdam.step takes no closure (the reference calls optimizers.step() bare, transform_base.py:488)')
        todo = []
        for group in self.param_groups:
            if group.get('weight_decay', 0) or group.get('amsgrad', False) or group.get('maximize', False):
                raise NotImplementedError('GuardedAdam: plain Adam only (the reference uses lr and betas, transform_base.py:329-331)')
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.grad.dtype != torch.float32:
                    raise _lib.L2IError('GuardedAdam updates contiguous float32 GPU parameters only')
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif not st['step'].is_cuda:                           # a state_dict saved by torch's own Adam keeps the counter on the host
                    st['step'] = st['step'].to(device=p.device, dtype=torch.float32)
                todo.append((p, group, st))
        if not todo:
            return None
        lib = _lib.load()
        state, scale = self._flags(todo[0][0].device)
        sc = self.scaler
        single = len(todo) == 1
        stream = _lib.stream_ptr()
        if not single:
            for p, _, _ in todo:
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                _lib.check(lib.l2i_nonfinite_flag_f32(_lib.fptr(g), g.numel(), _lib.ptr(state), stream), 'l2i_nonfinite_flag_f32')
        for i, (p, group, st) in enumerate(todo):
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            b1, b2 = group['betas']
            _lib.check(lib.l2i_adam_guarded_f32(_lib.fptr(p), _lib.fptr(g), _lib.fptr(st['exp_avg']), _lib.fptr(st['exp_avg_sq']), _lib.fptr(st['step']),
                                                p.numel(), float(group['lr']), float(b1), float(b2), float(group['eps']), 1 if single else 0,
                                                _lib.ptr(state), _lib.ptr(scale), float(sc.GROWTH if sc else 2.0), float(sc.BACKOFF if sc else 0.5),
                                                int(sc.growth_interval if sc else 0), float(sc.max_scale if sc else 1.0),
                                                1 if i == len(todo) - 1 else 0, stream), 'l2i_adam_guarded_f32')
        return None
