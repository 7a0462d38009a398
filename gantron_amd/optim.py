"""Flat-buffer Adam + gradient clipping on the GPU (reference train.py:242-243, 404-406).

The reference creates `torch.optim.Adam(params, lr, weight_decay=1e-6)` per network and, for the
generator, calls `clip_grad_norm_(params, 1.0)` before `step()`.  Here all parameters of a module
are re-homed as views of ONE contiguous fp32 buffer and all gradients land in ONE flat buffer, so:

* the update is two launches (gt_sumsq for the global norm, gt_adam_step with the clip coefficient
  read on the device: no host synchronisation),
* data-parallel training all-reduces that single flat gradient buffer (distributed.py).

`state_dict()` / `load_state_dict()` use torch.optim.Adam's layout (per-parameter `step`,
`exp_avg`, `exp_avg_sq`), so reference checkpoints (train.py:158-166) round-trip.
"""
import torch

from . import ops


def flatten_module(module):
    """Re-home every parameter of `module` as a view of one flat buffer; create the flat gradient
    buffer the autograd nodes of model.py write into.  Idempotent."""
    if getattr(module, '_flat_param', None) is not None:
        return module._flat_param, module._flat_grad
    params = [p for _, p in module.named_parameters()]
    assert params and all(p.dtype == torch.float32 for p in params)
    dev = params[0].device
    sizes = [p.numel() for p in params]
    offsets, off = [], 0
    for n in sizes:
        offsets.append(off)
        off += -(-n // 4) * 4          # keep every tensor 16-byte aligned inside the flat buffers
    flat = torch.zeros(off, device=dev, dtype=torch.float32)
    grad = torch.zeros(off, device=dev, dtype=torch.float32)
    views = {}
    for (name, p), o, n in zip(module.named_parameters(), offsets, sizes):
        flat[o:o + n].copy_(p.data.reshape(-1))
        p.data = flat[o:o + n].view_as(p.data)
        views[name] = grad[o:o + n].view_as(p.data)
    module._flat_param, module._flat_grad, module._grad_views = flat, grad, views
    module._flat_layout = list(zip([k for k, _ in module.named_parameters()], offsets, sizes))
    return flat, grad


def enable_direct_grads(module):
    """The autograd nodes of model.py then ACCUMULATE parameter gradients straight into the flat gradient buffer and
    point every p.grad at its view of it: no per-backward zero-filled gradient tensors, no copy into the flat buffer
    before the optimizer step or the all-reduce.  Because the buffer is the only home of the gradients, zeroing it is
    part of zero_grad(): the module's zero_grad is rebound accordingly."""
    flatten_module(module)
    if getattr(module, '_direct_grads', False):
        return
    module._direct_grads = True

    def zero_grad(set_to_none=True):
        module._flat_grad.zero_()
        for p in module.parameters():
            p.grad = None
    module.zero_grad = zero_grad


def flat_range(module, prefix):
    """[begin, end) of the flat buffers covered by the parameters whose names start with `prefix` (contiguous when the
    prefix names a sub-module)."""
    hits = [(o, o + -(-n // 4) * 4) for k, o, n in module._flat_layout if k.startswith(prefix)]
    if not hits:
        return None
    lo, hi = min(h[0] for h in hits), max(h[1] for h in hits)
    if sum(e - b for b, e in hits) != hi - lo:
        raise ValueError(f'parameters named {prefix}* are not contiguous in the flat buffer: [{lo}, {hi}) also holds others')
    return lo, hi


class FlatAdam:
    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.module = module
        self.flat, self.grad = flatten_module(module)
        enable_direct_grads(module)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.step_count = 0
        self._sumsq = torch.zeros(1, device=self.flat.device, dtype=torch.float32)

    def gather_grads(self):
        """Make sure the flat gradient buffer holds every parameter's .grad (autograd may hand back a
        copy instead of the view the backward node wrote)."""
        views = self.module._grad_views
        for name, p in self.module.named_parameters():
            if p.grad is None:
                continue
            if p.grad.data_ptr() != views[name].data_ptr():
                views[name].copy_(p.grad)

    def zero_grad(self):
        self.grad.zero_()
        for p in self.module.parameters():
            p.grad = None

    def grad_norm(self):
        """Global L2 norm of the flat gradient (0-dim tensor on the device)."""
        self._sumsq.zero_()
        ops.call('gt_sumsq', self.grad, self.grad.numel(), self._sumsq)
        return self._sumsq

    def step(self, max_norm=None):
        self.gather_grads()
        g = self.param_groups[0]
        self.step_count += 1
        sumsq = None
        if max_norm is not None:
            sumsq = self.grad_norm()
        # the device's hand-off flag guards the update: a step whose in-launch waits gave up (engine.handoff_flag) leaves
        # parameters and moments untouched, with no host synchronisation here; engine.check_handoffs() raises at the
        # caller's next synchronisation point
        from .engine import handoff_flag
        ops.call('gt_adam_step_guarded', self.flat, self.grad, self.m, self.v, self.flat.numel(), float(g['lr']),
                 float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['weight_decay']), self.step_count,
                 sumsq, float(max_norm) if max_norm is not None else 0.0, handoff_flag(self.flat.device))
        return sumsq

    # ---- torch.optim.Adam-compatible checkpoint format
    def state_dict(self):
        state = {}
        for i, (name, o, n) in enumerate(self.module._flat_layout):
            shape = dict(self.module.named_parameters())[name].shape
            state[i] = dict(step=torch.tensor(float(self.step_count)), exp_avg=self.m[o:o + n].view(shape).clone(),
                            exp_avg_sq=self.v[o:o + n].view(shape).clone())
        group = dict(self.param_groups[0], params=list(range(len(self.module._flat_layout))))
        return dict(state=state if self.step_count > 0 else {}, param_groups=[group])

    def load_state_dict(self, sd):
        for i, (name, o, n) in enumerate(self.module._flat_layout):
            st = sd['state'].get(i)
            if st is None:
                continue
            self.m[o:o + n].copy_(st['exp_avg'].reshape(-1))
            self.v[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            self.step_count = int(st['step'])
        for k in ('lr', 'betas', 'eps', 'weight_decay'):
            if k in sd['param_groups'][0]:
                self.param_groups[0][k] = sd['param_groups'][0][k]
