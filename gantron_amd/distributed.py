"""Data-parallel gradient averaging over RCCL/xGMI (reference distributed.py:126-173).

`apply_gradient_allreduce(module)` keeps the reference's contract -- rank 0's state is broadcast
at wrap time and after every backward each parameter's .grad holds the mean over ranks -- but the
exchange is re-designed for one-process-per-GPU MI355X nodes:

* parameters and gradients live in ONE flat fp32 buffer each (optim.flatten_module), so the
  reference's per-tensor broadcasts (one NCCL call per state tensor) become one broadcast and its
  flatten -> all_reduce -> unflatten copies (torch.cat of 130 MB per backward) disappear: the flat
  gradient buffer IS the all-reduce payload (130.0 MB generator / 49.3 MB discriminator);
* the 1/world scaling runs as a HIP kernel on the same stream, right behind the collective.

With xGMI's point-to-point links a ring all-reduce of 130 MB is bound by one link (~1.5 ms);
that is below 1 % of a generator step, so the collective is issued once per backward, as in the
reference, without bucketing.  The backend string comes from hparams.dist_backend ("nccl" is RCCL
on ROCm; "gloo" is used by the CPU tests of the wiring).
"""
import os

import torch
import torch.distributed as dist
from torch.autograd import Variable

from . import switches as sw
from .optim import flatten_module


def _tick(on):
    if not on:
        return 0.0
    import time
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.perf_counter()


def _scale(flat, world):
    if flat.is_cuda:
        from . import ops
        ops.call('gt_scale_dev', flat, None, 1.0 / world, flat, flat.numel(), 0)
    else:
        flat.div_(world)          # gloo wiring tests on the CPU


def _average_flat(module):
    """SUM all-reduce of the flat gradient buffer, then 1/world.  Ranges that a bucket already put on the wire while
    the backward pass was still running (module._early_buckets) are only waited for."""
    flat = module._flat_grad
    early = getattr(module, '_early_buckets', None) or []
    module._early_buckets = []
    covered = sorted((b, e) for b, e, _ in early)
    pos, rest = 0, []
    for b, e in covered:
        if b > pos:
            rest.append((pos, b))
        pos = max(pos, e)
    if pos < flat.numel():
        rest.append((pos, flat.numel()))
    for b, e in rest:
        dist.all_reduce(flat[b:e])
    for _, _, work in early:
        work.wait()                # the current stream waits for the side-stream collective
    _scale(flat, dist.get_world_size())
    if flat.is_cuda:
        # one rank whose in-launch hand-offs gave up (engine.handoff_flag) has just poisoned the average: every rank must
        # skip this optimizer step (gt_adam_step_guarded reads the flag), or the replicas drift apart
        from .engine import handoff_flag
        dist.all_reduce(handoff_flag(flat.device), op=dist.ReduceOp.MAX)


def apply_gradient_allreduce(module):
    flat, _ = flatten_module(module)
    dist.broadcast(flat, 0)
    for buf in module.buffers():
        dist.broadcast(buf, 0)

    timing = bool(sw.debug('dist_timing'))      # diagnostics: synchronising per-phase timer
    overlap = sw.schedule_on('dist_overlap')
    state = {'queued': False, 'comm': None}

    def allreduce_params():
        state['queued'] = False
        if not getattr(module, 'needs_reduction', False):
            return
        module.needs_reduction = False
        views = module._grad_views
        t0 = _tick(timing)
        for name, p in module.named_parameters():
            if p.grad is None:
                if not getattr(module, '_direct_grads', False):
                    views[name].zero_()
            elif p.grad.data_ptr() != views[name].data_ptr():     # autograd handed back its own tensor
                if getattr(module, '_direct_grads', False):
                    views[name].add_(p.grad)                      # the buffer already holds the engine's share
                else:
                    views[name].copy_(p.grad)
                p.grad = views[name]
            # else: the gradient was accumulated in place in the flat buffer already
        t1 = _tick(timing)
        _average_flat(module)
        t2 = _tick(timing)
        if timing:
            print(f'[dist] gather {1e3 * (t1 - t0):.2f} ms  all-reduce + scale {1e3 * (t2 - t1):.2f} ms '
                  f'({module._flat_grad.numel() * 4 / 1e6:.1f} MB)', flush=True)

    def allreduce_hook(*unused):
        if not state['queued']:        # one end-of-backward callback per backward pass (distributed.py:163-167)
            state['queued'] = True
            Variable._execution_engine.queue_callback(allreduce_params)

    for p in module.parameters():
        if p.requires_grad:
            p.register_hook(allreduce_hook)
    # model.py's autograd nodes write gradients straight into the flat buffer (optim.enable_direct_grads): autograd
    # never sees them, so no parameter hook fires -- the nodes call this instead
    module._after_backward = allreduce_hook

    def bucket_ready(prefix, event):
        """engine.GeneratorEngine.backward: the gradients under `prefix` are final once `event` has happened.  Their
        slice of the flat buffer goes on the wire now, from a side stream, while the decoder loop is still running."""
        if not (overlap and getattr(module, '_direct_grads', False) and module._flat_grad.is_cuda):
            return
        from .optim import flat_range
        rng = flat_range(module, prefix)
        if rng is None:
            return
        if state['comm'] is None:
            state['comm'] = torch.cuda.Stream()
        comm = state['comm']
        comm.wait_event(event)
        with torch.cuda.stream(comm):
            work = dist.all_reduce(module._flat_grad[rng[0]:rng[1]], async_op=True)
        module.__dict__.setdefault('_early_buckets', []).append((rng[0], rng[1], work))
    module._bucket_ready = bucket_ready

    def drain_stale():
        """A backward pass that raised after the hook queued its callback, or after a bucket went on the wire, must not
        leave the NEXT step without its all-reduce: every forward starts from a clean state."""
        state['queued'] = False
        for _, _, work in getattr(module, '_early_buckets', None) or []:
            work.wait()
        module._early_buckets = []

    def set_needs_reduction(mod, inputs, output):
        drain_stale()
        mod.needs_reduction = True

    module.register_forward_hook(set_needs_reduction)
    # The reference's Discriminator is only ever entered through adversarial_loss(), which bypasses
    # forward hooks (distributed.py:169-172 never fires for it); arm the reduction there as well.
    if hasattr(module, 'adversarial_loss'):
        inner = module.adversarial_loss

        def adversarial_loss(*a, **k):
            drain_stale()
            module.needs_reduction = True
            return inner(*a, **k)
        module.adversarial_loss = adversarial_loss
    return module


def reduce_tensor(tensor, n_gpus):
    """reference train.py:72-76"""
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= n_gpus
    return rt


def init_distributed(hparams, n_gpus, rank, group_name=None):
    """reference train.py:79-91; rendezvous from hparams.dist_url (tcp://host:port) or the
    torchrun environment (MASTER_ADDR/MASTER_PORT)."""
    if torch.cuda.is_available():
        torch.cuda.set_device(rank % torch.cuda.device_count())
    if dist.is_initialized():
        return
    import os
    init_method = 'env://' if 'MASTER_ADDR' in os.environ else hparams.dist_url
    dist.init_process_group(backend=hparams.dist_backend, init_method=init_method, world_size=n_gpus, rank=rank)
