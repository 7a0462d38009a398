"""Stand-alone timing of gt_plstm_fwd at the encoder's and the decoder LSTM's shapes (us per step), next to the per-step
launches it replaces (gt_bilstm_run's fused product + cell steps)."""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gantron_amd import engine
from gantron_amd._lib import GtPlstm, check, lib
from gantron_amd.ops import ptr

dev = torch.device('cuda:0')


def bench(H, ndir, T, blk, ld):
    Bp, N = 64, 4 * H
    W = (torch.randn(ndir, N, H, device=dev) / math.sqrt(H)).contiguous()
    pre = torch.randn(ndir, T, Bp, N, device=dev)
    gates = torch.empty_like(pre)
    c = torch.zeros(ndir, T + 1, Bp, H, device=dev)
    ring = 2 if ld == H else 0
    hx = torch.zeros(ndir, ring if ring else T, Bp, ld, device=dev)
    cnt = torch.zeros(ndir, dtype=torch.int32, device=dev)
    q = GtPlstm()
    q.B, q.Bp, q.H, q.T, q.ndir, q.hx_ring = 64, Bp, H, T, ndir, ring
    q.Whh, q.w_dstride, q.pre, q.pre_dstride = ptr(W), W.stride(0), ptr(pre), pre.stride(0)
    q.gates, q.gates_dstride, q.c, q.c_dstride = ptr(gates), gates.stride(0), ptr(c), c.stride(0)
    q.hx, q.hx_dstride, q.hx_step, q.hx_ld = ptr(hx), hx.stride(0), hx.stride(1), ld
    q.counters, q.timeout_flag = ptr(cnt), ptr(engine.handoff_flag(dev))

    def run():
        for s in range(0, T, blk):
            q.s_lo, q.s_hi = s, min(T, s + blk)
            check(lib().gt_plstm_fwd(ctypes.byref(q), torch.cuda.current_stream().cuda_stream), 'gt_plstm_fwd')
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    print(f'H={H} ndir={ndir} T={T} steps per launch {blk}: {us:9.1f} us total, {us / T:6.2f} us per step', flush=True)
    engine.check_handoffs()


bench(256, 2, 187, 187, 256)
bench(1024, 1, 160, 16, 2048)
bench(1024, 1, 160, 160, 2048)
