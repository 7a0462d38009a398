"""Stand-alone timing of the fused LSTM step against the two-launch path at the headline shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gantron_amd import ops
from gantron_amd._lib import lib
from gantron_amd.engine import pack_gates, pack_rec

dev = torch.device('cuda:0')
B = Bp = 64
H = 1024
REPS = int(os.environ.get('REPS', 200))


def timeit(name, fn, flops):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / REPS
    print(f'{name:40s} {us:8.2f} us  {flops / us / 1e6:7.2f} TFLOP/s', flush=True)


for K in (2048, 3072):
    W = torch.randn(4 * H, K, device=dev) * 0.02
    ld = int(lib().gt_fused_k_padded(K)) + int(os.environ.get('LDPAD', 0))
    X = torch.zeros(Bp, ld, device=dev); X[:, :K] = torch.randn(Bp, K, device=dev)
    pre = torch.randn(Bp, 4 * H, device=dev); c_prev = torch.randn(Bp, H, device=dev)
    keep = (torch.rand(Bp, H, device=dev) > 0.1).to(torch.uint8)
    gates = torch.empty(Bp, 4 * H, device=dev); c_out = torch.empty(Bp, H, device=dev); h_raw = torch.empty(Bp, H, device=dev)
    h0 = torch.empty(Bp, 2048, device=dev); h1 = torch.empty(Bp, 3072, device=dev)
    Wg = pack_gates(W, H, K)
    Wp = pack_rec(W, 4 * H, 4 * H, K)
    part = torch.empty(4, Bp, 4 * H, device=dev)
    fl = 2.0 * Bp * 4 * H * K

    def two():
        ops.call('gt_rec_gemm', X, ld, Wp, part, Bp // 16, 4 * H, K, 4, Bp * 4 * H, 1, 0, 0, 0)
        ops.call('gt_lstm_cell_fwd', ops.slabs(part, 4, Bp * 4 * H, 4 * H), pre, 4 * H, c_prev, None, 0, keep, 0.1, None,
                 gates, c_out, h_raw, h0, 2048, h1, 3072, None, 0, B, H)
    timeit(f'rec_gemm + cell   K={K}', two, fl)
    timeit(f'fused step        K={K}', lambda: ops.call('gt_lstm_step_fused', X, ld, Wg, K, pre, 4 * H, c_prev, keep, 0.1, gates, c_out,
                                                        h_raw, h0, 2048, h1, 3072, None, 0, B, Bp, H), fl)
