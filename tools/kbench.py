"""Micro-benchmark of the per-step decoder kernels at the headline shapes (B=64, Ti=187, E=1024),
through the C ABI, timed with events on torch's current stream.  Development tool, not a test."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gantron_amd import ops
from gantron_amd._lib import lib
from gantron_amd.engine import pack_rec

dev = torch.device('cuda:0')
B, Bp, Ti, E, A, F, Kc, Ha, Hd = 64, 64, 187, 1024, 128, 32, 31, 1024, 1024
REPS = int(os.environ.get('REPS', 100))


def timeit(name, fn, flops=0.0, bytes_=0.0):
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
    extra = ''
    if flops:
        extra += f'  {flops / us / 1e6:7.2f} TFLOP/s'
    if bytes_:
        extra += f'  {bytes_ / us / 1e3:7.1f} GB/s'
    print(f'{name:44s} {us:9.2f} us{extra}', flush=True)


def rec(name, N, K, S):
    W = torch.randn(N, K, device=dev) * 0.05
    Wp = pack_rec(W, N, N, K)
    ld = ops.roundup(K, 16)
    X = torch.randn(Bp, ld, device=dev)
    S = ops.pick_splits(K, S)
    part = torch.empty(S, Bp, N, device=dev)
    timeit(f'rec_gemm {name} N={N} K={K} S={S}', lambda: ops.call('gt_rec_gemm', X, ld, Wp, part, Bp // 16, N, K, S, Bp * N, 1, 0, 0, 0),
           flops=2.0 * Bp * N * K, bytes_=4.0 * (N * K + Bp * K + S * Bp * N))
    return part, S


def rec_bf16(name, N, K, S):
    from gantron_amd.engine import pack_rec_bf16
    W = torch.randn(N, K, device=dev) * 0.05
    Wp = pack_rec_bf16(W, N, N, K)
    ld = ops.roundup(K, 16)
    X = torch.randn(Bp, ld, device=dev)
    S = max(S, int(lib().gt_rec_bf16_min_splits(K)))
    kb = -(-K // 32)
    while -(-kb // -(-kb // S)) != S:
        S += 1
    part = torch.empty(S, Bp, N, device=dev)
    timeit(f'rec_gemm_bf16 {name} N={N} K={K} S={S}', lambda: ops.call('gt_rec_gemm_bf16', X, ld, Wp, part, Bp // 16, N, K, S, Bp * N, 1, 0, 0, 0),
           flops=2.0 * Bp * N * K, bytes_=2.0 * N * K + 4.0 * (Bp * K + S * Bp * N))


splits = [int(x) for x in os.environ.get('SPLITS', '4').split(',')]
if os.environ.get('BF16'):
    shapes = {'G1': (4096, 2048), 'G2': (4096, 3072), 'G2T': (3072, 4096), 'G1T': (2048, 4096)}
    for name, (N, K) in shapes.items():
        if os.environ.get('ONLY') and os.environ['ONLY'] != name:
            continue
        for S in splits:
            rec_bf16(name, N, K, S)
    sys.exit(0)
if os.environ.get('ONLY'):          # e.g. ONLY=G2 under rocprofv3: one shape, so that the stats file is per shape
    shapes = {'G1': (4096, 2048), 'G2': (4096, 3072), 'G2T': (3072, 4096), 'G1T': (2048, 4096)}
    for S in splits:
        rec(os.environ['ONLY'], *shapes[os.environ['ONLY']], S)
    sys.exit(0)
for S in splits:
    rec('G1 ', 4096, 2048, S)
    rec('G2 ', 4096, 3072, S)
    rec('G2T', 3072, 4096, S)
    rec('G1T', 2048, 4096, S)
rec('Gq ', 128, 1024, 16)
rec('GqT', 1024, 128, 2)

# cells
part, S = torch.randn(4, Bp, 4 * Ha, device=dev), 4
pre = torch.randn(Bp, 4 * Ha, device=dev); c_prev = torch.randn(Bp, Ha, device=dev)
keep = (torch.rand(Bp, Ha, device=dev) > 0.1).to(torch.uint8)
gates = torch.empty(Bp, 4 * Ha, device=dev); c_out = torch.empty(Bp, Ha, device=dev); h_raw = torch.empty(Bp, Ha, device=dev)
h0 = torch.empty(Bp, 2048, device=dev); h1 = torch.empty(Bp, 3072, device=dev)
timeit('lstm_cell_fwd', lambda: ops.call('gt_lstm_cell_fwd', ops.slabs(part, S, Bp * 4 * Ha, 4 * Ha), pre, 4 * Ha, c_prev, None, 0, keep, 0.1, None,
                                          gates, c_out, h_raw, h0, 2048, h1, 3072, None, 0, B, Ha), bytes_=4.0 * B * Ha * 26)
dgates = torch.empty(Bp, 4 * Ha, device=dev); dcp = torch.zeros(Bp, Ha, device=dev)
dsl = torch.randn(4, Bp, 3072, device=dev)
timeit('lstm_cell_bwd', lambda: ops.call('gt_lstm_cell_bwd', ops.slabs(dsl, 4, Bp * 3072, 3072), ops.NO_SLABS, ops.NO_SLABS, h_raw, Ha, dcp, gates, c_out,
                                          c_prev, keep, 0.1, None, dgates, dcp, None, B, Ha), bytes_=4.0 * B * Ha * 16)

# attention
d = ops.attn_dims(B, Ti, E, A, F, Kc)
qpart = torch.randn(16, Bp, A, device=dev)
w_prev = torch.softmax(torch.randn(B, Ti, device=dev), 1); w_cum = w_prev * 3
pm = torch.randn(B, Ti, A, device=dev); memory = torch.randn(B, Ti, E, device=dev)
Wc = torch.randn(F, 2, Kc, device=dev) * 0.2; Wd = torch.randn(A, F, device=dev) * 0.2; v = torch.randn(1, A, device=dev)
lens = torch.randint(60, Ti + 1, (B,), device=dev, dtype=torch.int32)
q_out = torch.empty(B, A, device=dev); en = torch.empty(B, Ti, device=dev)
timeit('attn_energies_fwd', lambda: ops.call('gt_attn_energies_fwd', d, ops.slabs(qpart, 16, Bp * A, A), w_prev, w_cum, pm, Wc, Wd, v, lens, q_out, en))
w_out = torch.empty(B, Ti, device=dev); cum_out = torch.empty(B, Ti, device=dev)
c0 = torch.empty(Bp, 2048, device=dev); c1 = torch.empty(Bp, 3072, device=dev); c2 = torch.empty(Bp, 2048, device=dev)
timeit('attn_context_fwd', lambda: ops.call('gt_attn_context_fwd', d, en, memory, E, lens, w_out, w_cum, cum_out, c0, 2048, c1, 3072, c2, 2048),
       bytes_=4.0 * B * Ti * E)
dsa = torch.randn(4, Bp, 3072, device=dev); dsb = torch.randn(4, Bp, 2048, device=dev); dhc = torch.randn(Bp, 2048, device=dev)
dwcat = torch.zeros(2, 2, B, Ti, device=dev); dcum = torch.zeros(B, Ti, device=dev); dctx_out = torch.empty(B, E, device=dev)
dq = torch.zeros(Bp, A, device=dev); dw = torch.empty(B, Ti, device=dev)
timeit('attn_context_bwd', lambda: ops.call('gt_attn_context_bwd', d, ops.slabs(dsa, 4, Bp * 3072, 3072, 1024), ops.slabs(dsb, 4, Bp * 2048, 2048),
                                             ops.ptr(dhc, 1024), 2048, memory, E, lens, dwcat[0], dcum, None, dctx_out, E, dq, A, dw), bytes_=4.0 * B * Ti * E)
dpm = torch.zeros(B, Ti, A, device=dev); acc = torch.zeros(int(lib().gt_attn_acc_elems(d)), device=dev)
timeit('attn_energies_bwd', lambda: ops.call('gt_attn_energies_bwd', d, dw, w_out, q_out, w_prev, w_cum, pm, Wc, Wd, v, lens, dq, A, dpm, dwcat[1], acc))
