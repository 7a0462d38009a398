"""Stand-alone throughput of gt_sgemm at the shapes of the training step (development tool)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gantron_amd import ops

dev = torch.device('cuda:0')
ops.BF16_GEMMS = bool(os.environ.get('BF16'))      # BF16=1: layer GEMMs with bf16 operands (hparams.fp16_run)
REPS = int(os.environ.get('REPS', 10))
if os.environ.get('SPLIT'):                         # SPLIT=3 / 2: fp32 products from that many bf16 planes (xgemm_kernel)
    ops.set_fp32_split(int(os.environ['SPLIT']))


def timeit(name, fn, flops):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / REPS
    print(f'{name:58s} {us:9.1f} us  {flops / us / 1e6:7.2f} TFLOP/s', flush=True)


B, To, Bp = 64, 870, 64
rows = B * To
if os.environ.get('PLANES'):     # PLANES=1: the GEMMs of the step on pre-split operands (pgemm_kernel), split time listed apart
    ops.set_fp32_split(3)
    P = ops.Planes
    dy = torch.randn(rows, 4096, device=dev); dyp = P(dy)
    timeit('gt_split3 of dG (rows x 4096): GB/s = (4 + 6) B/element', lambda: dyp.split(), 0.0)
    dyp.split()
    for N in (2048, 3072):
        x = torch.randn(rows, N, device=dev); xp = P(x).split(); out = torch.zeros(4096, N, device=dev)
        timeit(f'decoder dW dG^T.X  M=4096 N={N} K=rows  [planes]', lambda: ops.linear_dw(dy, x, rows, 4096, N, out, dy_planes=dyp, x_planes=xp), 2.0 * 4096 * N * rows)
        timeit(f'decoder dW dG^T.X  M=4096 N={N} K=rows  [in-kernel split]', lambda: ops.linear_dw(dy, x, rows, 4096, N, out), 2.0 * 4096 * N * rows)
    W = torch.randn(4096, 2048, device=dev) * 0.05; Wp = P(W).split()
    for blk in (16, 64):
        M = blk * 64
        xb = torch.randn(M, 2048, device=dev); xbp = P(xb).split()
        timeit(f'block GEMM fwd X2.W2ih^T M={M} N=4096 K=2048 [planes]', lambda: ops.linear(xb, W, M, 2048, 4096, x_planes=xbp, w_planes=Wp), 2.0 * M * 4096 * 2048)
        timeit(f'block GEMM fwd X2.W2ih^T M={M} N=4096 K=2048 [in-kernel]', lambda: ops.linear(xb, W, M, 2048, 4096), 2.0 * M * 4096 * 2048)
        gb = torch.randn(M, 4096, device=dev); gbp = P(gb).split()
        timeit(f'block GEMM bwd dG2.W2ih  M={M} N=2048 K=4096 [planes]', lambda: ops.linear_dx(gb, W, M, 4096, 2048, dy_planes=gbp, w_planes=Wp), 2.0 * M * 4096 * 2048)
        timeit(f'block GEMM bwd dG2.W2ih  M={M} N=2048 K=4096 [in-kernel]', lambda: ops.linear_dx(gb, W, M, 4096, 2048), 2.0 * M * 4096 * 2048)
    x = torch.randn(rows, 512, device=dev); Wc = torch.randn(512, 512, 5, device=dev) * 0.02
    Wf, Wb = ops.conv_weight_fwd(Wc), ops.conv_weight_bwd(Wc)
    xp, Wfp, Wbp = P(x).split(), P(Wf).split(), P(Wb).split()
    fl = 2.0 * rows * 512 * 512 * 5
    timeit('postnet conv fwd 512->512 k5 [planes]', lambda: ops.conv1d_fwd(x, Wf, None, B, To, 512, 512, 5, 1, x_planes=xp, w_planes=Wfp), fl)
    timeit('postnet conv fwd 512->512 k5 [in-kernel]', lambda: ops.conv1d_fwd(x, Wf, None, B, To, 512, 512, 5, 1), fl)
    timeit('postnet conv dx [planes]', lambda: ops.conv1d_dx(x, Wb, B, To, 512, 512, 5, 1, dy_planes=xp, w_planes=Wbp), fl)
    dW = torch.zeros(512, 512, 5, device=dev)
    timeit('postnet conv dW [planes]', lambda: ops.conv1d_dw(x, x, B, To, 512, 512, 5, 1, dW, dy_planes=xp, x_planes=xp), fl)
    timeit('postnet conv dW [in-kernel]', lambda: ops.conv1d_dw(x, x, B, To, 512, 512, 5, 1, dW), fl)
    sys.exit(0)
# linear: M x K times N x K
for name, M, K, N in (('pre1 (To*Bp x 256) . (4096 x 256)', rows, 256, 4096), ('proj dx (rows x 81) . W', rows, 81, 2048),
                      ('decoder dW dG^T.X  M=4096 N=2048 K=rows', 4096, rows, 2048), ('decoder dW M=4096 N=3072', 4096, rows, 3072)):
    if 'dW' in name:
        dy = torch.randn(rows, M, device=dev); x = torch.randn(rows, N, device=dev); out = torch.zeros(M, N, device=dev)
        timeit(name, lambda: ops.linear_dw(dy, x, rows, M, N, out), 2.0 * M * N * rows)
    else:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05
        timeit(name, lambda: ops.linear(x, w, M, K, N), 2.0 * M * N * K)
# postnet convs (channels-last): B x To positions, 512 -> 512, k = 5
x = torch.randn(rows, 512, device=dev); W = torch.randn(512, 512, 5, device=dev) * 0.02
Wf, Wb = ops.conv_weight_fwd(W), ops.conv_weight_bwd(W)
fl = 2.0 * rows * 512 * 512 * 5
timeit('postnet conv fwd 512->512 k5', lambda: ops.conv1d_fwd(x, Wf, None, B, To, 512, 512, 5, 1), fl)
timeit('postnet conv dx', lambda: ops.conv1d_dx(x, Wb, B, To, 512, 512, 5, 1), fl)
dW = torch.zeros(512, 512, 5, device=dev)
timeit('postnet conv dW', lambda: ops.conv1d_dw(x, x, B, To, 512, 512, 5, 1, dW), fl)
# discriminator first conv: (B*W positions) x (1600*5) -> 1024
Wn = 44
xd = torch.randn(B * Wn, 1600, device=dev); Wd = torch.randn(1024, 1600, 5, device=dev) * 0.01
Wdf = ops.conv_weight_fwd(Wd)
timeit('disc conv0 fwd 1600->1024 k5 (2816 positions)', lambda: ops.conv1d_fwd(xd, Wdf, None, B, Wn, 1600, 1024, 5, 1), 2.0 * B * Wn * 1600 * 1024 * 5)
# isolate: same shape as conv dW but plain (RC, RC) operands, and split-K sweep
dy = torch.randn(rows, 512, device=dev); xx = torch.randn(rows, 2560, device=dev); out = torch.zeros(512, 2560, device=dev)
for sk in (1, 2, 4, 8, 16):
    timeit(f'(RC,RC) M=512 N=2560 K=rows split_k={sk}', lambda: ops.sgemm(ops.Operand(dy, ops.RC, ld=512), ops.Operand(xx, ops.RC, ld=2560), out, 512, 2560, rows, accumulate=True, split_k=sk), 2.0 * 512 * 2560 * rows)
tmp = torch.zeros(512, 2560, device=dev)
for sk in (1, 2, 4, 8, 16):
    timeit(f'(RC,IM2COL_R) conv dW split_k={sk}', lambda: ops.sgemm(ops.Operand(x, ops.RC, ld=512), ops.Operand(x, ops.IM2COL_R, T=To, C=512, dil=1, pad=2), tmp, 512, 2560, rows, accumulate=True, split_k=sk), fl)
print('default_split_k', ops.default_split_k(512, 2560, rows))
dyd = torch.randn(rows, 4096, device=dev); xd2 = torch.randn(rows, 2048, device=dev); outd = torch.zeros(4096, 2048, device=dev)
for sk in (1, 2, 4):
    timeit(f'decoder dW M=4096 N=2048 K=rows split_k={sk}', lambda: ops.sgemm(ops.Operand(dyd, ops.RC, ld=4096), ops.Operand(xd2, ops.RC, ld=2048), outd, 4096, 2048, rows, accumulate=True, split_k=sk), 2.0 * 4096 * 2048 * rows)
# encoder conv dW: B*Ti = 64*187 positions
rows_e = 64 * 187
xe = torch.randn(rows_e, 512, device=dev); tmpe = torch.zeros(512, 2560, device=dev)
for sk in (1, 4, 8, 16, 32):
    timeit(f'encoder conv dW (K={rows_e}) split_k={sk}', lambda: ops.sgemm(ops.Operand(xe, ops.RC, ld=512), ops.Operand(xe, ops.IM2COL_R, T=187, C=512, dil=1, pad=2), tmpe, 512, 2560, rows_e, accumulate=True, split_k=sk), 2.0 * rows_e * 512 * 2560)
# prenet / small dW: M=256 N=256 K=rows ; projection dW M=81 N=2048
a1 = torch.randn(rows, 256, device=dev); o1 = torch.zeros(256, 256, device=dev)
for sk in (8, 32, 64):
    timeit(f'prenet dW M=256 N=256 K=rows split_k={sk}', lambda: ops.sgemm(ops.Operand(a1, ops.RC, ld=256), ops.Operand(a1, ops.RC, ld=256), o1, 256, 256, rows, accumulate=True, split_k=sk), 2.0 * 256 * 256 * rows)
print('defaults', ops.default_split_k(4096, 2048, rows), ops.default_split_k(512, 2560, rows_e), ops.default_split_k(256, 256, rows))
