"""Probe (round 4): the recurrent product X (64 x K) . W^T (N x K) of the decoder loops per weight format -- fp32 MFMA
(gt_rec_gemm), three bf16 planes (gt_rec_gemm_x3), two scaled fp16 planes (gt_rec_gemm_h2), bf16 operands (gt_rec_gemm_bf16) --
stand-alone: error against float64 and time per launch (back-to-back launches, events)."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gantron_amd import ops, engine
from gantron_amd._lib import lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
REPS = int(os.environ.get('REPS', 300))


def bench(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


for (N, K, xs, name) in ((1536, 4096, 1e-6, 'backward dG1 . W1 (N=1536, K=4096), gradients ~1e-6'),
                         (4096, 1536, 1.0, 'forward [ctx | h_att] . W1^T (N=4096, K=1536)')):
    M = 64
    W = (torch.randn(N, K) * 0.05).to(dev)
    X = (torch.randn(M, K) * xs).to(dev)
    ref = X.double().cpu() @ W.double().cpu().t()
    scale = float((X.double().abs().cpu() @ W.double().abs().cpu().t()).max())
    print(f'== {name}')
    for fmt in ('fp32', 'x3', 'h2', 'bf16'):
        if fmt == 'fp32':
            Wp = engine.pack_rec(W, N, N, K)
            S = engine.rec_splits(N, K)
        elif fmt == 'x3':
            Wp = engine.pack_rec_x3(W, N, N, K)
            S = engine.rec_splits_x3(N, K, int(os.environ.get('CUS', 256)))
        elif fmt == 'h2':
            Wp = torch.empty(int(lib().gt_rec_weight_h2_elems(N, K)), device=dev, dtype=torch.float16)
            ops.call('gt_pack_rec_weight_h2', W, K, 0, N, N, K, Wp)
            S = engine.rec_splits_x3(N, K, int(os.environ.get('CUS', 256)))
        else:
            Wp = engine.pack_rec_bf16(W, N, N, K)
            S = engine.rec_splits_bf16(N, K)
        for S_ in sorted({S} | ({int(s) for s in os.environ.get('SPLITS', '').split(',') if s})):
            part = torch.zeros(S_, M, N, device=dev)
            xmax = ops.absmax(X)
            st = torch.cuda.current_stream().cuda_stream
            if fmt == 'fp32':
                fn = lambda: ops.check(lib().gt_rec_gemm(ops.ptr(X), K, ops.ptr(Wp), ops.ptr(part), 4, N, K, S_, M * N, 1, 0, 0, 0, st), 'rec')
            elif fmt == 'x3':
                fn = lambda: ops.check(lib().gt_rec_gemm_x3(ops.ptr(X), K, ops.ptr(Wp), ops.ptr(part), 4, N, K, S_, M * N, 1, 0, 0, 0, st), 'rec')
            elif fmt == 'h2':
                fn = lambda: ops.check(lib().gt_rec_gemm_h2(ops.ptr(X), K, ops.ptr(Wp), ops.ptr(part), 4, N, K, S_, M * N, 1, 0, 0, ops.ptr(xmax), 0.0, st), 'rec')
            else:
                fn = lambda: ops.check(lib().gt_rec_gemm_bf16(ops.ptr(X), K, ops.ptr(Wp), ops.ptr(part), 4, N, K, S_, M * N, 1, 0, 0, 0, st), 'rec')
            try:
                us = bench(fn)
            except Exception as e:
                print(f'{fmt:5s} S={S_:2d}: {str(e)[:100]}')
                continue
            out = part.sum(0).double().cpu()
            err = float((out - ref).abs().max()) / scale
            print(f'{fmt:5s} S={S_:2d} ({(N // 64) * S_:3d} workgroups): {us:7.2f} us per launch   max |err| / max sum|x||w| = {err:.2e}', flush=True)
