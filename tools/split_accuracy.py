import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from gantron_amd import ops
import torch.nn.functional as F
dev = torch.device('cuda:0')
gen = torch.Generator().manual_seed(5)
def rnd(*s): return torch.randn(*s, generator=gen) * torch.exp(torch.randn(*s, generator=gen) * 2.0)
def err(C, ref, scale): return float(((C.cpu().double() - ref).abs() / scale).max())
for (M, N, K) in ((384, 512, 4096), (96, 130, 2816)):
    A, B = rnd(M, K), rnd(N, K)
    ref = A.double() @ B.double().t(); scale = A.double().abs() @ B.double().abs().t()
    for name, oa, ob in (('kc/kc', ops.Operand(A.to(dev), ops.KC, ld=K), ops.Operand(B.to(dev), ops.KC, ld=K)),
                         ('rc/rc', ops.Operand(A.t().contiguous().to(dev), ops.RC, ld=M), ops.Operand(B.t().contiguous().to(dev), ops.RC, ld=N)),
                         ('kc/rc', ops.Operand(A.to(dev), ops.KC, ld=K), ops.Operand(B.t().contiguous().to(dev), ops.RC, ld=N)),
                         ('rc/kc', ops.Operand(A.t().contiguous().to(dev), ops.RC, ld=M), ops.Operand(B.to(dev), ops.KC, ld=K))):
        out = []
        for prec in (0, 3, 2, 4):
            with ops.fp32_split(prec):
                C = torch.zeros(M, N, device=dev)
                ops.sgemm(oa, ob, C, M, N, K)
            out.append(err(C, ref, scale))
        print(f'{M}x{N}x{K} {name}: native {out[0]:.2e}  3 planes {out[1]:.2e}  2 planes {out[2]:.2e}  2 fp16 planes {out[3]:.2e}', flush=True)
# conv
for (Bn, T, Cin, Cout, Kk) in ((4, 150, 80, 512, 5), (64, 44, 64, 128, 5)):
    x, W, dy = rnd(Bn, Cin, T), rnd(Cout, Cin, Kk) * 0.2, rnd(Bn, Cout, T)
    xr, Wr = x.double().requires_grad_(True), W.double().requires_grad_(True)
    y = F.conv1d(xr, Wr, None, padding=2)
    gx, gw = torch.autograd.grad(y, (xr, Wr), dy.double())
    ya = F.conv1d(x.double().abs(), W.double().abs(), None, padding=2)
    xa, Wa = x.double().abs().requires_grad_(True), W.double().abs().requires_grad_(True)
    gxa, gwa = torch.autograd.grad(F.conv1d(xa, Wa, None, padding=2), (xa, Wa), dy.double().abs())
    xcl = x.permute(0, 2, 1).contiguous().view(Bn * T, Cin).to(dev); Wd = W.to(dev)
    dycl = dy.permute(0, 2, 1).contiguous().view(Bn * T, Cout).to(dev)
    res = {}
    for prec in (0, 3, 2, 4):
        with ops.fp32_split(prec):
            ycl = ops.conv1d_fwd(xcl, ops.conv_weight_fwd(Wd), None, Bn, T, Cin, Cout, Kk, 1)
            dx = ops.conv1d_dx(dycl, ops.conv_weight_bwd(Wd), Bn, T, Cin, Cout, Kk, 1)
            dW = torch.zeros(Cout, Cin, Kk, device=dev); ops.conv1d_dw(dycl, xcl, Bn, T, Cin, Cout, Kk, 1, dW)
        res[prec] = (err(ycl.view(Bn, T, Cout).permute(0, 2, 1), y.detach(), ya), err(dx.view(Bn, T, Cin).permute(0, 2, 1), gx, gxa), err(dW, gw, gwa))
    for i, nm in enumerate(('conv fwd', 'conv dx', 'conv dW')):
        print(f'B{Bn} T{T} {Cin}->{Cout} {nm}: native {res[0][i]:.2e}  3 planes {res[3][i]:.2e}  2 planes {res[2][i]:.2e}  2 fp16 planes {res[4][i]:.2e}', flush=True)
