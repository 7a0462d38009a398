"""Development aid: one Conv1d + BatchNorm(train) + ReLU + dropout layer, backward pieces against float64 torch."""
import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gantron_amd import engine, ops
from gantron_amd.ops import call
dev = torch.device('cuda:0')
B, T, C = int(os.environ.get('PB', 64)), int(os.environ.get('PT', 18)), 512
rows = B * T
g = torch.Generator().manual_seed(0)
y = torch.randn(rows, C, generator=g)              # conv output (BN input)
gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
keep = torch.rand(rows, C, generator=g) >= 0.5
dy = torch.randn(rows, C, generator=g)
yr = y.double().requires_grad_(True); gr = gamma.double().requires_grad_(True); br = beta.double().requires_grad_(True)
mean, var = yr.mean(0), yr.var(0, unbiased=False)
z = (yr - mean) / torch.sqrt(var + 1e-5) * gr + br
out = F.relu(z) * keep.double() * 2
gy, gg, gb = torch.autograd.grad(out, [yr, gr, br], dy.double())
dz_ref = dy.double() * keep.double() * 2 * (z > 0).double()
rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
yd, dyd, kd = y.to(dev), dy.to(dev), keep.to(torch.uint8).to(dev)
gd, bd = gamma.to(dev), beta.to(dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
stats = torch.empty(2 * C, device=dev); call('gt_col_stats', yd, yd, rows, C, stats, None)
mi = torch.empty(2 * C, device=dev); o = torch.empty(rows, C, device=dev)
call('gt_bn_act_drop_fwd', yd, stats, rows, C, gd, bd, rm, rv, 0.1, 1e-5, 1, kd, 0.5, o, mi, 1)
print(f'rows={rows}: fwd {rel(o, out.detach()):.2e}  mean {rel(mi[:C], mean.detach()):.2e}')
dz = torch.empty(rows, C, device=dev); sums = torch.empty(2 * C, device=dev)
call('gt_bn_act_drop_bwd1', dyd, yd, mi, gd, bd, rows, C, 1, kd, 0.5, dz, sums, None)
xhat = ((yr - mean) / torch.sqrt(var + 1e-5)).detach()
print(f'   dz {rel(dz, dz_ref.detach()):.2e}  sum dz {rel(sums[:C], dz_ref.sum(0).detach()):.2e}  sum dz*xhat {rel(sums[C:], (dz_ref * xhat).sum(0).detach()):.2e}')
dconv = torch.empty(rows, C, device=dev); Gg, Gb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
call('gt_bn_act_drop_bwd2', dz, yd, mi, gd, sums, rows, C, dconv, Gg, Gb)
print(f'   dconv {rel(dconv, gy):.2e}  dgamma {rel(Gg, gg):.2e}  dbeta {rel(Gb, gb):.2e}')
# conv pieces on these rows
W = torch.randn(C, C, 5, generator=g) * 0.03
x = torch.randn(B, T, C, generator=g)
xr = x.double().transpose(1, 2).requires_grad_(True); Wr = W.double().requires_grad_(True)
co = F.conv1d(xr, Wr, None, padding=2)
gx, gw = torch.autograd.grad(co, [xr, Wr], gy.view(B, T, C).transpose(1, 2))
Wd, xd = W.to(dev), x.view(rows, C).contiguous().to(dev)
cf = ops.conv1d_fwd(xd, ops.conv_weight_fwd(Wd), None, B, T, C, C, 5, 1)
dconv_exact = gy.float().contiguous().to(dev)
dx = ops.conv1d_dx(dconv_exact, ops.conv_weight_bwd(Wd), B, T, C, C, 5, 1)
dW = torch.zeros(C, C, 5, device=dev); ops.conv1d_dw(dconv_exact, xd, B, T, C, C, 5, 1, dW)
print(f'   conv fwd {rel(cf.view(B, T, C), co.detach().transpose(1, 2)):.2e}  conv dx {rel(dx.view(B, T, C), gx.transpose(1, 2)):.2e}  conv dW {rel(dW, gw):.2e}')
