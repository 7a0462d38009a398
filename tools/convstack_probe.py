"""Development aid: ConvBNStack (encoder convolutions) forward + backward on the GPU against float64 torch autograd."""
import os, sys, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gantron_amd import engine, ops
dev = torch.device('cuda:0')
B, T, C = int(os.environ.get('PB', 64)), int(os.environ.get('PT', 50)), 512
g = torch.Generator().manual_seed(0)
P, Pd = {}, {}
for i in range(3):
    P[f'encoder.convolutions.{i}.0.conv.weight'] = torch.randn(C, C, 5, generator=g) * 0.03
    P[f'encoder.convolutions.{i}.0.conv.bias'] = torch.randn(C, generator=g) * 0.1
    P[f'encoder.convolutions.{i}.1.weight'] = 1 + 0.1 * torch.randn(C, generator=g)
    P[f'encoder.convolutions.{i}.1.bias'] = 0.1 * torch.randn(C, generator=g)
    P[f'encoder.convolutions.{i}.1.running_mean'] = torch.zeros(C)
    P[f'encoder.convolutions.{i}.1.running_var'] = torch.ones(C)
x = torch.randn(B, T, C, generator=g)
keeps = [(torch.rand(B, T, C, generator=g) >= 0.5) for _ in range(3)]
dy = torch.randn(B, T, C, generator=g) * (torch.rand(B, T, 1, generator=g) > 0.3)       # some all-zero rows, like padding
# reference (float64)
Pr = {k: v.double().requires_grad_(k.endswith(('weight', 'bias'))) for k, v in P.items()}
h = x.double().transpose(1, 2).requires_grad_(True)
xin = h
zref = []
for i in range(3):
    h = F.conv1d(h, Pr[f'encoder.convolutions.{i}.0.conv.weight'], Pr[f'encoder.convolutions.{i}.0.conv.bias'], padding=2)
    h = F.batch_norm(h, None, None, Pr[f'encoder.convolutions.{i}.1.weight'], Pr[f'encoder.convolutions.{i}.1.bias'], True, 0.1, 1e-5)
    zref.append(h.detach().transpose(1, 2).reshape(B * T, C))
    h = F.relu(h) * keeps[i].transpose(1, 2).double() * 2.0
names = [k for k in Pr if Pr[k].requires_grad]
grads = torch.autograd.grad(h, [xin] + [Pr[k] for k in names], dy.double().transpose(1, 2))
# HIP
Pg = {k: v.to(dev) for k, v in P.items()}
G = {k: torch.zeros_like(v) for k, v in Pg.items()}
st = engine.ConvBNStack('encoder.convolutions', 3, [engine.ACT_RELU] * 3)
save = []
out = st.forward(Pg, x.view(B * T, C).contiguous().to(dev), B, T, [k.to(torch.uint8).contiguous().to(dev) for k in keeps], True, save)
dx = st.backward(Pg, G, dy.view(B * T, C).contiguous().to(dev), B, T, save)
engine.LEAF.join()
torch.cuda.synchronize()
rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
print(f'B={B} T={T}  out {rel(out.view(B, T, C), h.detach().transpose(1, 2)):.2e}  dx {rel(dx.view(B, T, C), grads[0].transpose(1, 2)):.2e}')
for k, gr in zip(names, grads[1:]):
    print(f'   {k}: {rel(G[k], gr):.2e}  (scale {float(gr.abs().max()):.2e})')

for i in range(3):
    yv, mi = save[i]['y'], save[i]['mi']
    z = (yv - mi[:C]) * mi[C:] * Pg[f'encoder.convolutions.{i}.1.weight'] + Pg[f'encoder.convolutions.{i}.1.bias']
    zr = zref[i].to(dev)
    flips = ((z > 0) != (zr > 0))
    print(f'layer {i}: {int(flips.sum())} ReLU sign flips between the fp32 HIP pre-activations and the fp64 reference'
          + (f' (|z| there: {zr[flips].abs().max().item():.2e}; kept by dropout: {int((flips & save[i]["keep"].bool().reshape(flips.shape)).sum())})' if flips.any() else ''))
