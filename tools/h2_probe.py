"""Probe (round 4): the fp16-plane product on operands shaped like the decoder loops' block GEMMs ([h | ctx] rows with exact
zeros and values at +-1, uniform +-1/32 weights), every mode against float64."""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from gantron_amd import ops
dev = torch.device('cuda:0')
gen = torch.Generator().manual_seed(7)
def err(C, ref, scale): return float(((C.cpu().double() - ref).abs() / scale).max())
for (M, N, K, ld) in ((1024, 4096, 2048, 2048), (256, 4096, 2048, 2048), (1024, 4096, 2048, 2112), (16 * 16, 4096, 2048, 2048)):
    h = torch.tanh(torch.randn(M, K // 2, generator=gen) * 2.0)
    h = h * (torch.rand(M, K // 2, generator=gen) > 0.1) / 0.9            # dropout 0.1
    ctx = torch.tanh(torch.randn(M, K // 2, generator=gen))
    A = torch.cat((h, ctx), 1)
    A[::7, :5] = 1.0; A[3::11, 5:9] = -1.0
    Apad = torch.zeros(M, ld); Apad[:, :K] = A
    B = (torch.rand(N, K, generator=gen) * 2 - 1) / 32
    bias = (torch.rand(N, generator=gen) * 2 - 1) / 32
    ref = A.double() @ B.double().t() + bias.double()
    scale = A.double().abs() @ B.double().abs().t() + bias.double().abs()
    out = []
    for prec in (0, 3, 2, 4):
        with ops.fp32_split(prec):
            C = torch.zeros(M, N, device=dev)
            ops.sgemm(ops.Operand(Apad.to(dev), ops.KC, ld=ld), ops.Operand(B.to(dev), ops.KC, ld=K), C, M, N, K, bias=bias.to(dev))
        out.append(err(C, ref, scale))
    print(f'{M}x{N}x{K} ld {ld}: native {out[0]:.2e}  3 bf16 planes {out[1]:.2e}  2 bf16 planes {out[2]:.2e}  2 fp16 planes {out[3]:.2e}', flush=True)
