"""Per-kernel checks of libgantron_hip.so on a real MI355X, each against an independent fp32/fp64
CPU computation (torch CPU ops or the oracle).  All calls go through the C ABI."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from harness import close, u01

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def g(*shape, seed=0, scale=1.0):
    gen = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=gen) * scale


# ------------------------------------------------------------------------------------- sgemm
def rb(t, bf16):
    """operand as the bf16 variant of gt_sgemm sees it (rounded to nearest even), in fp64"""
    return t.to(torch.bfloat16).double() if bf16 is True else t.double()


# GtGemm.bf16: False = fp32 MFMA, True = operands rounded to bf16, 3 / 2 = fp32 product from three / two bf16 planes per
# operand (xgemm_kernel), 4 = from two scaled fp16 planes (22 significand bits per operand).  Three bf16 planes and two fp16
# planes hold to the fp32 tolerances; two bf16 planes drop terms of 2^-16 |a||b|.
PRECISIONS = [False, True, 3, 2, 4]


def native_if_false(bf16):
    """`False` means the fp32 matrix pipe in these tests, whatever the process default (GANTRON_FP32_SPLIT) is."""
    import contextlib
    from gantron_amd import ops
    return ops.fp32_split(0) if bf16 is False else contextlib.nullcontext()


def tol(bf16, atol):
    return dict(rtol=1e-5, atol=atol) if bf16 != 2 else dict(rtol=1e-4, atol=100 * atol)


@pytest.mark.parametrize('bf16', PRECISIONS)
@pytest.mark.parametrize('M,N,K', [(200, 150, 70), (64, 64, 16), (129, 257, 333), (1, 80, 2048), (515, 1, 33), (300, 260, 1000),
                                   (2000, 1536, 200)])     # the last one runs on 128x128 tiles (>= 192 of them)
def test_sgemm_kc_kc(dev, M, N, K, bf16):
    from gantron_amd import ops
    A, B, bias = g(M, K), g(N, K, seed=1), g(N, seed=2)
    C = torch.empty(M, N, device=dev)
    with native_if_false(bf16):
        ops.sgemm(ops.Operand(A.to(dev), ops.KC, ld=K), ops.Operand(B.to(dev), ops.KC, ld=K), C, M, N, K, bias=bias.to(dev),
                  bf16=bf16)
    close(C, rb(A, bf16) @ rb(B, bf16).t() + bias.double(), what='sgemm kc/kc', **tol(bf16, 1e-5))


def test_sgemm_split_planes_accuracy(dev):
    """The three-plane product is as accurate as the fp32 matrix pipe (errors against fp64 within a factor of two of each
    other, over five decades of operand magnitude inside one reduction); two planes sit at the 2^-16 level."""
    from gantron_amd import ops
    M, N, K = 384, 512, 4096
    gen = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=gen) * torch.exp(torch.randn(M, K, generator=gen) * 2.5)
    B = torch.randn(N, K, generator=gen) * torch.exp(torch.randn(N, K, generator=gen) * 2.5)
    ref = A.double() @ B.double().t()
    scale = (A.double().abs() @ B.double().abs().t())           # sum |a||b|: what a rounding error is relative to
    err = {}
    for prec in (False, 3, 2, 4):
        C = torch.empty(M, N, device=dev)
        with native_if_false(prec):
            ops.sgemm(ops.Operand(A.to(dev), ops.KC, ld=K), ops.Operand(B.to(dev), ops.KC, ld=K), C, M, N, K, bf16=prec)
        err[prec] = float(((C.cpu().double() - ref).abs() / scale).max())
    print('max |error| / sum |a||b| against float64:', err)
    # fp32 accumulation over K = 4096 terms leaves a few 1e-6 of sum |a||b| on either pipe
    assert err[False] < 1e-5 and err[3] < 1e-5 and err[4] < 1e-5, err
    assert err[3] < 1.5 * err[False], err
    assert err[4] < 1.5 * err[False], err                       # two fp16 planes: the same bar as three bf16 planes
    assert err[2] < 4e-5, err                                   # 2^-16 per product


def test_sgemm_fp16_planes_scales_and_given_maxima(dev):
    """Mode 4 (two scaled fp16 planes): operands far outside fp16's range in both directions (gradient-sized 1e-9 against
    activations of 1e+3, the scales leave exactly), the caller-supplied maxima (gt_absmax -> GtOperand.amax: same result as the
    product's own pass), an all-zero operand, and a NaN that poisons its own output row only."""
    from gantron_amd import ops
    M, N, K = 200, 300, 1000
    for sa, sb in ((1e-9, 1e3), (3e5, 2e-7), (1.0, 1.0)):
        A, B = g(M, K) * sa, g(N, K, seed=1) * sb
        ref = A.double() @ B.double().t()
        Ad, Bd = A.to(dev), B.to(dev)
        C = torch.empty(M, N, device=dev)
        ops.sgemm(ops.Operand(Ad, ops.KC, ld=K), ops.Operand(Bd, ops.KC, ld=K), C, M, N, K, bf16=4)
        assert int(ops.lib().gt_sgemm_last_kernel()) == 5
        close(C, ref, rtol=1e-5, atol=0.0, what=f'scaled operands {sa:g} x {sb:g}')
        C2 = torch.empty(M, N, device=dev)
        ops.sgemm(ops.Operand(Ad, ops.KC, ld=K, amax=ops.absmax(Ad)), ops.Operand(Bd, ops.KC, ld=K, amax=ops.absmax(Bd, N, K, K)),
                  C2, M, N, K, bf16=4)
        assert torch.equal(C, C2), 'given maxima must reproduce the product\'s own pass bit for bit'
    Z = torch.zeros(M, K, device=dev)
    C = torch.full((M, N), 7.0, device=dev)
    ops.sgemm(ops.Operand(Z, ops.KC, ld=K), ops.Operand(Bd, ops.KC, ld=K), C, M, N, K, bf16=4)
    assert float(C.abs().max()) == 0.0
    An = A.clone()
    An[5, 17] = float('nan')
    C = torch.empty(M, N, device=dev)
    ops.sgemm(ops.Operand(An.to(dev), ops.KC, ld=K), ops.Operand(Bd, ops.KC, ld=K), C, M, N, K, bf16=4)
    assert bool(torch.isnan(C[5]).all()) and bool(torch.isfinite(C[torch.arange(M) != 5]).all())
    close(C[torch.arange(M) != 5], ref[torch.arange(M) != 5], rtol=1e-5, atol=0.0, what='rows beside the NaN row')


_WRAP_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from gantron_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
M = N = K = 64
def product(scale):
    A, B = (torch.randn(M, K) * scale).to(dev), torch.randn(N, K).to(dev)
    C = torch.empty(M, N, device=dev)
    ops.sgemm(ops.Operand(A, ops.KC, ld=K), ops.Operand(B, ops.KC, ld=K), C, M, N, K, bf16=4)
    ref = A.double() @ B.double().t()
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    assert torch.isfinite(C).all() and err < 1e-5, (scale, err)
# the tags run out after four calls (GANTRON_DEBUG=amax_wrap=4): slots 0..3 keep words with the largest tags there are
for _ in range(4):
    product(1e-3)
A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
C = torch.empty(M, N, device=dev)
for _ in range(8192 - 4):           # once round the ring
    ops.sgemm(ops.Operand(A, ops.KC, ld=K), ops.Operand(B, ops.KC, ld=K), C, M, N, K, bf16=4)
# slots 0..3 again, with small tags and operands 1e9 times larger: a stale maximum would overflow the fp16 planes
for _ in range(4):
    product(1e6)
torch.cuda.synchronize()
print('WRAP-OK')
"""


def test_sgemm_fp16_planes_survive_the_wrap_of_their_tags(tmp_path):
    """The maxima of mode 4 meet in words tagged by a 32-bit call counter; when it runs out (days of training) the ring is
    drained and cleared once.  GANTRON_DEBUG=amax_wrap=4 puts the wrap four calls into a fresh process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'wrap.py'
    script.write_text(_WRAP_SCRIPT)
    env = dict(os.environ, GANTRON_DEBUG='amax_wrap=4')
    r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'WRAP-OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize('bf16', PRECISIONS)
def test_sgemm_modes_accumulate_splitk_batch(dev, bf16):
    with native_if_false(bf16):
        _sgemm_modes(dev, bf16)


def _sgemm_modes(dev, bf16):
    from gantron_amd import ops
    import functools
    M, N, K = 96, 130, 700
    A, B = g(M, K), g(N, K, seed=1)
    ref = rb(A, bf16) @ rb(B, bf16).t()
    ops = type('O', (), {k: getattr(ops, k) for k in ('Operand', 'KC', 'RC')} | {'sgemm': functools.partial(ops.sgemm, bf16=bf16)})
    # RC/RC: both operands stored reduction-major
    C = torch.zeros(M, N, device=dev)
    ops.sgemm(ops.Operand(A.t().contiguous().to(dev), ops.RC, ld=M), ops.Operand(B.t().contiguous().to(dev), ops.RC, ld=N),
              C, M, N, K)
    close(C, ref, **tol(bf16, 1e-4), what='rc/rc')
    # KC/RC and RC/KC
    C.zero_()
    ops.sgemm(ops.Operand(A.to(dev), ops.KC, ld=K), ops.Operand(B.t().contiguous().to(dev), ops.RC, ld=N), C, M, N, K)
    close(C, ref, **tol(bf16, 1e-4), what='kc/rc')
    C.zero_()
    ops.sgemm(ops.Operand(A.t().contiguous().to(dev), ops.RC, ld=M), ops.Operand(B.to(dev), ops.KC, ld=K), C, M, N, K)
    close(C, ref, **tol(bf16, 1e-4), what='rc/kc')
    # accumulate + split-K on top of existing content, unaligned leading dimension (scalar path)
    C0 = g(M, N + 3, seed=5)
    Cd = C0.clone().to(dev)
    ops.sgemm(ops.Operand(A.to(dev), ops.KC, ld=K), ops.Operand(B.to(dev), ops.KC, ld=K), Cd, M, N, K, ldc=N + 3,
              accumulate=True, split_k=5, alpha=0.5)
    want = C0.double()
    want[:, :N] += 0.5 * ref
    close(Cd, want, **tol(bf16, 1e-4), what='accumulate split-k')
    # batched, operands at an element offset
    nb = 3
    Ab, Bb = g(nb, M, K, seed=7), g(nb, N, K, seed=8)
    Cb = torch.empty(nb, M, N, device=dev)
    Apad = torch.cat((torch.zeros(5), Ab.reshape(-1))).to(dev)
    ops.sgemm(ops.Operand(Apad, ops.KC, ld=K, offset=5, batch_stride=M * K),
              ops.Operand(Bb.to(dev), ops.KC, ld=K, batch_stride=N * K), Cb, M, N, K, batch=nb, c_batch_stride=M * N)
    close(Cb, torch.bmm(rb(Ab, bf16), rb(Bb, bf16).transpose(1, 2)), **tol(bf16, 1e-4), what='batched')
    # 128x128 tiles (>= 192 of them), reduction-major operands: the shape class of the weight-gradient GEMMs
    M2, N2, K2 = 1536, 2000, 150
    A2, B2 = g(K2, M2, seed=11), g(K2, N2, seed=12)
    C2 = torch.zeros(M2, N2, device=dev)
    ops.sgemm(ops.Operand(A2.to(dev), ops.RC, ld=M2), ops.Operand(B2.to(dev), ops.RC, ld=N2), C2, M2, N2, K2)
    close(C2, rb(A2, bf16).t() @ rb(B2, bf16), **tol(bf16, 1e-4), what='rc/rc 128x128 tiles')


def test_split3_planes_are_exact(dev):
    """gt_split3: x0 + x1 + x2 == x bit for bit (three 8-bit significands hold fp32's 24), over ten decades of magnitude,
    zeros and denormal-range residuals included."""
    from gantron_amd import ops
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(8 * 5000, generator=gen) * torch.exp(torch.randn(8 * 5000, generator=gen) * 6)
    x[::97] = 0
    P = ops.Planes(x.to(dev)).split()
    back = P.t.float().sum(0)                    # exact: each partial sum is representable
    assert torch.equal(back.cpu(), x)
    assert float((P.t[1].float().abs() > P.t[0].float().abs() * 2 ** -7).sum()) == 0       # plane 1 below plane 0's last bit


@pytest.mark.parametrize('modes', ['kc/kc', 'kc/rc', 'rc/rc', 'rc/kc'])
@pytest.mark.parametrize('M,N,K,sk', [(256, 192, 64, 1), (1536, 2048, 160, 1), (96, 136, 704, 5), (1024, 2048, 4096, 1),
                                      (2048, 2048, 160, 1), (1000, 1016, 704, 4)])      # the last two: 256 x 128 tiles (pgemm2)
def test_plane_gemm_matches_float64(dev, modes, M, N, K, sk):
    """pgemm_kernel (operands pre-split by gt_split3, tiles moved by LDS-DMA, swizzled images): every operand mode pair,
    64x64 and 128x128 tiles, ragged M / N (rows beyond the matrix re-read the last row), split-K with accumulation --
    against float64, at the tolerance the fp32 pipe is held to; and the kernel REALLY ran (the in-kernel-split result of
    the same call differs in summation order only, so equality of the two up to rounding is checked as well)."""
    from gantron_amd import ops
    am, bm = modes.split('/')
    A, B = g(M, K, seed=3), g(N, K, seed=4)
    ref = A.double() @ B.double().t()
    Ad = (A if am == 'kc' else A.t().contiguous()).to(dev)
    Bd = (B if bm == 'kc' else B.t().contiguous()).to(dev)

    def run(planes):
        oa = ops.Operand(Ad, ops.KC if am == 'kc' else ops.RC, ld=K if am == 'kc' else M, planes=ops.Planes(Ad).split() if planes else None)
        ob = ops.Operand(Bd, ops.KC if bm == 'kc' else ops.RC, ld=K if bm == 'kc' else N, planes=ops.Planes(Bd).split() if planes else None)
        C = torch.full((M, N), 0.5, device=dev)
        ops.sgemm(oa, ob, C, M, N, K, accumulate=True, split_k=sk, bf16=3)
        big = -(-M // 256) * -(-N // 128) * sk >= 128 and M >= 256 and N >= 128
        assert int(ops.lib().gt_sgemm_last_kernel()) == ((4 if big else 3) if planes else 2)
        return C.cpu().double() - 0.5
    got, inkernel = run(True), run(False)
    close(got, ref, rtol=1e-5, atol=1e-4, what=f'plane gemm {modes}')
    close(got, inkernel, rtol=2e-6, atol=2e-5, what=f'plane gemm vs in-kernel split {modes}')


@pytest.mark.parametrize('B,T,Cin,Cout,K,dil', [(3, 37, 32, 64, 5, 1), (2, 9, 64, 96, 5, 2), (4, 160, 512, 512, 5, 1), (8, 1024, 64, 512, 5, 1)])
def test_plane_gemm_convolutions(dev, B, T, Cin, Cout, K, dil):
    """The im2col operand modes of pgemm_kernel (taps outside [0, T) read zeros): Conv1d forward, input gradient and weight
    gradient with pre-split activations / weights against float64."""
    from gantron_amd import ops
    x, W = g(B, Cin, T), g(Cout, Cin, K, seed=1, scale=0.2)
    dy = g(B, Cout, T, seed=3)
    pad = dil * (K - 1) // 2
    xr, Wr = x.double().requires_grad_(True), W.double().requires_grad_(True)
    y = F.conv1d(xr, Wr, None, padding=pad, dilation=dil)
    gx, gw = torch.autograd.grad(y, (xr, Wr), dy.double())
    xcl = x.permute(0, 2, 1).contiguous().view(B * T, Cin).to(dev)
    dycl = dy.permute(0, 2, 1).contiguous().view(B * T, Cout).to(dev)
    Wd = W.to(dev)
    with ops.fp32_split(3):
        Wf, Wb = ops.conv_weight_fwd(Wd), ops.conv_weight_bwd(Wd)
        xp, dyp = ops.Planes(xcl).split(), ops.Planes(dycl).split()
        ycl = ops.conv1d_fwd(xcl, Wf, None, B, T, Cin, Cout, K, dil, x_planes=xp, w_planes=ops.Planes(Wf).split())
        kinds = [int(ops.lib().gt_sgemm_last_kernel())]
        close(ycl.view(B, T, Cout).permute(0, 2, 1), y, rtol=1e-5, atol=1e-4, what='conv fwd (planes)')
        dx = ops.conv1d_dx(dycl, Wb, B, T, Cin, Cout, K, dil, dy_planes=dyp, w_planes=ops.Planes(Wb).split())
        close(dx.view(B, T, Cin).permute(0, 2, 1), gx, rtol=1e-5, atol=1e-4, what='conv dx (planes)')
        dW = torch.zeros(Cout, Cin, K, device=dev)
        ops.conv1d_dw(dycl, xcl, B, T, Cin, Cout, K, dil, dW, dy_planes=dyp, x_planes=xp)
        kinds.append(int(ops.lib().gt_sgemm_last_kernel()))
        close(dW, gw, rtol=1e-5, atol=1e-4, what='conv dw (planes)')
        if (B * T) % 32 == 0:
            assert all(k in (3, 4) for k in kinds), kinds              # the pre-split kernels really ran
        if (B, T) == (8, 1024):
            assert kinds[0] == 4                                       # ... the 256 x 128 one for the big forward


@pytest.mark.parametrize('bf16', PRECISIONS)
@pytest.mark.parametrize('B,T,Cin,Cout,K,dil', [(3, 37, 16, 24, 5, 1), (2, 9, 32, 48, 5, 2), (4, 150, 80, 512, 5, 1)])
def test_conv1d_fwd_dx_dw(dev, B, T, Cin, Cout, K, dil, bf16):
    """bf16: ops.bf16_gemms (hparams.fp16_run) -- every product takes its two operands rounded to bf16, so the
    reference is the fp64 convolution / gradients of the rounded x, W and dy."""
    from gantron_amd import ops
    x, W, bias = g(B, Cin, T), g(Cout, Cin, K, seed=1, scale=0.2), g(Cout, seed=2)
    dy = g(B, Cout, T, seed=3)
    pad = dil * (K - 1) // 2
    xr, Wr = rb(x, bf16).requires_grad_(True), rb(W, bf16).requires_grad_(True)
    y = F.conv1d(xr, Wr, bias.double(), padding=pad, dilation=dil)
    gx, gw = torch.autograd.grad(y, (xr, Wr), rb(dy, bf16))
    xcl = x.permute(0, 2, 1).contiguous().view(B * T, Cin).to(dev)
    Wd = W.to(dev)
    with ops.bf16_gemms(bf16 is True), ops.fp32_split(0 if isinstance(bf16, bool) else bf16):
        ycl = ops.conv1d_fwd(xcl, ops.conv_weight_fwd(Wd), bias.to(dev), B, T, Cin, Cout, K, dil)
        close(ycl.view(B, T, Cout).permute(0, 2, 1), y, **tol(bf16, 1e-4), what='conv fwd')
        dycl = dy.permute(0, 2, 1).contiguous().view(B * T, Cout).to(dev)
        dx = ops.conv1d_dx(dycl, ops.conv_weight_bwd(Wd), B, T, Cin, Cout, K, dil)
        close(dx.view(B, T, Cin).permute(0, 2, 1), gx, **tol(bf16, 1e-4), what='conv dx')
        dW = torch.zeros(Cout, Cin, K, device=dev)
        ops.conv1d_dw(dycl, xcl, B, T, Cin, Cout, K, dil, dW)
        close(dW, gw, **tol(bf16, 1e-4), what='conv dw')


# ---------------------------------------------------------------------------- recurrent kernels
@pytest.mark.parametrize('Bsz,N,K,S', [(64, 256, 512, 4), (8, 64, 96, 3), (33, 128, 100, 2), (64, 4096, 2048, 4)])
def test_rec_gemm(dev, Bsz, N, K, S):
    from gantron_amd import ops
    from gantron_amd.engine import pack_rec
    Bp, ld = ops.roundup(Bsz, 16), ops.roundup(K, 16) + 16
    W = g(N, K, scale=0.1)
    X = torch.zeros(Bp, ld)
    X[:Bsz, :K] = g(Bsz, K, seed=1)
    Wp = pack_rec(W.to(dev), N, N, K)
    S = ops.pick_splits(K, S)
    part = torch.empty(S, Bp, N, device=dev)
    ops.call('gt_rec_gemm', X.to(dev), ld, Wp, part, Bp // 16, N, K, S, Bp * N, 1, 0, 0, 0)
    close(part.sum(0)[:Bsz], X[:Bsz, :K].double() @ W.double().t(), rtol=1e-5, atol=1e-4, what='rec_gemm')
    # transposed packing: out = X2 @ W (W is K2 x N2 with K2 = N, N2 = K padded to 16)
    N2 = ops.roundup(K, 16)
    WTp = pack_rec(W.to(dev), N2, K, N, transposed=True)
    X2 = torch.zeros(Bp, N)
    X2[:Bsz] = g(Bsz, N, seed=2)
    S2 = ops.pick_splits(N, 4)
    part2 = torch.empty(S2, Bp, N2, device=dev)
    ops.call('gt_rec_gemm', X2.to(dev), N, WTp, part2, Bp // 16, N2, N, S2, Bp * N2, 1, 0, 0, 0)
    close(part2.sum(0)[:Bsz, :K], X2[:Bsz].double() @ W.double(), rtol=1e-5, atol=1e-4, what='rec_gemm transposed')


@pytest.mark.parametrize('Bsz,N,K,S', [(64, 4096, 3072, 4), (64, 2048, 4096, 8), (5, 128, 40, 1), (33, 192, 1100, 2),
                                       (64, 4096, 2117, 3)])
def test_rec_gemm_bf16(dev, Bsz, N, K, S):
    """gt_rec_gemm_bf16: operands rounded to bf16 (nearest even), fp32 accumulation -- against the fp64 product of the
    SAME rounded operands (so the check is exact up to accumulation order), plain and transposed packing."""
    from gantron_amd import ops
    from gantron_amd.engine import pack_rec_bf16
    r = lambda t: t.to(torch.bfloat16).double()
    Bp, ld = ops.roundup(Bsz, 16), ops.roundup(K, 16)
    W = g(N, K, scale=0.1)
    X = torch.zeros(Bp, ld)
    X[:Bsz, :K] = g(Bsz, K, seed=1)
    Wp = pack_rec_bf16(W.to(dev), N, N, K)
    part = torch.full((S, Bp, N), float('nan'), device=dev)
    ops.call('gt_rec_gemm_bf16', X.to(dev), ld, Wp, part, Bp // 16, N, K, S, Bp * N, 1, 0, 0, 0)
    close(part.sum(0)[:Bsz], r(X[:Bsz, :K]) @ r(W).t(), rtol=1e-5, atol=1e-4, what='rec_gemm_bf16')
    N2 = ops.roundup(K, 16)
    WTp = pack_rec_bf16(W.to(dev), N2, K, N, transposed=True)
    X2 = torch.zeros(Bp, N)
    X2[:Bsz] = g(Bsz, N, seed=2)
    from gantron_amd._lib import lib
    S2 = max(int(lib().gt_rec_bf16_min_splits(N)), 1)
    part2 = torch.full((S2, Bp, N2), float('nan'), device=dev)
    ops.call('gt_rec_gemm_bf16', X2.to(dev), N, WTp, part2, Bp // 16, N2, N, S2, Bp * N2, 1, 0, 0, 0)
    close(part2.sum(0)[:Bsz, :K], r(X2[:Bsz]) @ r(W), rtol=1e-5, atol=1e-4, what='rec_gemm_bf16 transposed')


@pytest.mark.parametrize('Bsz,H,K,bias_row', [(64, 1024, 2048, False), (5, 32, 100, True), (33, 64, 333, False), (70, 16, 48, True)])
def test_lstm_step_fused(dev, Bsz, H, K, bias_row):
    """gt_lstm_step_fused (product + cell + dropout in one launch, gate-interleaved packing) against fp64."""
    from gantron_amd import ops
    from gantron_amd._lib import lib
    from gantron_amd.engine import pack_gates
    p = 0.1
    Bp = ops.roundup(Bsz, 16) if Bsz <= 64 else ops.roundup(Bsz, 64)
    ld = int(lib().gt_fused_k_padded(K))
    W = g(4 * H, K, scale=1.0 / math.sqrt(K))
    X = torch.zeros(Bp, ld)
    X[:Bsz, :K] = g(Bsz, K, seed=1)
    pre = g(1, 4 * H, seed=6, scale=0.5) if bias_row else g(Bp, 4 * H, seed=6, scale=0.5)
    c_prev = g(Bp, H, seed=2)
    keep = (torch.rand(Bp, H, generator=torch.Generator().manual_seed(3)) >= p)
    gates_in = X[:Bsz, :K].double() @ W.double().t() + pre.double()[:Bsz if not bias_row else 1]
    i, f, gg, o = gates_in.chunk(4, dim=1)
    c = torch.sigmoid(f) * c_prev[:Bsz].double() + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c) * keep[:Bsz].double() / (1 - p)
    Wg = pack_gates(W.to(dev), H, K)
    gates = torch.zeros(Bp, 4 * H, device=dev); c_out = torch.zeros(Bp, H, device=dev)
    h_raw = torch.zeros(Bp, H, device=dev); h0 = torch.zeros(Bp, H + 8, device=dev); h1 = torch.zeros(Bp, H, device=dev)
    ops.call('gt_lstm_step_fused', X.to(dev), ld, Wg, K, pre.to(dev), 0 if bias_row else 4 * H, c_prev.to(dev),
             keep.to(torch.uint8).to(dev), p, gates, c_out, h_raw, ops.ptr(h0, 4), H + 8, h1, H, None, 0, Bsz, Bp, H)
    close(gates[:Bsz], torch.cat((torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)), 1), rtol=1e-5,
          atol=1e-5, what='fused gates')
    close(c_out[:Bsz], c, rtol=1e-5, atol=1e-5, what='fused c')
    close(h_raw[:Bsz], h, rtol=1e-5, atol=1e-5, what='fused h')
    close(h0[:Bsz, 4:4 + H], h, rtol=1e-5, atol=1e-5, what='fused h strided')
    close(h1[:Bsz], h, rtol=1e-5, atol=1e-5, what='fused h second destination')
    assert float(c_out[Bsz:].abs().max()) == 0.0 if Bsz < Bp else True        # pad rows are never written


@pytest.mark.parametrize('Bsz,H,A,S', [(64, 1024, 128, 4), (5, 32, 8, 2), (70, 64, 32, 3)])
def test_cell_query_fused(dev, Bsz, H, A, S):
    """gt_cell_query_fwd = gt_lstm_cell_fwd + the query projection h . Wq^T (as H/32 slabs), one launch."""
    from gantron_amd import ops
    p = 0.1
    Bp = ops.roundup(Bsz, 16) if Bsz <= 64 else ops.roundup(Bsz, 64)
    Apad = ops.roundup(A, 16)
    part = g(S, Bp, 4 * H, scale=0.5)
    pre = g(Bp, 4 * H, seed=1, scale=0.5)
    c_prev = g(Bp, H, seed=2)
    keep = (torch.rand(Bp, H, generator=torch.Generator().manual_seed(3)) >= p)
    Wq = g(A, H, seed=4, scale=0.1)
    gates_in = (part.sum(0) + pre).double()[:Bsz]
    i, f, gg, o = gates_in.chunk(4, dim=1)
    c = torch.sigmoid(f) * c_prev[:Bsz].double() + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c) * keep[:Bsz].double() / (1 - p)
    gates = torch.zeros(Bp, 4 * H, device=dev); c_out = torch.zeros(Bp, H, device=dev); h_raw = torch.zeros(Bp, H, device=dev)
    h0 = torch.zeros(Bp, H + 8, device=dev); h1 = torch.zeros(Bp, H, device=dev)
    qpart = torch.full((H // 32, Bp, Apad), float('nan'), device=dev)
    part_d = part.to(dev)
    ops.call('gt_cell_query_fwd', ops.slabs(part_d, S, Bp * 4 * H, 4 * H), pre.to(dev), 4 * H, c_prev.to(dev),
             keep.to(torch.uint8).to(dev), p, gates, c_out, h_raw, ops.ptr(h0, 4), H + 8, h1, H, Wq.to(dev), H, qpart,
             Bp * Apad, Bsz, Bp, H, A, Apad)
    close(c_out[:Bsz], c, rtol=1e-5, atol=1e-5, what='c')
    close(h_raw[:Bsz], h, rtol=1e-5, atol=1e-5, what='h')
    close(h0[:Bsz, 4:4 + H], h, rtol=1e-5, atol=1e-5, what='h strided')
    close(h1[:Bsz], h, rtol=1e-5, atol=1e-5, what='h second destination')
    close(gates[:Bsz], torch.cat((torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)), 1), rtol=1e-5,
          atol=1e-5, what='gates')
    close(qpart.sum(0)[:Bsz, :A], h @ Wq.double().t(), rtol=1e-5, atol=1e-5, what='query')
    assert torch.isfinite(qpart).all()                   # every slab element was written


@pytest.mark.parametrize('Bsz,N,K', [(64, 4096, 1536), (64, 1536, 4096), (5, 128, 40), (33, 192, 1100), (64, 4096, 2117)])
def test_rec_gemm_x3(dev, Bsz, N, K):
    """gt_rec_gemm_x3: the fp32 product from three bf16 planes per operand (six exact bf16 MFMA products, fp32 accumulation)
    against float64 -- held to the accuracy of an fp32 product (a few 2^-24 of sum |a||b|), over operands that span five
    decades inside one reduction; plain and transposed packing."""
    from gantron_amd import ops
    from gantron_amd._lib import lib
    from gantron_amd.engine import pack_rec_x3, rec_splits_x3
    Bp, ld = ops.roundup(Bsz, 16), ops.roundup(K, 16)
    gen = torch.Generator().manual_seed(N + K)
    W = g(N, K, scale=0.1) * torch.pow(10.0, torch.randint(-3, 2, (N, K), generator=gen).float())
    X = torch.zeros(Bp, ld)
    X[:Bsz, :K] = g(Bsz, K, seed=1) * torch.pow(10.0, torch.randint(-2, 3, (Bsz, K), generator=gen).float())
    S = rec_splits_x3(N, K)
    assert S >= int(lib().gt_rec_x3_min_splits(K))
    Wp = pack_rec_x3(W.to(dev), N, N, K)
    part = torch.full((S, Bp, N), float('nan'), device=dev)
    ops.call('gt_rec_gemm_x3', X.to(dev), ld, Wp, part, Bp // 16, N, K, S, Bp * N, 1, 0, 0, 0)
    ref = X[:Bsz, :K].double() @ W.double().t()
    mag = X[:Bsz, :K].double().abs() @ W.double().abs().t()
    err = ((part.sum(0)[:Bsz].cpu().double() - ref).abs() / mag).max()
    assert float(err) < 4e-7, f'rec_gemm_x3: error {float(err):.3g} of sum |a||b|'
    N2 = ops.roundup(K, 16)
    WTp = pack_rec_x3(W.to(dev), N2, K, N, transposed=True)
    X2 = torch.zeros(Bp, N)
    X2[:Bsz] = g(Bsz, N, seed=2)
    S2 = rec_splits_x3(N2, N)
    part2 = torch.full((S2, Bp, N2), float('nan'), device=dev)
    ops.call('gt_rec_gemm_x3', X2.to(dev), N, WTp, part2, Bp // 16, N2, N, S2, Bp * N2, 1, 0, 0, 0)
    ref2 = X2[:Bsz].double() @ W.double()
    mag2 = X2[:Bsz].double().abs() @ W.double().abs()
    err2 = ((part2.sum(0)[:Bsz, :K].cpu().double() - ref2).abs() / mag2).max()
    assert float(err2) < 4e-7, f'rec_gemm_x3 transposed: error {float(err2):.3g} of sum |a||b|'
    if N2 > K:
        assert float(part2.sum(0)[:, K:].abs().max()) == 0.0          # padded columns meet zero weights


@pytest.mark.parametrize('Bsz,N,K,xs', [(64, 4096, 1536, 1.0), (64, 1536, 4096, 1e-7), (5, 128, 40, 3e4), (33, 192, 1100, 1.0),
                                        (64, 4096, 2117, 1e-3), (128, 1536, 4096, 1e-5)])
def test_rec_gemm_h2(dev, Bsz, N, K, xs):
    """gt_rec_gemm_h2: the fp32 product from two SCALED fp16 planes per operand (three exact fp16 MFMA products, fp32
    accumulation; the weights' scale from the image's trailer, the activations' from a gt_absmax word or a floor) against
    float64, held to the accuracy of an fp32 product over operands that span five decades inside one reduction and activations
    far outside fp16's range (gradient-sized 1e-7, 3e4); plain and transposed packing; two 64-row blocks on one weight image."""
    from gantron_amd import ops
    from gantron_amd._lib import lib
    from gantron_amd.engine import pack_rec_h2, rec_splits_x3
    Bp = ops.roundup(Bsz, 16) if Bsz <= 64 else ops.roundup(Bsz, 64)
    ld = ops.roundup(K, 16)
    mt, nz = (Bp // 16, 1) if Bp <= 64 else (4, Bp // 64)
    gen = torch.Generator().manual_seed(N + K)
    W = g(N, K, scale=0.1) * torch.pow(10.0, torch.randint(-3, 2, (N, K), generator=gen).float())
    X = torch.zeros(Bp, ld)
    X[:Bsz, :K] = g(Bsz, K, seed=1) * xs * torch.pow(10.0, torch.randint(-2, 3, (Bsz, K), generator=gen).float())
    S = rec_splits_x3(N, K)
    Wp = pack_rec_h2(W.to(dev), N, N, K)
    Xd = X.to(dev)
    ref = X[:Bsz, :K].double() @ W.double().t()
    mag = X[:Bsz, :K].double().abs() @ W.double().abs().t()
    for xmax, floor in ((ops.absmax(Xd), 0.0), (None, float(X.abs().max()) * 1.7)):        # a measured maximum; an a-priori bound
        part = torch.full((S, Bp, N), float('nan'), device=dev)
        ops.call('gt_rec_gemm_h2', Xd, ld, Wp, part, mt, N, K, S, Bp * N, nz, 64 * ld, 64 * N, xmax, floor)
        err = ((part.sum(0)[:Bsz].cpu().double() - ref).abs() / mag).max()
        assert float(err) < 4e-7, f'rec_gemm_h2: error {float(err):.3g} of sum |a||b|'
    N2 = ops.roundup(K, 16)
    WTp = pack_rec_h2(W.to(dev), N2, K, N, transposed=True)
    X2 = torch.zeros(Bp, N)
    X2[:Bsz] = g(Bsz, N, seed=2) * xs
    S2 = rec_splits_x3(N2, N)
    part2 = torch.full((S2, Bp, N2), float('nan'), device=dev)
    X2d = X2.to(dev)
    ops.call('gt_rec_gemm_h2', X2d, N, WTp, part2, mt, N2, N, S2, Bp * N2, nz, 64 * N, 64 * N2, ops.absmax(X2d), 0.0)
    ref2 = X2[:Bsz].double() @ W.double()
    mag2 = X2[:Bsz].double().abs() @ W.double().abs()
    err2 = ((part2.sum(0)[:Bsz, :K].cpu().double() - ref2).abs() / mag2).max()
    assert float(err2) < 4e-7, f'rec_gemm_h2 transposed: error {float(err2):.3g} of sum |a||b|'
    if N2 > K:
        assert float(part2.sum(0)[:, K:].abs().max()) == 0.0          # padded columns meet zero weights


def test_rec_h2_kernels_in_their_other_forms():
    """GANTRON_DEBUG=rec_h2_form=3: gt_rec_gemm_h2 through recurrent_bf16.hip's own NP = 2 instantiation (256-k chunks, 68 KB of
    LDS; faster alone, slower in the backward loop) instead of recurrent.hip's lean kernel, and gt_rec_cell_fwd_h2 through
    recurrent.hip's lean fused kernel instead of recurrent_bf16.hip's (the default there).  The switch is read once per
    process: the cases of test_rec_gemm_h2 and the h2 cases of test_rec_cell_fused run again in a child pytest."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GANTRON_DEBUG='rec_h2_form=3')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_kernels_gpu.py'), '-q', '-x', '-k',
                        'test_rec_gemm_h2 or (test_rec_cell_fused and 2-)', '-p', 'no:cacheprovider'], env=env, cwd=root,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize('Bsz,H,K,S,A', [(64, 1024, 2048, 4, 128), (5, 32, 100, 2, 8), (70, 64, 333, 3, 40), (64, 1024, 1024, 4, 0),
                                         (33, 48, 50, 1, 24), (64, 1024, 3072, 6, 0)])
@pytest.mark.parametrize('bf16', [False, True, 3, 2])
def test_rec_cell_fused(dev, Bsz, H, K, S, A, bf16):
    """gt_rec_cell_fwd = gt_rec_gemm + gt_lstm_cell_fwd (+ the query projection) in ONE launch: the K-splits of a
    column group meet through write-through slabs and a ticket, the last one runs the cell.  Three launches in a
    row on the same workspace (the tickets carry over from launch to launch), each with fresh inputs.
    bf16: gt_rec_cell_fwd_bf16 -- X and W rounded to bf16 (nearest even) for the product, everything else fp32."""
    from gantron_amd import ops
    from gantron_amd._lib import lib
    from gantron_amd.engine import pack_rec_cell
    p = 0.1
    Bp = ops.roundup(Bsz, 16) if Bsz <= 64 else ops.roundup(Bsz, 64)
    Apad = ops.roundup(max(A, 1), 16)
    ld = ops.roundup(K, 16) + 16
    W = g(4 * H, K, scale=0.1)
    if bf16 is True:
        kb = -(-K // 32)
        S = next(s for s in range(max(S, int(lib().gt_rec_bf16_min_splits(K))), 17) if -(-kb // -(-kb // s)) == s)
        W = W.bfloat16().float()
    elif bf16 in (3, 2):      # gt_rec_cell_fwd_x3 / _h2: the fp32 product from three bf16 / two scaled fp16 planes, K-slices of at most 1024
        kb = -(-K // 32)
        S = next(s for s in range(max(S, int(lib().gt_rec_x3_min_splits(K))), 17) if -(-kb // -(-kb // s)) == s)
    else:
        S = ops.pick_splits(K, S)
    Wp = pack_rec_cell(W.to(dev), H, K, bf16)
    fn = 'gt_rec_cell_fwd_bf16' if bf16 is True else 'gt_rec_cell_fwd_x3' if bf16 == 3 else 'gt_rec_cell_fwd_h2' if bf16 == 2 else 'gt_rec_cell_fwd'
    Wq = g(max(A, 1), H, seed=4, scale=0.1)
    part = torch.full((S, Bp, 4 * H), float('nan'), device=dev)
    counters = torch.zeros((Bp // 64 if Bp > 64 else 1) * (H // 16), dtype=torch.int32, device=dev)
    for rep in range(3):
        X = torch.zeros(Bp, ld)
        X[:Bsz, :K] = g(Bsz, K, seed=10 + rep)
        pre = g(Bp, 4 * H, seed=20 + rep, scale=0.5)
        c_prev = g(Bp, H, seed=30 + rep)
        keep = (torch.rand(Bp, H, generator=torch.Generator().manual_seed(40 + rep)) >= p)
        Xr = X.bfloat16().float() if bf16 is True else X
        gates_in = (Xr[:Bsz, :K].double() @ W.double().t() + pre[:Bsz].double())
        i, f, gg, o = gates_in.chunk(4, dim=1)
        c = torch.sigmoid(f) * c_prev[:Bsz].double() + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c) * keep[:Bsz].double() / (1 - p)
        gates = pre.clone().to(dev)                      # in place: the gate values overwrite the pre-activations
        c_out = torch.zeros(Bp, H, device=dev); h_raw = torch.zeros(Bp, H, device=dev)
        h0 = torch.zeros(Bp, H + 8, device=dev); h1 = torch.zeros(Bp, H, device=dev)
        qpart = torch.full((H // 16, Bp, Apad), float('nan'), device=dev)
        Xd = X.to(dev)
        # (h2: the activations' bound is a measured maximum on even launches, an a-priori one on odd launches)
        h2x = () if bf16 != 2 or bf16 is True else ((ops.absmax(Xd), 0.0) if rep % 2 == 0 else (None, float(X.abs().max()) * 1.3))
        ops.call(fn, Xd, ld, Wp, part, counters, Bp, Bsz, H, K, S, gates, 4 * H, c_prev.to(dev),
                 keep.to(torch.uint8).to(dev), p, gates, c_out, h_raw, ops.ptr(h0, 4), H + 8, h1, H, None, 0,
                 Wq.to(dev) if A else None, H, qpart if A else None, Bp * Apad, A, Apad, *h2x, 3)
        close(c_out[:Bsz], c, rtol=1e-5, atol=2e-5, what=f'c (launch {rep})')
        close(h_raw[:Bsz], h, rtol=1e-5, atol=2e-5, what=f'h (launch {rep})')
        close(h0[:Bsz, 4:4 + H], h, rtol=1e-5, atol=2e-5, what='h strided')
        close(h1[:Bsz], h, rtol=1e-5, atol=2e-5, what='h second destination')
        close(gates[:Bsz], torch.cat((torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)), 1), rtol=1e-5,
              atol=2e-5, what='gates')
        if Bsz < Bp:
            assert float(c_out[Bsz:].abs().max()) == 0.0 and float(h1[Bsz:].abs().max()) == 0.0      # pad rows untouched
        if A:
            close(qpart.sum(0)[:Bsz, :A], h @ Wq.double().t(), rtol=1e-5, atol=2e-5, what='query')
            assert torch.isfinite(qpart).all()               # every slab element was written
    assert int(counters.min()) == int(counters.max()) == (3 * S if S > 1 else 0)      # one ticket per workgroup and launch


@pytest.mark.parametrize('Bsz,H,S', [(64, 1024, 16), (5, 64, 2), (70, 128, 4), (33, 64, 1)])
def test_rec_cellbwd_fused(dev, Bsz, H, S):
    """gt_rec_cellbwd = the recurrent input-gradient product dG_t . W_hh + gt_lstm_cell_bwd of step t - 1 in ONE launch, run
    as a chain of 3 steps on the same workspace (tickets carry over) against the two-launch sequence."""
    from gantron_amd import ops
    from gantron_amd._lib import lib
    from gantron_amd.engine import pack_rec
    p = 0.1
    Bp = ops.roundup(Bsz, 16) if Bsz <= 64 else ops.roundup(Bsz, 64)
    K = 4 * H
    S = ops.pick_splits(K, S)
    T = 4
    Whh = g(4 * H, H, scale=0.1).to(dev)                              # (4H, H): d h = dG . Whh
    WT = pack_rec(Whh, H, H, K, transposed=True)                      # packed [H] x [4H]
    gates = torch.sigmoid(g(T, Bp, 4 * H, seed=1)).to(dev)
    c = g(T + 1, Bp, H, seed=2).to(dev)
    keep = (torch.rand(T, Bp, H, generator=torch.Generator().manual_seed(3)) >= p).to(torch.uint8).to(dev)
    dhdir = g(T, Bp, H + 8, seed=4, scale=0.3).to(dev)                # emitted-state gradients, offset columns

    def run(fused):
        dG = torch.zeros(T, Bp, 4 * H, device=dev); dc = torch.zeros(Bp, H, device=dev)
        part = torch.full((S, Bp, H), float('nan'), device=dev)
        counters = torch.zeros((Bp // 64 if Bp > 64 else 1) * (H // 64), dtype=torch.int32, device=dev)
        # last step: nothing recurrent yet
        ops.call('gt_lstm_cell_bwd', ops.NO_SLABS, ops.NO_SLABS, ops.NO_SLABS, ops.ptr(dhdir[T - 1], 4), H + 8, dc, gates[T - 1],
                 c[T], c[T - 1], keep[T - 1], p, None, dG[T - 1], dc, None, Bsz, H)
        for t in range(T - 1, 0, -1):            # product of step t, cell backward of step t - 1
            if fused:
                ops.call('gt_rec_cellbwd', dG[t], 4 * H, WT, part, counters, Bp, Bsz, H, K, S, ops.ptr(dhdir[t - 1], 4), H + 8,
                         dc, gates[t - 1], c[t], c[t - 1], keep[t - 1], p, dG[t - 1], 2)
            else:
                ops.call('gt_rec_gemm', dG[t], 4 * H, WT, part, Bp // 16 if Bp <= 64 else 4, H, K, S, Bp * H,
                         1 if Bp <= 64 else Bp // 64, 64 * 4 * H, 0, 64 * H)
                ops.call('gt_lstm_cell_bwd', ops.slabs(part, S, Bp * H, H), ops.NO_SLABS, ops.NO_SLABS, ops.ptr(dhdir[t - 1], 4),
                         H + 8, dc, gates[t - 1], c[t], c[t - 1], keep[t - 1], p, None, dG[t - 1], dc, None, Bsz, H)
        return dG, dc, counters

    ref, got = run(False), run(True)
    close(got[0][:, :Bsz], ref[0][:, :Bsz], rtol=1e-5, atol=1e-6, what='dG')
    close(got[1][:Bsz], ref[1][:Bsz], rtol=1e-5, atol=1e-6, what='dc')
    assert float(got[0][:, Bsz:].abs().max()) == 0.0 if Bsz < Bp else True       # pad rows untouched
    assert int(got[2].min()) == int(got[2].max()) == ((T - 1) * S if S > 1 else 0)


@pytest.mark.parametrize('Bsz,H,A,S', [(64, 1024, 128, 6), (5, 32, 8, 2), (70, 64, 40, 3)])
def test_cell_bwd_query_fused(dev, Bsz, H, A, S):
    """gt_cell_bwd_query = gt_lstm_cell_bwd whose dh also receives dq . Wq computed inside the kernel."""
    from gantron_amd import ops
    p = 0.1
    Bp = ops.roundup(Bsz, 16) if Bsz <= 64 else ops.roundup(Bsz, 64)
    Apad = ops.roundup(A, 16)
    gates = torch.sigmoid(g(Bp, 4 * H, seed=1)); c_new = g(Bp, H, seed=2); c_prev = g(Bp, H, seed=3)
    keep = (torch.rand(Bp, H, generator=torch.Generator().manual_seed(4)) >= p)
    da = g(S, Bp, H + 16, seed=5, scale=0.3); dcs = g(2, Bp, H, seed=6, scale=0.3)
    dq = torch.zeros(Bp, Apad); dq[:Bsz, :A] = g(Bsz, A, seed=7)
    Wq = g(A, H, seed=8, scale=0.1)
    dc_next = g(Bp, H, seed=9, scale=0.3)
    dh = (da[:, :, 8:8 + H].sum(0) + dcs.sum(0) + dq[:, :A] @ Wq).double()
    gi, gf, gg, go = gates.double().chunk(4, dim=1)
    dhd = dh * keep.double() / (1 - p)
    tc = torch.tanh(c_new.double())
    dcell = dc_next.double() + dhd * go * (1 - tc * tc)
    want = torch.cat((dcell * gg * gi * (1 - gi), dcell * c_prev.double() * gf * (1 - gf), dcell * gi * (1 - gg * gg),
                      dhd * tc * go * (1 - go)), 1)
    dgates = torch.zeros(Bp, 4 * H, device=dev); dcp = dc_next.clone().to(dev)
    da_d, dcs_d = da.to(dev), dcs.to(dev)
    ops.call('gt_cell_bwd_query', ops.slabs(da_d, S, Bp * (H + 16), H + 16, offset=8), ops.slabs(dcs_d, 2, Bp * H, H),
             dq.to(dev), Apad, Wq.to(dev), H, dcp, gates.to(dev), c_new.to(dev), c_prev.to(dev),
             keep.to(torch.uint8).to(dev), p, dgates, dcp, Bsz, Bp, H, A)
    close(dgates[:Bsz], want[:Bsz], rtol=1e-5, atol=1e-5, what='dgates')
    close(dcp[:Bsz], (dcell * gf)[:Bsz], rtol=1e-5, atol=1e-5, what='dc_prev')


def test_rec_gemm2_two_problems_one_launch(dev):
    """gt_rec_gemm2: two products of different shapes in one launch == two gt_rec_gemm calls."""
    from gantron_amd import ops
    from gantron_amd.engine import pack_rec
    Bp = 64
    outs = []
    args = []
    for (N, K, S, seed) in ((2048, 4096, 8, 1), (3072, 4096, 5, 2)):
        W = g(N, K, seed=seed, scale=0.05)
        X = g(Bp, K, seed=seed + 10)
        Wp = pack_rec(W.to(dev), N, N, K)
        part = torch.full((S, Bp, N), float('nan'), device=dev)
        args += [X.to(dev), K, Wp, part, N, K, S, Bp * N]
        outs.append((X.double() @ W.double().t(), part))
    ops.call('gt_rec_gemm2', *args, Bp // 16)
    for want, part in outs:
        close(part.sum(0), want, rtol=1e-5, atol=1e-3, what='rec_gemm2')


def test_lstm_cell_fwd_bwd(dev):
    from gantron_amd import ops
    B, H, S, p = 5, 48, 3, 0.1
    part = g(S, B, 4 * H, scale=0.5)
    pre = g(B, 4 * H, seed=1, scale=0.5)
    c_prev = g(B, H, seed=2)
    keep = (torch.rand(B, H, generator=torch.Generator().manual_seed(3)) >= p)
    gates_in = (part.sum(0) + pre).double().requires_grad_(True)
    cp = c_prev.double().requires_grad_(True)
    i, f, gg, o = gates_in.chunk(4, dim=1)
    c = torch.sigmoid(f) * cp + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c) * keep.double() / (1 - p)
    gates = torch.empty(B, 4 * H, device=dev); c_out = torch.empty(B, H, device=dev)
    h_raw = torch.empty(B, H, device=dev); h0 = torch.zeros(B, H + 7, device=dev)
    part_d = part.to(dev)     # slabs()/ptr() keep raw addresses only: the tensor must outlive the call
    ops.call('gt_lstm_cell_fwd', ops.slabs(part_d, S, B * 4 * H, 4 * H), pre.to(dev), 4 * H, c_prev.to(dev), None, 0,
             keep.to(torch.uint8).to(dev), p, None, gates, c_out, h_raw, ops.ptr(h0, 3), H + 7, None, 0, None, 0, B, H)
    close(c_out, c, what='cell c')
    close(h_raw, h, what='cell h')
    close(h0[:, 3:3 + H], h, what='cell h strided')
    dh, dc = g(B, H, seed=4), g(B, H, seed=5)
    gg_, gc = torch.autograd.grad((h * dh.double()).sum() + (c * dc.double()).sum(), (gates_in, cp))
    dgates = torch.empty(B, 4 * H, device=dev); dcp = torch.empty(B, H, device=dev)
    dh_sl = torch.stack((0.25 * dh, 0.75 * dh)).contiguous().to(dev)
    ops.call('gt_lstm_cell_bwd', ops.slabs(dh_sl, 2, B * H, H), ops.NO_SLABS, ops.NO_SLABS, None, 0, dc.to(dev), gates, c_out,
             c_prev.to(dev), keep.to(torch.uint8).to(dev), p, None, dgates, dcp, None, B, H)
    close(dgates, gg_, what='cell dgates')
    close(dcp, gc, what='cell dc_prev')


# ------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize('B,Ti,E,A,Fn,Kc', [(3, 45, 40, 8, 4, 5), (2, 70, 96, 128, 32, 31), (2, 300, 64, 128, 32, 31),
                                            (3, 45, 1024, 128, 32, 31)])
def test_attention_step_fwd_bwd(dev, B, Ti, E, A, Fn, Kc):
    from gantron_amd import ops
    from gantron_amd._lib import lib
    from oracle import gantron_ref as R
    pre = 'decoder.attention_layer.'
    P = {pre + 'query_layer.linear_layer.weight': g(A, 16, scale=0.3),
         pre + 'location_layer.location_conv.conv.weight': g(Fn, 2, Kc, seed=1, scale=0.3),
         pre + 'location_layer.location_dense.linear_layer.weight': g(A, Fn, seed=2, scale=0.3),
         pre + 'v.linear_layer.weight': g(1, A, seed=3, scale=0.5)}
    P = {k: v.double().requires_grad_(True) for k, v in P.items()}
    lens = torch.tensor([Ti, Ti - 7, max(Ti - 20, 3)][:B])
    memory = g(B, Ti, E, seed=4).double().requires_grad_(True)
    pm = g(B, Ti, A, seed=5, scale=0.5).double().requires_grad_(True)
    query = g(B, 16, seed=6).double().requires_grad_(True)
    w_prev = torch.softmax(g(B, Ti, seed=7), 1).double().requires_grad_(True)
    w_cum = (torch.softmax(g(B, Ti, seed=8), 1) * 3).double().requires_grad_(True)
    pad = ~R.length_mask(lens, Ti)
    ctx, w = R.attention_step(P, query, memory, pm, w_prev, w_cum, pad)
    q_ref = query @ P[pre + 'query_layer.linear_layer.weight'].t()

    d = ops.attn_dims(B, Ti, E, A, Fn, Kc)
    f32 = lambda t: t.detach().float().contiguous().to(dev)
    qd = f32(q_ref)
    Wc, Wd, v = (f32(P[pre + 'location_layer.location_conv.conv.weight']),
                 f32(P[pre + 'location_layer.location_dense.linear_layer.weight']), f32(P[pre + 'v.linear_layer.weight']))
    q_out = torch.empty(B, A, device=dev); en = torch.empty(B, Ti, device=dev)
    lens32 = lens.to(torch.int32).to(dev)
    ops.call('gt_attn_energies_fwd', d, ops.slabs(qd, 1, B * A, A), f32(w_prev), f32(w_cum), f32(pm), Wc, Wd, v, lens32, q_out, en)
    w_out = torch.empty(B, Ti, device=dev); cum_out = torch.empty(B, Ti, device=dev)
    c0 = torch.empty(B, E, device=dev); c1 = torch.zeros(B, E + 5, device=dev)
    ops.call('gt_attn_context_fwd', d, en, f32(memory), E, lens32, w_out, f32(w_cum), cum_out, c0, E, ops.ptr(c1, 5), E + 5, None, 0)
    close(w_out, w, rtol=1e-4, atol=1e-6, what='attention weights')
    close(c0, ctx, rtol=1e-4, atol=1e-5, what='context')
    close(c1[:, 5:], ctx, rtol=1e-4, atol=1e-5, what='context strided')
    close(cum_out, w_cum + w, rtol=1e-4, atol=1e-6, what='cumulative')

    # backward: upstream grads on ctx and (externally) on w; carried alignment grads from "step t+1"
    dctx, dw_ext = g(B, E, seed=9).double(), g(B, Ti, seed=10, scale=0.3).double()
    carry_prev, carry_cum, cum_acc0 = g(B, Ti, seed=11, scale=0.2), g(B, Ti, seed=12, scale=0.2), g(B, Ti, seed=13, scale=0.2)
    loss = (ctx * dctx).sum() + (w * (dw_ext + carry_prev.double() + carry_cum.double() + cum_acc0.double())).sum()
    names = list(P)
    grads = torch.autograd.grad(loss, [query, memory, pm, w_prev, w_cum] + [P[k] for k in names])
    gq_in, gmem, gpm, gwp, gwc = grads[:5]
    gP = dict(zip(names, grads[5:]))

    dwcat_in = torch.stack((carry_prev, carry_cum)).contiguous().to(dev)
    dcum_acc = cum_acc0.clone().to(dev)
    dctx_out = torch.empty(B, E, device=dev); dq = torch.full((B, A), 7.0, device=dev); dw = torch.empty(B, Ti, device=dev)
    half = (0.5 * dctx).float().contiguous().to(dev)
    ops.call('gt_attn_context_bwd', d, ops.slabs(half, 1, B * E, E), ops.NO_SLABS, half, E, f32(memory), E, lens32, dwcat_in, dcum_acc,
             f32(dw_ext), dctx_out, E, dq, A, dw)
    close(dctx_out, dctx, what='dctx total')
    assert float(dwcat_in.abs().max()) == 0.0 and float(dq.abs().max()) == 0.0
    valid = R.length_mask(lens, Ti)          # padded positions carry no gradient: their bookkeeping is skipped
    close(dcum_acc.cpu() * valid, (cum_acc0 + carry_cum) * valid, what='dcum_acc')
    dpm = torch.zeros(B, Ti, A, device=dev); dwcat_out = torch.zeros(2, B, Ti, device=dev)
    acc = torch.zeros(int(lib().gt_attn_acc_elems(d)), device=dev)
    ops.call('gt_attn_energies_bwd', d, dw, w_out, q_out, f32(w_prev), f32(w_cum), f32(pm), Wc, Wd, v, lens32, dq, A, dpm, dwcat_out, acc)
    close(dpm, gpm, rtol=2e-3, atol=1e-6, what='dpm')
    close(dwcat_out[0], gwp, rtol=2e-3, atol=1e-6, what='d w_prev')
    dv = torch.zeros(1, A, device=dev); dWd = torch.zeros(A, Fn, device=dev); dWc = torch.zeros(Fn, 2, Kc, device=dev)
    ops.call('gt_attn_acc_reduce', d, acc, dv, dWd, dWc)
    close(dv, gP[pre + 'v.linear_layer.weight'], rtol=2e-3, atol=1e-6, what='dv')
    close(dWd, gP[pre + 'location_layer.location_dense.linear_layer.weight'], rtol=2e-3, atol=1e-6, what='dWd')
    close(dWc, gP[pre + 'location_layer.location_conv.conv.weight'], rtol=2e-3, atol=1e-6, what='dWc')
    # d memory from the context product: dmem[b,t,:] = w[b,t] * dctx[b,:]
    close(w_out[:, :, None].cpu().double() * dctx[:, None, :], gmem, rtol=2e-3, atol=1e-6, what='dmemory identity')
    # d w_cum as seen by the kernel = conv-transpose part only (the loss' direct w_cum terms are zero here)
    close(dwcat_out[1], gwc, rtol=2e-3, atol=1e-6, what='d w_cum')
    # dq: gradient w.r.t. the processed query = (d query) projected back is ill-posed; check via q grad directly
    q2 = q_ref.detach().clone().requires_grad_(True)
    # recompute energies with q as a leaf
    cat = torch.stack((w_prev.detach(), w_cum.detach()), dim=1)
    wconv = P[pre + 'location_layer.location_conv.conv.weight'].detach()
    loc = F.conv1d(cat, wconv, None, padding=(Kc - 1) // 2).transpose(1, 2) @ \
        P[pre + 'location_layer.location_dense.linear_layer.weight'].detach().t()
    e = (torch.tanh(q2[:, None, :] + loc + pm.detach()) @ P[pre + 'v.linear_layer.weight'].detach().t()).squeeze(-1)
    e = e.masked_fill(pad, -float('inf'))
    w2 = torch.softmax(e, 1)
    ctx2 = torch.bmm(w2.unsqueeze(1), memory.detach()).squeeze(1)
    l2 = (ctx2 * dctx).sum() + (w2 * (dw_ext + carry_prev.double() + carry_cum.double() + cum_acc0.double())).sum()
    close(dq, torch.autograd.grad(l2, q2)[0], rtol=2e-3, atol=1e-6, what='dq')


@pytest.mark.parametrize('B,Ti,E,nslabs', [(3, 150, 1024, 64), (64, 187, 1024, 64), (11, 77, 1096, 3), (2, 33, 64, 1), (70, 256, 512, 5)])
def test_attention_step_fused(dev, B, Ti, E, nslabs):
    """gt_attn_step_fwd = gt_attn_energies_fwd_u + gt_attn_context_fwd in ONE launch (the row's workgroups exchange their
    energies inside it): weights, cumulative weights, context (three destinations), query and tanh history against the
    oracle's attention_step and against the two-launch path, for three consecutive steps on the same arrival counters
    (ragged lengths: rows with 1 .. NC member workgroups, column slices of every width)."""
    from gantron_amd import ops
    from gantron_amd._lib import lib
    from oracle import gantron_ref as R
    A, Fn, Kc = 128, 32, 31
    d = ops.attn_dims(B, Ti, E, A, Fn, Kc)
    assert lib().gt_attn_step_supported(d)
    pre = 'decoder.attention_layer.'
    P = {pre + 'query_layer.linear_layer.weight': g(A, 16, scale=0.3),
         pre + 'location_layer.location_conv.conv.weight': g(Fn, 2, Kc, seed=1, scale=0.3),
         pre + 'location_layer.location_dense.linear_layer.weight': g(A, Fn, seed=2, scale=0.3),
         pre + 'v.linear_layer.weight': g(1, A, seed=3, scale=0.5)}
    gen = torch.Generator().manual_seed(B * 1000 + Ti)
    lens = torch.randint(1, Ti + 1, (B,), generator=gen)
    lens[0] = Ti
    if B > 1:
        lens[1] = 1
    memory = g(B, Ti, E, seed=4)
    pm = g(B, Ti, A, seed=5, scale=0.5)
    pad = ~R.length_mask(lens, Ti)
    f32 = lambda t: t.detach().float().contiguous().to(dev)
    Wc, Wd, v = (f32(P[pre + 'location_layer.location_conv.conv.weight']),
                 f32(P[pre + 'location_layer.location_dense.linear_layer.weight']), f32(P[pre + 'v.linear_layer.weight']))
    wpack = torch.empty(int(lib().gt_attn_wpack_elems(d)), device=dev)
    ops.call('gt_attn_pack_weights', d, Wc, Wd, wpack)
    lens32 = lens.to(torch.int32).to(dev)
    mem_d, pm_d = f32(memory), f32(pm)
    erow = torch.full((B, Ti), float('nan'), device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    tflag = torch.zeros(1, dtype=torch.int32, device=dev)
    for step in range(3):
        query = g(B, 16, seed=60 + step)
        w_prev = torch.softmax(g(B, Ti, seed=70 + step), 1) * (~pad)
        w_cum = torch.softmax(g(B, Ti, seed=80 + step), 1) * 3 * (~pad)
        ctx, w = R.attention_step({k: t.double() for k, t in P.items()}, query.double(), memory.double(), pm.double(),
                                  w_prev.double(), w_cum.double(), pad)
        q_ref = query.double() @ P[pre + 'query_layer.linear_layer.weight'].double().t()
        # the query arrives as `nslabs` partial slabs
        parts = torch.randn(nslabs, B, A, generator=gen) * 0.1
        parts[0] += q_ref.float() - parts.sum(0)
        qd = parts.contiguous().to(dev)
        q_out = torch.empty(B, A, device=dev); u_out = torch.empty(B, Ti, A, device=dev)
        w_out = torch.empty(B, Ti, device=dev); cum_out = torch.empty(B, Ti, device=dev)
        c0 = torch.full((B, E), float('nan'), device=dev); c1 = torch.zeros(B, E + 8, device=dev); c2 = torch.zeros(B, E, device=dev)
        ops.call('gt_attn_step_fwd', d, ops.slabs(qd, nslabs, B * A, A), f32(w_prev), f32(w_cum), pm_d, Wc, Wd, v, lens32, q_out,
                 wpack, u_out, erow, cnt, step + 1, tflag, mem_d, E, w_out, f32(w_cum), cum_out, c0, E, ops.ptr(c1, 8), E + 8, c2, E)
        assert int(tflag) == 0, 'the in-launch hand-off timed out'
        close(w_out, w, rtol=1e-4, atol=1e-6, what=f'attention weights (step {step})')
        close(c0, ctx, rtol=1e-4, atol=1e-5, what=f'context (step {step})')
        close(c1[:, 8:], ctx, rtol=1e-4, atol=1e-5, what='context strided')
        close(c2, ctx, rtol=1e-4, atol=1e-5, what='context third destination')
        close(cum_out, w_cum.double() + w, rtol=1e-4, atol=1e-6, what='cumulative')
        close(q_out, q_ref, rtol=1e-4, atol=1e-5, what='query')
        # against the two-launch path (bit-level agreement is not required: the context is summed in another order)
        q2 = torch.empty(B, A, device=dev); en = torch.empty(B, Ti, device=dev); u2 = torch.empty(B, Ti, A, device=dev)
        ops.call('gt_attn_energies_fwd_u', d, ops.slabs(qd, nslabs, B * A, A), f32(w_prev), f32(w_cum), pm_d, Wc, Wd, v, lens32, q2, en,
                 wpack, u2)
        w2 = torch.empty(B, Ti, device=dev); cu2 = torch.empty(B, Ti, device=dev); cc = torch.empty(B, E, device=dev)
        ops.call('gt_attn_context_fwd', d, en, mem_d, E, lens32, w2, f32(w_cum), cu2, cc, E, None, 0, None, 0)
        close(w_out, w2, rtol=1e-6, atol=1e-9, what='weights vs two-launch path')
        close(c0, cc, rtol=1e-5, atol=1e-6, what='context vs two-launch path')
        valid = (~pad)[:, :, None].to(dev)
        close(u_out * valid, u2 * valid, rtol=0, atol=0, what='tanh history')
    want = ((lens + 31) // 32 * 3).to(torch.int32)
    assert torch.equal(cnt.cpu(), want)          # every member workgroup of every step arrived exactly once


@pytest.mark.parametrize('B,Ti,E,nslabs', [(64, 187, 1024, 64), (9, 150, 1024, 5), (5, 77, 512, 1), (2, 33, 1024, 64), (17, 256, 640, 3)])
def test_attention_step2_partial_contexts(dev, B, Ti, E, nslabs):
    """gt_attn_step2_fwd = gt_attn_energies_fwd_u + gt_attn_context_fwd in ONE launch with NO in-launch wait: every chunk
    workgroup forms softmax numerators against its own maximum and a partial context, the row's last arriver combines
    them (online softmax).  Weights, cumulative weights, context (three destinations), query and tanh history against the
    oracle's attention_step and against the two-launch path, for three consecutive steps on the same tickets (ragged
    lengths: rows with 1 .. NC member workgroups; one row with a huge energy spread between its chunks)."""
    from gantron_amd import ops
    from gantron_amd._lib import lib
    from oracle import gantron_ref as R
    A, Fn, Kc = 128, 32, 31
    d = ops.attn_dims(B, Ti, E, A, Fn, Kc)
    assert lib().gt_attn_step2_supported(d)
    pre = 'decoder.attention_layer.'
    P = {pre + 'query_layer.linear_layer.weight': g(A, 16, scale=0.3),
         pre + 'location_layer.location_conv.conv.weight': g(Fn, 2, Kc, seed=1, scale=0.3),
         pre + 'location_layer.location_dense.linear_layer.weight': g(A, Fn, seed=2, scale=0.3),
         pre + 'v.linear_layer.weight': g(1, A, seed=3, scale=0.5)}
    gen = torch.Generator().manual_seed(B * 1000 + Ti)
    lens = torch.randint(1, Ti + 1, (B,), generator=gen)
    lens[0] = Ti
    if B > 1:
        lens[1] = 1
    memory = g(B, Ti, E, seed=4)
    pm = g(B, Ti, A, seed=5, scale=0.5)
    if Ti > 64:
        pm[0, 40:64] += 4.0          # chunk 1 of row 0 far above the others: the combine must rescale by e^{m_c - M}
    pad = ~R.length_mask(lens, Ti)
    f32 = lambda t: t.detach().float().contiguous().to(dev)
    Wc, Wd, v = (f32(P[pre + 'location_layer.location_conv.conv.weight']),
                 f32(P[pre + 'location_layer.location_dense.linear_layer.weight']), f32(P[pre + 'v.linear_layer.weight']))
    wpack = torch.empty(int(lib().gt_attn_wpack_elems(d)), device=dev)
    ops.call('gt_attn_pack_weights', d, Wc, Wd, wpack)
    lens32 = lens.to(torch.int32).to(dev)
    mem_d, pm_d = f32(memory), f32(pm)
    nch = int(lib().gt_attn_chunks(Ti))
    xbuf = torch.full((B * nch * int(lib().gt_attn_step2_xstride(d)),), float('nan'), device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    for step in range(3):
        query = g(B, 16, seed=60 + step)
        w_prev = torch.softmax(g(B, Ti, seed=70 + step), 1) * (~pad)
        w_cum = torch.softmax(g(B, Ti, seed=80 + step), 1) * 3 * (~pad)
        ctx, w = R.attention_step({k: t.double() for k, t in P.items()}, query.double(), memory.double(), pm.double(),
                                  w_prev.double(), w_cum.double(), pad)
        q_ref = query.double() @ P[pre + 'query_layer.linear_layer.weight'].double().t()
        parts = torch.randn(nslabs, B, A, generator=gen) * 0.1
        parts[0] += q_ref.float() - parts.sum(0)
        qd = parts.contiguous().to(dev)
        q_out = torch.empty(B, A, device=dev); u_out = torch.zeros(B, Ti, A, device=dev)
        w_out = torch.full((B, Ti), float('nan'), device=dev); cum_out = torch.full((B, Ti), float('nan'), device=dev)
        c0 = torch.full((B, E), float('nan'), device=dev); c1 = torch.zeros(B, E + 8, device=dev); c2 = torch.zeros(B, E, device=dev)
        ops.call('gt_attn_step2_fwd', d, ops.slabs(qd, nslabs, B * A, A), f32(w_prev), f32(w_cum), pm_d, Wc, Wd, v, lens32, q_out,
                 wpack, u_out, xbuf, cnt, step + 1, mem_d, E, w_out, f32(w_cum), cum_out, c0, E, ops.ptr(c1, 8), E + 8, c2, E)
        close(w_out, w, rtol=1e-4, atol=1e-6, what=f'attention weights (step {step})')
        close(c0, ctx, rtol=1e-4, atol=1e-5, what=f'context (step {step})')
        close(c1[:, 8:], ctx, rtol=1e-4, atol=1e-5, what='context strided')
        close(c2, ctx, rtol=1e-4, atol=1e-5, what='context third destination')
        close(cum_out, w_cum.double() + w, rtol=1e-4, atol=1e-6, what='cumulative')
        close(q_out, q_ref, rtol=1e-4, atol=1e-5, what='query')
        q2 = torch.empty(B, A, device=dev); en = torch.empty(B, Ti, device=dev); u2 = torch.zeros(B, Ti, A, device=dev)
        ops.call('gt_attn_energies_fwd_u', d, ops.slabs(qd, nslabs, B * A, A), f32(w_prev), f32(w_cum), pm_d, Wc, Wd, v, lens32, q2, en,
                 wpack, u2)
        w2 = torch.empty(B, Ti, device=dev); cu2 = torch.empty(B, Ti, device=dev); cc = torch.empty(B, E, device=dev)
        ops.call('gt_attn_context_fwd', d, en, mem_d, E, lens32, w2, f32(w_cum), cu2, cc, E, None, 0, None, 0)
        close(w_out, w2, rtol=1e-5, atol=1e-8, what='weights vs two-launch path')
        close(c0, cc, rtol=1e-5, atol=1e-6, what='context vs two-launch path')
        valid = (~pad)[:, :, None].to(dev)
        close(u_out * valid, u2 * valid, rtol=0, atol=0, what='tanh history')
    want = ((lens + 31) // 32 * 3).to(torch.int32)
    assert torch.equal(cnt.cpu(), want)          # every member workgroup of every step drew exactly one ticket


@pytest.mark.parametrize('Bsz,H,E,S,Sx,B,Ti', [(64, 1024, 1024, 3, 3, 64, 187), (64, 1024, 1024, 2, 2, 3, 40), (5, 32, 64, 2, 2, 5, 77),
                                               (33, 64, 48, 1, 1, 33, 150), (48, 128, 512, 3, 4, 9, 256)])
def test_rec_tail_rides_in_the_attention_step(dev, Bsz, H, E, S, Sx, B, Ti):
    """GtRecTail: the h half of [ctx | h] . W^T is cut into (column group, K-slice) items that the workgroups of an attention
    step pick up when they have nothing (more) of their own to do (gt_attn_step2_fwd_tail), and the LSTM step behind it
    multiplies the ctx half only and adds those slabs (gt_rec_cell_fwd_part).  Three rounds on the same workspaces: the
    attention step's own results equal those of the launch without a tail, every item is computed exactly once (the
    slabs start as NaN, the work counter ends at items + workgroups), and cell state / h / gates equal the float64 LSTM
    step over ALL of K.  (B, Ti): the attention batch -- few rows = few workgroups for many items; ragged lengths = chunk
    workgroups that lie in the padding and start on the items at once.)"""
    from gantron_amd import ops
    from gantron_amd._lib import GtRecTail, lib
    from gantron_amd.engine import pack_rec_cell
    A, Fn, Kc = 128, 32, 31
    d = ops.attn_dims(B, Ti, 1024, A, Fn, Kc)
    gen = torch.Generator().manual_seed(B * 1000 + Ti + H)
    lens = torch.randint(1, Ti + 1, (B,), generator=gen)
    lens[0] = Ti
    f32 = lambda t: t.detach().float().contiguous().to(dev)
    Wc, Wd, v = f32(g(Fn, 2, Kc, seed=1, scale=0.3)), f32(g(A, Fn, seed=2, scale=0.3)), f32(g(1, A, seed=3, scale=0.5))
    wpack = torch.empty(int(lib().gt_attn_wpack_elems(d)), device=dev)
    ops.call('gt_attn_pack_weights', d, Wc, Wd, wpack)
    lens32 = lens.to(torch.int32).to(dev)
    mem_d, pm_d = f32(g(B, Ti, 1024, seed=4)), f32(g(B, Ti, A, seed=5, scale=0.5))
    nch = int(lib().gt_attn_chunks(Ti))
    xs = int(lib().gt_attn_step2_xstride(d))
    n_wg = -(-B // 8) * 8 * nch
    # the LSTM step
    K = E + H
    Bp = ops.roundup(Bsz, 16)
    ld = ops.roundup(K, 16) + 16
    N = 4 * H
    W = g(N, K, scale=0.1)
    Wp = pack_rec_cell(W.to(dev), H, K, False)
    S = ops.pick_splits(E, S)
    Sx = ops.pick_splits(H, Sx)
    kbs = -(-(H // 16) // Sx)
    items = (N // 64) * Sx
    part = torch.full((S + Sx, Bp, N), float('nan'), device=dev)
    counters = torch.zeros(H // 16, dtype=torch.int32, device=dev)
    p = 0.1
    for rep in range(3):
        X = torch.zeros(Bp, ld)
        X[:Bsz, :K] = g(Bsz, K, seed=10 + rep)
        Xd = X.to(dev)
        outs = []
        work = torch.zeros(1, dtype=torch.int32, device=dev)
        part[S:] = float('nan')
        for with_tail in (False, True):
            tl = GtRecTail()
            if with_tail:
                tl.X, tl.ldx, tl.Wp, tl.out, tl.out_sstride = ops.ptr(Xd), ld, ops.ptr(Wp), ops.ptr(part, S * Bp * N), Bp * N
                tl.N, tl.KB, tl.kb0, tl.kb1, tl.kbs, tl.cg0, tl.groups, tl.mtiles = N, -(-K // 16), E // 16, -(-K // 16), kbs, 0, N // 64, Bp // 16
                tl.work = ops.ptr(work)
            xbuf = torch.full((B * nch * xs,), float('nan'), device=dev)
            cnt = torch.zeros(B, dtype=torch.int32, device=dev)
            qd = f32(g(2, B, A, seed=60 + rep, scale=0.1))
            w_prev, w_cum = f32(torch.softmax(g(B, Ti, seed=70 + rep), 1)), f32(torch.softmax(g(B, Ti, seed=80 + rep), 1) * 3)
            q_out = torch.empty(B, A, device=dev); u_out = torch.zeros(B, Ti, A, device=dev)
            w_out = torch.zeros(B, Ti, device=dev); cum_out = torch.zeros(B, Ti, device=dev)
            c0 = torch.zeros(B, 1024, device=dev)
            ops.call('gt_attn_step2_fwd_tail', d, ops.slabs(qd, 2, B * A, A), w_prev, w_cum, pm_d, Wc, Wd, v, lens32, q_out, wpack,
                     u_out, xbuf, cnt, 1, mem_d, 1024, w_out, w_cum, cum_out, c0, 1024, None, 0, None, 0, tl)
            outs.append((w_out, cum_out, c0, q_out, u_out))
        for a_, b_, what in zip(outs[0], outs[1], ('weights', 'cumulative weights', 'context', 'query', 'tanh history')):
            if what == 'context':       # summed in the order the row's LAST arriver sees its partials: not the same in every launch
                close(a_, b_, rtol=1e-5, atol=1e-6, what='context with / without a tail')
                continue
            assert torch.equal(a_, b_), f'the attention step changed its {what} when it carried a tail'
        assert int(work.item()) == items + n_wg, (int(work.item()), items, n_wg)      # every workgroup drew until the counter ran dry
        # slab columns are in the packed order of gt_pack_rec_cell_weight: column 64 cg + 16 gate + u = gate `gate` of unit 16 cg + u
        n_p = torch.arange(N)
        nat = ((n_p // 16) % 4) * H + 16 * (n_p // 64) + n_p % 16
        tail_ref = (X[:Bsz, E:K].double() @ W[:, E:].double().t())[:, nat]
        assert torch.isfinite(part[S:, :Bsz]).all()
        close(part[S:].sum(0)[:Bsz], tail_ref, rtol=1e-5, atol=2e-5, what=f'tail slabs (round {rep})')
        if Bsz < Bp:
            assert float(part[S:, Bsz:].abs().max()) == 0.0           # zero operand rows give zero slab rows
        # the LSTM step over the ctx half + the slabs
        pre = g(Bp, N, seed=20 + rep, scale=0.5)
        c_prev = g(Bp, H, seed=30 + rep)
        keep = (torch.rand(Bp, H, generator=torch.Generator().manual_seed(40 + rep)) >= p)
        gates_in = X[:Bsz, :K].double() @ W.double().t() + pre[:Bsz].double()
        i, f, gg, o = gates_in.chunk(4, dim=1)
        c = torch.sigmoid(f) * c_prev[:Bsz].double() + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c) * keep[:Bsz].double() / (1 - p)
        gates = pre.clone().to(dev)
        c_out = torch.zeros(Bp, H, device=dev); h_raw = torch.zeros(Bp, H, device=dev)
        ops.call('gt_rec_cell_fwd_part', Xd, ld, Wp, part, counters, Bp, Bsz, H, K, S, gates, N, c_prev.to(dev),
                 keep.to(torch.uint8).to(dev), p, gates, c_out, h_raw, None, 0, None, 0, None, 0, None, 0, None, 0, 0, 16, E, Sx, 3)
        close(c_out[:Bsz], c, rtol=1e-5, atol=2e-5, what=f'c (round {rep})')
        close(h_raw[:Bsz], h, rtol=1e-5, atol=2e-5, what=f'h (round {rep})')
        close(gates[:Bsz], torch.cat((torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)), 1), rtol=1e-5,
              atol=2e-5, what='gates')


@pytest.mark.parametrize('B,Ti', [(3, 150), (5, 77), (2, 33)])
def test_attention_hoisted_backward_equals_in_loop(dev, B, Ti):
    """gt_attn_energies_fwd_u + gt_attn_energies_bwd_rec per step + ONE gt_attn_param_grads over the steps == the
    in-loop gt_attn_energies_bwd of every step (itself checked against autograd above): dq, d[w_prev; w_cum] per
    step, and d(processed memory), dv, dWd, dWc summed over 4 steps."""
    from gantron_amd import ops
    from gantron_amd._lib import lib
    A, Fn, Kc, S = 128, 32, 31, 4
    d = ops.attn_dims(B, Ti, 64, A, Fn, Kc)
    assert lib().gt_attn_hoist_supported(d) == 1
    lens = torch.tensor([Ti, Ti - 7, max(Ti - 20, 3), Ti - 1, 5][:B])
    lens32 = lens.to(torch.int32).to(dev)
    Wc, Wd, v = g(Fn, 2, Kc, seed=1, scale=0.3).to(dev), g(A, Fn, seed=2, scale=0.3).to(dev), g(1, A, seed=3, scale=0.5).to(dev)
    pm = g(B, Ti, A, seed=5, scale=0.5).to(dev)
    wpack = torch.empty(int(lib().gt_attn_wpack_elems(d)), device=dev)
    ops.call('gt_attn_pack_weights', d, Wc, Wd, wpack)
    w_prev = torch.stack([torch.softmax(g(B, Ti, seed=20 + s), 1) for s in range(S)]).to(dev)
    w_cum = torch.stack([torch.softmax(g(B, Ti, seed=30 + s), 1) * (s + 1) for s in range(S)]).to(dev)
    q = torch.stack([g(B, A, seed=40 + s) for s in range(S)]).to(dev)
    dw = torch.stack([g(B, Ti, seed=50 + s, scale=0.3) for s in range(S)]).to(dev)
    w = torch.empty(S, B, Ti, device=dev)
    u = torch.full((S, B, Ti, A), float('nan'), device=dev)
    en = torch.empty(B, Ti, device=dev); q_out = torch.empty(B, A, device=dev)
    for s in range(S):
        ops.call('gt_attn_energies_fwd_u', d, ops.slabs(q[s], 1, B * A, A), w_prev[s], w_cum[s], pm, Wc, Wd, v, lens32, q_out, en,
                 wpack, u[s])
        w[s] = torch.softmax(en, 1)
    # in-loop reference
    dpm0 = torch.zeros(B, Ti, A, device=dev); acc0 = torch.zeros(int(lib().gt_attn_acc_elems(d)), device=dev)
    dq0 = torch.zeros(S, B, A, device=dev); dwc0 = torch.zeros(S, 2, B, Ti, device=dev)
    for s in range(S):
        ops.call('gt_attn_energies_bwd_w', d, dw[s], w[s], q[s], w_prev[s], w_cum[s], pm, Wc, Wd, v, lens32, dq0[s], A, dpm0,
                 dwc0[s], acc0, wpack)
    # hoisted
    dpm1 = torch.zeros(B, Ti, A, device=dev); acc1 = torch.zeros_like(acc0)
    dq1 = torch.zeros(S, B, A, device=dev); dwc1 = torch.zeros(S, 2, B, Ti, device=dev)
    de = torch.full((S, B, Ti), float('nan'), device=dev)
    for s in range(S):
        ops.call('gt_attn_energies_bwd_rec', d, dw[s], w[s], u[s], Wc, Wd, v, lens32, dq1[s], A, dwc1[s], de[s], wpack)
    ops.call('gt_attn_param_grads', d, 2, u, de, w_prev, w_cum, B * Ti, Wc, Wd, v, lens32, dpm1, acc1, wpack)       # steps 0-1
    ops.call('gt_attn_param_grads', d, S - 2, u[2:], de[2:], w_prev[2:], w_cum[2:], B * Ti, Wc, Wd, v, lens32, dpm1, acc1, wpack)
    close(dq1, dq0, rtol=1e-5, atol=1e-7, what='dq')
    close(dwc1, dwc0, rtol=1e-5, atol=1e-7, what='d[w_prev; w_cum]')
    close(dpm1, dpm0, rtol=1e-5, atol=1e-7, what='dpm')
    outs = []
    for acc in (acc0, acc1):
        dv = torch.zeros(1, A, device=dev); dWd = torch.zeros(A, Fn, device=dev); dWc = torch.zeros(Fn, 2, Kc, device=dev)
        ops.call('gt_attn_acc_reduce', d, acc, dv, dWd, dWc)
        outs.append((dv, dWd, dWc))
    for name, a, b in zip(('dv', 'dWd', 'dWc'), outs[1], outs[0]):
        close(a, b, rtol=1e-5, atol=1e-6, what=name)


@pytest.mark.parametrize('B,Ti,E', [(3, 150, 1024), (5, 77, 64), (2, 33, 1096), (64, 187, 1024)])
def test_attention_bwd_step_fused(dev, B, Ti, E):
    """gt_attn_bwd_wterms (once) + gt_attn_bwd_step per step (+ gt_attn_bwd_fixup) == gt_attn_context_bwd +
    gt_attn_energies_bwd_rec per step, over a CHAIN of 4 backward steps (the carried alignment gradients of a step come out
    of the step after it): the softmax backward's row sum is applied by the consumers instead of inside the step."""
    from gantron_amd import ops
    from gantron_amd._lib import lib
    A, Fn, Kc, S = 128, 32, 31, 4
    d = ops.attn_dims(B, Ti, E, A, Fn, Kc)
    nch = int(lib().gt_attn_chunks(Ti))
    lens = torch.tensor(([Ti, Ti - 7, max(Ti - 20, 3), Ti - 1, 5] * 13)[:B])
    lens32 = lens.to(torch.int32).to(dev)
    Wc, Wd, v = g(Fn, 2, Kc, seed=1, scale=0.3).to(dev), g(A, Fn, seed=2, scale=0.3).to(dev), g(1, A, seed=3, scale=0.5).to(dev)
    wpack = torch.empty(int(lib().gt_attn_wpack_elems(d)), device=dev)
    ops.call('gt_attn_pack_weights', d, Wc, Wd, wpack)
    memory = g(B, Ti, E, seed=4, scale=0.5).to(dev)
    mask = (torch.arange(Ti)[None, :] < lens[:, None])
    w = torch.stack([torch.softmax(g(B, Ti, seed=20 + s).masked_fill(~mask, float('-inf')), 1) for s in range(S)]).to(dev)
    u = torch.stack([torch.tanh(g(B, Ti, A, seed=30 + s)) for s in range(S)]).to(dev)
    dctx_dir = torch.stack([g(B, E, seed=40 + s, scale=0.2) for s in range(S)]).to(dev)
    dslab = torch.stack([g(3, B, E + 8, seed=50 + s, scale=0.2) for s in range(S)]).to(dev)      # 3 K-split slabs, offset columns
    dalign = torch.stack([g(B, Ti, seed=60 + s, scale=0.1) for s in range(S)]).to(dev)
    Apad = 144

    def run(fused):
        dwcat = torch.zeros(2, 2, B, Ti, device=dev); dcum = torch.zeros(B, Ti, device=dev)
        dq = torch.zeros(S, B, Apad, device=dev); de = torch.full((S, B, Ti), float('nan'), device=dev)
        dctx_out = torch.full((S, B, E), float('nan'), device=dev)
        dw = torch.empty(B, Ti, device=dev)
        if fused:
            sums = torch.full((S, B, nch), float('nan'), device=dev)
            dq2 = torch.zeros(S, B, Apad, device=dev); dwc2 = torch.zeros(S, 2, B, Ti, device=dev)
            ph = torch.full((S, B, Ti, 64), float('nan'), device=dev)
            ops.call('gt_attn_bwd_wterms', d, S, w, B * Ti, u, Wc, Wd, v, lens32, dq2, B * Apad, Apad, dwc2, ph, wpack)
        for s in reversed(range(S)):             # backward order
            din, dout = dwcat[(s + 1) & 1], dwcat[s & 1]
            sl = ops.slabs(dslab[s], 3, B * (E + 8), E + 8, offset=4)
            if fused:
                prev = s + 1 < S
                ops.call('gt_attn_bwd_step', d, sl, ops.NO_SLABS, dctx_dir[s], E, memory, E, lens32, din, dcum,
                         dalign[s], dctx_out[s], E, w[s], u[s], v, ph[s], sums[s + 1] if prev else None,
                         dwc2[s + 1] if prev else None, sums[s], dq[s], Apad, dout, de[s])
            else:
                ops.call('gt_attn_context_bwd', d, sl, ops.NO_SLABS, dctx_dir[s], E, memory, E, lens32, din, dcum,
                         dalign[s], dctx_out[s], E, dq[s], Apad, dw)
                ops.call('gt_attn_energies_bwd_rec', d, dw, w[s], u[s], Wc, Wd, v, lens32, dq[s], Apad, dout, de[s], wpack)
        if fused:
            ops.call('gt_attn_bwd_fixup', d, S, sums, w, B * Ti, de, dq, B * Apad, Apad, dq2)
            last = dwcat[0] - sums[0].sum(1)[None, :, None] * dwc2[0]     # what a step -1 would read as its carried planes
        else:
            last = dwcat[0].clone()
        return dq, de, dctx_out, dcum, last

    ref, got = run(False), run(True)
    for name, a, bb in zip(('dq', 'de', 'dctx', 'dcum_acc', 'carried d[w_prev; w_cum]'), got, ref):
        close(a, bb, rtol=2e-5, atol=2e-6, what=name)


@pytest.mark.parametrize('B,Ti,E,H', [(3, 150, 1024, 1024), (5, 77, 64, 48), (64, 187, 1024, 1024), (7, 40, 512, 1280)])
def test_attention_bwd_step_with_cell(dev, B, Ti, E, H):
    """gt_attn_bwd_step_cell == gt_attn_bwd_step + gt_cell_bwd_query_c over a chain of 4 backward steps that share the row
    counters (launch = 1 .. 4) and carry dc from step to step: gate gradients and dc of the attention LSTM, computed by each
    row's last workgroup inside the attention step's launch, against the two-launch form (same dq: atomics, so 1e-5)."""
    import ctypes
    from gantron_amd import ops
    from gantron_amd._lib import lib, STRUCTS
    A, Fn, Kc, S = 128, 32, 31, 4
    d = ops.attn_dims(B, Ti, E, A, Fn, Kc)
    nch = int(lib().gt_attn_chunks(Ti))
    Bp = 16 * ((B + 15) // 16)
    lens = torch.tensor(([Ti, Ti - 7, max(Ti - 20, 3), Ti - 1, 5] * 13)[:B])
    lens32 = lens.to(torch.int32).to(dev)
    Wc, Wd, v = g(Fn, 2, Kc, seed=1, scale=0.3).to(dev), g(A, Fn, seed=2, scale=0.3).to(dev), g(1, A, seed=3, scale=0.5).to(dev)
    wpack = torch.empty(int(lib().gt_attn_wpack_elems(d)), device=dev)
    ops.call('gt_attn_pack_weights', d, Wc, Wd, wpack)
    memory = g(B, Ti, E, seed=4, scale=0.5).to(dev)
    mask = (torch.arange(Ti)[None, :] < lens[:, None])
    w = torch.stack([torch.softmax(g(B, Ti, seed=20 + s).masked_fill(~mask, float('-inf')), 1) for s in range(S)]).to(dev)
    u = torch.stack([torch.tanh(g(B, Ti, A, seed=30 + s)) for s in range(S)]).to(dev)
    dctx_dir = torch.stack([g(B, E, seed=40 + s, scale=0.2) for s in range(S)]).to(dev)
    dslab = torch.stack([g(3, B, E + 8, seed=50 + s, scale=0.2) for s in range(S)]).to(dev)
    Apad = 144
    # the cell's side: gates in (0, 1) / (-1, 1), cell states, a dropout mask, d h_att from two sources, Wq
    gates = torch.stack([torch.cat((torch.sigmoid(g(Bp, 2 * H, seed=70 + s)), torch.tanh(g(Bp, H, seed=75 + s)),
                                    torch.sigmoid(g(Bp, H, seed=80 + s))), 1) for s in range(S)]).to(dev)
    cst = g(S + 1, Bp, H, seed=85).to(dev)
    keep = (torch.rand(S, Bp, H, generator=torch.Generator().manual_seed(86)) >= 0.1).to(torch.uint8).to(dev)
    dha = g(S, Bp, H + 4, seed=87, scale=0.2).to(dev)                    # one slab, ld = H + 4
    dhc = g(S, 2, Bp, H + 8, seed=88, scale=0.2).to(dev)                 # two slabs, offset columns
    Wq = g(A, H, seed=89, scale=0.1).to(dev)

    def run(cell_inside):
        dwcat = torch.zeros(2, 2, B, Ti, device=dev); dcum = torch.zeros(B, Ti, device=dev)
        dq = torch.zeros(S, Bp, Apad, device=dev); de = torch.full((S, B, Ti), float('nan'), device=dev)
        dctx_out = torch.full((S, B, E), float('nan'), device=dev)
        sums = torch.full((S, B, nch), float('nan'), device=dev)
        dq2 = torch.zeros(S, Bp, Apad, device=dev); dwc2 = torch.zeros(S, 2, B, Ti, device=dev)
        ph = torch.full((S, B, Ti, 64), float('nan'), device=dev)
        ops.call('gt_attn_bwd_wterms', d, S, w, B * Ti, u, Wc, Wd, v, lens32, dq2, Bp * Apad, Apad, dwc2, ph, wpack)
        dG = torch.full((S, Bp, 4 * H), 7.0, device=dev)
        dc = torch.zeros(Bp, H, device=dev)
        cnt = torch.zeros(B, dtype=torch.int32, device=dev)
        amax = torch.zeros(S, dtype=torch.int64, device=dev)    # GtCellBwdTail.dg_amax: max |dgates| of each step
        for n, s in enumerate(reversed(range(S))):             # backward order
            din, dout = dwcat[(s + 1) & 1], dwcat[s & 1]
            sl = ops.slabs(dslab[s], 3, B * (E + 8), E + 8, offset=4)
            prev = s + 1 < S
            a_sl = ops.slabs(dha[s], 1, 0, H + 4)
            c_sl = ops.slabs(dhc[s], 2, Bp * (H + 8), H + 8, offset=4) if prev else ops.NO_SLABS
            common = (d, sl, ops.NO_SLABS, dctx_dir[s], E, memory, E, lens32, din, dcum, None, dctx_out[s], E, w[s], u[s], v, ph[s],
                      sums[s + 1] if prev else None, dwc2[s + 1] if prev else None, sums[s], dq[s], Apad, dout, de[s])
            if cell_inside:
                c = STRUCTS['GtCellBwdTail']()
                c.dh_a, c.dh_c = a_sl, c_sl
                c.dq2, c.Wq, c.ldwq = ops.ptr(dq2[s]), ops.ptr(Wq), H
                c.dc_next, c.gates, c.c_new, c.c_prev = ops.ptr(dc), ops.ptr(gates[s]), ops.ptr(cst[s + 1]), ops.ptr(cst[s])
                c.keep, c.p_drop, c.dgates, c.dc_prev, c.H = ops.ptr(keep[s]), 0.1, ops.ptr(dG[s]), ops.ptr(dc), H
                c.row_cnt, c.launch = ops.ptr(cnt), n + 1
                c.dg_amax = ops.ptr(amax, s)
                ops.call('gt_attn_bwd_step_cell', *common, ctypes.byref(c))
            else:
                ops.call('gt_attn_bwd_step', *common)
                ops.call('gt_cell_bwd_query_c', a_sl, c_sl, dq[s], Apad, sums[s], nch, dq2[s], Wq, H, dc, gates[s], cst[s + 1], cst[s],
                         keep[s], 0.1, dG[s], dc, B, Bp, H, A)
        if cell_inside:         # the words: tag 1 in the high half, the bits of the step's largest |gate gradient| in the low one
            want = dG[:, :B].abs().amax(dim=(1, 2)).view(torch.int32).to(torch.int64) | (1 << 32)
            assert torch.equal(amax, want), (amax, want)
        return dG[:, :B], dc[:B], dq[:, :B], de

    ref, got = run(False), run(True)
    for name, a, bb in zip(('dgates', 'dc', 'dq', 'e1'), got, ref):
        close(a, bb, rtol=2e-5, atol=2e-6, what=name)
    assert bool((got[0] != 7.0).all()), 'every gate gradient of the valid rows must have been written'


# ------------------------------------------------------------------------- BN / elementwise
@pytest.mark.parametrize('det', [True, False])
@pytest.mark.parametrize('rows,C,act', [(300, 24, 1), (1000, 80, 2), (77, 8, 0)])
def test_bn_act_drop(dev, rows, C, act, det):
    from gantron_amd import ops
    x = (g(rows, C) * 2 + 5).requires_grad_(True)
    gamma, beta = (1 + 0.1 * g(C, seed=1)).requires_grad_(True), (0.1 * g(C, seed=2)).requires_grad_(True)
    keep = torch.rand(rows, C, generator=torch.Generator().manual_seed(3)) >= 0.5
    rm, rv = g(C, seed=4) * 0.1, torch.ones(C) * 0.9
    xd = x.double()
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    z = (xd - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
    a = torch.relu(z) if act == 1 else (torch.tanh(z) if act == 2 else z)
    y = a * keep.double() * 2
    dy = g(rows, C, seed=5)
    gx, gg, gb = torch.autograd.grad(y, (x, gamma, beta), dy.double())
    xg = x.detach().to(dev); stats = torch.empty(2 * C, device=dev)
    ws = ops.col_ws(rows, C, dev) if det else None
    ops.call('gt_col_stats', xg, xg, rows, C, stats, ws)
    yo = torch.empty(rows, C, device=dev); mi = torch.empty(2 * C, device=dev)
    rmd, rvd = rm.clone().to(dev), rv.clone().to(dev)
    k8 = keep.to(torch.uint8).to(dev)
    ops.call('gt_bn_act_drop_fwd', xg, stats, rows, C, gamma.detach().to(dev), beta.detach().to(dev), rmd, rvd, 0.1, 1e-5, act,
             k8, 0.5, yo, mi, 1)
    close(yo, y, rtol=1e-4, atol=1e-5, what='bn fwd')
    close(rmd, 0.9 * rm + 0.1 * mean, rtol=1e-4, atol=1e-6, what='running mean')
    close(rvd, 0.9 * rv + 0.1 * xd.var(0, unbiased=True), rtol=1e-4, atol=1e-6, what='running var')
    dz = torch.empty(rows, C, device=dev); sums = torch.empty(2 * C, device=dev)
    ops.call('gt_bn_act_drop_bwd1', dy.to(dev), xg, mi, gamma.detach().to(dev), beta.detach().to(dev), rows, C, act, k8, 0.5, dz, sums,
             ops.col_ws(rows, C, dev) if det else None)
    dx = torch.empty(rows, C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    ops.call('gt_bn_act_drop_bwd2', dz, xg, mi, gamma.detach().to(dev), sums, rows, C, dx, dg, db)
    close(dx, gx, rtol=1e-3, atol=1e-5, what='bn dx')
    close(dg, gg, rtol=1e-3, atol=1e-4, what='bn dgamma')
    close(db, gb, rtol=1e-3, atol=1e-4, what='bn dbeta')


def test_elementwise_misc(dev):
    from gantron_amd import ops
    # bias + dropout + tanh (DiscConv1d order) and relu + dropout (prenet order), fwd/bwd
    rows, C = 50, 12
    x, bias = g(rows, C).requires_grad_(True), g(C, seed=1)
    keep = torch.rand(rows, C, generator=torch.Generator().manual_seed(2)) >= 0.5
    dy = g(rows, C, seed=3)
    y1 = torch.tanh((x.double() + bias.double()) * keep.double() * 2)
    y0 = torch.relu(x.double()) * keep.double() * 2
    g1, = torch.autograd.grad(y1, x, dy.double(), retain_graph=True)
    g0, = torch.autograd.grad(y0, x, dy.double())
    k8 = keep.to(torch.uint8).to(dev)
    for order, act, b, yref, gref in ((1, 2, bias.to(dev), y1, g1), (0, 1, None, y0, g0)):
        yo = torch.empty(rows, C, device=dev)
        ops.call('gt_bias_act_drop_fwd', x.detach().to(dev), b, rows, C, act, order, k8, 0.5, yo)
        close(yo, yref, rtol=1e-5, atol=1e-6, what=f'bias_act_drop fwd {order}')
        dx = torch.empty(rows, C, device=dev)
        ops.call('gt_bias_act_drop_bwd', dy.to(dev), yo, rows, C, act, order, k8, 0.5, dx)
        close(dx, gref, rtol=1e-4, atol=1e-6, what=f'bias_act_drop bwd {order}')
    # embedding gather / scatter-add
    ids = torch.tensor([[3, 1, 3], [0, 2, 1]])
    table = g(5, 8)
    out = torch.empty(6, 8, device=dev)
    ops.call('gt_embedding_fwd', ids.to(dev), table.to(dev), 6, 8, out, 8)
    close(out, table[ids.view(-1)], rtol=0, atol=0, what='embedding')
    dt = torch.zeros(5, 8, device=dev)
    do = g(6, 8, seed=4)
    ops.call('gt_embedding_bwd', ids.to(dev), do.to(dev), 8, 6, 8, dt)
    want = torch.zeros(5, 8).index_add_(0, ids.view(-1), do)
    close(dt, want, rtol=1e-6, atol=1e-6, what='embedding bwd')
    # gather rows with -1 and accumulate
    src = g(7, 6)
    idx = torch.tensor([2, -1, 6, 0], dtype=torch.int32)
    o = torch.ones(4, 9, device=dev)
    ops.call('gt_gather_rows', src.to(dev), 6, idx.to(dev), 4, 6, o, 9, 1)
    want = torch.ones(4, 9)
    want[:, :6] += torch.stack([src[2], torch.zeros(6), src[6], src[0]])
    close(o, want, rtol=0, atol=1e-7, what='gather rows')
    # mask outputs
    mel, post, gate = g(2, 4, 6).to(dev), g(2, 4, 6, seed=1).to(dev), g(2, 6, seed=2).to(dev)
    m0 = mel.clone().cpu()
    ops.call('gt_mask_outputs', mel, post, gate, torch.tensor([6, 4]).to(dev), 2, 4, 6)
    m0[1, :, 4:] = 0
    close(mel, m0, rtol=0, atol=0, what='mask mel')
    assert float(post[1, :, 4:].abs().max()) == 0 and torch.all(gate[1, 4:] == 1e3) and torch.all(gate[0].cpu() != 1e3)


def test_rng_matches_numpy(dev):
    from gantron_amd import ops
    n, seed, off = 10000, 1234, 77
    u = torch.empty(n, device=dev)
    ops.call('gt_uniform', seed, off, n, u)
    ref = u01(seed, off, n)
    assert np.array_equal(u.cpu().numpy(), ref)
    k = torch.empty(n, dtype=torch.uint8, device=dev)
    ops.call('gt_keep_mask', seed, off, 0.1, n, k)
    assert np.array_equal(k.cpu().numpy(), (ref >= np.float32(0.1)).astype(np.uint8))
    assert 0.88 < k.float().mean().item() < 0.92


# --------------------------------------------------------------------- discriminator pieces, losses
def test_disc_gather_and_adv_mean(dev, golden_dir):
    import os
    from gantron_amd import ops
    from oracle import gantron_ref as R
    z = np.load(os.path.join(golden_dir, 'ops.npz'))
    x = torch.from_numpy(z['disc_reshape_in'])                 # (2, 47, 80) frames x mel
    mel = x.permute(0, 2, 1).contiguous()                       # (B, n_mel, T)
    r = torch.empty(2 * 3, 1600, device=dev)
    ops.call('gt_disc_gather_fwd', mel.to(dev), 2, 80, 47, 20, r)
    want = torch.from_numpy(z['disc_reshape_out'])              # (2, 1600, 3)
    close(r.view(2, 3, 1600).permute(0, 2, 1), want, rtol=0, atol=0, what='disc gather vs reference fixture')
    # backward = transpose of the gather
    for T in (47, 60, 20, 41):
        B, W = 2, T // 20 + (1 if T % 20 else 0)
        m = g(B, 80, T).double().requires_grad_(True)
        rr = R.disc_window_input(m.transpose(1, 2), 20)         # (B, 1600, W)
        dr = g(B, 1600, W, seed=1).double()
        gm, = torch.autograd.grad(rr, m, dr)
        dm = torch.empty(B, 80, T, device=dev)
        ops.call('gt_disc_gather_bwd', dr.permute(0, 2, 1).float().contiguous().to(dev), B, 80, T, 20, dm, 0)
        close(dm, gm, rtol=1e-6, atol=1e-6, what=f'disc gather bwd T={T}')
    outs, lens = torch.from_numpy(z['adv_outs']), torch.from_numpy(z['adv_lens'])
    loss = torch.empty(1, device=dev)
    ops.call('gt_adv_mean_fwd', outs.to(dev), lens.to(dev), 7, 3, 20, loss)
    close(loss, torch.tensor([float(z['adv_batch_loss'])]), rtol=1e-5, atol=1e-6, what='adv mean vs reference fixture')
    dout = torch.empty(7, 3, device=dev)
    ops.call('gt_adv_mean_bwd', torch.tensor([2.0]).to(dev), lens.to(dev), 7, 3, 20, 0.5, dout)
    nb = torch.ceil(lens.double() / 20)
    want = torch.zeros(7, 3, dtype=torch.double)
    for b in range(7):
        want[b, :int(nb[b])] = 1.0 / (7 * nb[b])
    close(dout, want, rtol=1e-5, atol=1e-7, what='adv mean bwd')


def test_losses(dev):
    from gantron_amd.loss_function import Tacotron2Loss
    from oracle import gantron_ref as R
    B, n_mel, To, Ti = 3, 8, 21, 9
    in_len, out_len = torch.tensor([9, 7, 4]), torch.tensor([21, 15, 11])
    mel, post = g(B, n_mel, To), g(B, n_mel, To, seed=1)
    gate = g(B, To, seed=2) * 3
    align = torch.softmax(g(B, To, Ti, seed=3), dim=2)
    mel_t, gate_t = g(B, n_mel, To, seed=4), (torch.rand(B, To, generator=torch.Generator().manual_seed(5)) > 0.7).float()
    leaves = [t.clone().requires_grad_(True) for t in (mel, post, gate, align)]
    ml, gl, al = R.tacotron_loss(leaves, mel_t, gate_t, in_len, out_len)
    ref_g = torch.autograd.grad(ml + 2 * gl + 10 * al, leaves)
    dl = [t.to(dev).requires_grad_(True) for t in (mel, post, gate, align)]
    m2, g2, a2 = Tacotron2Loss()(dl, (mel_t.to(dev), gate_t.to(dev)), in_len.to(dev), out_len.to(dev))
    close(m2, ml, rtol=1e-5, atol=1e-6, what='mel loss')
    close(g2, gl, rtol=1e-5, atol=1e-6, what='gate loss')
    close(a2, al, rtol=1e-4, atol=1e-6, what='attention loss')
    (m2 + 2 * g2 + 10 * a2).backward()
    for t, r, name in zip(dl, ref_g, ('dmel', 'dpost', 'dgate', 'dalign')):
        close(t.grad, r, rtol=1e-4, atol=1e-7, what=name)


def test_adam_and_clip(dev):
    from gantron_amd import ops
    from oracle import gantron_ref as R
    n = 5000
    p0, gr = g(n), g(n, seed=1) * 3
    P, st = {'w': p0.clone()}, {}
    grads = {'w': gr.clone()}
    R.clip_grad_norm(grads, 1.0)
    R.adam_step(P, grads, st, 1e-3, 1)
    grads2 = {'w': gr.clone() * 0.5}
    R.clip_grad_norm(grads2, 1.0)
    R.adam_step(P, grads2, st, 1e-3, 2)
    pd, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step, gsrc in ((1, gr), (2, gr * 0.5)):
        ss = torch.zeros(1, device=dev)
        gd = gsrc.to(dev)
        ops.call('gt_sumsq', gd, n, ss)
        ops.call('gt_adam_step', pd, gd, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 1e-6, step, ss, 1.0)
    close(pd, P['w'], rtol=1e-5, atol=1e-6, what='adam')


def test_adam_step_is_skipped_when_the_handoff_flag_is_up(dev):
    """gt_adam_step_guarded: with the flag at 0 it is gt_adam_step bit for bit; with the flag raised (an in-launch hand-off of
    the step gave up: its gradients are undefined) parameters and both moments stay untouched."""
    from gantron_amd import ops
    n = 100003
    p0, g0 = g(n, seed=1).to(dev), g(n, seed=2).to(dev)
    m0, v0 = g(n, seed=3, scale=0.1).to(dev), g(n, seed=4).abs().to(dev)
    ss = (g0.double() ** 2).sum().float().reshape(1).to(dev)
    outs = []
    for flag_val in (None, 0, 1):
        p, m, v = p0.clone(), m0.clone(), v0.clone()
        if flag_val is None:
            ops.call('gt_adam_step', p, g0, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 1e-6, 3, ss, 1.0)
        else:
            flag = torch.full((1,), flag_val, dtype=torch.int32, device=dev)
            ops.call('gt_adam_step_guarded', p, g0, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 1e-6, 3, ss, 1.0, flag)
        outs.append((p, m, v))
    for a_, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a_, b_)
    assert not torch.equal(outs[0][0], p0)
    for a_, b_ in zip(outs[2], (p0, m0, v0)):
        assert torch.equal(a_, b_)


def test_stft_mel(dev, golden_dir):
    import os
    from gantron_amd.stft import TacotronSTFT
    from oracle import gantron_ref as R
    z = np.load(os.path.join(golden_dir, 'ops.npz'))
    audio = torch.from_numpy(z['stft_audio'])
    stft = TacotronSTFT().to(dev)
    mag = stft.magnitudes(audio.to(dev))
    close(mag, z['stft_mag'], rtol=1e-4, atol=1e-4, what='stft magnitude vs reference fixture')
    mel = stft.mel_spectrogram(audio.to(dev))
    close(mel, R.mel_spectrogram(audio), rtol=1e-4, atol=1e-3, what='mel vs oracle')


def test_column_sums_are_bit_reproducible_with_a_workspace(dev):
    """gt_col_stats / gt_bn_act_drop_bwd1 with their scratch (what BatchNorm runs on): the row blocks' partial sums are
    added in row order by the last block to arrive, so the statistics carry the same bits in every run and under any
    concurrent load -- here: 20 runs at the encoder's and the PostNet's sizes while a GEMM keeps another stream busy.
    (With fp32 atomics the last bits follow the dispatch order; a pre-activation within rounding of the ReLU threshold
    then flips between runs: tools/enc_grad_probe.py.)"""
    from gantron_amd import ops
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev)
    for rows, C in [(64 * 187, 512), (55680, 80), (1000, 24)]:
        x = torch.randn(rows, C, device=dev) * 3 + 1
        outs = []
        for it in range(20):
            if it % 2:
                with torch.cuda.stream(side):
                    ops.linear(a, a, 4096, 4096, 4096)
            out = torch.empty(2 * C, device=dev)
            ops.call('gt_col_stats', x, x, rows, C, out, ops.col_ws(rows, C, dev))
            outs.append(out.clone())
        torch.cuda.synchronize()
        assert all(torch.equal(o, outs[0]) for o in outs[1:]), (rows, C)
        ref = torch.stack(((x - x[0]).double().sum(0), ((x - x[0]).double() ** 2).sum(0))).reshape(-1)
        close(outs[0], ref, rtol=1e-5, atol=1e-3, what=f'column sums {rows}x{C}')


def _lstm_reference(pre, Whh, keep, active, c0):
    """fp64 recurrence G_s = pre_s + h_{s-1} Whh^T with packed-sequence pass-through; returns gates, c (T+1), h_raw, h_carried."""
    T, Bp, N = pre.shape
    H = N // 4
    c, h = c0.double().clone(), torch.zeros(Bp, H, dtype=torch.float64)
    gates, cs, hraw, hcar = [], [c.clone()], [], []
    for s in range(T):
        G = pre[s].double() + h @ Whh.double().t()
        i, f, g, o = torch.sigmoid(G[:, :H]), torch.sigmoid(G[:, H:2 * H]), torch.tanh(G[:, 2 * H:3 * H]), torch.sigmoid(G[:, 3 * H:])
        cn = f * c + i * g
        hn = o * torch.tanh(cn)
        if keep is not None:
            hn = hn * keep[s].double() * (1.0 / 0.9)
        a = torch.ones(Bp, 1, dtype=torch.float64) if active is None else active[s].double()[:, None]
        gates.append(torch.cat((i, f, g, o), 1))
        c = a * cn + (1 - a) * c
        hraw.append(a * hn)
        h = a * hn + (1 - a) * h
        cs.append(c.clone())
        hcar.append(h.clone())
    return torch.stack(gates), torch.stack(cs), torch.stack(hraw), torch.stack(hcar)


def _plstm(dev, H, ndir, T, blocks, packed, dropout, B):
    import ctypes
    from gantron_amd import engine
    from gantron_amd._lib import GtPlstm, check, lib
    from gantron_amd.ops import ptr
    Bp, N = 64, 4 * H
    assert lib().gt_plstm_supported(Bp, H)
    gen = torch.Generator().manual_seed(H + T)
    Whh = torch.randn(ndir, N, H, generator=gen) * (1.5 / math.sqrt(H))
    pre = torch.randn(ndir, T, Bp, N, generator=gen)
    pre[:, :, B:] = 0
    keep = (torch.rand(T, Bp, H, generator=gen) >= 0.1).to(torch.uint8) if dropout else None
    lens = torch.randint(1, T + 1, (Bp,), generator=gen)
    lens[0] = T
    active = (torch.arange(T)[:, None] < lens[None, :]).to(torch.int32) if packed else None
    if active is not None:
        active[:, B:] = 0
    ld = 2 * H if not packed else H                   # decoder use: h lives in columns [0, H) of the (T, Bp, 2H) HC history
    d = lambda t: None if t is None else t.contiguous().to(dev)
    Wd, pred, keepd, actd = d(Whh), d(pre), d(keep), d(active)
    gates = torch.full((ndir, T, Bp, N), float('nan'), device=dev)
    c = torch.full((ndir, T + 1, Bp, H), float('nan'), device=dev)
    c[:, 0] = 0
    hs = torch.full((ndir, T, Bp, H), float('nan'), device=dev) if packed else None
    ring = 2 if packed else 0
    hx = torch.full((ndir, ring if ring else T, Bp, ld), float('nan'), device=dev)
    cnt = torch.zeros(ndir, dtype=torch.int32, device=dev)
    q = GtPlstm()
    q.B, q.Bp, q.H, q.T, q.ndir, q.hx_ring, q.p_drop = B, Bp, H, T, ndir, ring, 0.1 if dropout else 0.0
    q.Whh, q.w_dstride, q.pre, q.pre_dstride = ptr(Wd), Wd.stride(0), ptr(pred), pred.stride(0)
    q.gates, q.gates_dstride, q.c, q.c_dstride = ptr(gates), gates.stride(0), ptr(c), c.stride(0)
    if hs is not None:
        q.hs, q.hs_dstride = ptr(hs), hs.stride(0)
    q.hx, q.hx_dstride, q.hx_step, q.hx_ld = ptr(hx), hx.stride(0), hx.stride(1), ld
    q.keep, q.active, q.counters, q.timeout_flag = ptr(keepd), ptr(actd), ptr(cnt), ptr(engine.handoff_flag(dev))
    s = 0
    for n in blocks:                                   # several launches: state is picked up from c / hx of the step before
        q.s_lo, q.s_hi = s, s + n
        check(lib().gt_plstm_fwd(ctypes.byref(q), torch.cuda.current_stream().cuda_stream), 'gt_plstm_fwd')
        s += n
    assert s == T
    torch.cuda.synchronize()
    engine.check_handoffs()
    for dd in range(ndir):
        g_ref, c_ref, hraw_ref, hcar_ref = _lstm_reference(pre[dd], Whh[dd], keep, active, torch.zeros(Bp, H))
        close(gates[dd, :, :B], g_ref[:, :B], rtol=1e-5, atol=1e-5, what=f'gates dir {dd}')
        close(c[dd, :, :B], c_ref[:, :B], rtol=1e-5, atol=1e-5, what=f'c dir {dd}')
        if hs is not None:
            close(hs[dd, :, :B], hraw_ref[:, :B], rtol=1e-5, atol=1e-5, what=f'emitted h dir {dd}')
            close(hx[dd, (T - 1) % 2, :B], hcar_ref[T - 1, :B], rtol=1e-5, atol=1e-5, what=f'carried h dir {dd}')
        else:
            close(hx[dd, :, :B, :H], hcar_ref[:, :B], rtol=1e-5, atol=1e-5, what=f'h history dir {dd}')
        assert B == Bp or float(c[dd, :, B:].abs().max()) == 0.0           # pad rows stay zero


@pytest.mark.parametrize('B', [64, 50])
def test_plstm_encoder_bilstm(dev, B):
    """gt_plstm_fwd as the encoder BiLSTM runs it: two directions in one persistent launch, H = 256, packed-sequence
    pass-through (ragged lengths), 57 steps -- gates, cell states, emitted and carried h against a float64 recurrence."""
    _plstm(dev, 256, 2, 57, [57], packed=True, dropout=False, B=B)


def test_plstm_decoder_lstm_blocks(dev):
    """gt_plstm_fwd as the decoder loop runs it: H = 1024 (W_hh resident in 64 workgroups' registers), one launch per
    block of steps with the state picked up from the histories, dropout on h, h published into a (T, Bp, 2H) history."""
    _plstm(dev, 1024, 1, 40, [16, 16, 4, 4], packed=False, dropout=True, B=64)


def _plstm_bwd(dev, H, ndir, T, blocks, packed, dropout, B):
    """gt_plstm_bwd against a float64 restatement of the LSTM backward recurrence (model.py:233-236, 340-343 through autograd:
    dh_s = direct + dG_{s+1} W_hh, the cell's gate gradients, dc carried; rows past their length leave dG = 0, dc as is)."""
    import ctypes
    from gantron_amd import engine
    from gantron_amd._lib import GtPlstmBwd, check, lib
    from gantron_amd.ops import ptr
    Bp, N = 64, 4 * H
    gen = torch.Generator().manual_seed(3 * H + T)
    Whh = torch.randn(ndir, N, H, generator=gen) * (1.5 / math.sqrt(H))
    sg = torch.sigmoid(torch.randn(ndir, T, Bp, N, generator=gen))
    gates = sg.clone()
    gates[..., 2 * H:3 * H] = 2 * sg[..., 2 * H:3 * H] - 1
    c = torch.randn(ndir, T + 1, Bp, H, generator=gen)
    ld = 2 * H if not packed else H                   # decoder use: the direct gradient lives in columns [0, H) of a (T, Bp, 2H) history
    dd = torch.randn(ndir, T, Bp, ld, generator=gen)
    keep = (torch.rand(T, Bp, H, generator=gen) >= 0.1).to(torch.uint8) if dropout else None
    lens = torch.randint(1, T + 1, (Bp,), generator=gen)
    lens[0] = T
    active = (torch.arange(T)[:, None] < lens[None, :]).to(torch.int32) if packed else None
    if active is not None:
        active[:, B:] = 0
    dc0 = torch.randn(ndir, Bp, H, generator=gen) * 0.1
    d = lambda t: None if t is None else t.contiguous().to(dev)
    Wd, gd, cd, ddd, keepd, actd, dcd = d(Whh), d(gates), d(c), d(dd), d(keep), d(active), d(dc0)
    dG = torch.full((ndir, T, Bp, N), float('nan'), device=dev)
    xe = lib().gt_plstm_bwd_xbuf_elems(H)
    xbuf = torch.full((ndir, xe), float('nan'), device=dev)
    cnt = torch.zeros(ndir, dtype=torch.int32, device=dev)
    q = GtPlstmBwd()
    q.B, q.Bp, q.H, q.T, q.ndir, q.p_drop = B, Bp, H, T, ndir, 0.1 if dropout else 0.0
    q.Whh, q.w_dstride, q.gates, q.gates_dstride, q.c, q.c_dstride = ptr(Wd), Wd.stride(0), ptr(gd), gd.stride(0), ptr(cd), cd.stride(0)
    q.dh_direct, q.dd_dstride, q.dd_step, q.dd_ld = ptr(ddd), ddd.stride(0), ddd.stride(1), ld
    q.dG, q.dG_dstride, q.dc, q.dc_dstride, q.xbuf, q.x_dstride = ptr(dG), dG.stride(0), ptr(dcd), dcd.stride(0), ptr(xbuf), xbuf.stride(0)
    q.keep, q.active, q.counters, q.timeout_flag = ptr(keepd), ptr(actd), ptr(cnt), ptr(engine.handoff_flag(dev))
    s = T
    for n in blocks:                                   # highest steps first; later launches pick the exchange tiles up
        q.s_lo, q.s_hi, q.recurrent_in = s - n, s, int(s < T)
        check(lib().gt_plstm_bwd(ctypes.byref(q), torch.cuda.current_stream().cuda_stream), 'gt_plstm_bwd')
        s -= n
    assert s == 0
    torch.cuda.synchronize()
    engine.check_handoffs()
    dG, dcd = dG.cpu(), dcd.cpu()
    for k in range(ndir):
        W, g, cc = Whh[k].double(), gates[k].double(), c[k].double()
        dcr, dGn = dc0[k].double().clone(), torch.zeros(Bp, N, dtype=torch.float64)
        for s in range(T - 1, -1, -1):
            dh = dd[k, s, :, :H].double() + dGn @ W
            if keep is not None:
                dh = dh * keep[s].double() * (1.0 / 0.9)
            i, f, gg, o = g[s, :, :H], g[s, :, H:2 * H], g[s, :, 2 * H:3 * H], g[s, :, 3 * H:]
            tc = torch.tanh(cc[s + 1])
            dcell = dcr + dh * o * (1 - tc * tc)
            a = torch.zeros(Bp, 1, dtype=torch.float64)
            a[:B] = 1.0 if active is None else active[s, :B].double()[:, None]
            dGn = a * torch.cat((dcell * gg * i * (1 - i), dcell * cc[s] * f * (1 - f), dcell * i * (1 - gg * gg),
                                 dh * tc * o * (1 - o)), 1)
            dcr = a * dcell * f + (1 - a) * dcr
            close(dG[k, s], dGn, rtol=2e-5, atol=2e-5 * float(dGn.abs().max()) + 1e-12, what=f'dG dir {k} step {s}')
        close(dcd[k], dcr, rtol=2e-5, atol=2e-5 * float(dcr.abs().max()) + 1e-12, what=f'carried dc dir {k}')


@pytest.mark.parametrize('B', [64, 50])
def test_plstm_bwd_encoder_bilstm(dev, B):
    """gt_plstm_bwd as the encoder BiLSTM backward runs it: two directions in one persistent launch, H = 256, ragged lengths."""
    _plstm_bwd(dev, 256, 2, 57, [57], packed=True, dropout=False, B=B)


def test_plstm_bwd_decoder_lstm_blocks(dev):
    """gt_plstm_bwd as the decoder backward loop runs it: H = 1024, one launch per block of steps (highest first) with the
    exchange tiles of the previous launch picked up, dropout on h, the direct gradient in a (T, Bp, 2H) history."""
    _plstm_bwd(dev, 1024, 1, 40, [4, 4, 16, 16], packed=False, dropout=True, B=64)
