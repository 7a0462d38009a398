"""Tensor-level wrappers over the C ABI (include/gantron_hip.h).

Every function takes CUDA (ROCm) fp32 torch tensors, passes raw device pointers + sizes to
libgantron_hip.so on torch's current stream, and raises GantronHipError on a non-zero status.
torch is used here for device memory only (allocation, views); all arithmetic is in the HIP
kernels.
"""
import ctypes
import os as _os

import torch

from . import _lib
from . import switches as sw
from ._lib import GtAttnDims, GtGemm, GtOperand, GtSlabs, check, lib

KC, RC, IM2COL_K, IM2COL_R = 0, 1, 2, 3
# Workgroups a weight-gradient GEMM should reach (default_split_k).  Two regimes, both measured on the benchmark
# batch: a GEMM nothing else overlaps wants ~512 workgroups of 128x128 tiles (two per CU for the three-plane kernel; the
# fp32-pipe kernel liked ~1024: postnet conv dW 41 -> 103 TFLOP/s; pair 133.3 -> 131.9 ms with 512 and 12 backward chunks);
# the same setting for the GEMMs that run on the leaf stream UNDER the decoder backward loop makes the pair 2 ms
# slower (they crowd out the latency-bound loop kernels), so the engine lowers the target there.
SPLITK_EXPOSED = 512
SPLITK_OVERLAPPED = 128
SPLITK_TARGET = SPLITK_EXPOSED
FWD_SPLITK = sw.schedule_on('fwd_splitk')


class splitk_target:
    """with ops.splitk_target(n): ... -- temporary target for default_split_k."""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        global SPLITK_TARGET
        self.old, SPLITK_TARGET = SPLITK_TARGET, self.n

    def __exit__(self, *exc):
        global SPLITK_TARGET
        SPLITK_TARGET = self.old
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2

# hparams.fp16_run: the layer GEMMs (linear / conv1d forward, input gradient, weight gradient below) round their
# operands to bf16 inside gt_sgemm; the engines switch it on for the duration of their entry points.  Raw sgemm()
# calls stay fp32 unless they ask.
BF16_GEMMS = False


class bf16_gemms:
    """with ops.bf16_gemms(flag): ... -- layer GEMMs with bf16 operands inside the block."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        global BF16_GEMMS
        self.old, BF16_GEMMS = BF16_GEMMS, self.on

    def __exit__(self, *exc):
        global BF16_GEMMS
        BF16_GEMMS = self.old


# How fp32 GEMMs are computed (gt_sgemm_set_fp32_split): 0 = fp32 matrix pipe, 3 = three bf16 planes per operand and
# six bf16 MFMAs (fp32-exact operands, 16/6 of the fp32 pipe's rate), 2 = two bf16 planes / three MFMAs (2^-16 accuracy
# class), 4 = two SCALED fp16 planes per operand / three fp16 MFMAs (round 4, the default: 22 significand bits per operand,
# measured error against float64 below the fp32 pipe's; include/gantron_hip.h).
# GANTRON_FP32_SPLIT picks the process default, applied when the first GEMM is issued.
FP32_SPLIT = int(_os.environ.get('GANTRON_FP32_SPLIT', '4'))
_split_applied = None


def set_fp32_split(planes):
    global FP32_SPLIT, _split_applied
    check(lib().gt_sgemm_set_fp32_split(int(planes)), 'gt_sgemm_set_fp32_split')
    FP32_SPLIT = _split_applied = int(planes)


def apply_fp32_split():
    """Push FP32_SPLIT into the library if it is not there yet (cheap; called on the GEMM entry points and by the engines
    before they enter the C loops, which issue their own fp32 GEMMs)."""
    if _split_applied != FP32_SPLIT:
        set_fp32_split(FP32_SPLIT)


class fp32_split:
    """with ops.fp32_split(mode): ... -- fp32 GEMMs inside the block are computed in that mode (0 / 2 / 3 / 4, above)."""

    def __init__(self, planes):
        self.planes = int(planes)

    def __enter__(self):
        self.old = FP32_SPLIT
        set_fp32_split(self.planes)

    def __exit__(self, *exc):
        set_fp32_split(self.old)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def h2d(data, dtype, device):
    """Small host -> device copy that does NOT stall the host: `torch.tensor(data, device=...)` and `.to(device)` from
    pageable memory synchronise the stream after the copy, i.e. the host waits until the GPU has drained everything issued
    so far and the kernels issued next start from an idle device (measured in a kernel trace of the step: ~12 such points
    per G+D pair, ~14 ms of idle GPU per pair under the profiler).  Pinned staging + non_blocking copy instead; the caching
    host allocator keeps the staging block alive until the copy has run."""
    t = data if isinstance(data, torch.Tensor) else torch.tensor(data, dtype=dtype)
    t = t.to(dtype).contiguous()
    if torch.device(device).type != 'cuda':
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def ptr(t, offset=0):
    """Device address of element `offset` of tensor t (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr() + offset * t.element_size()


def call(name, *args):
    apply_fp32_split()
    conv = [a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args]
    check(getattr(lib(), name)(*conv, _stream()), name)


def _require(t, dtype=torch.float32):
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t


class Planes:
    """An fp32 tensor as three bf16 planes x = x0 + x1 + x2 (exact; gt_split3): what the pure three-plane GEMM kernel
    reads.  The planes are indexed like the flat fp32 tensor they were made from, so the element offsets / leading
    dimensions of an Operand apply unchanged.  split(lo, hi) fills the planes of the flat element range [lo, hi)."""

    def __init__(self, src):
        assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and src.numel() % 8 == 0, \
            (src.dtype, src.is_contiguous(), src.numel())
        self.src, self.n = src, src.numel()
        self.t = torch.empty(3, self.n, device=src.device, dtype=torch.bfloat16)

    def split(self, lo=0, hi=None):
        hi = self.n if hi is None else hi
        assert 0 <= lo < hi <= self.n and lo % 8 == 0 and (hi - lo) % 8 == 0, (lo, hi, self.n)
        check(lib().gt_split3(ptr(self.src, lo), hi - lo, ptr(self.t, lo), self.n, _stream()), 'gt_split3')
        return self


# GANTRON_EXPERIMENTAL=plane_gemms: operands of the decoder's weight-gradient GEMMs are split into bf16 planes ONCE (gt_split3 on the
# leaf stream) and those GEMMs run the kernel that has no split left to do (pgemm_kernel).  Measured on MI355X (round 3,
# DESIGN.md section 3) and NOT adopted: stand-alone the two kernels tie (190-197 vs 191-202 TFLOP/s on the decoder shapes: with
# 6 instead of 4 bytes per element the plane kernel is bound by what moves L2 -> LDS, 43.8 GB per GEMM, not by the split),
# and in the step the split passes cost what nothing pays back: 135.2-135.7 vs 133.8-133.9 ms per pair.  Off by default.
PLANE_GEMMS = bool(sw.experimental('plane_gemms'))


class Operand:
    """Addressing of one GEMM operand (see GtOperand).  planes: a Planes of the tensor `t` is a view of (same offsets)."""

    def __init__(self, t, mode, ld=0, T=0, C=0, dil=1, pad=0, offset=0, batch_stride=0, planes=None, amax=None):
        self.t, self.mode, self.ld, self.T, self.C = t, mode, ld, T, C
        self.dil, self.pad, self.offset, self.batch_stride = dil, pad, offset, batch_stride
        self.planes = planes
        self.amax = amax          # int64[1] from absmax(): the fp16-plane product then skips its own pass over this operand

    def c(self):
        o = GtOperand()
        o.p = ptr(self.t, self.offset)
        o.mode, o.T, o.C, o.dil, o.pad = self.mode, self.T, self.C, self.dil, self.pad
        o.ld, o.batch_stride = self.ld, self.batch_stride
        if self.planes is not None:
            o.planes, o.plane_stride = ptr(self.planes.t, self.offset), self.planes.n
        if self.amax is not None:
            assert self.amax.is_cuda and self.amax.dtype == torch.int64 and self.amax.numel() == 1
            o.amax = self.amax.data_ptr()
        return o


def absmax(t, rows=None, cols=None, ld=None, offset=0):
    """gt_absmax: max |x| over the finite elements of a (rows, cols) matrix with row stride ld that starts at element `offset`
    of the fp32 tensor t (default: all of t), as the device word GtOperand.amax takes -- for an operand that several fp16-plane
    products share (the word of a superset of what a product addresses is as good: the scale only has to bound the operand)."""
    _require(t)
    rows, cols = (1, t.numel()) if rows is None else (rows, cols)
    out = torch.empty(1, dtype=torch.int64, device=t.device)
    check(lib().gt_absmax(ptr(t, offset), rows, cols, cols if ld is None else ld, ptr(out), _stream()), 'gt_absmax')
    return out


def h2_active():
    """True when the tiled fp32 GEMMs issued now run from scaled fp16 planes (mode 4), i.e. when a shared operand's maximum is
    worth computing once (absmax) instead of once per product."""
    return FP32_SPLIT == 4 and not BF16_GEMMS


def default_split_k(M, N, K, batch=1):
    """Split the reduction of a weight-gradient GEMM until about 1024 workgroups of the tile size gt_sgemm will pick
    exist (two co-resident workgroups per CU, two rounds for the tail).  Measured on the postnet conv dW
    (M=512, N=2560, K=55680): 41 TFLOP/s unsplit on 64x64 tiles, 103 TFLOP/s with 16 splits of 128x128 tiles."""
    if M > 64 and N > 64:
        tiles = -(-M // 128) * -(-N // 128) * batch
    else:
        tiles = -(-M // 64) * -(-N // 64) * batch
    if tiles >= SPLITK_TARGET or K < 1024:
        return 1
    return max(1, min(-(-SPLITK_TARGET // tiles), K // 512, 64))


def forward_split_k(M, N, K):
    """Split-K for an activation GEMM whose output grid is too small to fill the device (discriminator and encoder
    convolutions: 88-376 tiles of 128x128): the reduction is split until ~768 workgroups exist, each keeping at
    least 1024 of K; the output is zero-filled and accumulated with fp32 atomics.  1 = no split."""
    if not FWD_SPLITK or M <= 64 or N <= 64 or K < 2048:
        return 1
    tiles = -(-M // 128) * -(-N // 128)
    if tiles >= 512:
        return 1
    return max(1, min(-(-768 // tiles), K // 1024))


def sgemm(A, B, C, M, N, K, ldc=None, bias=None, accumulate=False, split_k=1, batch=1,
          c_batch_stride=0, alpha=1.0, c_offset=0, bf16=False):
    g = GtGemm()
    g.A, g.B = A.c(), B.c()
    g.C = ptr(C, c_offset)
    g.ldc = N if ldc is None else ldc
    g.c_batch_stride = c_batch_stride
    g.bias = ptr(bias)
    g.M, g.N, g.K, g.batch = M, N, K, batch
    g.accumulate, g.split_k, g.alpha = int(accumulate), split_k, alpha
    g.bf16 = int(bf16)              # False / True, or 2 / 3: that many bf16 planes whatever the process setting
    apply_fp32_split()
    if split_k > 1 and not accumulate:
        raise ValueError('split_k > 1 accumulates: zero C and pass accumulate=True')
    check(lib().gt_sgemm(ctypes.byref(g), _stream()), 'gt_sgemm')


def slabs(t, n, stride, ld, offset=0):
    s = GtSlabs()
    s.p, s.n, s.slab_stride, s.ld = ptr(t, offset), n, stride, ld
    return s


NO_SLABS = GtSlabs()


def attn_dims(B, Ti, E, A, F, Kc):
    d = GtAttnDims()
    d.B, d.Ti, d.E, d.A, d.F, d.Kc = B, Ti, E, A, F, Kc
    return d


def roundup(x, m):
    return -(-x // m) * m


def pick_splits(K, want):
    """Largest split count <= want that leaves no empty K-slab (gt_rec_gemm requirement)."""
    kb = -(-K // 16)
    for s in range(min(want, kb), 0, -1):
        if -(-kb // -(-kb // s)) == s:
            return s
    return 1


# ---------------------------------------------------------------------------------- layers
def linear(x, W, M, K, N, bias=None, out=None, ldw=None, w_offset=0, ldx=None, x_offset=0, x_planes=None, w_planes=None):
    """out[M,N] = x[M,K] @ W[N,K]^T (+ bias).  *_planes: ops.Planes of the WHOLE tensors x / W (optional)."""
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    sgemm(Operand(x, KC, ld=ldx or K, offset=x_offset, planes=x_planes),
          Operand(W, KC, ld=ldw or K, offset=w_offset, planes=w_planes), out, M, N, K, bias=bias, bf16=BF16_GEMMS)
    return out


def linear_dx(dy, W, M, N, K, out=None, ldw=None, w_offset=0, lddy=None, dy_offset=0, dy_planes=None, w_planes=None, dy_amax=None):
    """out[M,K] = dy[M,N] @ W[N,K]."""
    if out is None:
        out = torch.empty(M, K, device=dy.device, dtype=torch.float32)
    sgemm(Operand(dy, KC, ld=lddy or N, offset=dy_offset, planes=dy_planes, amax=dy_amax),
          Operand(W, RC, ld=ldw or K, offset=w_offset, planes=w_planes), out, M, K, N, bf16=BF16_GEMMS)
    return out


def linear_dw(dy, x, rows, N, K, dW, lddy=None, dy_offset=0, ldx=None, x_offset=0, ldc=None, c_offset=0, dy_planes=None,
              x_planes=None, dy_amax=None, x_amax=None):
    """dW[N,K] += dy[rows,N]^T @ x[rows,K]  (split-K over rows when the output grid is small).  *_amax: ops.absmax words of
    (a superset of) the operands, when several products share them."""
    sk = default_split_k(N, K, rows)
    sgemm(Operand(dy, RC, ld=lddy or N, offset=dy_offset, planes=dy_planes, amax=dy_amax),
          Operand(x, RC, ld=ldx or K, offset=x_offset, planes=x_planes, amax=x_amax), dW, N, K, rows,
          ldc=ldc, accumulate=True, split_k=sk, c_offset=c_offset, bf16=BF16_GEMMS)


def col_ws(rows, C, device):
    """Scratch of the bit-reproducible column sums (gt_col_stats / gt_bn_act_drop_bwd1 with ws: BatchNorm statistics)."""
    return torch.empty(int(lib().gt_col_ws_elems(rows, C)), device=device, dtype=torch.float32)


def col_sums(x, rows, C):
    """Per-column sums of a (rows, C) matrix -> first C entries of a (2C,) tensor."""
    out = torch.empty(2 * C, device=x.device, dtype=torch.float32)
    call('gt_col_stats', x, None, rows, C, out, None)
    return out[:C]


def conv_weight_fwd(W):
    """(Cout, Cin, K) -> (Cout, K, Cin): the k-contiguous B operand of the forward implicit GEMM."""
    co, ci, k = W.shape
    out = torch.empty(co, k, ci, device=W.device, dtype=torch.float32)
    call('gt_permute3', W, ci * k, 1, k, co, k, ci, out, 0, 0, 0, 1.0)
    return out


def conv_weight_bwd(W):
    """(Cout, Cin, K) -> (Cin, K, Cout) with flipped taps: B operand of the input-gradient GEMM."""
    co, ci, k = W.shape
    out = torch.empty(ci, k, co, device=W.device, dtype=torch.float32)
    check(lib().gt_permute3(ptr(W, k - 1), k, -1, ci * k, ci, k, co, out.data_ptr(), 0, 0, 0, 1.0, _stream()),
          'gt_permute3')
    return out


def conv1d_fwd(x, Wf, bias, B, T, Cin, Cout, K, dil, x_planes=None, w_planes=None):
    """Channels-last Conv1d with 'same' padding: x (B*T, Cin) -> (B*T, Cout)."""
    pad = dil * (K - 1) // 2
    sk = forward_split_k(B * T, Cout, K * Cin)
    out = (torch.zeros if sk > 1 else torch.empty)(B * T, Cout, device=x.device, dtype=torch.float32)
    sgemm(Operand(x, IM2COL_K, T=T, C=Cin, dil=dil, pad=pad, planes=x_planes), Operand(Wf, KC, ld=K * Cin, planes=w_planes),
          out, B * T, Cout, K * Cin, bias=bias, accumulate=sk > 1, split_k=sk, bf16=BF16_GEMMS)
    return out


def conv1d_dx(dy, Wb, B, T, Cin, Cout, K, dil, dy_planes=None, w_planes=None):
    pad = dil * (K - 1) // 2
    sk = forward_split_k(B * T, Cin, K * Cout)
    out = (torch.zeros if sk > 1 else torch.empty)(B * T, Cin, device=dy.device, dtype=torch.float32)
    sgemm(Operand(dy, IM2COL_K, T=T, C=Cout, dil=dil, pad=pad, planes=dy_planes), Operand(Wb, KC, ld=K * Cout, planes=w_planes),
          out, B * T, Cin, K * Cout, accumulate=sk > 1, split_k=sk, bf16=BF16_GEMMS)
    return out


def conv1d_dw(dy, x, B, T, Cin, Cout, K, dil, dW, dy_planes=None, x_planes=None):
    """dW (Cout, Cin, K) += correlation of dy (B*T, Cout) with x (B*T, Cin)."""
    pad = dil * (K - 1) // 2
    tmp = torch.zeros(Cout, K * Cin, device=dy.device, dtype=torch.float32)
    sk = default_split_k(Cout, K * Cin, B * T)
    sgemm(Operand(dy, RC, ld=Cout, planes=dy_planes), Operand(x, IM2COL_R, T=T, C=Cin, dil=dil, pad=pad, planes=x_planes),
          tmp, Cout, K * Cin, B * T, accumulate=True, split_k=sk, bf16=BF16_GEMMS)
    # tmp[co][k][ci] -> dW[co][ci][k]
    call('gt_permute3', tmp, K * Cin, 1, Cin, Cout, Cin, K, dW, 0, 0, 1, 1.0)
