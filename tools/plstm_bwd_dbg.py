"""Debug driver for gt_plstm_bwd: one launch, then dump the counters and which steps of dG were written."""
import ctypes, math, sys, torch
sys.path.insert(0, '.')
from gantron_amd import engine
from gantron_amd._lib import GtPlstmBwd, check, lib
from gantron_amd.ops import ptr
dev = torch.device('cuda:0')
H, ndir, T, B, Bp = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 1, 6, 64, 64
N = 4 * H
Whh = (torch.randn(ndir, N, H) / math.sqrt(H)).to(dev)
gates = torch.rand(ndir, T, Bp, N).to(dev)
c = torch.randn(ndir, T + 1, Bp, H).to(dev)
dd = torch.randn(ndir, T, Bp, H).to(dev)
dc = torch.zeros(ndir, Bp, H, device=dev)
dG = torch.full((ndir, T, Bp, N), float('nan'), device=dev)
xbuf = torch.full((ndir, lib().gt_plstm_bwd_xbuf_elems(H)), float('nan'), device=dev)
cnt = torch.zeros(ndir, dtype=torch.int32, device=dev)
q = GtPlstmBwd()
q.B, q.Bp, q.H, q.T, q.ndir, q.p_drop = B, Bp, H, T, ndir, 0.0
q.Whh, q.w_dstride, q.gates, q.gates_dstride, q.c, q.c_dstride = ptr(Whh), Whh.stride(0), ptr(gates), gates.stride(0), ptr(c), c.stride(0)
q.dh_direct, q.dd_dstride, q.dd_step, q.dd_ld = ptr(dd), dd.stride(0), dd.stride(1), H
q.dG, q.dG_dstride, q.dc, q.dc_dstride, q.xbuf, q.x_dstride = ptr(dG), dG.stride(0), ptr(dc), dc.stride(0), ptr(xbuf), xbuf.stride(0)
q.counters, q.timeout_flag = ptr(cnt), ptr(engine.handoff_flag(dev))
q.s_lo, q.s_hi, q.recurrent_in = 0, T, 0
check(lib().gt_plstm_bwd(ctypes.byref(q), torch.cuda.current_stream().cuda_stream), 'gt_plstm_bwd')
torch.cuda.synchronize()
print('counters', cnt.tolist(), 'flag', engine.handoff_flag(dev).item(), 'NW', H // 16)
print('dG nan per step', [int(torch.isnan(dG[0, s]).sum()) for s in range(T)])
xb = xbuf[0].view(2, H // 16, H // 16, 64, 16)
print('xbuf nan per slot/dest', [[int(torch.isnan(xb[sl, d]).sum()) for d in range(min(4, H // 16))] for sl in range(2)])
