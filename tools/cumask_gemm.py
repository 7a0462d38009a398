"""The decoder weight-gradient GEMM on CU-masked streams (development tool): hipExtStreamCreateWithCUMask with mask words
[lo, hi) of 8 set (tools/cumask_probe.hip: each word = 4 CUs in every XCD, one per shader engine).  Round 3 measurement
(MI355X): all 256 CUs 4.64 ms (201 TFLOP/s), 192 CUs 8.03 ms, 128 CUs 7.02 ms (133 TFLOP/s: a third more per CU), 64 CUs
13.96 ms.  Confining the leaf stream's GEMMs this way did not help the training step (131.0 ms unconfined, 132.1 on 128 CUs,
131.3 on 160, 135.1 on 96: the loops are slowed by the GEMMs' memory traffic, not by their occupancy), so the product has no
such switch.  NB such streams are BLOCKING streams: they serialise against the legacy default stream."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gantron_amd import ops
dev = torch.device('cuda:0')
torch.cuda.init()
hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(lo, hi):
    mask = (ctypes.c_uint32 * 8)(*[0xffffffff if lo <= i < hi else 0 for i in range(8)])
    h = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, mask) == 0
    return torch.cuda.ExternalStream(h.value)


rows, N = 64 * 870, 2048
dy, x, out = torch.randn(rows, 4096, device=dev), torch.randn(rows, N, device=dev), torch.zeros(4096, N, device=dev)
fl = 2.0 * 4096 * N * rows
for name, st in [('default', torch.cuda.current_stream()), ('torch side', torch.cuda.Stream())] + \
        [(f'words [0,{n})', masked_stream(0, n)) for n in (8, 6, 4, 2)] + [('words [4,8)', masked_stream(4, 8))]:
    with torch.cuda.stream(st):
        for _ in range(2):
            ops.linear_dw(dy, x, rows, 4096, N, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.linear_dw(dy, x, rows, 4096, N, out)
        e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    print(f'{name:14s} {us:9.1f} us  {fl / us / 1e6:7.2f} TFLOP/s', flush=True)
