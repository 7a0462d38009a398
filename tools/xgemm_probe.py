"""A few tiled-GEMM shapes of the training step in one precision mode (SPLIT=0/2/3), for rocprofv3 --pmc passes
(development tool):  SPLIT=3 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY ... -- python3 tools/xgemm_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gantron_amd import ops

dev = torch.device('cuda:0')
ops.set_fp32_split(int(os.environ.get('SPLIT', '0')))
REPS = int(os.environ.get('REPS', 3))
B, To = 64, 870
rows = B * To
x = torch.randn(rows, 512, device=dev); W = torch.randn(512, 512, 5, device=dev) * 0.02
Wf = ops.conv_weight_fwd(W)
dy = torch.randn(rows, 4096, device=dev); xd = torch.randn(rows, 2048, device=dev); out = torch.zeros(4096, 2048, device=dev)
xa = torch.randn(rows, 2048, device=dev); wa = torch.randn(4096, 2048, device=dev) * 0.05
for _ in range(REPS):
    ops.conv1d_fwd(x, Wf, None, B, To, 512, 512, 5, 1)                       # (IM2COL_K, KC)
    ops.linear_dw(dy, xd, rows, 4096, 2048, out)                             # (RC, RC)
    ops.linear(xa, wa, rows, 2048, 4096)                                     # (KC, KC)
torch.cuda.synchronize()
