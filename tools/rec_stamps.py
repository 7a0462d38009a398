"""Where does the attention LSTM's fused step spend its time IN the training step?  (development tool)
GANTRON_DEBUG=rec_stamps python tools/rec_stamps.py : runs two pairs, then prints the phase stamps of the last
gt_rec_cell_fwd launch of the attention LSTM (decoder step To - 1 of the last forward), per phase: mean / max over workgroups."""
import ctypes, os, sys
os.environ['GANTRON_DEBUG'] = (os.environ.get('GANTRON_DEBUG', '') + ',rec_stamps,attn_stamps').lstrip(',')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from gantron_amd._lib import lib
from gantron_amd.hparams import HParams
from gantron_amd.model import Tacotron2

dev = torch.device('cuda:0')
hp = HParams(bench.HPARAMS + ',batch_size=64')
torch.manual_seed(hp.seed)
G = Tacotron2(hp).to(dev).train()
batch = bench.ljspeech_shaped_batch(hp, 64, seed=1234)
batch = tuple(t if i in (1, 6) else t.to(dev) for i, t in enumerate(batch))
x, _ = G.parse_batch(batch)
for _ in range(3):
    with torch.no_grad():
        G(x)
torch.cuda.synchronize()
buf = (ctypes.c_int64 * (8 * 4096))()
n = lib().gt_debug_rec_stamps(buf)
st = np.array(buf[:8 * n], dtype=np.int64).reshape(n, 8).astype(np.float64) * 0.01      # us
t0 = st[:, 0].min()
names = ['start', 'first operands in LDS', 'MFMAs done', 'tile published', 'other slabs added', 'cell done', 'query issued']
print(f'{n} workgroups; times in us relative to the first workgroup start')
for k, name in enumerate(names):
    v = st[:, k][st[:, k] > 0] - t0
    if len(v):
        print(f'{name:24s} n={len(v):4d}  min {v.min():6.2f}  mean {v.mean():6.2f}  max {v.max():6.2f}')
last = st[st[:, 5] > 0]
print('last arrivers: publish -> slabs added', np.mean(last[:, 4] - last[:, 3]).round(2), ' slabs -> cell done', np.mean(last[:, 5] - last[:, 4]).round(2),
      ' cell -> query', np.mean(last[:, 6] - last[:, 5]).round(2))
print('shader clock during the product loop: %.2f GHz (clock64 ticks / wall-clock time, mean over workgroups)' % float(np.mean(st[:, 7] / 0.01 / np.maximum(st[:, 2] - st[:, 1], 1e-9)) / 1e3))
print('per workgroup: start -> first operands', np.mean(st[:, 1] - st[:, 0]).round(2), ' -> MFMAs done', np.mean(st[:, 2] - st[:, 1]).round(2),
      ' -> published', np.mean(st[:, 3] - st[:, 2]).round(2))

n = lib().gt_debug_attn_stamps(buf)
if n:
    st = np.array(buf[:8 * n], dtype=np.int64).reshape(n, 8).astype(np.float64) * 0.01
    st = st[st[:, 0] > 0]                                   # workgroups of chunks beyond a row's length leave at once
    t0 = st[:, 0].min()
    print(f'\nattention step (gt_attn_step2_fwd): {len(st)} member workgroups of {n}')
    for k, name in enumerate(['start', '(unused)', 'inputs staged', 'location conv done', 'energies done',
                              'partial context done', 'published + ticket', 'last arriver done']):
        v = st[:, k][st[:, k] > 0] - t0
        if len(v):
            print(f'{name:24s} n={len(v):4d}  min {v.min():6.2f}  mean {v.mean():6.2f}  max {v.max():6.2f}')

n = lib().gt_debug_attn_tail_stamps(buf)
if n:
    ts = np.array(buf[:4 * n], dtype=np.int64).reshape(n, 4).astype(np.float64)
    ts[:, :3] *= 0.01
    ran = ts[ts[:, 1] > 0]
    if len(ran):
        print(f'\ntail of the attention step (GtRecTail): {int(ts[:, 3].sum())} items over {int((ts[:, 3] > 0).sum())} of {n} workgroups')
        t0 = ts[:, 0][ts[:, 0] > 0].min()
        for k, name in enumerate(['workgroup start', 'first draw', 'counter dry']):
            v = ran[:, k] - t0
            print(f'{name:24s} n={len(v):4d}  min {v.min():6.2f}  mean {v.mean():6.2f}  max {v.max():6.2f}')
        w = ran[ran[:, 3] > 0]
        print('per working workgroup: items %.2f, us per item %.2f; early workers (first draw < 2 us): %d' % (
            w[:, 3].mean(), np.mean((w[:, 2] - w[:, 1]) / w[:, 3]), int(((ran[:, 1] - t0) < 2.0).sum())))
        order = np.argsort(ts[:, 0])
        print('workgroup starts (us, every 32nd in start order):', np.round(ts[order[::32], 0] - t0, 2).tolist())
