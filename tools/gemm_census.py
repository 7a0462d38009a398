"""Census of the gt_sgemm calls of one generator + discriminator step pair: shape, operand modes, split-K and the
serialised duration of each call (development tool; synchronises around every GEMM)."""
import collections, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['GANTRON_SCHEDULE'] = 'no_leaf_stream'
import bench
from gantron_amd import ops
from gantron_amd.hparams import HParams
from gantron_amd.loss_function import Tacotron2Loss
from gantron_amd.model import Discriminator, Tacotron2
from gantron_amd.optim import FlatAdam
from gantron_amd.train_steps import discriminator_step, generator_step

dev = torch.device('cuda:0')
hp = HParams(bench.HPARAMS)
G, D = Tacotron2(hp).to(dev).train(), Discriminator(hp).to(dev).train()
g_opt, d_opt = FlatAdam(G, lr=1e-3), FlatAdam(D, lr=1e-3)
crit = Tacotron2Loss()
batch = tuple(t if i in (1, 6) else t.to(dev) for i, t in enumerate(bench.ljspeech_shaped_batch(hp, 64, seed=1234)))
log = collections.OrderedDict()
orig = ops.sgemm
recording = False


def timed(A, B, C, M, N, K, **kw):
    if not recording:
        return orig(A, B, C, M, N, K, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    orig(A, B, C, M, N, K, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    key = (M, N, K, A.mode, B.mode, kw.get('split_k', 1), kw.get('batch', 1))
    e = log.setdefault(key, [0, 0.0])
    e[0] += 1; e[1] += dt


ops.sgemm = timed
import gantron_amd.engine as E, gantron_amd.stft as S
for it in range(2):
    recording = it == 1
    _, gen = generator_step(G, D, g_opt, crit, batch, hp, iteration=it)
    discriminator_step(G, D, d_opt, batch, gen, hp)
torch.cuda.synchronize()
tot = sum(v[1] for v in log.values())
print(f'{len(log)} distinct GEMMs, {sum(v[0] for v in log.values())} calls, {tot * 1e3:.1f} ms serialised')
for key, (n, t) in sorted(log.items(), key=lambda kv: -kv[1][1])[:28]:
    M, N, K, am, bm, sk, nb = key
    fl = 2.0 * M * N * K * nb * n
    print(f'M={M:6d} N={N:5d} K={K:6d} modes=({am},{bm}) split={sk:2d} batch={nb:2d} calls={n:3d}  {t * 1e3:7.2f} ms  {fl / t / 1e12:6.1f} TFLOP/s')
