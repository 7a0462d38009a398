import sys, os, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.getcwd())
import bench
from gantron_amd.hparams import HParams
from gantron_amd.loss_function import Tacotron2Loss
from gantron_amd.model import Discriminator, Tacotron2
from gantron_amd.optim import FlatAdam
from gantron_amd.train_steps import discriminator_step, generator_step
dev = torch.device('cuda:0')
hp = HParams(bench.HPARAMS + ',batch_size=64')
torch.manual_seed(1234)
G, D = Tacotron2(hp).to(dev).train(), Discriminator(hp).to(dev).train()
g_opt = FlatAdam(G, lr=1e-3, weight_decay=1e-6); d_opt = FlatAdam(D, lr=7e-4, weight_decay=1e-6)
crit = Tacotron2Loss()
batch = bench.ljspeech_shaped_batch(hp, 64, seed=1234)
batch = tuple(t if i in (1, 6) else t.to(dev) for i, t in enumerate(batch))
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(4):
    t0 = sync()
    losses, gen = generator_step(G, D, g_opt, crit, batch, hp, iteration=it)
    t1 = sync()
    discriminator_step(G, D, d_opt, batch, gen, hp)
    t2 = sync()
    print(f'G step {1e3*(t1-t0):.1f} ms   D step {1e3*(t2-t1):.1f} ms', flush=True)
