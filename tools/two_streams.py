"""Feasibility probe (round 3): is the chip idle enough beside ONE latency-bound decoder chain that a second, independent
chain fits in?  Generator FORWARD passes (encoder, decoder loop, PostNet) of two independent 32-row batches, each on a stream
of its own issued from a thread of its own, against one of them alone and against one 64-row batch.  Throughput only.

Measured (MI355X, round 3): one batch of 64: 50.9 ms per forward; one batch of 32: 36.2 ms; two batches of 32 on two streams:
50.9 ms for both -- exactly the 64-row time.  The second chain does not fit in: the loop kernels hold a CU each (512 threads,
64+ KB of LDS, most of the register file) while they wait on memory latency and hand-offs, so kernels of two chains take
turns instead of overlapping.  (tools/dispatch_probe.hip rules the dispatcher out: an empty launch costs 4.0 us of kernel time
and 2.7-2.9 us back to back whatever its grid, 1 to 8192 waves.  Two PROCESSES with 32 rows each, tools/two_halves.sh,
time-slice: 290 ms per pair each.)"""
import os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from gantron_amd.hparams import HParams
from gantron_amd.model import Tacotron2

dev = torch.device('cuda:0')


def instance(B, seed):
    hp = HParams(bench.HPARAMS + f',batch_size={B}')
    torch.manual_seed(hp.seed)
    G = Tacotron2(hp).to(dev).train()
    batch = bench.ljspeech_shaped_batch(hp, B, seed=seed)
    batch = tuple(t if i in (1, 6) else t.to(dev) for i, t in enumerate(batch))
    x, _ = G.parse_batch(batch)
    st = torch.cuda.Stream()

    def fwd():
        with torch.cuda.stream(st), torch.no_grad():
            G(x)
    return fwd


def run(name, fns, steps=4, warm=2):
    for _ in range(warm):
        for f in fns:
            f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=lambda f=f: [f() for _ in range(steps)]) for f in fns]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    print(f'{name:34s} {(time.perf_counter() - t0) / steps * 1e3:7.2f} ms per forward round', flush=True)


run('one batch of 64', [instance(64, 1234)])
run('one batch of 32', [instance(32, 1234)])
run('two batches of 32, two streams', [instance(32, 1234), instance(32, 4321)])
