"""Development aid: the fused forward attention step (gt_attn_step_fwd) against energies + context as two launches, at
the benchmark's shapes (B=64 LJSpeech-shaped lengths, Ti=187, E=1024); event-timed over REPS back-to-back steps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gantron_amd import ops
from gantron_amd._lib import lib
from gantron_amd.hparams import HParams
dev = torch.device('cuda:0')
REPS = int(os.environ.get('REPS', 300))
hp = HParams(bench.HPARAMS)
batch = bench.ljspeech_shaped_batch(hp, 64, seed=1234)
lens = batch[1]
B, Ti, E, A = 64, int(lens.max()), 1024, 128
d = ops.attn_dims(B, Ti, E, A, 32, 31)
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
Wc, Wd, v = r(32, 2, 31) * 0.3, r(128, 32) * 0.3, r(1, 128) * 0.5
wpack = torch.empty(int(lib().gt_attn_wpack_elems(d)), device=dev); ops.call('gt_attn_pack_weights', d, Wc, Wd, wpack)
memory, pm = r(B, Ti, E), r(B, Ti, A) * 0.5
nsl = 64
qd = r(nsl, B, A) * 0.05
w_prev = torch.softmax(r(B, Ti), 1); w_cum = torch.softmax(r(B, Ti), 1) * 3
lens32 = lens.to(torch.int32).to(dev)
q_out = torch.empty(B, A, device=dev); u_out = torch.empty(B, Ti, A, device=dev); en = torch.empty(B, Ti, device=dev)
w_out = torch.empty(B, Ti, device=dev); cum_out = torch.empty(B, Ti, device=dev)
c0 = torch.empty(B, E, device=dev); c1 = torch.empty(B, 2 * E, device=dev); c2 = torch.empty(B, 2 * E, device=dev)
cnt = torch.zeros(B, dtype=torch.int32, device=dev); tf = torch.zeros(1, dtype=torch.int32, device=dev)
qs = ops.slabs(qd, nsl, B * A, A)
step = [0]
def fused():
    step[0] += 1
    ops.call('gt_attn_step_fwd', d, qs, w_prev, w_cum, pm, Wc, Wd, v, lens32, q_out, wpack, u_out, en, cnt, step[0], tf, memory, E,
             w_out, w_cum, cum_out, c0, E, c1, 2 * E, c2, 2 * E)
nch = int(lib().gt_attn_chunks(Ti))
xbuf = torch.empty(B * nch * int(lib().gt_attn_step2_xstride(d)), device=dev)
cnt2 = torch.zeros(B, dtype=torch.int32, device=dev)
step2 = [0]
def fused2():
    step2[0] += 1
    ops.call('gt_attn_step2_fwd', d, qs, w_prev, w_cum, pm, Wc, Wd, v, lens32, q_out, wpack, u_out, xbuf, cnt2, step2[0], memory, E,
             w_out, w_cum, cum_out, c0, E, c1, 2 * E, c2, 2 * E)
def energies_only():
    ops.call('gt_attn_energies_fwd_u', d, qs, w_prev, w_cum, pm, Wc, Wd, v, lens32, q_out, en, wpack, u_out)
def two():
    ops.call('gt_attn_energies_fwd_u', d, qs, w_prev, w_cum, pm, Wc, Wd, v, lens32, q_out, en, wpack, u_out)
    ops.call('gt_attn_context_fwd', d, en, memory, E, lens32, w_out, w_cum, cum_out, c0, E, c1, 2 * E, c2, 2 * E)
for name, fn in (('two launches', two), ('energies only', energies_only), ('fused (wait)', fused), ('fused2 (no wait)', fused2)):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name:14s} {e0.elapsed_time(e1) * 1e3 / REPS:7.2f} us per step   (timeout flag {int(tf)})')
