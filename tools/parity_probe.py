"""Development aid: the `full_dims_b64_t200` parity case (tests/test_parity_gpu.py) outside pytest -- prints the
relative error of every generator gradient against the CPU oracle (worst first) and, with REPEAT=1, the run-to-run
difference of two identical HIP runs.  Test infrastructure only (imports the oracle)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import common
from test_parity_gpu import run_pair_hip
from gantron_amd.hparams import HParams
from oracle import gantron_ref as R

dev = torch.device('cuda:0')
hp = HParams('use_labels=False,use_noise=True')
g = torch.Generator().manual_seed(17)
in_lens = sorted(torch.randint(12, 51, (64,), generator=g).tolist(), reverse=True)
out_lens = torch.randint(96, 201, (64,), generator=g).tolist()
batch = common.synthetic_batch(hp, in_lens, out_lens, seed=5)
Gp, Dp = R.init_params(hp, seed=3)
Gc = dict(Gp)
for k in list(Gp):
    if k.endswith('running_var'):
        Gc[k.replace('running_var', 'num_batches_tracked')] = torch.zeros((), dtype=torch.long)
res = run_pair_hip(hp, Gc, Dp, batch, 77, dev)
if os.environ.get('REPEAT') == '1':
    res2 = run_pair_hip(hp, Gc, Dp, batch, 77, dev)
    rows = sorted(((float((res['g_grad'][k] - res2['g_grad'][k]).abs().max() / (res['g_grad'][k].abs().max() + 1e-30)), k)
                   for k in res['g_grad'] if float(res['g_grad'][k].abs().max()) > 1e-5), reverse=True)
    print('run-to-run:', ', '.join(f'{k} {e:.2e}' for e, k in rows[:6]))
text, il, mel, gate, spk, emo, ol = batch
stream = common.MaskStream(77)
B, Ti, To = len(in_lens), max(in_lens), max(out_lens)
rng = common.draw_generator_rng(stream, hp, B, Ti, To)
rng['disc_keep'] = common.draw_disc_keep(stream, hp, B, To)
out, losses, grads, _ = R.generator_step_grads(Gp, Dp, (text, il, mel, gate, ol), rng, hp)
for name, t in zip(('mel', 'mel_post', 'gate', 'align'), out):
    print(f'out.{name} rel err {float((res["out"][name] - t).abs().max() / t.abs().max()):.2e}')
rows = sorted(((float((res['g_grad'][k] - grads[k]).abs().max() / (grads[k].abs().max() + 1e-30)), k, float(grads[k].abs().max()))
               for k in grads if float(grads[k].abs().max()) > 1e-5), reverse=True)
for e, k, s in rows[:8]:
    print(f'{e:.2e}  {k}  (scale {s:.2e})')
# discriminator step of the same pair (fake = the oracle's own mel_post)
kr, kf = common.draw_disc_keep(stream, hp, B, To), common.draw_disc_keep(stream, hp, B, To)
dl, dgrads = R.discriminator_step_grads(Dp, mel, ol, out[1], ol, kr, kf, hp)
for k in dgrads:
    d = (res['d_grad'][k].double() - dgrads[k].double())
    print(f'd_grad {k}: max err {float(d.abs().max()):.2e} rel L2 {float(d.norm() / (dgrads[k].double().norm() + 1e-30)):.2e} (scale {float(dgrads[k].abs().max()):.2e})')
