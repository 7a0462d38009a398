"""Build-container only: time the REAL reference (imported in place with the same shims as
make_golden.py) against the oracle on bench.py's CPU-baseline sample, same threads, same batch.
SURVEY.md 8(d) asks that the restatement's time be within +-10 % of the reference's so that the
`cpu_baseline` figure bench.py reports from the oracle is not sand-bagged.

    python tests/golden/time_reference.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden as MG  # noqa: E402  (installs the shims, imports the reference)
import common  # noqa: E402
import bench  # noqa: E402
from oracle import gantron_ref as R  # noqa: E402


def main():
    torch.set_num_threads(bench.host_threads())
    hp, G, D = MG.build(bench.HPARAMS)
    text, il, mel, gate, ol = bench.cpu_sample_batch(hp, 8)
    B, Ti, To = text.shape[0], text.shape[1], mel.shape[2]
    batch = (text, il, mel, gate, torch.zeros(B, dtype=torch.long), torch.zeros(B, 1), ol)
    crit = MG.ref_loss.Tacotron2Loss()

    def ref_pair():
        for p in D.parameters():
            p.requires_grad = False
        G.zero_grad()
        x, y = G.parse_batch(batch)
        y_pred = G(x)
        mel_loss, gate_loss, attn_loss = crit(y_pred, y, x[1], x[-1])
        total = mel_loss + gate_loss + D.adversarial_loss(y_pred[1], x[-1]) + 10 * attn_loss
        total.backward()
        torch.nn.utils.clip_grad_norm_(G.parameters(), 1.0)
        for p in D.parameters():
            p.requires_grad = True
        D.zero_grad()
        fake = y_pred[1].detach()
        d_loss = (D.adversarial_loss(x[2], x[-1]) - D.adversarial_loss(fake, x[-1])) / 2
        d_loss.backward()

    Gp = {k: v.detach().clone() for k, v in G.state_dict().items() if not k.endswith('num_batches_tracked')}
    Dp = {k: v.detach().clone() for k, v in D.state_dict().items()}
    stream = common.MaskStream(5)
    rng = common.draw_generator_rng(stream, hp, B, Ti, To)
    rng['disc_keep'] = common.draw_disc_keep(stream, hp, B, To)
    kr, kf = common.draw_disc_keep(stream, hp, B, To), common.draw_disc_keep(stream, hp, B, To)

    def oracle_pair():
        out, _, grads, _ = R.generator_step_grads(Gp, Dp, (text, il, mel, gate, ol), rng, hp)
        R.clip_grad_norm(grads, 1.0)
        R.discriminator_step_grads(Dp, mel, ol, out[1], ol, kr, kf, hp)

    def med(fn, n=3):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    frames = int(ol.sum())
    t_ref, t_or = med(ref_pair), med(oracle_pair)
    print(f'threads {torch.get_num_threads()}  B={B} Ti<={Ti} To<={To}  {frames} valid frames')
    print(f'reference (shimmed, in place): {t_ref:.2f} s/pair = {frames / t_ref:.0f} frames/s')
    print(f'oracle/gantron_ref.py        : {t_or:.2f} s/pair = {frames / t_or:.0f} frames/s   (oracle/reference time = {t_or / t_ref:.2f})')


if __name__ == '__main__':
    main()
