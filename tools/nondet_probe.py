"""Probe (round 4): the evaluation-mode forward of the benchmark-shaped batch twice with identical inputs; for every history
tensor of the decoder loop, the first step at which the two runs differ (bitwise) and by how much."""
import os, sys
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [root, os.path.join(root, 'tests'), os.path.join(root, 'tests', 'golden')]
import torch
import bench
from gantron_amd import ops
from gantron_amd.hparams import HParams
from gantron_amd.model import Tacotron2
dev = torch.device('cuda:0')
hp = HParams('use_labels=False,use_noise=True')
torch.manual_seed(0)
G = Tacotron2(hp).to(dev).eval()
batch = bench.ljspeech_shaped_batch(hp, 64, seed=7)
text, il, mel, gate, spk, emo, ol = batch
To, Ti = mel.shape[2], text.shape[1]
g = torch.Generator().manual_seed(3)
keeps = [(torch.rand(To, 64, hp.prenet_dim, generator=g) >= 0.5).to(torch.uint8).to(dev) for _ in range(2)]
noise = torch.rand(64, hp.noise_size, generator=g).to(dev)
x, _ = G.parse_batch(batch)
P = {k: p.detach() for k, p in G.named_parameters()}
P.update({k: v for k, v in G.named_buffers()})
runs = []
for r in range(int(os.environ.get('RUNS', 3))):
    rng = dict(prenet_keep=keeps, noise=noise)
    with torch.no_grad():
        outs, sv = G.engine.forward(P, x[0].contiguous(), il.tolist(), x[2].contiguous(), ol.tolist(), rng, None, None,
                                    training=False, mask_padding=True)
    torch.cuda.synchronize()
    h = {k: v.detach().clone() for k, v in sv['hist'].items() if torch.is_tensor(v)}
    h['out_mel'] = outs[0].detach().clone()
    runs.append(h)
a = runs[0]
for r, b in enumerate(runs[1:], 1):
    print(f'--- run {r} against run 0')
    for k in sorted(a):
        if a[k].shape != b[k].shape or a[k].dim() < 2:
            continue
        d = (a[k] != b[k])
        if not bool(d.any()):
            print(f'{k:12s} {tuple(a[k].shape)} identical')
            continue
        steps = d.reshape(d.shape[0], -1).any(1).nonzero().flatten()
        t0 = int(steps[0])
        diff = (a[k][t0].double() - b[k][t0].double()).abs()
        rows = d[t0].reshape(d.shape[1], -1).any(1).nonzero().flatten().tolist() if a[k].dim() >= 3 else []
        print(f'{k:12s} {tuple(a[k].shape)} first differs at index {t0} of dim 0 ({len(steps)} differ): max |diff| {float(diff.max()):.3e} '
              f'of {float(a[k][t0].abs().max()):.3e}; rows {rows[:12]}; overall max {float((a[k].double() - b[k].double()).abs().max()):.3e}')

# ---- which run is right?  gates2[t] recomputed in float64 from the histories of the same run
def ref_gates2(h, t):
    Wih, Whh = P['decoder.decoder_rnn.weight_ih'].double(), P['decoder.decoder_rnn.weight_hh'].double()
    b = (P['decoder.decoder_rnn.bias_ih'] + P['decoder.decoder_rnn.bias_hh']).double()
    Hd = Whh.shape[1]
    x = torch.cat((h['hatt'][t].double(), h['HC'][t][:, Hd:].double()), 1)
    hprev = h['HC'][t - 1][:, :Hd].double()
    pre = x @ Wih.t() + hprev @ Whh.t() + b
    i, f, g_, o = pre.chunk(4, 1)
    return torch.cat((torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g_), torch.sigmoid(o)), 1)
for t in (15, 16, 17, 33):
    for r, h in enumerate(runs):
        ref = ref_gates2(h, t)
        err = (h['gates2'][t].double() - ref).abs()
        bad_rows = (err.max(1).values > 1e-4).nonzero().flatten().tolist()
        print(f'gates2[{t}] run {r}: max err vs float64 of own inputs {float(err.max()):.3e}; rows off by > 1e-4: {bad_rows[:20]} ({len(bad_rows)})')

# ---- ctx[t] (the HC history's context columns) against float64 sum_i w[t+1][b, i] memory[b, i, :] of the same run
mem = sv['memory'].double()                                  # (B, Ti, E)
Hd = P['decoder.decoder_rnn.weight_hh'].shape[1]
for t in (15, 16, 17):
    for r, h in enumerate(runs):
        w = h['w_hist'][t + 1].double()                      # (B, Ti)
        ref = torch.einsum('bi,bie->be', w, mem)
        ctx = h['HC'][t][:, Hd:].double()
        err = (ctx - ref).abs()
        bad = (err.max(1).values > 1e-4).nonzero().flatten().tolist()
        print(f'ctx[{t}] run {r}: max err {float(err.max()):.3e}; rows off by > 1e-4: {bad[:24]} ({len(bad)})')
        for b in bad[:3]:
            cols = (err[b] > 1e-4).nonzero().flatten()
            # which chunk's partial is stale?  the error should equal (partial of step t-1) - (partial of step t) of that chunk
            wp = h['w_hist'][t].double()
            expl = []
            for c in range((int(il[b]) + 31) // 32):
                sl = slice(32 * c, min(32 * c + 32, mem.shape[1]))
                d_now = torch.einsum('i,ie->e', w[b, sl], mem[b, sl])
                expl.append(float((ctx[b] - ref[b]).dot(d_now) / d_now.dot(d_now).clamp_min(1e-30)))
            print(f'    row {b} (Ti {int(il[b])}): {len(cols)} columns off, first {int(cols[0])} last {int(cols[-1])}; '
                  f'projection of the error on each chunk partial: {[round(x, 3) for x in expl]}')
