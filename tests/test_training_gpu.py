"""Integration checks on a real MI355X: evaluation-mode forward, a few optimizer steps against the
oracle (loss-curve and parameter agreement), and the reference-compatible train loop end to end
(schedule, checkpoint write / resume) on a small synthetic dataset."""
import os

import numpy as np
import pytest
import torch

import common
from harness import close, kernel_disc_keep, kernel_rng

pytestmark = pytest.mark.gpu

MID = ('use_labels=False,use_noise=True,n_mel_channels=16,encoder_embedding_dim=32,symbols_embedding_dim=32,'
       'attention_rnn_dim=64,decoder_rnn_dim=64,prenet_dim=16,attention_dim=32,attention_location_n_filters=8,'
       'attention_location_kernel_size=9,postnet_embedding_dim=32,noise_size=32,discriminator_dim=32,'
       'discriminator_window=6')


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def _with_counters(Gp):
    out = dict(Gp)
    for k in list(Gp):
        if k.endswith('running_var'):
            out[k.replace('running_var', 'num_batches_tracked')] = torch.zeros((), dtype=torch.long)
    return out


def test_eval_mode_forward_matches_oracle(dev):
    """model.eval(): BatchNorm uses running statistics, dropout is off everywhere except the Prenet
    (reference model.py:104: training=True hard-coded), exactly what validate() runs (train.py:172-183)."""
    from gantron_amd.hparams import HParams
    from gantron_amd.model import Tacotron2
    from oracle import gantron_ref as R
    hp = HParams(MID)
    in_lens, out_lens = [21, 18, 7], [30, 41, 12]
    batch = common.synthetic_batch(hp, in_lens, out_lens, seed=9)
    Gp, _ = R.init_params(hp, seed=4)
    g = torch.Generator().manual_seed(1)
    for k in Gp:                                   # non-trivial running statistics
        if k.endswith('running_mean'):
            Gp[k] = torch.randn(Gp[k].shape, generator=g) * 0.1
        if k.endswith('running_var'):
            Gp[k] = torch.rand(Gp[k].shape, generator=g) + 0.5
    G = Tacotron2(hp).to(dev).eval()
    G.load_state_dict(_with_counters(Gp))
    stream = common.MaskStream(3)
    rng = common.draw_generator_rng(stream, hp, 3, max(in_lens), max(out_lens))
    G.rng_override = kernel_rng(rng, 3, dev)
    with torch.no_grad():
        x, _ = G.parse_batch(batch)
        got = G(x)
    text, il, mel, gate, spk, emo, ol = batch
    want = R.generator_forward(Gp, text, il, mel, ol, rng, training=False)
    for name, a, b in zip(('mel', 'mel_post', 'gate', 'align'), got, want):
        close(a, b, rtol=1e-3, atol=1e-5, what=f'eval {name}')
    for k, v in G.state_dict().items():            # evaluation must not touch the running statistics
        if k.endswith(('running_mean', 'running_var')):
            close(v, Gp[k], rtol=0, atol=0, what=k)


def test_forward_uses_the_lengths_it_is_given(dev):
    """x1 = parse_batch(b1); parse_batch(b2); model(x1) must run with b1's lengths (the reference derives every mask
    from the tensors inside `inputs`, model.py:644-673), and hand-built inputs without parse_batch work as well."""
    from gantron_amd.hparams import HParams
    from gantron_amd.model import Tacotron2
    hp = HParams(MID)
    b1 = common.synthetic_batch(hp, [21, 18, 7], [30, 41, 12], seed=9)
    b2 = common.synthetic_batch(hp, [21, 9, 5], [41, 20, 33], seed=10)       # same B, Ti, To: other lengths
    torch.manual_seed(0)
    G = Tacotron2(hp).to(dev).eval()
    rng = common.draw_generator_rng(common.MaskStream(3), hp, 3, 21, 41)
    G.rng_override = kernel_rng(rng, 3, dev)
    with torch.no_grad():
        x1, _ = G.parse_batch(b1)
        want = [o.clone() for o in G(x1)]
        x1b, _ = G.parse_batch(b1)
        G.parse_batch(b2)                               # a later parse must not leak into x1b
        got = G(x1b)
        hand = tuple(t.clone() if torch.is_tensor(t) else t for t in x1)    # fresh tensors: no host copies attached
        got2 = G(hand)
        other = G(G.parse_batch(b2)[0])
    for name, a, b, c in zip(('mel', 'mel_post', 'gate', 'align'), got, want, got2):
        close(a, b, rtol=0, atol=0, what=f'{name} after a second parse_batch')
        close(c, b, rtol=0, atol=0, what=f'{name} from hand-built inputs')
    assert float((other[2] - want[2]).abs().max()) > 1.0        # b2's lengths really give another padding mask


def test_three_optimizer_steps_track_the_oracle(dev):
    """G,D,G schedule with Adam + clipping: losses of every step and the final parameters agree with the
    oracle driven by the same masks (train.py:361-423, 302-360)."""
    from gantron_amd.hparams import HParams
    from gantron_amd.loss_function import Tacotron2Loss
    from gantron_amd.model import Discriminator, Tacotron2
    from gantron_amd.optim import FlatAdam
    from gantron_amd.train_steps import discriminator_step, generator_step
    from oracle import gantron_ref as R
    hp = HParams(MID)
    in_lens, out_lens = [17, 12, 12, 5], [26, 33, 19, 8]
    batch = common.synthetic_batch(hp, in_lens, out_lens, seed=2)
    text, il, mel, gate, spk, emo, ol = batch
    B, Ti, To = 4, max(in_lens), max(out_lens)
    Gp, Dp = R.init_params(hp, seed=8)
    G, D = Tacotron2(hp).to(dev).train(), Discriminator(hp).to(dev).train()
    G.load_state_dict(_with_counters(Gp))
    D.load_state_dict(Dp)
    g_opt = FlatAdam(G, lr=hp.g_learning_rate, weight_decay=hp.weight_decay)
    d_opt = FlatAdam(D, lr=hp.d_learning_rate, weight_decay=hp.weight_decay)
    crit = Tacotron2Loss()
    stream_h, stream_o = common.MaskStream(21), common.MaskStream(21)
    g_state, d_state = {}, {}
    Go, Do = dict(Gp), dict(Dp)
    g_step = d_step = 0
    fake_h = fake_o = None
    for it, kind in enumerate(('G', 'D', 'G')):
        if kind == 'G':
            rng = common.draw_generator_rng(stream_h, hp, B, Ti, To)
            dk = common.draw_disc_keep(stream_h, hp, B, To)
            G.rng_override, D.rng_override = kernel_rng(rng, B, dev), kernel_disc_keep(dk, dev)
            losses, fake_h = generator_step(G, D, g_opt, crit, batch, hp, iteration=it)
            rng_o = common.draw_generator_rng(stream_o, hp, B, Ti, To)
            rng_o['disc_keep'] = common.draw_disc_keep(stream_o, hp, B, To)
            out, lo, grads, new_stats = R.generator_step_grads(Go, Do, (text, il, mel, gate, ol), rng_o, hp)
            R.clip_grad_norm(grads, hp.grad_clip_thresh)
            g_step += 1
            R.adam_step(Go, grads, g_state, hp.g_learning_rate, g_step, wd=hp.weight_decay)
            Go.update(new_stats)
            fake_o = out[1]
            close(losses['total'], lo['total'], rtol=2e-3, atol=1e-5, what=f'step {it} G loss')
        else:
            kr, kf = common.draw_disc_keep(stream_h, hp, B, To), common.draw_disc_keep(stream_h, hp, B, To)
            # adversarial_loss is called twice per discriminator step: real masks first, then fake masks
            masks = [kernel_disc_keep(kr, dev), kernel_disc_keep(kf, dev)]
            orig = D._keeps
            D._keeps = lambda *a, **k: masks.pop(0)
            losses = discriminator_step(G, D, d_opt, batch, fake_h, hp)
            D._keeps = orig
            kr_o, kf_o = common.draw_disc_keep(stream_o, hp, B, To), common.draw_disc_keep(stream_o, hp, B, To)
            lo, grads = R.discriminator_step_grads(Do, mel, ol, fake_o, ol, kr_o, kf_o, hp)
            d_step += 1
            R.adam_step(Do, grads, d_state, hp.d_learning_rate, d_step, wd=hp.weight_decay)
            close(losses['total'], lo['total'], rtol=2e-3, atol=1e-5, what=f'step {it} D loss')
    def close_params(p, ref, lr, steps, what):
        """Adam's first steps move a weight by lr * g / (|g| + eps): where |g| is at the level of fp32 rounding noise the
        step (up to +-lr) is decided by that noise, on the oracle's side as much as on ours -- the gradients themselves
        are compared in test_parity_gpu.py.  So: every element within the tolerance, except a handful (<= 2 per million)
        that may differ by what the optimizer can move them in `steps` steps."""
        a, b = p.detach().double().cpu(), ref.double()
        d = (a - b).abs()
        tol = 2e-5 + 5e-3 * float(b.abs().max())
        n_bad = int((d > tol).sum())
        assert n_bad <= max(2, int(2e-6 * d.numel())), f'{what}: {n_bad} of {d.numel()} elements beyond {tol:.2e} (max {float(d.max()):.2e})'
        assert float(d.max()) <= 2.2 * lr * steps, f'{what}: max err {float(d.max()):.3e} > what {steps} steps of lr {lr} can move'

    bad = []
    for k, p in G.named_parameters():
        if k.endswith('.0.conv.bias'):
            # a conv bias in front of a BatchNorm has a mathematically zero gradient; both sides hold ~1e-7 of
            # rounding noise there, which Adam turns into +-lr steps of arbitrary sign: not comparable
            continue
        try:
            close_params(p, Go[k], hp.g_learning_rate, g_step, f'G param {k}')
        except AssertionError as e:
            bad.append(str(e))
    for k, p in D.named_parameters():
        try:
            close_params(p, Do[k], hp.d_learning_rate, d_step, f'D param {k}')
        except AssertionError as e:
            bad.append(str(e))
    assert not bad, '\n'.join(bad)


def test_train_loop_end_to_end(dev, tmp_path):
    """gantron_amd.train.train() on precomputed mels: schedule runs, a checkpoint with the reference's
    dict schema is written, and training resumes from it."""
    from gantron_amd import train as T
    from gantron_amd.hparams import HParams
    rs = np.random.default_rng(0)
    wavs = tmp_path / 'mels'
    wavs.mkdir()
    lines = []
    for i in range(12):
        n = int(rs.integers(12, 30))
        np.save(wavs / f'u{i}.npy', np.clip(rs.normal(-5.5, 2.2, (16, n)), -11.5, 2).astype(np.float32))
        words = ' '.join(rs.choice(['alpha', 'beta', 'gamma', 'delta', 'tau'], size=int(rs.integers(2, 5))))
        lines.append(f'u{i}.npy|{words}.')
    flist = tmp_path / 'train.txt'
    flist.write_text('\n'.join(lines) + '\n')
    hp = HParams(MID + ',batch_size=4,iters_per_checkpoint=6,disc_warmp_up=8,attn_steps=4,epochs=100,load_mel_from_disk=True')
    hp.training_files = [str(flist), str(flist)]
    hp.validation_files = [str(flist), str(flist)]
    hp.iterations = 12
    out = tmp_path / 'out'
    T.train(str(out), None, False, 1, 0, 'g', hp, str(wavs) + '/')
    ckpts = sorted(os.listdir(out))
    assert ckpts, 'no checkpoint written'
    ck = torch.load(out / ckpts[-1], map_location='cpu')
    assert set(ck) == {'iteration', 'state_dict', 'g_optimizer', 'g_learning_rate', 'd_optimizer', 'd_learning_rate'}
    assert 'decoder.attention_rnn.weight_ih' in ck['state_dict'] and not any(k.startswith('discriminator') for k in ck['state_dict'])
    assert all(torch.isfinite(v).all() for v in ck['state_dict'].values() if v.dtype.is_floating_point)
    hp2 = HParams(MID + ',batch_size=4,iters_per_checkpoint=100,disc_warmp_up=8,attn_steps=4,epochs=100,load_mel_from_disk=True')
    hp2.training_files, hp2.validation_files = hp.training_files, hp.validation_files
    hp2.iterations = ck['iteration'] + 3
    out2 = tmp_path / 'out2'
    T.train(str(out2), str(out / ckpts[-1]), False, 1, 0, 'g', hp2, str(wavs) + '/')
    assert os.listdir(out2)


def _pad_keeps(keeps, steps, Bp, dev):
    out = []
    for k in keeps:                                   # (n, B, P) bool -> (steps, Bp, P) uint8
        t = torch.zeros(steps, Bp, k.shape[2], dtype=torch.uint8)
        t[:k.shape[0], :k.shape[1]] = torch.as_tensor(k).to(torch.uint8)
        out.append(t.to(dev))
    return out


@pytest.mark.parametrize('tag,frames', [('stop', 5), ('max', 40)])
def test_inference_matches_reference_golden(dev, golden_dir, tag, frames):
    """Tacotron2.inference against the reference's own run (tests/golden/tiny_inference.npz): the stop gate
    fires on the same frame / the loop is cut at max_decoder_steps (crossing a host-check chunk boundary)."""
    from gantron_amd.hparams import HParams
    from gantron_amd.model import Tacotron2
    z = np.load(os.path.join(golden_dir, 'tiny_inference.npz'))
    hp = HParams(str(z['hparams']))
    G = Tacotron2(hp).to(dev).eval()
    G.load_state_dict(common.unpack_dict(z, f'{tag}/state'))
    G.rng_override = dict(prenet_keep=_pad_keeps([z[f'{tag}/keep0'], z[f'{tag}/keep1']], hp.max_decoder_steps, 16, dev))
    out = G.inference(torch.from_numpy(z['text']), style=torch.from_numpy(z['style']))
    ref = common.unpack_dict(z, f'{tag}/out')
    assert out[0].shape[2] == frames
    for name, t in zip(('mel', 'mel_post', 'gate', 'align'), out):
        close(t, ref[name], rtol=1e-3, atol=1e-5, what=f'inference[{tag}].{name}')


@pytest.mark.parametrize('bf16', [False, True])
def test_inference_batch_against_oracle(dev, bf16):
    """Free-running decode at a mid-size configuration, B=3 rows (padded to 16), 37 steps: every frame of
    every item against the oracle fed the same Prenet masks and style.  bf16: hparams.fp16_run (bf16 operands in the
    two LSTM products, emulated operand for operand by the oracle; 37 free-running steps feed rounding differences
    back, hence the looser tolerance)."""
    from gantron_amd.hparams import HParams
    from gantron_amd.model import Tacotron2
    from oracle import gantron_ref as R
    hp = HParams(MID + ',max_decoder_steps=37' + (',fp16_run=True' if bf16 else ''))
    Gp, _ = R.init_params(hp, seed=6)
    Gp['decoder.gate_layer.linear_layer.bias'] = torch.full((1,), -30.0)      # never stop: compare all 37 frames
    g = torch.Generator().manual_seed(5)
    B, Ti = 3, 19
    text = torch.randint(1, hp.n_symbols, (B, Ti), generator=g)
    style = torch.rand(B, hp.noise_size, generator=g)
    keeps = [torch.rand(37, B, hp.prenet_dim, generator=g) >= 0.5 for _ in range(2)]
    G = Tacotron2(hp).to(dev).eval()
    G.load_state_dict(_with_counters(Gp))
    G.rng_override = dict(prenet_keep=_pad_keeps(keeps, 37, 16, dev))
    got = G.inference(text, style=style)
    want = R.generator_inference(Gp, text, style, keeps, max_decoder_steps=37, bf16_products=bf16)
    assert got[0].shape[2] == 37
    for name, a, b in zip(('mel', 'mel_post', 'gate', 'align'), got, want):
        close(a, b, rtol=5e-3 if bf16 else 1e-3, atol=1e-5, what=f'inference {name}')


def test_inference_with_encoder_inputs(dev):
    """hparams.encoder_inputs=True (labels + style as channels of the first encoder convolution, speaker embedding in
    the memory): free-running decode against the oracle; the style is passed in the reference's (B, noise, 1) layout."""
    from gantron_amd.hparams import HParams
    from gantron_amd.model import Tacotron2
    from oracle import gantron_ref as R
    hp = HParams(common.ENCIN_HPARAMS + ',max_decoder_steps=21')
    Tacotron2(hp)                                   # applies the reference's hparams side effects (model.py:589-591)
    Gp, _ = R.init_params(hp, seed=8)
    Gp['decoder.gate_layer.linear_layer.bias'] = torch.full((1,), -30.0)
    g = torch.Generator().manual_seed(9)
    B, Ti = 2, 11
    text = torch.randint(1, hp.n_symbols, (B, Ti), generator=g)
    style = torch.rand(B, hp.noise_size, generator=g)
    emotions = torch.eye(5)[torch.tensor([1, 3])]
    speaker = torch.tensor([4, 7])
    keeps = [torch.rand(21, B, hp.prenet_dim, generator=g) >= 0.5 for _ in range(2)]
    G = Tacotron2(hp).to(dev).eval()
    G.load_state_dict(_with_counters(Gp))
    G.rng_override = dict(prenet_keep=_pad_keeps(keeps, 21, 16, dev))
    got = G.inference(text, style=style[:, :, None], emotions=emotions, speaker=speaker)
    want = R.generator_inference(Gp, text, style, keeps, speaker=speaker, emotions=emotions, max_decoder_steps=21,
                                 encoder_inputs=True)
    for name, a, b in zip(('mel', 'mel_post', 'gate', 'align'), got, want):
        close(a, b, rtol=1e-3, atol=1e-5, what=f'inference(encoder_inputs) {name}')


def _write_wavs(root, n, rs, sr=22050):
    from scipy.io import wavfile
    lines = []
    for i in range(n):
        m = int(rs.integers(3000, 9000))
        audio = np.clip(rs.normal(0, 0.25, m), -1, 1)
        wavfile.write(root / f'w{i}.wav', sr, (audio * 32767).astype(np.int16))
        words = ' '.join(rs.choice(['alpha', 'beta', 'gamma', 'delta', 'tau'], size=int(rs.integers(2, 6))))
        lines.append(f'w{i}.wav|{words}.')
    return lines


def test_device_batches_match_per_item_loader(dev, tmp_path):
    """SURVEY 8f-1 data path: raw audio collated on the host, ONE batched STFT-mel per batch on the device
    (ragged reflect padding) == the reference-style per-item loader + TextMelCollate on the same items."""
    from torch.utils.data import DataLoader
    from gantron_amd.data_utils import (DeviceBatches, TextAudioCollate, TextAudioLoader, TextMelCollate, TextMelLoader)
    from gantron_amd.hparams import HParams
    from gantron_amd.stft import TacotronSTFT
    rs = np.random.default_rng(4)
    wavs = tmp_path / 'wavs'
    wavs.mkdir()
    flist = tmp_path / 'list.txt'
    flist.write_text('\n'.join(_write_wavs(wavs, 10, rs)) + '\n')
    hp = HParams('batch_size=5')
    files = [str(flist), str(flist)]
    per_item = DataLoader(TextMelLoader(files, hp, str(wavs) + '/'), batch_size=5, shuffle=False,
                          collate_fn=TextMelCollate(1))
    raw = TextAudioLoader(files, hp, str(wavs) + '/')
    batched = DeviceBatches(DataLoader(raw, batch_size=5, shuffle=False, num_workers=1, pin_memory=True,
                                       collate_fn=TextAudioCollate()), TacotronSTFT(*raw._stft_args).to(dev))
    n = 0
    for want, got in zip(per_item, batched):
        n += 1
        text, il, mel, gate, spk, emo, ol = want
        assert got[1].device.type == 'cpu' and got[6].device.type == 'cpu'       # lengths stay on the host
        assert torch.equal(got[0].cpu(), text) and torch.equal(got[1], il) and torch.equal(got[6], ol)
        close(got[2], mel, rtol=0, atol=2e-4, what='mel_padded')
        assert torch.equal(got[3].cpu(), gate)
    assert n == 2


def test_train_loop_on_raw_audio(dev, tmp_path):
    """train() on wav files: DataLoader worker decodes, DeviceBatches computes the mels, the schedule runs."""
    from gantron_amd import train as T
    from gantron_amd.hparams import HParams
    rs = np.random.default_rng(1)
    wavs = tmp_path / 'wavs'
    wavs.mkdir()
    flist = tmp_path / 'train.txt'
    flist.write_text('\n'.join(_write_wavs(wavs, 8, rs)) + '\n')
    hp = HParams(MID + ',batch_size=4,iters_per_checkpoint=100,disc_warmp_up=7,attn_steps=4,epochs=100')
    hp.training_files = hp.validation_files = [str(flist), str(flist)]
    hp.iterations = 9
    out = tmp_path / 'out'
    T.train(str(out), None, False, 1, 0, 'g', hp, str(wavs) + '/')
    assert os.path.isdir(out)


_DP_WORKER = r'''
import os, sys
rank, port, root, out = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4]
for p in (root, os.path.join(root, 'tests'), os.path.join(root, 'tests', 'golden')):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
import common
from gantron_amd.distributed import apply_gradient_allreduce
from gantron_amd.hparams import HParams
from gantron_amd.loss_function import Tacotron2Loss
from gantron_amd.model import Discriminator, Tacotron2
from gantron_amd.optim import FlatAdam
from gantron_amd.train_steps import discriminator_step, generator_step
world = int(sys.argv[5])
hp = HParams(sys.argv[6])
backend = sys.argv[7] if len(sys.argv) > 7 else 'gloo'
dev = torch.device('cuda', rank if backend == 'nccl' else 0)       # RCCL: one GPU per rank; gloo rehearsal: both on device 0
torch.cuda.set_device(dev)
if world > 1:
    kw = dict(device_id=dev) if backend == 'nccl' else {}
    dist.init_process_group(backend, init_method=f'tcp://127.0.0.1:{port}', world_size=world, rank=rank, **kw)
torch.manual_seed(7 + 13 * (rank if world > 1 else 0))     # different initial weights per rank: the wrap must broadcast rank 0's
G, D = Tacotron2(hp).to(dev).train(), Discriminator(hp).to(dev).train()
g_opt, d_opt = FlatAdam(G, lr=1e-3), FlatAdam(D, lr=1e-3)
if world > 1:
    G, D = apply_gradient_allreduce(G), apply_gradient_allreduce(D)
crit = Tacotron2Loss()
lens = [([17, 12, 12, 5], [26, 33, 19, 8]), ([15, 9, 6, 4], [31, 22, 14, 9])][rank]
batch = common.synthetic_batch(hp, lens[0], lens[1], seed=20 + rank)
G.zero_grad(); D.zero_grad()
_, gen = generator_step(G, D, g_opt, crit, batch, hp, iteration=0)
g_grad = G._flat_grad.detach().cpu().numpy().copy()
discriminator_step(G, D, d_opt, batch, gen, hp)
d_grad = D._flat_grad.detach().cpu().numpy().copy()
from gantron_amd.optim import flat_range
np.savez(os.path.join(out, f'w{world}_rank{rank}.npz'), g_grad=g_grad, d_grad=d_grad, postnet_range=np.array(flat_range(G, 'postnet.')),
         g_param=G._flat_param.detach().cpu().numpy(), d_param=D._flat_param.detach().cpu().numpy())
if world > 1:
    dist.destroy_process_group()
'''


def test_data_parallel_step_two_processes_on_one_gpu(dev, tmp_path):
    """The distributed training step with world_size 2 (two processes sharing the one GPU of the test box, gloo
    carrying the CUDA flat gradient buffers): both ranks start from rank 0's weights, and after the generator /
    discriminator backward every rank holds the MEAN of the two per-rank gradients, hence identical parameters after
    the optimizer step (reference distributed.py:126-173, train.py:79-91)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'dp_worker.py'
    script.write_text(_DP_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), '29587', root, str(tmp_path), '2', MID], env=env)
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    r0, r1 = np.load(tmp_path / 'w2_rank0.npz'), np.load(tmp_path / 'w2_rank1.npz')
    for k in ('g_grad', 'd_grad', 'g_param', 'd_param'):
        close(torch.from_numpy(r0[k]), torch.from_numpy(r1[k]), rtol=1e-6, atol=1e-7, what=f'rank0 vs rank1 {k}')
    assert float(np.abs(r0['g_grad']).max()) > 0 and float(np.abs(r0['d_grad']).max()) > 0
    # ... and that common value IS the mean of the two single-process gradients (each rank's batch run alone from rank
    # 0's initial weights): a bucket reduced twice, or left out of the all-reduce, or a flat range that covers the wrong
    # parameters would keep the ranks equal to each other but not to the mean.  The early (side-stream) bucket of
    # `postnet.` is part of what this drives.
    for r in range(2):
        p = subprocess.Popen([sys.executable, str(script), str(r), '0', root, str(tmp_path), '1', MID], env=env)
        assert p.wait(timeout=600) == 0
    s0, s1 = np.load(tmp_path / 'w1_rank0.npz'), np.load(tmp_path / 'w1_rank1.npz')
    close(torch.from_numpy(r0['g_grad']), torch.from_numpy((s0['g_grad'] + s1['g_grad']) / 2), rtol=1e-4, atol=1e-7,
          what='averaged generator gradient vs mean of the single-process gradients')
    layout = np.load(tmp_path / 'w2_rank0.npz')['postnet_range']
    assert layout[1] > layout[0] and float(np.abs(r0['g_grad'][layout[0]:layout[1]]).max()) > 0


def test_data_parallel_step_rccl_two_gpus(tmp_path):
    """The same experiment over RCCL ('nccl' backend), one process per GPU: runs wherever two MI355X are visible and skips
    itself on a one-GPU box.  Besides rank 0 == rank 1 it checks that the averaged gradient IS the mean of the two
    single-process gradients (ranks re-run alone, world_size 1, from rank 0's initial weights)."""
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL all-reduce between two ranks)')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'dp_worker.py'
    script.write_text(_DP_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, str(script), str(r), '29589', root, str(tmp_path), '2', MID, 'nccl'], env=env)
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    r0, r1 = np.load(tmp_path / 'w2_rank0.npz'), np.load(tmp_path / 'w2_rank1.npz')
    for k in ('g_grad', 'd_grad', 'g_param', 'd_param'):
        close(torch.from_numpy(r0[k]), torch.from_numpy(r1[k]), rtol=1e-6, atol=1e-7, what=f'RCCL rank0 vs rank1 {k}')
    assert float(np.abs(r0['g_grad']).max()) > 0 and float(np.abs(r0['d_grad']).max()) > 0
    # the same two batches, one process each, from rank 0's initial weights: the mean of their gradients is what the
    # data-parallel ranks must hold (the discriminator step sees the generator output of the un-averaged first step
    # in both settings, so the generator gradient is compared)
    for r in range(2):
        p = subprocess.Popen([sys.executable, str(script), str(r), '0', root, str(tmp_path), '1', MID], env=env)
        assert p.wait(timeout=600) == 0
    s0, s1 = np.load(tmp_path / 'w1_rank0.npz'), np.load(tmp_path / 'w1_rank1.npz')
    close(torch.from_numpy(r0['g_grad']), torch.from_numpy((s0['g_grad'] + s1['g_grad']) / 2), rtol=1e-4, atol=1e-7,
          what='averaged generator gradient')


_CABI_ALLREDUCE = r'''
import ctypes, os, sys
root, rank, world, idfile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, root)
import torch
from gantron_amd._lib import check, lib
rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'))
class UniqueId(ctypes.Structure):
    _fields_ = [('internal', ctypes.c_char * 128)]
torch.cuda.set_device(rank % torch.cuda.device_count())
uid = UniqueId()
if rank == 0:
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    open(idfile + '.tmp', 'wb').write(ctypes.string_at(ctypes.byref(uid), 128))
    os.replace(idfile + '.tmp', idfile)
else:
    import time
    while not os.path.exists(idfile):
        time.sleep(0.05)
    ctypes.memmove(ctypes.byref(uid), open(idfile, 'rb').read(), 128)
comm = ctypes.c_void_p()
rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
assert rccl.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) == 0
n = 1000003                                       # not a multiple of 4: the scalar tail of the scaling pass
buf = torch.arange(n, device='cuda', dtype=torch.float32) * (rank + 1) * 1e-3
want = torch.arange(n, dtype=torch.float64) * 1e-3 * sum(r + 1 for r in range(world)) / world
st = torch.cuda.current_stream().cuda_stream
check(lib().gt_allreduce_avg(comm, buf.data_ptr(), n, st), 'gt_allreduce_avg')
torch.cuda.synchronize()
err = float((buf.cpu().double() - want).abs().max() / want.abs().max())
assert err < 1e-6, err
# the same through the caller's own entry points (a host whose RCCL the library cannot see)
buf2 = torch.arange(n, device='cuda', dtype=torch.float32) * (rank + 1) * 1e-3
fn = lambda f: ctypes.cast(f, ctypes.c_void_p)
check(lib().gt_allreduce_avg_with(comm, buf2.data_ptr(), n, fn(rccl.ncclAllReduce), fn(rccl.ncclCommCount), st), 'gt_allreduce_avg_with')
torch.cuda.synchronize()
assert torch.equal(buf, buf2)
rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
rccl.ncclCommDestroy(comm)
print('OK', rank, err)
'''


@pytest.mark.parametrize('world', [1, 2])
def test_c_abi_gradient_average_over_rccl(tmp_path, world):
    """gt_allreduce_avg (the collective behind the C ABI, distributed.py:137-161): a communicator made with RCCL's
    own C API, no torch.distributed -- one rank on the one-GPU box (sum over one rank, no scaling), two ranks with one GPU
    each wherever two are visible (sum, then / 2; n is not a multiple of 4)."""
    import subprocess
    import sys
    if world > torch.cuda.device_count():
        pytest.skip('needs two GPUs (one RCCL rank per GPU)')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'cabi_allreduce.py'
    script.write_text(_CABI_ALLREDUCE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, str(script), root, str(r), str(world), str(tmp_path / 'nccl_id')], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0 and 'OK' in out, out[-2000:]


def test_steps_with_linear_discriminator(dev):
    """hparams.discriminator_type = 'linear': the generator step (adversarial gradient into the spectrogram) and the
    discriminator step run through the same step bodies and produce finite losses and non-zero gradients."""
    from gantron_amd import train as T
    from gantron_amd.hparams import HParams
    from gantron_amd.loss_function import Tacotron2Loss
    from gantron_amd.model import LinearDiscriminator
    from gantron_amd.optim import FlatAdam
    from gantron_amd.train_steps import discriminator_step, generator_step
    hp = HParams(MID.replace('discriminator_window=6', 'discriminator_window=8') + ',discriminator_type=linear')
    G, D = T.load_model(hp)
    assert isinstance(D, LinearDiscriminator)
    g_opt, d_opt = FlatAdam(G, lr=1e-3), FlatAdam(D, lr=1e-3)
    batch = common.synthetic_batch(hp, [17, 12, 12, 5], [26, 33, 19, 9], seed=2)
    losses, gen = generator_step(G.train(), D.train(), g_opt, Tacotron2Loss(), batch, hp, iteration=0)
    dl = discriminator_step(G, D, d_opt, batch, gen, hp)
    assert all(torch.isfinite(v).all() for v in losses.values() if v is not None)
    assert torch.isfinite(dl['total']) and float(D._flat_grad.abs().max()) > 0 and float(G._flat_grad.abs().max()) > 0


def test_discriminator_step_with_gradient_penalty(dev):
    """hparams.clipping_value=0, gradient_penalty_lambda=10 (train.py:326-333): the D step adds lambda * penalty; its
    D-parameter gradient equals the plain step's gradient + lambda * the penalty's gradient (same masks)."""
    from gantron_amd import train as T
    from gantron_amd.hparams import HParams
    from gantron_amd.optim import FlatAdam
    from gantron_amd.train_steps import discriminator_step
    grads = {}
    for lam in (0, 10):
        hp = HParams(MID + f',clipping_value=0,gradient_penalty_lambda={lam}')
        G, D = T.load_model(hp)
        d_opt = FlatAdam(D, lr=0.0)
        batch = common.synthetic_batch(hp, [17, 12, 12, 5], [26, 33, 19, 9], seed=2)
        g = torch.Generator().manual_seed(3)
        fake = (torch.randn(batch[2].shape, generator=g) * 2 - 5).to(dev)
        out = discriminator_step(G, D.train(), d_opt, batch, [fake, batch[6].to(dev)], hp)
        assert ('gradient_penalty' in out) == (lam > 0)
        assert torch.isfinite(out['total'])
        grads[lam] = D._flat_grad.clone()
    assert float((grads[10] - grads[0]).abs().max()) > 0


@pytest.mark.parametrize('launcher', ['self', 'torchrun'])
def test_bench_two_ranks_rehearsal(dev, tmp_path, launcher):
    """bench.py's N > 1 control flow (per-rank batches, barriers, MAX / SUM reductions of the timings, rank 0 printing
    the one JSON line) on the 1-GPU test box: two ranks share device 0 and gloo carries the CUDA tensors
    (GANTRON_BENCH_DEVICE / GANTRON_BENCH_BACKEND exist for this rehearsal only; the driver's launch uses RCCL)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GANTRON_BENCH_DEVICE='0', GANTRON_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    # 'self': the plain command, `python bench.py --gpus 2` with no launcher and no WORLD_SIZE -- bench.py starts its own
    # two ranks (before anything in the parent touches the GPU); 'torchrun': the driver's documented N > 1 launch
    env.pop('WORLD_SIZE', None), env.pop('RANK', None), env.pop('LOCAL_RANK', None)
    tail = [os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--batch', '8']
    cmd = [sys.executable] + tail if launcher == 'self' else \
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', '29611'] + tail
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]                # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 1 and d['scaling'] == 'weak' and d['cpu_baseline'] is None
    assert d['dist']['backend'] == 'gloo' and d['dist']['world_size'] == 2 and d['dist']['grad_checksum_equal_on_all_ranks']
    assert len(d['dist']['ms_per_step_by_rank']) == 2 and d['dist']['ms_per_step_max'] <= d['ms_per_step'] * 1.0001
    assert d['dist']['ms_per_step_min'] > 0 and len(set(d['dist']['valid_frames_by_rank'])) == 2       # per-rank batches differ
    assert d['config']['global_batch'] == 16 and d['value'] > 0 and d['roofline'] is not None


def test_bf16_mode_trains_like_fp32(dev):
    """hparams.fp16_run (bf16 operands in the LSTM products and layer GEMMs, fp32 master weights and accumulation): 10
    generator+discriminator pairs from the same initial weights, batch and dropout seeds follow the fp32 run -- the
    generator loss of every step stays within 2 % of the fp32 trajectory and both go down."""
    from gantron_amd import train as T
    from gantron_amd.hparams import HParams
    from gantron_amd.loss_function import Tacotron2Loss
    from gantron_amd.optim import FlatAdam
    from gantron_amd.train_steps import discriminator_step, generator_step
    traj = {}
    for mode in ('fp32', 'bf16'):
        hp = HParams(MID + (',fp16_run=True' if mode == 'bf16' else ''))
        torch.manual_seed(hp.seed)
        G, D = T.load_model(hp)
        g_opt, d_opt = FlatAdam(G, lr=1e-3, weight_decay=hp.weight_decay), FlatAdam(D, lr=1e-3, weight_decay=hp.weight_decay)
        batch = common.synthetic_batch(hp, [17, 12, 12, 5], [26, 33, 19, 9], seed=2)
        crit = Tacotron2Loss()
        losses = []
        for it in range(10):
            out, gen = generator_step(G.train(), D.train(), g_opt, crit, batch, hp, iteration=it)
            discriminator_step(G, D, d_opt, batch, gen, hp)
            losses.append(float(out['total']))
        traj[mode] = losses
    a, b = np.array(traj['fp32']), np.array(traj['bf16'])
    assert np.all(np.isfinite(b)) and b[-1] < b[0] and a[-1] < a[0], (a, b)
    assert np.max(np.abs(a - b) / np.abs(a)) < 2e-2, (a, b)
