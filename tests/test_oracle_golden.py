"""Pin the CPU oracle (oracle/gantron_ref.py) against vectors produced by the reference's own
classes (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import common
from gantron_amd.hparams import HParams
from oracle import gantron_ref as R


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _close(a, b, rtol=1e-4, atol=1e-6, what=''):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item() if a.numel() else 0.0
    scale = b.abs().max().item() if b.numel() else 0.0
    assert err <= atol + rtol * scale, f'{what}: max err {err:.3e} vs scale {scale:.3e}'


def _run_oracle_pair(z, hp, G, Dp):
    in_lens, out_lens = z['in_lens'].tolist(), z['out_lens'].tolist()
    vesus = bool(z['vesus'])
    batch = common.synthetic_batch(hp, in_lens, out_lens, seed=int(z['batch_seed']), vesus=vesus)
    text, il, mel, gate, spk, emo, ol = batch
    stream = common.MaskStream(int(z['mask_seed']))
    B, Ti, To = len(in_lens), max(in_lens), max(out_lens)
    rng = common.draw_generator_rng(stream, hp, B, Ti, To)
    rng['disc_keep'] = common.draw_disc_keep(stream, hp, B, To)
    out, losses, grads, new_stats = R.generator_step_grads(
        G, Dp, (text, il, mel, gate, ol, spk.long(), emo), rng, hp)
    keep_real = common.draw_disc_keep(stream, hp, B, To)
    keep_fake = common.draw_disc_keep(stream, hp, B, To)
    dl, dg = R.discriminator_step_grads(Dp, mel, ol, out[1], ol, keep_real, keep_fake, hp)
    return out, losses, grads, new_stats, dl, dg


@pytest.mark.parametrize('name', ['tiny_pair.npz', 'tiny_encin_pair.npz'])
def test_tiny_pair_matches_reference(golden_dir, name):
    """Every output, loss, gradient and BN buffer of one G step + one D step; tiny_encin_pair.npz is the
    encoder_inputs=True conditioning (labels + noise into the first encoder convolution) with VESUS speakers."""
    z = _load(golden_dir, name)
    hp = HParams(str(z['hparams']))
    G = common.unpack_dict(z, 'g_state_before')
    Dp = common.unpack_dict(z, 'd_state')
    out, losses, grads, new_stats, dl, dg = _run_oracle_pair(z, hp, G, Dp)
    ref_out = common.unpack_dict(z, 'out')
    for name, t in zip(('mel', 'mel_post', 'gate', 'align'), out):
        _close(t, ref_out[name], what=f'out.{name}')
    for k, v in common.unpack_dict(z, 'g_loss').items():
        _close(losses[k], v, what=f'g_loss.{k}')
    ref_grad = common.unpack_dict(z, 'g_grad')
    assert set(ref_grad) == set(grads)
    for k, v in ref_grad.items():
        _close(grads[k], v, rtol=2e-4, atol=2e-6, what=f'g_grad.{k}')
    for k, v in common.unpack_dict(z, 'g_buffers_after').items():
        _close(new_stats[k], v, what=f'buffer.{k}')
    for k, v in common.unpack_dict(z, 'd_loss').items():
        _close(dl[k], v, what=f'd_loss.{k}')
    for k, v in common.unpack_dict(z, 'd_grad').items():
        _close(dg[k], v, rtol=2e-4, atol=2e-6, what=f'd_grad.{k}')


def test_postnet_conv0_weight_grad_sees_masked_mel(golden_dir):
    """The autograd-invisible mask (model.py:638-640) changes the tensor saved by the first
    postnet conv: the fixture's gradient must differ from a 'clean' computation."""
    z = _load(golden_dir, 'tiny_pair.npz')
    g = common.unpack_dict(z, 'g_grad')['postnet.convolutions.0.0.conv.weight']
    assert torch.isfinite(g).all() and g.abs().max() > 0


@pytest.mark.parametrize('name', ['full_pair_summary.npz', 'vesus_pair_summary.npz'])
def test_full_size_summaries(golden_dir, name):
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.skip('summary fixture not generated')
    z = _load(golden_dir, name)
    hp = HParams(str(z['hparams']))
    hp.n_symbols = 148
    shapes_g = {k[len('shape_g/'):]: tuple(z[k]) for k in z.files if k.startswith('shape_g/')}
    shapes_d = {k[len('shape_d/'):]: tuple(z[k]) for k in z.files if k.startswith('shape_d/')}
    G = common.closed_form_weights(shapes_g, int(z['weight_seed']))
    Dp = common.closed_form_weights(shapes_d, int(z['weight_seed']) + 1)
    out, losses, grads, new_stats, dl, dg = _run_oracle_pair(z, hp, G, Dp)
    got = dict(out=dict(zip(('mel', 'mel_post', 'gate', 'align'), out)), g_grad=grads, d_grad=dg,
               g_buffers_after=new_stats)
    for key, tensors in got.items():
        for tname, t in tensors.items():
            pre = f'{key}/{tname}/'
            if pre + 'l2' not in z.files:
                continue
            s = common.summarize(t)
            scale = float(z[pre + 'absmax']) + 1e-12
            assert abs(s['l2'] - float(z[pre + 'l2'])) <= 5e-4 * float(z[pre + 'l2']) + 1e-5, pre
            assert np.abs(s['probe_val'] - z[pre + 'probe_val']).max() <= 5e-4 * scale + 1e-6, pre
    for k in ('mel', 'gate', 'attn', 'adv', 'total'):
        _close(losses[k], z[f'g_loss/{k}'], rtol=2e-4, what=k)
    for k in ('real', 'fake', 'total'):
        _close(dl[k], z[f'd_loss/{k}'], rtol=2e-4, what=k)


def test_op_fixtures(golden_dir):
    z = _load(golden_dir, 'ops.npz')
    _close(R.attention_target(5, 9), z['attn_mask_5_9'], what='attn 5x9')
    _close(R.attention_target(7, 23), z['attn_mask_7_23'], what='attn 7x23')
    x = torch.from_numpy(z['disc_reshape_in'])
    r = R.disc_window_input(x, 20)
    assert r.shape == (2, 1600, 3)
    assert torch.equal(r, torch.from_numpy(z['disc_reshape_out']))
    # adversarial reduction: per-item means over ceil(len/window) windows
    outs, lens = torch.from_numpy(z['adv_outs']), torch.from_numpy(z['adv_lens'])
    import math
    item = [outs[b, :int(math.ceil(float(lens[b]) / 20))].mean().item() for b in range(len(lens))]
    _close(item, z['adv_item_loss'], what='adv per item')
    _close(sum(item) / len(item), z['adv_batch_loss'], what='adv batch')
    # STFT magnitude of real audio + probes of the windowed DFT basis
    mag = R.stft_magnitude(torch.from_numpy(z['stft_audio']))
    _close(mag, z['stft_mag'], rtol=1e-4, atol=1e-5, what='stft magnitude')
    basis = R.stft_basis(1024)
    vals = [basis[i, j].item() for i, j in z['stft_basis_probe_idx']]
    _close(vals, z['stft_basis_probe_val'], rtol=1e-5, atol=1e-7, what='stft basis')


def test_hparams_mirror():
    hp = HParams('use_labels=False,use_noise=True,vesus_path=/a/b,batch_size=64,foo=bar')
    assert hp.use_labels is False and hp.use_noise is True and hp.vesus_path == '/a/b'
    assert hp.batch_size == 64 and hp.foo == 'bar' and hp.n_symbols == 148
    import argparse
    ns = argparse.Namespace(hparams='batch_size=8', batch_size=16, wavs_path=None, n_gpus=2)
    hp.add_params(ns)
    assert hp.batch_size == 8 and hp.n_gpus == 2


@pytest.mark.parametrize('tag,frames', [('stop', 5), ('max', 40)])
def test_inference_matches_reference(golden_dir, tag, frames):
    """Tacotron2.inference (model.py:675-723) of the reference, B=1: the stop-gate run ends on the same
    frame, the max_decoder_steps run is cut at 40; outputs of every frame agree."""
    z = _load(golden_dir, 'tiny_inference.npz')
    hp = HParams(str(z['hparams']))
    P = {k: v for k, v in common.unpack_dict(z, f'{tag}/state').items() if not k.endswith('num_batches_tracked')}
    keeps = [torch.from_numpy(z[f'{tag}/keep0']), torch.from_numpy(z[f'{tag}/keep1'])]
    out = R.generator_inference(P, torch.from_numpy(z['text']), torch.from_numpy(z['style']), keeps,
                                gate_threshold=hp.gate_threshold, max_decoder_steps=hp.max_decoder_steps)
    ref = common.unpack_dict(z, f'{tag}/out')
    assert out[0].shape[2] == frames == ref['mel'].shape[2]
    for name, t in zip(('mel', 'mel_post', 'gate', 'align'), out):
        assert tuple(t.shape) == tuple(ref[name].shape), name       # gate is (B,T,1) on this path
        _close(t, ref[name], rtol=2e-4, atol=1e-5, what=f'{tag}.{name}')


def test_linear_discriminator_matches_reference(golden_dir):
    """LinearDiscriminator.adversarial_loss of the reference (model.py:543-583): loss, every parameter gradient and
    the gradient w.r.t. the input spectrogram; the window starts replay the reference's random.randint sequence."""
    import random
    from gantron_amd.model import LinearDiscriminator
    z = _load(golden_dir, 'linear_disc.npz')
    hp = HParams(str(z['hparams']))
    P = {k: v.clone().requires_grad_(True) for k, v in common.unpack_dict(z, 'state').items()}
    mel = torch.from_numpy(z['mel']).clone().requires_grad_(True)
    random.seed(int(z['random_seed']))
    starts = LinearDiscriminator(hp).window_starts(z['lengths'].tolist())
    keeps = [torch.from_numpy(z[f'keep{i}']) for i in range(3)]
    loss = R.linear_discriminator_loss(P, mel, starts, hp.discriminator_window, keeps)
    loss.backward()
    _close(loss.detach(), z['loss'], what='loss')
    _close(mel.grad, z['dmel'], rtol=2e-4, atol=1e-7, what='d mel')
    for k, v in common.unpack_dict(z, 'grad').items():
        _close(P[k].grad, v, rtol=2e-4, atol=1e-7, what=f'grad {k}')


def test_gradient_penalty_matches_reference(golden_dir):
    """train.gradient_penalty (train.py:32-69): the penalty and d(lambda * penalty)/d(every D parameter), i.e. the
    double backward through the conv discriminator, against the reference's own run."""
    z = _load(golden_dir, 'gradient_penalty.npz')
    hp = HParams(str(z['hparams']))
    Dp = {k: v.clone().requires_grad_(True) for k, v in common.unpack_dict(z, 'd_state').items()}
    real, fake = torch.from_numpy(z['real']), torch.from_numpy(z['fake'])
    lens = torch.from_numpy(z['out_lens'])
    stream = common.MaskStream(int(z['mask_seed']))
    alpha = stream.uniform((real.size(0), 1, 1)).view(-1)
    keeps = common.draw_disc_keep(stream, hp, real.size(0), real.size(2))
    gp = R.gradient_penalty(Dp, real, fake, lens, lens, alpha, hp.discriminator_window, keeps)
    _close(gp, z['gp'], what='gp')
    names = list(Dp)
    grads = torch.autograd.grad(float(z['lam']) * gp, [Dp[k] for k in names], allow_unused=True)
    ref = common.unpack_dict(z, 'd_grad')
    for k, g in zip(names, grads):
        _close(g if g is not None else torch.zeros_like(Dp[k]), ref[k], rtol=2e-4, atol=2e-6, what=f'd_grad.{k}')
