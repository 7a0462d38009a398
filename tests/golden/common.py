"""Shared helpers for golden-vector generation and the tests that read the vectors.

Nothing here touches /root/reference.  `make_golden.py` (build container only) uses these
helpers to drive the reference's own classes; the tests use the same helpers to rebuild the
identical inputs for the oracle / the HIP path.
"""
import math

import numpy as np
import torch

TINY_HPARAMS = ("use_labels=False,use_noise=True,n_mel_channels=8,encoder_embedding_dim=16,"
                "symbols_embedding_dim=16,attention_rnn_dim=32,decoder_rnn_dim=32,prenet_dim=8,"
                "attention_dim=8,attention_location_n_filters=4,attention_location_kernel_size=5,"
                "postnet_embedding_dim=16,noise_size=16,discriminator_dim=16,discriminator_window=4")
FULL_HPARAMS = "use_labels=False,use_noise=True"
VESUS_HPARAMS = "use_labels=True,use_noise=True,vesus_path=/nonexistent/vesus/"
# labels + noise fed to the encoder's first convolution instead of the decoder memory (hparams.py:102)
ENCIN_HPARAMS = (TINY_HPARAMS.replace('use_labels=False', 'use_labels=True') +
                 ",vesus_path=/nonexistent/vesus/,speakers_embedding=4,encoder_inputs=True")


class MaskStream:
    """Deterministic source of dropout keep-masks and style noise, consumed in call order."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)

    def keep(self, shape, p):
        return torch.rand(tuple(shape), generator=self.g) >= p

    def uniform(self, shape):
        return torch.rand(tuple(shape), generator=self.g)


def draw_generator_rng(stream, hp, B, Ti, To, train=True):
    """Masks in the order the reference's Tacotron2.forward calls F.dropout / torch.rand
    (model.py:197, 275, 104, 393-394/408-409 per step, 148-149)."""
    E = hp.encoder_embedding_dim
    rng = {}
    noise = hp.noise_size if hp.use_noise else 0
    if getattr(hp, 'encoder_inputs', False):       # Encoder.append_noise draws (B, noise, 1) first (model.py:184-187)
        rng['noise'] = stream.uniform((B, noise, 1))[:, :, 0] if noise > 0 else None
    rng['enc_keep'] = [stream.keep((B, E, Ti), 0.5) for _ in range(hp.encoder_n_convolutions)]
    if not getattr(hp, 'encoder_inputs', False):
        rng['noise'] = stream.uniform((B, 1, noise))[:, 0, :] if noise > 0 else None
    rng['prenet_keep'] = [stream.keep((To + 1, B, hp.prenet_dim), 0.5) for _ in range(2)]
    att, dec = [], []
    for _ in range(To):
        att.append(stream.keep((B, hp.attention_rnn_dim), hp.p_attention_dropout))
        dec.append(stream.keep((B, hp.decoder_rnn_dim), hp.p_decoder_dropout))
    rng['att_keep'] = torch.stack(att)
    rng['dec_keep'] = torch.stack(dec)
    chans = [hp.postnet_embedding_dim] * (hp.postnet_n_convolutions - 1) + [hp.n_mel_channels]
    rng['post_keep'] = [stream.keep((B, c, To), 0.5) for c in chans]
    return rng


def disc_channels(hp):
    in_dim = hp.discriminator_window * hp.n_mel_channels
    first = min((in_dim // hp.discriminator_dim + 1) * hp.discriminator_dim, 1024)
    return [in_dim, first, hp.discriminator_dim, hp.discriminator_dim, hp.n_mel_channels]


def disc_windows(T, window):
    return T // window + (1 if T % window else 0)


def draw_disc_keep(stream, hp, B, T):
    W = disc_windows(T, hp.discriminator_window)
    return [stream.keep((B, c, W), 0.5) for c in disc_channels(hp)[1:]]


def synthetic_batch(hp, in_lens, out_lens, seed, vesus=False):
    """A collated batch in the layout of TextMelCollate (data_utils.py:88-131): sorted by text
    length (descending, caller's job), zero padded, gate = 1 from the last valid frame on."""
    g = torch.Generator().manual_seed(seed)
    B, Ti, To = len(in_lens), max(in_lens), max(out_lens)
    text = torch.zeros(B, Ti, dtype=torch.long)
    mel = torch.zeros(B, hp.n_mel_channels, To)
    gate = torch.zeros(B, To)
    for b in range(B):
        text[b, :in_lens[b]] = torch.randint(1, hp.n_symbols, (in_lens[b],), generator=g)
        mel[b, :, :out_lens[b]] = (torch.randn(hp.n_mel_channels, out_lens[b], generator=g) * 2.2 - 5.5
                                   ).clamp(-11.52, 2.0)
        gate[b, out_lens[b] - 1:] = 1
    spk = torch.zeros(B)
    emo = torch.zeros(B, 5)
    if vesus:
        spk = torch.randint(0, 11, (B,), generator=g).float()
        emo = torch.eye(5)[torch.randint(0, 5, (B,), generator=g)]
    return (text, torch.tensor(in_lens), mel, gate, spk, emo, torch.tensor(out_lens))


def closed_form_weights(shapes, seed):
    """Weights for the full-size fixtures: N(0,1)/sqrt(fan_in)-ish values drawn from a seeded
    generator in sorted-name order; BN running_var positive, num_batches_tracked zero."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        if name.endswith('num_batches_tracked'):
            out[name] = torch.zeros(shape, dtype=torch.long)
            continue
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        w = torch.randn(shape, generator=g) / math.sqrt(max(fan_in, 1))
        if len(shape) == 1:
            w = torch.randn(shape, generator=g) * 0.1
        if name.endswith('running_var'):
            w = w.abs() + 0.5
        if name.endswith('.1.weight'):       # BN gamma near 1
            w = 1.0 + w
        out[name] = w
    return out


def summarize(t, n_probe=64, seed=7):
    """mean / abs-max / L2 / sparse probes of a tensor (committed instead of the full tensor)."""
    t = t.detach().double().reshape(-1)
    g = torch.Generator().manual_seed(seed + t.numel())
    idx = torch.randint(0, t.numel(), (min(n_probe, t.numel()),), generator=g)
    return dict(mean=np.float64(t.mean()), absmax=np.float64(t.abs().max()),
                l2=np.float64(t.norm()), probe_idx=idx.numpy(), probe_val=t[idx].numpy())


def pack_dict(prefix, d):
    return {f'{prefix}/{k}': (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def unpack_dict(npz, prefix):
    pre = prefix + '/'
    return {k[len(pre):]: torch.from_numpy(np.asarray(npz[k])) for k in npz.files if k.startswith(pre)}


# ---- loader / collate fixture (loader_collate.npz): a small dataset both the reference and the repo read
LOADER_TEXTS = ['Printing, in the only sense with which we are at present concerned,',
                'differs from most if not from all the arts and crafts represented in the Exhibition',
                'Mr. Smith and Dr. Jones met Mrs. Brown; "quoted" text -- dashes?',
                'in being comparatively modern.',
                'Hello!',
                'For although the Chinese took impressions from wood blocks engraved in relief for centuries',
                'produced the block books, which were the immediate predecessors of the true printed book,',
                'A b c']
VESUS_TEXTS = ['I am happy today.', 'Leave me alone!', 'What a day.']


def write_loader_dataset(root, n_mel=8, seed=3):
    """Writes mels (.npy, load_mel_from_disk) and two filelists under `root`; returns (lj_list, vesus_list)."""
    import os
    g = torch.Generator().manual_seed(seed)
    os.makedirs(os.path.join(root, 'mels'), exist_ok=True)
    os.makedirs(os.path.join(root, 'vesus', 'Happy'), exist_ok=True)
    os.makedirs(os.path.join(root, 'vesus', 'Angry'), exist_ok=True)
    lj = os.path.join(root, 'lj.txt')
    with open(lj, 'w') as f:
        for i, t in enumerate(LOADER_TEXTS):
            frames = int(torch.randint(5, 40, (1,), generator=g))
            np.save(os.path.join(root, 'mels', f'a{i}.npy'), torch.randn(n_mel, frames, generator=g).numpy())
            f.write(f'mels/a{i}.npy|{t}\n')
    ve = os.path.join(root, 'vesus.txt')
    with open(ve, 'w') as f:
        for i, t in enumerate(VESUS_TEXTS):
            emo = 'Happy' if i % 2 == 0 else 'Angry'
            frames = int(torch.randint(5, 40, (1,), generator=g))
            np.save(os.path.join(root, 'vesus', emo, f'v{i}.npy'), torch.randn(n_mel, frames, generator=g).numpy())
            votes = ', '.join(str(v) for v in torch.rand(5, generator=g).round(decimals=2).tolist())
            f.write(f'vesus/{emo}/v{i}.npy|{t}|{i + 3}|{votes}\n')
    return lj, ve
