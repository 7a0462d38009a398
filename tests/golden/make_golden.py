"""Generate the golden vectors under tests/golden/ by running the reference's OWN classes.

Build-container only (needs /root/reference; never runs on the GPU box).  The reference is
imported in place (nothing is copied): third-party modules that are absent from this image
(librosa, unidecode, inflect) are registered as in-memory stand-ins because the hot path
never calls them, hard-coded `.cuda()` calls are redirected to the CPU, and the stochastic
inputs (F.dropout masks, the style noise from torch.rand) are replaced by a deterministic
stream so that the oracle and the HIP path can be fed the identical masks.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import common  # noqa: E402

REF = '/root/reference'
sys.dont_write_bytecode = True


# --------------------------------------------------------------------------- reference import
def _install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def _absent(*a, **k):
        raise RuntimeError('third-party function absent in this image (not on the hot path)')

    lib = mod('librosa', load=_absent)
    lib.filters = mod('librosa.filters', mel=lambda *a, **k: np.zeros((a[2], a[1] // 2 + 1), dtype=np.float32))
    lib.util = mod('librosa.util', pad_center=lambda w, n: w, tiny=lambda x: 1e-30, normalize=_absent)
    mod('unidecode', unidecode=lambda s: s)
    mod('inflect', engine=lambda: types.SimpleNamespace(number_to_words=lambda *a, **k: ''))
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.IntTensor = torch.IntTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    for fn_name in ('zeros', 'linspace'):
        orig = getattr(torch, fn_name)

        def wrapped(*a, _orig=orig, **k):
            k.pop('device', None)
            return _orig(*a, **k)
        setattr(torch, fn_name, wrapped)


_install_shims()
import hparams as ref_hparams        # noqa: E402
import model as ref_model            # noqa: E402
import loss_function as ref_loss     # noqa: E402
import stft as ref_stft              # noqa: E402


def _cpu_mask_from_lengths(lengths):
    max_len = int(torch.max(lengths).item())
    return torch.arange(0, max_len)[None, :] < lengths[:, None]


ref_model.get_mask_from_lengths = _cpu_mask_from_lengths


class Injector:
    """Routes F.dropout and torch.rand through a MaskStream and records what was drawn."""

    def __init__(self, stream):
        self.stream = stream
        self.drops = []

    def __enter__(self):
        self._dropout, self._rand = F.dropout, torch.rand

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            keep = self.stream.keep(x.shape, p)
            self.drops.append(keep)
            return x * (keep.to(x.dtype) / (1.0 - p))

        def rand(*size, **kw):
            if 'generator' in kw:
                return self._rand(*size, **kw)
            return self.stream.uniform(size if not isinstance(size[0], (tuple, list)) else size[0])

        F.dropout = dropout
        torch.nn.functional.dropout = dropout
        torch.rand = rand
        return self

    def __exit__(self, *exc):
        F.dropout = self._dropout
        torch.nn.functional.dropout = self._dropout
        torch.rand = self._rand


def build(hp_string, weight_seed=None):
    hp = ref_hparams.HParams(hp_string)
    torch.manual_seed(1234)
    G = ref_model.Tacotron2(hp)
    D = ref_model.Discriminator(hp)
    if weight_seed is not None:
        for m, seed in ((G, weight_seed), (D, weight_seed + 1)):
            sd = m.state_dict()
            m.load_state_dict(common.closed_form_weights({k: v.shape for k, v in sd.items()}, seed))
    return hp, G.train(), D.train()


def run_pair(hp, G, D, batch, mask_seed):
    """One generator step and one discriminator step exactly as train.py:361-397 / 302-344
    compute their losses and gradients (no optimizer update)."""
    res = {}
    stream = common.MaskStream(mask_seed)
    crit = ref_loss.Tacotron2Loss()
    sd_before = {k: v.clone() for k, v in G.state_dict().items()}
    with Injector(stream):
        for p in D.parameters():
            p.requires_grad = False
        G.zero_grad()
        x, y = G.parse_batch(batch)
        y_pred = G(x)
        mel_loss, gate_loss, attn_loss = crit(y_pred, y, x[1], x[-1])
        adv = D.adversarial_loss(y_pred[1], x[-1])
        total = mel_loss + gate_loss + adv + 10 * attn_loss
        total.backward()
    res['out'] = dict(mel=y_pred[0].detach().clone(), mel_post=y_pred[1].detach().clone(),
                      gate=y_pred[2].detach().clone(), align=y_pred[3].detach().clone())
    res['g_loss'] = dict(mel=mel_loss.detach(), gate=gate_loss.detach(), attn=attn_loss.detach(),
                         adv=adv.detach(), total=total.detach())
    res['g_grad'] = {k: p.grad.detach().clone() for k, p in G.named_parameters()}
    res['g_state_before'] = sd_before
    res['g_buffers_after'] = {k: v.clone() for k, v in G.state_dict().items()
                              if k.endswith(('running_mean', 'running_var'))}
    # discriminator step on (real, cached fake)
    fake = y_pred[1].detach()
    with Injector(stream):
        for p in D.parameters():
            p.requires_grad = True
        D.zero_grad()
        real_loss = 1 * D.adversarial_loss(x[2], x[-1])
        fake_loss = -1 * D.adversarial_loss(fake, x[-1])
        d_loss = (real_loss + fake_loss) / 2
        d_loss.backward()
    res['d_loss'] = dict(real=real_loss.detach(), fake=fake_loss.detach(), total=d_loss.detach())
    res['d_grad'] = {k: p.grad.detach().clone() for k, p in D.named_parameters()}
    res['d_state'] = {k: v.clone() for k, v in D.state_dict().items()}
    return res


def save_full(path, hp_string, in_lens, out_lens, res, mask_seed, batch_seed, vesus=False):
    blob = dict(hparams=np.array(hp_string), in_lens=np.array(in_lens), out_lens=np.array(out_lens),
                mask_seed=np.array(mask_seed), batch_seed=np.array(batch_seed), vesus=np.array(vesus))
    for key in ('out', 'g_loss', 'g_grad', 'g_state_before', 'g_buffers_after', 'd_loss', 'd_grad', 'd_state'):
        blob.update(common.pack_dict(key, res[key]))
    np.savez_compressed(path, **blob)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def save_summary(path, hp_string, in_lens, out_lens, res, mask_seed, batch_seed, weight_seed, vesus):
    blob = dict(hparams=np.array(hp_string), in_lens=np.array(in_lens), out_lens=np.array(out_lens),
                mask_seed=np.array(mask_seed), batch_seed=np.array(batch_seed),
                weight_seed=np.array(weight_seed), vesus=np.array(vesus))
    for key in ('out', 'g_grad', 'd_grad', 'g_buffers_after'):
        for name, t in res[key].items():
            for stat, v in common.summarize(t).items():
                blob[f'{key}/{name}/{stat}'] = v
    for key in ('g_loss', 'd_loss'):
        blob.update(common.pack_dict(key, res[key]))
    blob.update({f'shape_g/{k}': np.array(v.shape) for k, v in res['g_state_before'].items()})
    blob.update({f'shape_d/{k}': np.array(v.shape) for k, v in res['d_state'].items()})
    np.savez_compressed(path, **blob)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def tiny_case():
    hp, G, D = build(common.TINY_HPARAMS)
    in_lens, out_lens = [7, 5, 4], [11, 9, 6]
    batch = common.synthetic_batch(hp, in_lens, out_lens, seed=11)
    res = run_pair(hp, G, D, batch, mask_seed=101)
    save_full(os.path.join(HERE, 'tiny_pair.npz'), common.TINY_HPARAMS, in_lens, out_lens, res, 101, 11)


def encoder_inputs_case():
    """hparams.encoder_inputs=True with labels + noise (VESUS conditioning): full tensors on the tiny config."""
    hp, G, D = build(common.ENCIN_HPARAMS)
    in_lens, out_lens = [7, 5, 4], [11, 9, 6]
    batch = common.synthetic_batch(hp, in_lens, out_lens, seed=13, vesus=True)
    res = run_pair(hp, G, D, batch, mask_seed=303)
    save_full(os.path.join(HERE, 'tiny_encin_pair.npz'), common.ENCIN_HPARAMS, in_lens, out_lens, res, 303, 13,
              vesus=True)


def gradient_penalty_case():
    """train.gradient_penalty (train.py:32-69) on the tiny conv discriminator: the penalty and the gradient of
    lambda * penalty with respect to every discriminator parameter (double backward through D)."""
    m = types.ModuleType('wandb')
    m.log = lambda *a, **k: None
    sys.modules.setdefault('wandb', m)
    import train as ref_train
    ref_train.get_mask_from_lengths = _cpu_mask_from_lengths
    hp, G, D = build(common.TINY_HPARAMS)
    out_lens = [11, 9, 6]
    batch = common.synthetic_batch(hp, [7, 5, 4], out_lens, seed=14)
    real = batch[2]
    g = torch.Generator().manual_seed(15)
    fake = (torch.randn(real.shape, generator=g) * 2.0 - 5.0)
    lens = torch.tensor(out_lens)
    lam = 10.0
    with Injector(common.MaskStream(404)):
        D.zero_grad()
        gp = ref_train.gradient_penalty(D, real, fake, lens, lens)
        (lam * gp).backward()
    blob = dict(hparams=np.array(common.TINY_HPARAMS), out_lens=np.array(out_lens), mask_seed=np.array(404),
                real=real.numpy(), fake=fake.numpy(), lam=np.array(lam), gp=gp.detach().numpy())
    blob.update(common.pack_dict('d_state', {k: v.clone() for k, v in D.state_dict().items()}))
    # the last bias does not influence d D / d x: autograd leaves its .grad at None, stored as zeros
    blob.update(common.pack_dict('d_grad', {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
                                            for k, p in D.named_parameters()}))
    path = os.path.join(HERE, 'gradient_penalty.npz')
    np.savez_compressed(path, **blob)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB', 'gp =', float(gp))


def full_case(name, hp_string, vesus):
    hp, G, D = build(hp_string, weight_seed=2024)
    in_lens, out_lens = [21, 17], [45, 38]
    batch = common.synthetic_batch(hp, in_lens, out_lens, seed=12, vesus=vesus)
    res = run_pair(hp, G, D, batch, mask_seed=202)
    save_summary(os.path.join(HERE, name), hp_string, in_lens, out_lens, res, 202, 12, 2024, vesus)


def op_fixtures():
    blob = {}
    # L2: guided-attention target (loss_function.py:45-52)
    blob['attn_mask_5_9'] = ref_loss.attention_mask(torch.tensor(5), torch.tensor(9)).numpy()
    blob['attn_mask_7_23'] = ref_loss.attention_mask(torch.tensor(7), torch.tensor(23)).numpy()
    # D1: window fix-up + raw reshape (model.py:523-526), T=47 -> T'=60, W=3
    hp = ref_hparams.HParams(common.FULL_HPARAMS)
    x = torch.arange(2 * 47 * 80, dtype=torch.float32).reshape(2, 47, 80)
    win = hp.discriminator_window
    xin = torch.cat((x[:, :-(x.size(1) % win)], x[:, -win:]), dim=1)
    blob['disc_reshape_in'] = x.numpy()
    blob['disc_reshape_out'] = xin.reshape(xin.size(0), win * 80, -1).numpy()
    # D2: per-item window counts used by adversarial_loss (model.py:532-538): drive the
    # reference's own reduction with a known D output and record the per-item means.
    D = ref_model.Discriminator(ref_hparams.HParams(common.FULL_HPARAMS))
    lens = torch.tensor([40, 41, 59, 60, 20, 21, 3])
    g = torch.Generator().manual_seed(5)
    outs = torch.randn(7, 3, generator=g)
    item = []
    for b in range(7):
        D.forward = lambda inputs, _b=b: outs[_b:_b + 1]
        item.append(float(D.adversarial_loss(torch.zeros(1, 80, 60), lens[b:b + 1])))
    D.forward = lambda inputs: outs
    blob['adv_outs'] = outs.numpy()
    blob['adv_lens'] = lens.numpy()
    blob['adv_item_loss'] = np.array(item)
    blob['adv_batch_loss'] = np.array(float(D.adversarial_loss(torch.zeros(7, 80, 60), lens)))
    # S1: STFT magnitude of the first second of demo.wav (stft.py:77-105)
    from scipy.io import wavfile
    sr, wav = wavfile.read(os.path.join(REF, 'demo.wav'))
    wav = wav.astype(np.float32)
    assert sr == 22050 and np.abs(wav).max() <= 1.0
    audio = torch.from_numpy(wav[:22050])[None]
    stft = ref_stft.STFT(1024, 256, 1024)
    mag, _ = stft.transform(audio)
    blob['stft_audio'] = audio.numpy()
    blob['stft_mag'] = mag.numpy()
    blob['stft_basis_probe_idx'] = np.array([[0, 0], [1, 5], [512, 1023], [513, 1], [700, 300], [1025, 1023]])
    fb = stft.forward_basis[:, 0, :]
    blob['stft_basis_probe_val'] = np.array([fb[i, j].item() for i, j in blob['stft_basis_probe_idx']])
    np.savez_compressed(os.path.join(HERE, 'ops.npz'), **blob)
    print('wrote ops.npz')


def _allreduce_worker(rank, world, port, q):
    import torch.distributed as dist
    _install_shims()
    import distributed as ref_dist
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', world_size=world, rank=rank)
    torch.manual_seed(100 + rank)
    lin = torch.nn.Linear(4, 3)
    lin = ref_dist.apply_gradient_allreduce(lin)
    w_after_bcast = lin.weight.detach().clone()
    x = torch.full((2, 4), float(rank + 1))
    lin(x).sum().backward()
    q.put((rank, w_after_bcast.numpy(), lin.weight.grad.numpy().copy(), lin.bias.grad.numpy().copy()))
    dist.destroy_process_group()


def allreduce_fixture():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_allreduce_worker, args=(r, 2, 29571, q)) for r in range(2)]
    [p.start() for p in procs]
    got = sorted([q.get(timeout=120) for _ in procs])
    [p.join() for p in procs]
    blob = {}
    for rank, w, gw, gb in got:
        blob[f'w_after_bcast_{rank}'] = w
        blob[f'grad_w_{rank}'] = gw
        blob[f'grad_b_{rank}'] = gb
    np.savez_compressed(os.path.join(HERE, 'allreduce.npz'), **blob)
    print('wrote allreduce.npz')


def inference_case():
    """Tacotron2.inference (model.py:675-723) on the tiny configuration, B=1: eval mode, explicit style
    vector, Prenet dropout masks recorded in call order.  The gate bias is shifted so that the stop
    condition fires inside the loop (random weights alone never cross the threshold)."""
    hp_string = common.TINY_HPARAMS + ',max_decoder_steps=40,gate_threshold=0.5'
    hp, G, D = build(hp_string)
    G.eval()
    g = torch.Generator().manual_seed(77)
    for k, v in G.state_dict().items():                 # non-trivial running statistics
        if k.endswith('running_mean'):
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
        if k.endswith('running_var'):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
    text = torch.randint(1, hp.n_symbols, (1, 9), generator=g)
    style = torch.rand(1, 1, hp.noise_size, generator=g).repeat_interleave(9, dim=1)
    blob = dict(hparams=np.array(hp_string), text=text.numpy(), style=style.numpy())
    for tag in ('max', 'stop'):
        if tag == 'max':
            G.decoder.gate_layer.linear_layer.weight.data.mul_(40.0)      # a livelier gate trajectory
            G.decoder.gate_layer.linear_layer.bias.data.fill_(-50.0)
        else:
            # the gate does not feed back, so the trajectory of the 'max' run tells where a shifted bias
            # crosses the threshold: stop at the first running maximum after frame 8
            raw = out[2].flatten() + 50.0
            best = None                     # (margin, sign, cut): first crossing in frames 3..34 with the widest margin
            for sign in (1.0, -1.0):
                r = sign * raw
                vals = torch.sort(r).values
                for cut in ((vals[1:] + vals[:-1]) / 2).tolist():
                    first = int(torch.nonzero(r > cut)[0])
                    margin = float((r[:first + 1] - cut).abs().min())
                    if 3 <= first < 35 and (best is None or margin > best[0]):
                        best = (margin, sign, cut)
            margin, sign, cut = best
            print(f'stop placement: margin {margin:.4f}')
            G.decoder.gate_layer.linear_layer.weight.data.mul_(sign)
            G.decoder.gate_layer.linear_layer.bias.data.fill_(-cut)
        inj = Injector(common.MaskStream(303))
        with torch.no_grad(), inj:
            out = G.inference(text, style=style)
        n = out[0].shape[2]
        print(f'inference[{tag}]: {n} frames, last gate {out[2].flatten()[-1].item():.4f}')
        blob.update(common.pack_dict(f'{tag}/state', {k: v.clone() for k, v in G.state_dict().items()}))
        blob.update(common.pack_dict(f'{tag}/out', dict(mel=out[0], mel_post=out[1], gate=out[2], align=out[3])))
        blob[f'{tag}/keep0'] = torch.stack(inj.drops[0::2]).numpy()
        blob[f'{tag}/keep1'] = torch.stack(inj.drops[1::2]).numpy()
    path = os.path.join(HERE, 'tiny_inference.npz')
    np.savez_compressed(path, **blob)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def linear_disc_case():
    """LinearDiscriminator.adversarial_loss (model.py:543-583) on a small configuration: loss, parameter gradients and
    the gradient w.r.t. the input spectrogram.  Window starts come from Python's global `random` (seeded here; the
    HIP module and the oracle replay the same sequence); dropout masks are recorded per item and layer."""
    import random
    hp_string = 'discriminator_type=linear,n_mel_channels=8,discriminator_window=8,discriminator_dim=16'
    hp = ref_hparams.HParams(hp_string)
    torch.manual_seed(4321)
    D = ref_model.LinearDiscriminator(hp).train()
    g = torch.Generator().manual_seed(5)
    lengths = torch.tensor([40, 27, 17])
    mel = torch.randn(3, 8, 40, generator=g)
    for b in range(3):
        mel[b, :, int(lengths[b]):] = 0
    mel.requires_grad_(True)
    random.seed(99)
    inj = Injector(common.MaskStream(404))
    with inj:
        loss = D.adversarial_loss(mel, lengths)
        loss.backward()
    blob = dict(hparams=np.array(hp_string), mel=mel.detach().numpy(), lengths=lengths.numpy(), random_seed=np.array(99),
                loss=loss.detach().numpy(), dmel=mel.grad.numpy())
    blob.update(common.pack_dict('state', {k: v.detach().clone() for k, v in D.state_dict().items()}))
    blob.update(common.pack_dict('grad', {k: p.grad.clone() for k, p in D.named_parameters()}))
    for i in range(3):                                   # per layer: the masks of all items, in item order
        blob[f'keep{i}'] = torch.cat(inj.drops[i::3]).numpy()
    path = os.path.join(HERE, 'linear_disc.npz')
    np.savez_compressed(path, **blob)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB', 'loss', float(loss))


def loader_collate_case():
    """The reference's TextMelLoader (text front-end, seeded index shuffle, load_mel_from_disk, VESUS speakers/labels)
    and TextMelCollate (data_utils.py:13-131) on a small dataset both sides can build (common.write_loader_dataset)."""
    import tempfile
    import data_utils as ref_data
    out = {}
    for tag, hp_extra in (('lj', ''), ('vesus', ',vesus_path=ROOT/,use_intended_labels=True'),
                          ('vesus_multi', ',vesus_path=ROOT/,use_intended_labels=False')):
        with tempfile.TemporaryDirectory() as root:
            lj, ve = common.write_loader_dataset(root)
            hp = ref_hparams.HParams(('use_labels=True,use_noise=True,n_mel_channels=8,load_mel_from_disk=True,seed=77'
                                      + hp_extra).replace('ROOT', root))
            ds = ref_data.TextMelLoader([lj, ve], hp, root + '/')
            items = [ds[i] for i in range(len(ds))]
            batch = ref_data.TextMelCollate(1)(items[:5] if tag == 'lj' else items)
            out[f'{tag}/idx'] = np.asarray(ds.idx)
            out[f'{tag}/n'] = np.asarray(len(ds))
            for i, it in enumerate(items):
                out[f'{tag}/text{i}'] = it[0].numpy()
                if len(it) == 4:
                    out[f'{tag}/speaker{i}'] = np.asarray(int(it[2]))
                    out[f'{tag}/emotions{i}'] = it[3].numpy()
            names = ('text_padded', 'input_lengths', 'mel_padded', 'gate_padded', 'speaker_ids', 'emotions', 'output_lengths')
            for nme, t in zip(names, batch):
                if tag == 'lj' and nme in ('speaker_ids', 'emotions'):
                    continue                     # uninitialised memory in the reference when the items carry no labels
                out[f'{tag}/batch/{nme}'] = t.numpy()
    path = os.path.join(HERE, 'loader_collate.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    which = sys.argv[1:] or ['tiny', 'ops', 'full', 'vesus', 'allreduce', 'inference', 'linear', 'encin', 'gp', 'loader']
    if 'loader' in which:
        loader_collate_case()
    if 'gp' in which:
        gradient_penalty_case()
    if 'encin' in which:
        encoder_inputs_case()
    if 'linear' in which:
        linear_disc_case()
    if 'inference' in which:
        inference_case()
    if 'tiny' in which:
        tiny_case()
    if 'ops' in which:
        op_fixtures()
    if 'full' in which:
        full_case('full_pair_summary.npz', common.FULL_HPARAMS, vesus=False)
    if 'vesus' in which:
        full_case('vesus_pair_summary.npz', common.VESUS_HPARAMS, vesus=True)
    if 'allreduce' in which:
        allreduce_fixture()
