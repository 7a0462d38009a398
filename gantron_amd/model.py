"""`model.Tacotron2` / `model.Discriminator` with the reference's Python surface over the HIP engine.

Drop-in for reference model.py:586-723 (Tacotron2) and :500-540 (Discriminator):

* same constructor argument (an HParams object), same sub-module tree, therefore the same
  `state_dict()` keys and shapes (`decoder.attention_rnn.weight_ih`,
  `postnet.convolutions.0.0.conv.weight`, `discriminator.0.module.0.weight`, ...) and the same
  initialisers (layers.py:14-16,35-36,52; model.py:603-605);
* same calls: `parse_batch(batch)`, `generator(x) -> [mel, mel_postnet, gate, alignments]`,
  `discriminator.adversarial_loss(mel, lengths)`, `discriminator(x)`, nn.Module protocol.

The sub-modules (nn.Conv1d, nn.LSTMCell, ...) only hold parameters; their forward is never called.
`forward` is ONE autograd node whose forward/backward run the hand-written HIP kernels through
the C ABI (engine.py).  There is no PyTorch/CPU fallback: without the extension the call raises.
"""
from math import sqrt

import torch
from torch import nn

from . import ops
from .engine import DiscriminatorEngine, GeneratorEngine, LinearDiscriminatorEngine
from .layers import ConvNorm, DiscConv1d, DiscDense, LinearNorm
from .utils import to_gpu


class DropoutSource:
    """Counter-based dropout masks / style noise (gt_keep_mask, gt_uniform): (seed, running offset).
    tests reproduce the same stream on the CPU (tests/harness.py:u01)."""

    def __init__(self, seed):
        self.seed, self.offset = int(seed) & 0xFFFFFFFFFFFFFFFF, 0

    def keep(self, shape, p, device):
        out = torch.empty(shape, dtype=torch.uint8, device=device)
        ops.call('gt_keep_mask', self.seed, self.offset, float(p), out.numel(), out)
        self.offset += out.numel()
        return out

    def uniform(self, shape, device):
        out = torch.empty(shape, dtype=torch.float32, device=device)
        ops.call('gt_uniform', self.seed, self.offset, out.numel(), out)
        self.offset += out.numel()
        return out


def _named_tensors(module):
    names, tensors = [], []
    for k, v in module.named_parameters():
        names.append(k)
        tensors.append(v)
    return names, tensors


def _direct(module):
    return getattr(module, '_direct_grads', False) and getattr(module, '_grad_views', None) is not None


def _publish_direct(module):
    """After an in-place accumulation into the flat gradient buffer: every p.grad is its view of that buffer (a gradient
    another autograd node left in a tensor of its own is folded in), then the data-parallel wrapper -- which cannot rely
    on parameter hooks here, autograd never sees these gradients -- is told that a backward pass produced gradients."""
    views = module._grad_views
    for name, p in module.named_parameters():
        v = views[name]
        if p.grad is None:
            p.grad = v
        elif p.grad.data_ptr() != v.data_ptr():
            v.add_(p.grad)
            p.grad = v
    hook = getattr(module, '_after_backward', None)
    if hook is not None:
        hook()


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, text, in_lens, mels, out_lens, rng, speaker_ids, emotions, *params):
        P = dict(zip(module._param_names, (p.detach() for p in params)))
        P.update({k: v for k, v in module.named_buffers()})
        outs, sv = module.engine.forward(P, text, in_lens, mels, out_lens, rng, speaker_ids, emotions,
                                         training=module.training, mask_padding=module.mask_padding)
        ctx.module, ctx.sv, ctx.P = module, sv, P
        ctx.needs = [p.requires_grad for p in params]
        return outs

    @staticmethod
    def backward(ctx, dmel, dpost, dgate, dalign):
        module, P = ctx.module, ctx.P
        if _direct(module) and all(ctx.needs):
            # gradients are accumulated in place in the flat gradient buffer (optim.enable_direct_grads)
            module.engine.backward(P, module._grad_views, ctx.sv, dmel, dpost, dgate, dalign,
                                   on_ready=getattr(module, '_bucket_ready', None))
            _publish_direct(module)
            return (None,) * (8 + len(ctx.needs))
        G = {k: torch.zeros_like(P[k]) for k in module._param_names}
        module.engine.backward(P, G, ctx.sv, dmel, dpost, dgate, dalign)
        grads = tuple(G[k] if need else None for k, need in zip(module._param_names, ctx.needs))
        return (None,) * 8 + grads


class Postnet(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.convolutions = nn.ModuleList()
        n, ch, k = hparams.postnet_n_convolutions, hparams.postnet_embedding_dim, hparams.postnet_kernel_size
        dims = [hparams.n_mel_channels] + [ch] * (n - 1) + [hparams.n_mel_channels]
        for i in range(n):
            self.convolutions.append(nn.Sequential(
                ConvNorm(dims[i], dims[i + 1], kernel_size=k, stride=1, padding=(k - 1) // 2, dilation=1,
                         w_init_gain='tanh' if i < n - 1 else 'linear'),
                nn.BatchNorm1d(dims[i + 1])))


class Encoder(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        dim, k = hparams.encoder_embedding_dim, hparams.encoder_kernel_size
        extra = 0
        if hparams.encoder_inputs:                      # labels / style noise enter here (model.py:162-170)
            extra = (hparams.n_labels if hparams.use_labels else 0) + (hparams.noise_size if hparams.use_noise else 0)
        self.convolutions = nn.ModuleList([
            nn.Sequential(ConvNorm(dim + (extra if i == 0 else 0), dim, kernel_size=k, stride=1,
                                   padding=(k - 1) // 2, dilation=1, w_init_gain='relu'), nn.BatchNorm1d(dim))
            for i in range(hparams.encoder_n_convolutions)])
        self.lstm = nn.LSTM(dim, dim // 2, 1, batch_first=True, bidirectional=True)


class LocationLayer(nn.Module):
    def __init__(self, n_filters, kernel_size, attention_dim):
        super().__init__()
        self.location_conv = ConvNorm(2, n_filters, kernel_size=kernel_size, padding=(kernel_size - 1) // 2,
                                      bias=False, stride=1, dilation=1)
        self.location_dense = LinearNorm(n_filters, attention_dim, bias=False, w_init_gain='tanh')


class Attention(nn.Module):
    def __init__(self, attention_rnn_dim, embedding_dim, attention_dim, n_filters, kernel_size):
        super().__init__()
        self.query_layer = LinearNorm(attention_rnn_dim, attention_dim, bias=False, w_init_gain='tanh')
        self.memory_layer = LinearNorm(embedding_dim, attention_dim, bias=False, w_init_gain='tanh')
        self.v = LinearNorm(attention_dim, 1, bias=False)
        self.location_layer = LocationLayer(n_filters, kernel_size, attention_dim)
        self.score_mask_value = -float('inf')


class Prenet(nn.Module):
    def __init__(self, in_dim, sizes):
        super().__init__()
        self.layers = nn.ModuleList([LinearNorm(i, o, bias=False) for i, o in zip([in_dim] + sizes[:-1], sizes)])


class Decoder(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        if hparams.n_frames_per_step != 1:
            raise NotImplementedError('n_frames_per_step must be 1 (as in the reference, hparams.py:71)')
        noise = 0 if hparams.encoder_inputs else hparams.noise_size          # model.py:240-246
        din = hparams.encoder_embedding_dim + noise
        if hparams.vesus_path:
            din += hparams.speakers_embedding + (5 if (hparams.use_labels and not hparams.encoder_inputs) else 0)
        self.decoder_in_dim = din
        self.prenet = Prenet(hparams.n_mel_channels, [hparams.prenet_dim, hparams.prenet_dim])
        self.attention_rnn = nn.LSTMCell(hparams.prenet_dim + din, hparams.attention_rnn_dim)
        self.attention_layer = Attention(hparams.attention_rnn_dim, din, hparams.attention_dim,
                                         hparams.attention_location_n_filters,
                                         hparams.attention_location_kernel_size)
        self.decoder_rnn = nn.LSTMCell(hparams.attention_rnn_dim + din, hparams.decoder_rnn_dim, 1)
        self.linear_projection = LinearNorm(hparams.decoder_rnn_dim + din, hparams.n_mel_channels)
        self.gate_layer = LinearNorm(hparams.decoder_rnn_dim + din, 1, bias=True, w_init_gain='sigmoid')


class Tacotron2(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        if not hparams.use_noise:                       # same side effects as reference model.py:589-591
            hparams.noise_size = 0
        hparams.use_labels = hparams.use_labels if hparams.vesus_path else False
        self.hparams = hparams
        self.mask_padding = hparams.mask_padding
        self.n_mel_channels = hparams.n_mel_channels
        self.embedding = nn.Embedding(hparams.n_symbols, hparams.symbols_embedding_dim)
        val = sqrt(3.0) * sqrt(2.0 / (hparams.n_symbols + hparams.symbols_embedding_dim))
        self.embedding.weight.data.uniform_(-val, val)
        self.vesus = bool(hparams.vesus_path)
        self.use_labels = hparams.use_labels
        if self.vesus:
            self.speaker_embedding = nn.Embedding(123, hparams.speakers_embedding)
            self.speaker_embedding.weight.data.uniform_(-val, val)
        self.encoder = Encoder(hparams)
        self.decoder = Decoder(hparams)
        self.postnet = Postnet(hparams)
        self.engine = GeneratorEngine(hparams)
        self._param_names = [k for k, _ in self.named_parameters()]
        self.dropout_source = DropoutSource(hparams.seed)
        self.rng_override = None          # tests inject explicit masks / noise here

    def parse_batch(self, batch):
        """reference model.py:617-630.  Lengths additionally stay on the host (python lists) so the
        engine never synchronises on them (the reference calls .item()/.cpu() instead)."""
        text_padded, input_lengths, mel_padded, gate_padded, speaker_ids, emotions, output_lengths = batch
        host_in, host_out = input_lengths.tolist(), output_lengths.tolist()
        max_len = int(max(host_in))
        text_padded = to_gpu(text_padded).long()
        input_lengths = to_gpu(input_lengths).long()
        speaker_ids = to_gpu(speaker_ids).long()
        mel_padded = to_gpu(mel_padded).float()
        gate_padded = to_gpu(gate_padded).float()
        emotions = to_gpu(emotions).float()
        output_lengths = to_gpu(output_lengths).long()
        # the host copies travel WITH the device tensors they describe (forward() reads them back from the very
        # tensor objects it is handed, and falls back to .tolist() for tensors that did not come from here)
        input_lengths._gt_host, output_lengths._gt_host = host_in, host_out
        return ((text_padded, input_lengths, mel_padded, max_len, speaker_ids, emotions, output_lengths),
                (mel_padded, gate_padded))

    def draw_rng(self, B, Ti, To, device):
        """Dropout masks and style noise for one forward, in kernel layout (engine.py docstring)."""
        hp, src = self.hparams, self.dropout_source
        Bp = ops.roundup(B, 16) if B <= 64 else ops.roundup(B, 64)
        rng = {'prenet_keep': [src.keep((To, Bp, hp.prenet_dim), 0.5, device) for _ in range(2)]}
        if self.training:
            dim = hp.encoder_embedding_dim
            rng['enc_keep'] = [src.keep((B, Ti, dim), 0.5, device) for _ in range(hp.encoder_n_convolutions)]
            rng['att_keep'] = src.keep((To, Bp, hp.attention_rnn_dim), hp.p_attention_dropout, device)
            rng['dec_keep'] = src.keep((To, Bp, hp.decoder_rnn_dim), hp.p_decoder_dropout, device)
            n = hp.postnet_n_convolutions
            chans = [hp.postnet_embedding_dim] * (n - 1) + [hp.n_mel_channels]
            rng['post_keep'] = [src.keep((B, To, c), 0.5, device) for c in chans]
        if hp.noise_size > 0:
            rng['noise'] = src.uniform((B, hp.noise_size), device)
        return rng

    def forward(self, inputs):
        text, text_lengths, mels, max_len, speaker_ids, emotions, output_lengths = inputs
        # lengths as python lists: the copies parse_batch attached to these very tensors, else one device read
        in_lens = getattr(text_lengths, '_gt_host', None)
        out_lens = getattr(output_lengths, '_gt_host', None)
        if in_lens is None or len(in_lens) != text.size(0):
            in_lens = text_lengths.tolist()
        if out_lens is None or len(out_lens) != text.size(0):
            out_lens = output_lengths.tolist()
        rng = self.rng_override if self.rng_override is not None else \
            self.draw_rng(text.size(0), text.size(1), mels.size(2), text.device)
        params = [p for _, p in self.named_parameters()]
        outs = _GeneratorFn.apply(self, text.contiguous(), in_lens, mels.contiguous(), out_lens, rng,
                                  speaker_ids if self.vesus else None,
                                  emotions.contiguous() if self.use_labels else None, *params)
        return list(outs)

    @torch.no_grad()
    def inference(self, text_inputs, style=None, emotions=None, speaker=None):
        """reference model.py:675-723: text (1,Ti) ids or a raw string; style (B,Ti,noise) / (B,1,noise) /
        (B,noise) or None (drawn U[0,1) as :469 does) -- with hparams.encoder_inputs the reference's layout is
        (B,noise,Ti) / (B,noise,1) (:184-191); emotions (B,5); speaker LongTensor (B,).
        Returns [mel, mel_postnet, gate (B,T,1), alignments] cut at the stop gate."""
        hp = self.hparams
        dev = self.embedding.weight.device
        if isinstance(text_inputs, str):
            import numpy as np
            from .text import text_to_sequence
            text_inputs = torch.from_numpy(np.array(text_to_sequence(text_inputs, ['english_cleaners']))[None, :])
        text = text_inputs.to(dev).long().contiguous()
        B, Ti = text.shape
        src = self.dropout_source
        if hp.noise_size > 0:
            if style is None:
                style = src.uniform((B, hp.noise_size), dev)
            style = style.to(dev).float()
            if hp.encoder_inputs and style.dim() == 3:  # Encoder.append_noise concatenates on channels: (B, noise, Ti|1)
                style = style.transpose(1, 2)
            if style.dim() == 3 and style.size(1) == 1:
                style = style[:, 0]
            style = style.contiguous()
        else:
            style = None
        if self.use_labels and emotions is None:
            emotions = src.uniform((B, 5), dev)
        if self.vesus and speaker is None:
            raise ValueError('this model was built with vesus_path: inference needs `speaker` (model.py:698-702)')
        Bp = ops.roundup(B, 16) if B <= 64 else ops.roundup(B, 64)
        steps = hp.max_decoder_steps
        if self.rng_override is not None:
            keeps = self.rng_override['prenet_keep']
        else:
            keeps = [src.keep((steps, Bp, hp.prenet_dim), 0.5, dev) for _ in range(2)]
        P = {k: v.detach() for k, v in self.named_parameters()}
        P.update({k: v for k, v in self.named_buffers()})
        outs = self.engine.inference(P, text, style, keeps,
                                     speaker.to(dev).long().contiguous() if self.vesus else None,
                                     emotions.to(dev).float().contiguous() if self.use_labels else None,
                                     gate_threshold=hp.gate_threshold, max_decoder_steps=steps)
        return list(outs)


class _AdversarialFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, mel, lengths, keeps, *params):
        P = dict(zip(module._param_names, (p.detach() for p in params)))
        loss, sv = module.engine.forward(P, mel.detach(), lengths, keeps, training=module.training)
        ctx.module, ctx.sv, ctx.P = module, sv, P
        ctx.need_in = mel.requires_grad
        ctx.needs = [p.requires_grad for p in params]
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        module, P = ctx.module, ctx.P
        need_p = any(ctx.needs)
        direct = need_p and all(ctx.needs) and _direct(module)
        G = module._grad_views if direct else ({k: torch.zeros_like(P[k]) for k in module._param_names} if need_p else None)
        dmel = module.engine.backward(P, G, ctx.sv, dloss.contiguous().view(1), 1.0, need_dinput=ctx.need_in,
                                      need_dparams=need_p)
        if direct:
            _publish_direct(module)
            return (None, dmel, None, None) + (None,) * len(ctx.needs)
        grads = tuple(G[k] if need else None for k, need in zip(module._param_names, ctx.needs)) if need_p \
            else (None,) * len(ctx.needs)
        return (None, dmel, None, None) + grads


class _GradientPenaltyFn(torch.autograd.Function):
    """The penalty and its parameter gradient are computed together (the gradient IS a second backward pass through
    the discriminator, engine.DiscriminatorEngine.gradient_penalty); backward only scales it by the upstream."""

    @staticmethod
    def forward(ctx, module, real, fake, lengths, alpha, keeps, *params):
        P = dict(zip(module._param_names, (p.detach() for p in params)))
        G = {k: torch.zeros_like(P[k]) for k in module._param_names}
        gp = module.engine.gradient_penalty(P, G, real.detach(), fake.detach(), lengths, alpha, keeps,
                                            training=module.training)
        ctx.G, ctx.names = G, module._param_names
        ctx.needs = [p.requires_grad for p in params]
        return gp.view(())

    @staticmethod
    def backward(ctx, dgp):
        d = dgp.contiguous().view(1)
        grads = []
        for k, need in zip(ctx.names, ctx.needs):
            if need:
                ops.call('gt_scale_dev', ctx.G[k], d, 1.0, ctx.G[k], ctx.G[k].numel(), 0)
            grads.append(ctx.G[k] if need else None)
        return (None,) * 6 + tuple(grads)


class _DiscForwardFn(torch.autograd.Function):
    """Discriminator.forward as one autograd node: mel (B, n_mel, T) -> (B, W) window outputs."""

    @staticmethod
    def forward(ctx, module, mel, keeps, *params):
        P = dict(zip(module._param_names, (p.detach() for p in params)))
        lengths = torch.full((mel.size(0),), mel.size(2), dtype=torch.int64, device=mel.device)
        _, sv = module.engine.forward(P, mel.detach(), lengths, keeps, training=module.training)
        ctx.module, ctx.sv, ctx.P = module, sv, P
        ctx.need_in = mel.requires_grad
        ctx.needs = [p.requires_grad for p in params]
        return sv['out'].view(mel.size(0), -1).clone()

    @staticmethod
    def backward(ctx, dout):
        module, P = ctx.module, ctx.P
        need_p = any(ctx.needs)
        G = {k: torch.zeros_like(P[k]) for k in module._param_names} if need_p else None
        dmel = module.engine.backward(P, G, ctx.sv, None, 1.0, need_dinput=ctx.need_in, need_dparams=need_p,
                                      dout=dout.contiguous().view(-1, 1))
        grads = tuple(G[k] if need else None for k, need in zip(module._param_names, ctx.needs)) if need_p \
            else (None,) * len(ctx.needs)
        return (None, dmel, None) + grads


class Discriminator(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.window = hparams.discriminator_window
        self.in_dim = self.window * hparams.n_mel_channels
        dd = hparams.discriminator_dim
        first_hidden = min((int(self.in_dim / dd) + 1) * dd, 1024)
        self.discriminator = nn.Sequential(
            DiscConv1d(self.in_dim, first_hidden, 5),
            DiscConv1d(first_hidden, dd, 5, dilation=2),
            DiscConv1d(dd, dd, 5, dilation=2),
            DiscConv1d(dd, hparams.n_mel_channels, 5, dilation=2),
            nn.Conv1d(hparams.n_mel_channels, 1, 1, bias=True))
        self.engine = DiscriminatorEngine(hparams)
        self._param_names = [k for k, _ in self.named_parameters()]
        self.dropout_source = DropoutSource(hparams.seed + 1)
        self.rng_override = None

    def _keeps(self, B, T, device):
        if self.rng_override is not None:
            return self.rng_override
        if not self.training:
            return None
        W = T // self.window + (1 if T % self.window else 0)
        chans = [self.discriminator[i].module[0].out_channels for i in range(4)]
        return [self.dropout_source.keep((B, W, c), 0.5, device) for c in chans]

    def adversarial_loss(self, inputs, target_length):
        """inputs (B, n_mel, T); reference model.py:529-540."""
        params = [p for _, p in self.named_parameters()]
        keeps = self._keeps(inputs.size(0), inputs.size(2), inputs.device)
        return _AdversarialFn.apply(self, inputs.contiguous(), target_length.contiguous().long(), keeps, *params)

    def gradient_penalty(self, real_data, generated_data, real_lengths, generated_lengths):
        """train.gradient_penalty (reference train.py:32-69) for this discriminator: real/generated (B, n_mel, T);
        returns a 0-dim tensor whose backward adds d penalty / d theta to the parameters' .grad.
        rng_override (tests): dict(alpha=(B,) tensor, keeps=[4 keep masks])."""
        B = real_data.size(0)
        if real_data.size(2) < generated_data.size(2):                      # train.py:37-44
            T, lengths = real_data.size(2), real_lengths
        else:
            T, lengths = generated_data.size(2), generated_lengths
        real = real_data[:, :, :T].contiguous().float()
        fake = generated_data[:, :, :T].contiguous().float()
        dev = real.device
        if isinstance(self.rng_override, dict):
            alpha, keeps = self.rng_override['alpha'], self.rng_override['keeps']
        else:
            alpha = self.dropout_source.uniform((B,), dev)
            keeps = self._keeps(B, T, dev)
        params = [p for _, p in self.named_parameters()]
        return _GradientPenaltyFn.apply(self, real, fake, lengths.to(dev).contiguous().long(), alpha.contiguous(),
                                        keeps, *params)

    def forward(self, inputs):
        """inputs (B, T, n_mel) -> (B, W) per-window outputs; reference model.py:514-527.  Differentiable with respect
        to the input and the parameters (train.gradient_penalty, train.py:52-58, differentiates exactly this call)."""
        params = [p for _, p in self.named_parameters()]
        mel = inputs.transpose(1, 2).contiguous()
        keeps = self._keeps(mel.size(0), mel.size(2), mel.device)
        return _DiscForwardFn.apply(self, mel, keeps, *params)


class _LinearAdvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, mel, starts, keeps, *params):
        P = dict(zip(module._param_names, (p.detach() for p in params)))
        loss, sv = module.engine.forward(P, mel.detach(), starts, keeps, training=module.training)
        ctx.module, ctx.sv, ctx.P = module, sv, P
        ctx.need_in = mel.requires_grad
        ctx.needs = [p.requires_grad for p in params]
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        module, P = ctx.module, ctx.P
        need_p = any(ctx.needs)
        direct = need_p and all(ctx.needs) and _direct(module)
        G = module._grad_views if direct else ({k: torch.zeros_like(P[k]) for k in module._param_names} if need_p else None)
        dmel = module.engine.backward(P, G, ctx.sv, dloss.contiguous().view(1), 1.0, need_dinput=ctx.need_in,
                                      need_dparams=need_p)
        if direct:
            _publish_direct(module)
            return (None, dmel, None, None) + (None,) * len(ctx.needs)
        grads = tuple(G[k] if need else None for k, need in zip(module._param_names, ctx.needs)) if need_p \
            else (None,) * len(ctx.needs)
        return (None, dmel, None, None) + grads


class _LinearFwdFn(torch.autograd.Function):
    """LinearDiscriminator.forward as one autograd node: windows (NW, window*n_mel) -> (NW, 1)."""

    @staticmethod
    def forward(ctx, module, x, keeps, *params):
        from . import ops as _ops
        P = dict(zip(module._param_names, (p.detach() for p in params)))
        with _ops.bf16_gemms(module.engine.bf16):
            out, sv = module.engine.layers_forward(P, x.detach(), keeps, module.training)
        ctx.module, ctx.sv, ctx.P = module, sv, P
        ctx.need_in = x.requires_grad
        ctx.needs = [p.requires_grad for p in params]
        return out.clone()

    @staticmethod
    def backward(ctx, dout):
        from . import ops as _ops
        module, P = ctx.module, ctx.P
        need_p = any(ctx.needs)
        G = {k: torch.zeros_like(P[k]) for k in module._param_names} if need_p else None
        with _ops.bf16_gemms(module.engine.bf16):
            dx = module.engine.layers_backward(P, G, ctx.sv, dout.contiguous(), need_dinput=ctx.need_in, need_dparams=need_p)
        grads = tuple(G[k] if need else None for k, need in zip(module._param_names, ctx.needs)) if need_p \
            else (None,) * len(ctx.needs)
        return (None, dx, None) + grads


class LinearDiscriminator(nn.Module):
    """reference model.py:543-583 (hparams.discriminator_type == 'linear'): windows of `discriminator_window` frames
    with random overlap (Python's global `random`, drawn in the reference's order) through three DiscDense layers and
    a Linear(1); the adversarial loss is the batch mean of the per-item window means."""

    def __init__(self, hparams):
        super().__init__()
        dd = hparams.discriminator_dim
        self.discriminator = nn.Sequential(
            DiscDense(hparams.discriminator_window * hparams.n_mel_channels, dd), DiscDense(dd, dd), DiscDense(dd, dd),
            nn.Linear(dd, 1))
        self.n_mel_channels = hparams.n_mel_channels
        self.windows_size = hparams.discriminator_window
        self.max_window_overlap = 6
        self.engine = LinearDiscriminatorEngine(hparams)
        self._param_names = [k for k, _ in self.named_parameters()]
        self.dropout_source = DropoutSource(hparams.seed + 1)
        self.rng_override = None

    def window_starts(self, lengths):
        """Window start frames per item, consuming random.randint exactly as model.py:571-574 does."""
        import random
        starts = []
        for n in lengths:
            st, s0 = [], 0
            while s0 + self.windows_size < n:
                st.append(s0)
                s0 += self.windows_size - random.randint(0, self.max_window_overlap)
            st.append(n - self.windows_size)
            starts.append(st)
        return starts

    def adversarial_loss(self, inputs, target_length):
        """inputs (B, n_mel, T); target_length (B,) (host or device)."""
        lengths = [int(v) for v in target_length.tolist()]
        assert min(lengths) >= self.windows_size, 'every item needs at least one full discriminator window'
        starts = self.window_starts(lengths)
        keeps = self.rng_override
        if keeps is None and self.training:
            nw = sum(len(s) for s in starts)
            dims = [self.discriminator[i].module[0].out_features for i in range(3)]
            keeps = [self.dropout_source.keep((nw, c), 0.5, inputs.device) for c in dims]
        params = [p for _, p in self.named_parameters()]
        return _LinearAdvFn.apply(self, inputs.contiguous(), starts, keeps, *params)

    def forward(self, inputs):
        """inputs (NW, window * n_mel) already-cut windows (frame-major) -> (NW, 1); reference model.py:556-565.
        Differentiable with respect to the windows and the parameters."""
        keeps = self.rng_override
        if keeps is None and self.training:
            dims = [self.discriminator[i].module[0].out_features for i in range(3)]
            keeps = [self.dropout_source.keep((inputs.size(0), c), 0.5, inputs.device) for c in dims]
        params = [p for _, p in self.named_parameters()]
        return _LinearFwdFn.apply(self, inputs.contiguous().float(), keeps, *params)
