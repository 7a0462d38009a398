"""CPU oracle for the GANtron generator+discriminator training step.

TEST INFRASTRUCTURE ONLY.  Nothing in `gantron_amd/` may import this file: only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` use it,
and only as the checker (or as the timed CPU baseline), never as the product path.

It restates, in plain fp32 PyTorch-CPU ops, the algorithm of the reference
(`/root/reference`, cited per function as file:line).  All stochastic inputs of the
reference (dropout masks, the style noise drawn on the host generator) are explicit
arguments, so the same inputs can be fed to the HIP path and to the reference itself.

Pinned by `tests/golden/*.npz`: vectors produced by importing the reference's own
classes in the build container (`tests/golden/make_golden.py`), see
`tests/test_oracle_golden.py`.  Third-party arithmetic that is absent from
/root/reference (librosa 0.7.2 mel filterbank, resampy, unidecode/inflect text
normalisation) is "parity unpinned"; see DESIGN.md.

Parameters are passed as a dict keyed by the reference's state_dict names
(SURVEY.md §8b), in PyTorch layouts.
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------- helpers
def keep_scale(x: Tensor, keep: Optional[Tensor], p: float) -> Tensor:
    """Inverted dropout with an explicit keep mask: x * keep / (1-p).

    The reference calls F.dropout (model.py:104,148-149,197,393-394,408-409,
    layers.py:55); torch draws `keep ~ Bernoulli(1-p)` and multiplies by keep/(1-p).
    """
    if keep is None or p == 0.0:
        return x
    return x * (keep.to(x.dtype) / (1.0 - p))


def length_mask(lengths: Tensor, max_len: Optional[int] = None) -> Tensor:
    """True where position < length.  utils.py:20-24."""
    if max_len is None:
        max_len = int(lengths.max())
    ids = torch.arange(max_len, device=lengths.device)
    return ids[None, :] < lengths[:, None]


def batch_norm_train(x: Tensor, P: Dict[str, Tensor], prefix: str, training: bool,
                     new_stats: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """nn.BatchNorm1d over (B, C, T): batch statistics over B*T *including padding*
    in training mode (model.py:123,176); running stats updated with momentum 0.1 and
    the unbiased variance."""
    w, b = P[prefix + '.weight'], P[prefix + '.bias']
    rm, rv = P[prefix + '.running_mean'], P[prefix + '.running_var']
    if not training:
        return (x - rm[None, :, None]) / torch.sqrt(rv[None, :, None] + BN_EPS) * w[None, :, None] \
            + b[None, :, None]
    mean = x.mean(dim=(0, 2))
    var = x.var(dim=(0, 2), unbiased=False)
    if new_stats is not None:
        n = x.size(0) * x.size(2)
        new_stats[prefix + '.running_mean'] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach()
        new_stats[prefix + '.running_var'] = (1 - BN_MOMENTUM) * rv + \
            BN_MOMENTUM * var.detach() * (n / max(n - 1, 1))
    xhat = (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + BN_EPS)
    return xhat * w[None, :, None] + b[None, :, None]


def _bf16(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


# ---- hparams.fp16_run on the HIP path (BASELINE config 3): bf16 OPERANDS, fp32 accumulation, in
#   * the layer GEMMs (Linear / Conv1d forward, input gradient and weight gradient: gt_sgemm with GtGemm.bf16),
#   * the two decoder LSTM products of every step (gt_rec_gemm_bf16) and their input gradients;
# products that live inside fp32 kernels (BiLSTM recurrence, attention query / location / v, context) stay fp32, but
# their weight gradients, being layer GEMMs over saved activations, take rounded operands too.  The oracle states
# exactly that arithmetic, so the *_bf16 parity tests stay operand-for-operand checks.
_FP16_RUN = False


class fp16_run:
    """with fp16_run(flag): ... -- products below follow the HIP path's hparams.fp16_run arithmetic."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        global _FP16_RUN
        self.old, _FP16_RUN = _FP16_RUN, self.on

    def __exit__(self, *exc):
        global _FP16_RUN
        _FP16_RUN = self.old


class _Prod(torch.autograd.Function):
    """y = x . W^T with per-product operand rounding: r_fwd (forward), r_dx (dx = dy . W), r_dw (dW = dy^T . x)."""

    @staticmethod
    def forward(ctx, x, W, r_fwd, r_dx, r_dw):
        ctx.save_for_backward(x, W)
        ctx.flags = (r_dx, r_dw)
        return (_bf16(x) @ _bf16(W).t()) if r_fwd else (x @ W.t())

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        r_dx, r_dw = ctx.flags
        dy2, x2 = dy.reshape(-1, dy.size(-1)), x.reshape(-1, x.size(-1))
        dx = (_bf16(dy2) @ _bf16(W)) if r_dx else (dy2 @ W)
        dW = (_bf16(dy2).t() @ _bf16(x2)) if r_dw else (dy2.t() @ x2)
        return dx.reshape(x.shape), dW, None, None, None


def mm(x: Tensor, W: Tensor, kind: str = 'gemm') -> Tensor:
    """x . W^T.  kind 'gemm': a layer GEMM (all three products rounded under fp16_run); 'kernel32': forward and input
    gradient inside an fp32 kernel, weight gradient a layer GEMM; 'rec': a decoder LSTM product (gt_rec_gemm_bf16)."""
    if not _FP16_RUN:
        return x @ W.t()
    if kind == 'kernel32':
        return _Prod.apply(x, W, False, False, True)
    return _Prod.apply(x, W, True, True, True)


class _Conv(torch.autograd.Function):
    """Conv1d whose three implicit GEMMs (forward, input gradient, weight gradient) take bf16-rounded operands."""

    @staticmethod
    def forward(ctx, x, w, padding, dilation):
        ctx.save_for_backward(x, w)
        ctx.cfg = (padding, dilation)
        return F.conv1d(_bf16(x), _bf16(w), None, padding=padding, dilation=dilation)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        padding, dilation = ctx.cfg
        dyr = _bf16(dy)
        dx = torch.nn.grad.conv1d_input(x.shape, _bf16(w), dyr, padding=padding, dilation=dilation)
        dw = torch.nn.grad.conv1d_weight(_bf16(x), w.shape, dyr, padding=padding, dilation=dilation)
        return dx, dw, None, None


def conv1d(x: Tensor, w: Tensor, bias: Optional[Tensor], padding: int = 0, dilation: int = 1) -> Tensor:
    if not _FP16_RUN:
        return F.conv1d(x, w, bias, padding=padding, dilation=dilation)
    y = _Conv.apply(x, w, padding, dilation)
    return y if bias is None else y + bias[None, :, None]


def lstm_cell(x: Tensor, h: Tensor, c: Tensor, w_ih: Tensor, w_hh: Tensor,
              b_ih: Tensor, b_hh: Tensor, bf16_from: Optional[int] = None, head_fp32: bool = False):
    """nn.LSTMCell, gate order i,f,g,o (model.py:252-254,261-263).  Default (encoder BiLSTM): the input product is a
    layer GEMM, the recurrent one runs inside gt_rec_gemm (fp32).  bf16_from = n (decoder cells under fp16_run): input
    columns < n (the Prenet part, a batched layer GEMM) and columns >= n together with the recurrent product (one
    gt_rec_gemm_bf16 product) all take bf16 operands."""
    if bf16_from is None:
        gates = mm(x, w_ih) + b_ih + mm(h, w_hh, 'kernel32') + b_hh
    else:
        n = bf16_from
        gates = b_ih + b_hh + mm(torch.cat((x[:, n:], h), dim=1), torch.cat((w_ih[:, n:], w_hh), dim=1), 'rec')
        if n > 0:       # head_fp32: the free-running decoder's per-step Prenet product (fp32 in its C++ loop)
            gates = gates + (x[:, :n] @ w_ih[:, :n].t() if head_fp32 else mm(x[:, :n], w_ih[:, :n]))
    i, f, g, o = gates.chunk(4, dim=1)
    c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h_new = torch.sigmoid(o) * torch.tanh(c_new)
    return h_new, c_new


# ----------------------------------------------------------------------------- encoder
def encoder_forward(P, embedded: Tensor, lengths: Tensor, enc_keep: Optional[List[Tensor]],
                    training: bool = True, new_stats=None, relu_pos: Optional[List[Tensor]] = None,
                    relu_report: Optional[list] = None) -> Tensor:
    """Encoder.forward, model.py:193-212.  embedded: (B, 512, Ti).
    3x[conv5 -> BN -> ReLU -> dropout .5] then a length-packed BiLSTM whose outputs
    are zero beyond each length.

    relu_pos (parity tests at >= 1024 encoder rows): per block a bool (B, C, Ti) tensor saying which KEPT elements the
    implementation under test let through its ReLU.  Of B*Ti*512 pre-activations a few lie within fp32 rounding of zero,
    and on which side such an element falls depends on the summation order of the convolution before it; one flipped
    element moves a whole channel's slice of the weight gradients below it.  With relu_pos the restatement takes the
    other side's DECISION (x * pos instead of relu(x)) wherever its own pre-activation is that close to zero, so that both
    differentiate the same piecewise-linear function; every disagreement is appended to relu_report as (block, count,
    largest |pre-activation| among the disagreeing elements, scale) for the caller to bound -- a decision that differs
    anywhere else is a real error and shows there (and in the outputs)."""
    x = embedded
    n_conv = len([k for k in P if k.startswith('encoder.convolutions.') and k.endswith('.0.conv.weight')])
    for i in range(n_conv):
        w = P[f'encoder.convolutions.{i}.0.conv.weight']
        x = conv1d(x, w, P[f'encoder.convolutions.{i}.0.conv.bias'], padding=(w.size(2) - 1) // 2)
        x = batch_norm_train(x, P, f'encoder.convolutions.{i}.1', training, new_stats)
        keep = enc_keep[i] if (enc_keep is not None and training) else None
        if relu_pos is not None:
            kept = torch.ones_like(x, dtype=torch.bool) if keep is None else keep.to(torch.bool)
            differs = ((x.detach() > 0) != relu_pos[i]) & kept
            if relu_report is not None:
                n = int(differs.sum())
                relu_report.append((i, n, float(x.detach().abs()[differs].max()) if n else 0.0, float(x.detach().abs().max())))
            x = keep_scale(x * (relu_pos[i] & kept).to(x.dtype), keep, 0.5)
            continue
        x = keep_scale(F.relu(x), keep, 0.5)
    x = x.transpose(1, 2)  # (B, Ti, C)
    return bilstm_packed(P, x, lengths)


def bilstm_packed(P, x: Tensor, lengths: Tensor) -> Tensor:
    """pack_padded_sequence -> nn.LSTM(bidirectional) -> pad_packed_sequence
    (model.py:202-210) restated as masked recurrences: item b runs t = 0..len_b-1
    forward and t = len_b-1..0 backward, both from zero state; outputs beyond len_b are 0."""
    B, T, _ = x.shape
    T = int(lengths.max())
    H = P['encoder.lstm.weight_hh_l0'].size(1)
    act = length_mask(lengths, T).to(x.dtype)  # (B, T)
    outs = []
    for sfx, order in (('', range(T)), ('_reverse', range(T - 1, -1, -1))):
        w_ih, w_hh = P['encoder.lstm.weight_ih_l0' + sfx], P['encoder.lstm.weight_hh_l0' + sfx]
        b_ih, b_hh = P['encoder.lstm.bias_ih_l0' + sfx], P['encoder.lstm.bias_hh_l0' + sfx]
        h = x.new_zeros(B, H)
        c = x.new_zeros(B, H)
        out = [None] * T
        for t in order:
            hn, cn = lstm_cell(x[:, t], h, c, w_ih, w_hh, b_ih, b_hh)
            a = act[:, t:t + 1]
            h = a * hn + (1 - a) * h
            c = a * cn + (1 - a) * c
            out[t] = a * hn
        outs.append(torch.stack(out, dim=1))
    return torch.cat(outs, dim=2)  # (B, T, 2H)


# ----------------------------------------------------------------------------- decoder
def prenet_forward(P, x: Tensor, prenet_keep: List[Tensor], fp32: bool = False) -> Tensor:
    """Prenet.forward, model.py:102-105: dropout p=.5 is ALWAYS on.  fp32: the free-running decoder computes the
    Prenet of one frame per step inside its C++ loop with fp32 products whatever hparams.fp16_run says."""
    i = 0
    while f'decoder.prenet.layers.{i}.linear_layer.weight' in P:
        W = P[f'decoder.prenet.layers.{i}.linear_layer.weight']
        x = keep_scale(F.relu(x @ W.t() if fp32 else mm(x, W)), prenet_keep[i], 0.5)
        i += 1
    return x


def attention_step(P, query: Tensor, memory: Tensor, processed_memory: Tensor,
                   w_prev: Tensor, w_cum: Tensor, pad_mask: Optional[Tensor],
                   score_mask_value: float = -float('inf')):
    """Attention.forward + LocationLayer.forward, model.py:27-31,48-91.
    query (B,Ha); memory (B,Ti,E); processed_memory (B,Ti,A); w_prev/w_cum (B,Ti);
    pad_mask True on padded positions."""
    pre = 'decoder.attention_layer.'
    pq = mm(query, P[pre + 'query_layer.linear_layer.weight'], 'kernel32')     # (B, A)
    wconv = P[pre + 'location_layer.location_conv.conv.weight']             # (F, 2, K)
    cat = torch.stack((w_prev, w_cum), dim=1)                                # (B, 2, Ti)
    loc = F.conv1d(cat, wconv, None, padding=(wconv.size(2) - 1) // 2)       # (B, F, Ti)
    loc = loc.transpose(1, 2) @ P[pre + 'location_layer.location_dense.linear_layer.weight'].t()
    e = torch.tanh(pq[:, None, :] + loc + processed_memory) @ P[pre + 'v.linear_layer.weight'].t()
    e = e.squeeze(-1)                                                        # (B, Ti)
    if pad_mask is not None:
        e.data.masked_fill_(pad_mask, score_mask_value)                      # model.py:84-85
    w = F.softmax(e, dim=1)
    ctx = torch.bmm(w.unsqueeze(1), memory).squeeze(1)
    return ctx, w


def decoder_forward(P, memory: Tensor, mels: Tensor, memory_lengths: Tensor,
                    noise: Optional[Tensor], prenet_keep: List[Tensor],
                    att_keep: Optional[Tensor], dec_keep: Optional[Tensor],
                    p_att: float = 0.1, p_dec: float = 0.1, training: bool = True, bf16_products: bool = False):
    """Decoder.forward (teacher forcing), model.py:419-454 with decode() :378-417.
    memory (B,Ti,Eenc); mels (B,n_mel,To); noise (B,noise_size) U[0,1) style vector
    repeated over Ti (model.py:273-279).  Returns mel (B,n_mel,To), gate (B,To),
    alignments (B,To,Ti) laid out as parse_decoder_outputs (model.py:349-376) does."""
    B, Ti, _ = memory.shape
    if noise is not None:
        memory = torch.cat([memory, noise[:, None, :].expand(B, Ti, noise.size(1))], dim=-1)
    n_mel = mels.size(1)
    dec_in = mels.transpose(1, 2).transpose(0, 1)                            # (To, B, n_mel)
    dec_in = torch.cat((mels.new_zeros(1, B, n_mel), dec_in), dim=0)         # go frame
    dec_in = prenet_forward(P, dec_in, prenet_keep)
    To = dec_in.size(0) - 1

    Ha = P['decoder.attention_rnn.weight_hh'].size(1)
    Hd = P['decoder.decoder_rnn.weight_hh'].size(1)
    h_a = memory.new_zeros(B, Ha); c_a = memory.new_zeros(B, Ha)
    h_d = memory.new_zeros(B, Hd); c_d = memory.new_zeros(B, Hd)
    w = memory.new_zeros(B, Ti); w_cum = memory.new_zeros(B, Ti)
    ctx = memory.new_zeros(B, memory.size(2))
    pm = mm(memory, P['decoder.attention_layer.memory_layer.linear_layer.weight'])
    pad = ~length_mask(memory_lengths, Ti)

    mel_out, gate_out, aligns = [], [], []
    for t in range(To):
        h_a, c_a = lstm_cell(torch.cat((dec_in[t], ctx), -1), h_a, c_a,
                             P['decoder.attention_rnn.weight_ih'], P['decoder.attention_rnn.weight_hh'],
                             P['decoder.attention_rnn.bias_ih'], P['decoder.attention_rnn.bias_hh'],
                             bf16_from=dec_in.size(2) if bf16_products else None)
        h_a = keep_scale(h_a, att_keep[t] if (training and att_keep is not None) else None, p_att)
        ctx, w = attention_step(P, h_a, memory, pm, w, w_cum, pad)
        w_cum = w_cum + w
        h_d, c_d = lstm_cell(torch.cat((h_a, ctx), -1), h_d, c_d,
                             P['decoder.decoder_rnn.weight_ih'], P['decoder.decoder_rnn.weight_hh'],
                             P['decoder.decoder_rnn.bias_ih'], P['decoder.decoder_rnn.bias_hh'],
                             bf16_from=0 if bf16_products else None)
        h_d = keep_scale(h_d, dec_keep[t] if (training and dec_keep is not None) else None, p_dec)
        hc = torch.cat((h_d, ctx), dim=1)
        mel_out.append(mm(hc, P['decoder.linear_projection.linear_layer.weight'])
                       + P['decoder.linear_projection.linear_layer.bias'])
        gate_out.append((mm(hc, P['decoder.gate_layer.linear_layer.weight'])
                         + P['decoder.gate_layer.linear_layer.bias']).squeeze(1))
        aligns.append(w)
    aligns = torch.stack(aligns).transpose(0, 1)
    gate_out = torch.stack(gate_out).transpose(0, 1).contiguous()
    mel_out = torch.stack(mel_out).transpose(0, 1).contiguous()
    mel_out = mel_out.view(B, -1, n_mel).transpose(1, 2)
    return mel_out, gate_out, aligns


# ----------------------------------------------------------------------------- postnet
def postnet_forward(P, x: Tensor, post_keep: Optional[List[Tensor]], training: bool = True,
                    new_stats=None) -> Tensor:
    """Postnet.forward, model.py:146-151: tanh on all but the last conv, dropout on all."""
    n = len([k for k in P if k.startswith('postnet.convolutions.') and k.endswith('.0.conv.weight')])
    for i in range(n):
        w = P[f'postnet.convolutions.{i}.0.conv.weight']
        x = conv1d(x, w, P[f'postnet.convolutions.{i}.0.conv.bias'], padding=(w.size(2) - 1) // 2)
        x = batch_norm_train(x, P, f'postnet.convolutions.{i}.1', training, new_stats)
        if i < n - 1:
            x = torch.tanh(x)
        x = keep_scale(x, post_keep[i] if (training and post_keep is not None) else None, 0.5)
    return x


# ----------------------------------------------------------------------------- generator
def generator_forward(P, text: Tensor, text_lengths: Tensor, mels: Tensor, output_lengths: Tensor,
                      rng: Dict[str, object], speaker_ids: Optional[Tensor] = None,
                      emotions: Optional[Tensor] = None, use_labels: bool = False,
                      training: bool = True, mask_padding: bool = True, new_stats=None,
                      p_att: float = 0.1, p_dec: float = 0.1, encoder_inputs: bool = False,
                      bf16_products: bool = False):
    """Tacotron2.forward + parse_output, model.py:632-673.
    rng: dict with 'enc_keep' [3x(B,C,Ti)], 'prenet_keep' [2x(To+1,B,D)], 'att_keep' (To,B,Ha),
    'dec_keep' (To,B,Hd), 'post_keep' [5x(B,C,To)], 'noise' (B,noise_size) or None.
    encoder_inputs (hparams.py:102): labels and style noise are concatenated to the embedded text as extra
    channels of the first encoder convolution (model.py:649-651, 184-191) instead of to the decoder memory."""
    emb = F.embedding(text, P['embedding.weight']).transpose(1, 2)
    dec_noise = rng.get('noise')
    if encoder_inputs:
        if use_labels:
            emb = torch.cat((emb, emotions[:, :, None].repeat(1, 1, emb.size(-1))), dim=1)
        if dec_noise is not None:
            emb = torch.cat((emb, dec_noise[:, :, None].repeat(1, 1, emb.size(-1))), dim=1)
        dec_noise = None
    enc = encoder_forward(P, emb, text_lengths, rng.get('enc_keep'), training, new_stats,
                          relu_pos=rng.get('enc_relu_pos'), relu_report=rng.get('enc_relu_report'))
    if 'speaker_embedding.weight' in P:                                      # model.py:655-663
        spk = F.embedding(speaker_ids, P['speaker_embedding.weight'])[:, None].repeat(1, enc.size(1), 1)
        if use_labels and not encoder_inputs:
            emo = emotions[:, None].repeat(1, enc.size(1), 1)
            enc = torch.cat((enc, spk, emo), dim=2)
        else:
            enc = torch.cat((enc, spk), dim=2)
    mel, gate, align = decoder_forward(P, enc, mels, text_lengths, dec_noise,
                                       rng['prenet_keep'], rng.get('att_keep'), rng.get('dec_keep'),
                                       p_att, p_dec, training, bf16_products)
    post = postnet_forward(P, mel, rng.get('post_keep'), training, new_stats)
    mel_post = mel + post
    if mask_padding and output_lengths is not None:
        # Autograd-invisible in-place fill (model.py:638-640).  NB: `mel` is the tensor the
        # first postnet conv saved for its weight gradient, so that gradient sees masked frames.
        pad = ~length_mask(output_lengths, mel.size(2))
        pad3 = pad[:, None, :].expand(-1, mel.size(1), -1)
        mel.data.masked_fill_(pad3, 0.0)
        mel_post.data.masked_fill_(pad3, 0.0)
        gate.data.masked_fill_(pad, 1e3)
    return mel, mel_post, gate, align


# ----------------------------------------------------------------------------- inference
def decoder_inference(P, memory: Tensor, style: Optional[Tensor], prenet_keep: List[Tensor],
                      gate_threshold: float = 0.5, max_decoder_steps: int = 500, bf16_products: bool = False):
    """Decoder.inference, model.py:456-497: free-running decode.  The previous mel frame goes back
    through the Prenet (dropout still on, model.py:104), no attention mask (mask=None), no LSTM dropout
    (eval), stop as soon as sigmoid(gate) > gate_threshold (the stopping frame is kept) or after
    max_decoder_steps frames.  memory (B,Ti,Eenc); style (B,Ti,noise) or (B,noise) or None;
    prenet_keep [2 x (steps,B,D)].  Returns mel (B,n_mel,T), gate (B,T,1), alignments (B,T,Ti)."""
    B, Ti, _ = memory.shape
    if style is not None:
        if style.dim() == 2:
            style = style[:, None, :].expand(B, Ti, style.size(1))
        memory = torch.cat([memory, style], dim=-1)
    n_mel = P['decoder.linear_projection.linear_layer.weight'].size(0)
    Ha = P['decoder.attention_rnn.weight_hh'].size(1)
    Hd = P['decoder.decoder_rnn.weight_hh'].size(1)
    h_a = memory.new_zeros(B, Ha); c_a = memory.new_zeros(B, Ha)
    h_d = memory.new_zeros(B, Hd); c_d = memory.new_zeros(B, Hd)
    w = memory.new_zeros(B, Ti); w_cum = memory.new_zeros(B, Ti)
    ctx = memory.new_zeros(B, memory.size(2))
    pm = mm(memory, P['decoder.attention_layer.memory_layer.linear_layer.weight'])
    pad = torch.zeros(B, Ti, dtype=torch.bool)
    frame = memory.new_zeros(B, n_mel)
    mel_out, gate_out, aligns = [], [], []
    while True:
        t = len(mel_out)
        x = prenet_forward(P, frame, [k[t] for k in prenet_keep], fp32=True)
        h_a, c_a = lstm_cell(torch.cat((x, ctx), -1), h_a, c_a,
                             P['decoder.attention_rnn.weight_ih'], P['decoder.attention_rnn.weight_hh'],
                             P['decoder.attention_rnn.bias_ih'], P['decoder.attention_rnn.bias_hh'],
                             bf16_from=x.size(1) if bf16_products else None, head_fp32=True)
        ctx, w = attention_step(P, h_a, memory, pm, w, w_cum, pad)
        w_cum = w_cum + w
        h_d, c_d = lstm_cell(torch.cat((h_a, ctx), -1), h_d, c_d,
                             P['decoder.decoder_rnn.weight_ih'], P['decoder.decoder_rnn.weight_hh'],
                             P['decoder.decoder_rnn.bias_ih'], P['decoder.decoder_rnn.bias_hh'],
                             bf16_from=0 if bf16_products else None)
        hc = torch.cat((h_d, ctx), dim=1)
        frame = hc @ P['decoder.linear_projection.linear_layer.weight'].t() + P['decoder.linear_projection.linear_layer.bias']
        # (B,1) per step: unlike Decoder.forward (model.py:446), inference does not squeeze the gate, so
        # parse_decoder_outputs returns it as (B,T,1)
        gate = hc @ P['decoder.gate_layer.linear_layer.weight'].t() + P['decoder.gate_layer.linear_layer.bias']
        mel_out.append(frame); gate_out.append(gate); aligns.append(w)
        if bool((torch.sigmoid(gate) > gate_threshold).all()) or len(mel_out) == max_decoder_steps:
            break
    aligns = torch.stack(aligns).transpose(0, 1)
    gate_out = torch.stack(gate_out).transpose(0, 1).contiguous()
    mel_out = torch.stack(mel_out).transpose(0, 1).transpose(1, 2).contiguous()
    return mel_out, gate_out, aligns


def generator_inference(P, text: Tensor, style: Optional[Tensor], prenet_keep: List[Tensor],
                        speaker: Optional[Tensor] = None, emotions: Optional[Tensor] = None,
                        gate_threshold: float = 0.5, max_decoder_steps: int = 500, encoder_inputs: bool = False,
                        bf16_products: bool = False):
    """Tacotron2.inference, model.py:675-723: eval-mode encoder without packing
    (Encoder.inference :214-224), memory = [encoder | speaker | emotions | style], free-running decoder,
    PostNet residual; parse_output without lengths leaves the outputs unmasked.  With encoder_inputs the
    emotions and the style (B,noise) are channels of the embedded text instead (:694-701) and the decoder
    gets neither."""
    with fp16_run(bf16_products):
        emb = F.embedding(text, P['embedding.weight']).transpose(1, 2)
        if encoder_inputs:
            if emotions is not None:
                emb = torch.cat((emb, emotions[:, :, None].repeat(1, 1, emb.size(-1))), dim=1)
            if style is not None:
                emb = torch.cat((emb, style[:, :, None].repeat(1, 1, emb.size(-1))), dim=1)
            style = emotions = None
        lengths = torch.full((text.size(0),), text.size(1), dtype=torch.long)
        enc = encoder_forward(P, emb, lengths, None, training=False)
        if speaker is not None and 'speaker_embedding.weight' in P:
            enc = torch.cat((enc, F.embedding(speaker, P['speaker_embedding.weight'])[:, None].repeat(1, enc.size(1), 1)), dim=2)
        if emotions is not None:
            enc = torch.cat((enc, emotions[:, None].repeat(1, enc.size(1), 1)), dim=2)
        mel, gate, align = decoder_inference(P, enc, style, prenet_keep, gate_threshold, max_decoder_steps, bf16_products)
        mel_post = mel + postnet_forward(P, mel, None, training=False)
    return mel, mel_post, gate, align


# ----------------------------------------------------------------------------- discriminator
def disc_window_input(x: Tensor, window: int) -> Tensor:
    """Discriminator.forward input shaping, model.py:523-526.  x: (B, T, n_mel).
    Drops the T%window tail, appends the last `window` frames, then a RAW row-major
    reshape to (B, window*n_mel, W): r[b,c,w] = X'[b].flat[c*W + w]."""
    if x.size(1) % window != 0:
        x = torch.cat((x[:, :-(x.size(1) % window)], x[:, -window:]), dim=1)
    return x.reshape(x.size(0), window * x.size(2), -1)


def discriminator_forward(P, x: Tensor, window: int, disc_keep: Optional[List[Tensor]],
                          training: bool = True) -> Tensor:
    """Discriminator.forward, model.py:514-527; DiscConv1d = conv -> Dropout(.5) -> Tanh
    (layers.py:43-60); dilations 1,2,2,2; last layer Conv1d(n_mel,1,1)."""
    r = disc_window_input(x, window)
    dil = [1, 2, 2, 2]
    for i in range(4):
        w = P[f'discriminator.{i}.module.0.weight']
        r = conv1d(r, w, P[f'discriminator.{i}.module.0.bias'],
                     padding=dil[i] * (w.size(2) - 1) // 2, dilation=dil[i])
        r = torch.tanh(keep_scale(r, disc_keep[i] if (training and disc_keep is not None) else None, 0.5))
    r = conv1d(r, P['discriminator.4.weight'], P['discriminator.4.bias'])
    return r.squeeze(1)


def adversarial_loss(P, mel: Tensor, lengths: Tensor, window: int, disc_keep, training=True) -> Tensor:
    """Discriminator.adversarial_loss, model.py:529-540: per item mean over the first
    ceil(len/window) outputs, then the batch mean.  mel: (B, n_mel, T)."""
    out = discriminator_forward(P, mel.transpose(1, 2), window, disc_keep, training)
    loss = 0
    for b in range(lengths.size(0)):
        n = int(math.ceil(float(lengths[b]) / window))
        loss = loss + out[b, :n].mean()
    return loss / lengths.size(0)


# ----------------------------------------------------------------------------- losses
def gradient_penalty(P, real: Tensor, fake: Tensor, real_len: Tensor, fake_len: Tensor, alpha: Tensor,
                     window: int, disc_keep, training: bool = True) -> Tensor:
    """gradient_penalty, train.py:32-69 (WGAN-GP on the conv discriminator): x = alpha real + (1 - alpha) fake cropped
    to the shorter tensor and zeroed beyond the lengths of that tensor; g = d sum(D(x)) / d x (all windows, dropout
    on), masked again; mean over items of (sqrt(sum g^2 + 1e-12) - 1)^2.  alpha: (B,) U[0,1)."""
    B = real.size(0)
    a = alpha.view(B, 1, 1)
    if real.size(2) < fake.size(2):
        x = a * real + (1 - a) * fake[:, :, :real.size(2)]
        mask = length_mask(real_len)
    else:
        x = a * real[:, :, :fake.size(2)] + (1 - a) * fake
        mask = length_mask(fake_len)
    mask = mask[:, None, :].expand_as(x)
    x = x.detach().masked_fill(~mask, 0.0).requires_grad_(True)
    prob = discriminator_forward(P, x.transpose(1, 2), window, disc_keep, training)
    g = torch.autograd.grad(prob, x, torch.ones_like(prob), create_graph=True)[0]
    g = g.masked_fill(~mask, 0.0)
    norm = torch.sqrt((g.reshape(B, -1) ** 2).sum(dim=1) + 1e-12)
    return ((norm - 1) ** 2).mean()


def linear_discriminator_loss(P, mel: Tensor, starts, window: int, keeps, training: bool = True) -> Tensor:
    """LinearDiscriminator.adversarial_loss (model.py:567-583) with the window starts given (the reference draws them
    with random.randint; the caller replays the same sequence).  mel (B, n_mel, T); keeps: 3 masks (NW, dim) over the
    windows of all items in order."""
    wins = []
    counts = []
    for b, st in enumerate(starts):
        x = mel[b].t()                                   # (T, n_mel)
        for s0 in st:
            wins.append(x[s0:s0 + window].flatten())
        counts.append(len(st))
    h = torch.stack(wins)
    for i in range(3):
        h = mm(h, P[f'discriminator.{i}.module.0.weight']) + P[f'discriminator.{i}.module.0.bias']
        h = torch.tanh(keep_scale(h, keeps[i] if (keeps is not None and training) else None, 0.5))
    out = (mm(h, P['discriminator.3.weight']) + P['discriminator.3.bias']).squeeze(1)
    loss, o = 0.0, 0
    for n in counts:
        loss = loss + out[o:o + n].mean()
        o += n
    return loss / len(counts)


def attention_target(n_chars: int, n_frames: int, width: float = 0.5, sig: float = 3.0) -> Tensor:
    """attention_mask/gaussian, loss_function.py:40-52: row n is exp(-(t-c_n)^2/(width*sig)),
    c_n = n*(n_frames-1)//(n_chars-1) (integer floor)."""
    t = torch.linspace(0, n_frames - 1, n_frames)
    rows = []
    for n in range(n_chars):
        c = n * (n_frames - 1) // (n_chars - 1)
        rows.append(torch.exp(-torch.pow(t - c, 2.0) / (width * sig)))
    return torch.stack(rows)


def attention_loss(align: Tensor, in_len: Tensor, out_len: Tensor) -> Tensor:
    """loss_function.py:25-32: BCE(align[b].T[:Ti_b,:To_b], target) averaged over the batch."""
    total = 0
    for b in range(align.size(0)):
        ni, no = int(in_len[b]), int(out_len[b])
        total = total + F.binary_cross_entropy(align[b].T[:ni, :no], attention_target(ni, no).to(align.dtype))
    return total / align.size(0)


def tacotron_loss(outputs, mel_target: Tensor, gate_target: Tensor, in_len: Tensor, out_len: Tensor):
    """Tacotron2Loss.forward, loss_function.py:10-22."""
    mel, mel_post, gate, align = outputs
    mel_loss = F.mse_loss(mel, mel_target) + F.mse_loss(mel_post, mel_target)
    gate_loss = F.binary_cross_entropy_with_logits(gate.reshape(-1, 1), gate_target.reshape(-1, 1))
    return mel_loss, gate_loss, attention_loss(align, in_len, out_len)


# ----------------------------------------------------------------------------- STFT -> mel
def hann_periodic(n: int) -> torch.Tensor:
    """scipy.signal.get_window('hann', n, fftbins=True) (stft.py:66)."""
    k = torch.arange(n, dtype=torch.float64)
    return 0.5 - 0.5 * torch.cos(2.0 * math.pi * k / n)


def stft_basis(filter_length: int = 1024) -> Tensor:
    """STFT.__init__, stft.py:53-75: rows 0..cutoff-1 = Re, cutoff.. = Im of fft(eye(N)),
    each multiplied by the periodic Hann window; float32."""
    n = filter_length
    cutoff = n // 2 + 1
    k = torch.arange(cutoff, dtype=torch.float64)[:, None]
    t = torch.arange(n, dtype=torch.float64)[None, :]
    ang = 2.0 * math.pi * k * t / n
    basis = torch.cat([torch.cos(ang), -torch.sin(ang)], dim=0).float()
    return basis * hann_periodic(n).float()[None, :]


def slaney_mel_basis(sr: int = 22050, n_fft: int = 1024, n_mels: int = 80,
                     fmin: float = 0.0, fmax: float = 8000.0) -> Tensor:
    """librosa 0.7.2 `filters.mel(sr, n_fft, n_mels, fmin, fmax)` (htk=False, norm=1), called
    at layers.py:84-85.  librosa is not vendored in the reference and absent here, so this
    follows the published Slaney definition: PARITY UNPINNED for these coefficients."""
    def hz_to_mel(f):
        f = torch.as_tensor(f, dtype=torch.float64)
        f_sp = 200.0 / 3
        mel = f / f_sp
        min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
        min_log_mel = min_log_hz / f_sp
        return torch.where(f >= min_log_hz, min_log_mel + torch.log(f.clamp(min=1e-10) / min_log_hz) / logstep, mel)

    def mel_to_hz(m):
        f_sp = 200.0 / 3
        min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
        min_log_mel = min_log_hz / f_sp
        return torch.where(m >= min_log_mel, min_log_hz * torch.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = torch.linspace(0, sr / 2.0, n_fft // 2 + 1, dtype=torch.float64)
    mel_f = mel_to_hz(torch.linspace(float(hz_to_mel(fmin)), float(hz_to_mel(fmax)), n_mels + 2,
                                     dtype=torch.float64))
    fdiff = mel_f[1:] - mel_f[:-1]
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = torch.clamp(torch.minimum(lower, upper), min=0.0)
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).float()


def stft_magnitude(audio: Tensor, filter_length: int = 1024, hop: int = 256) -> Tensor:
    """STFT.transform, stft.py:77-105: reflect pad N/2, conv1d with the windowed DFT
    basis at stride hop, magnitude sqrt(re^2+im^2).  audio (B, n) -> (B, N/2+1, n//hop+1)."""
    x = F.pad(audio[:, None, None, :], (filter_length // 2, filter_length // 2, 0, 0), mode='reflect').squeeze(1)
    ft = F.conv1d(x, stft_basis(filter_length)[:, None, :], stride=hop)
    cutoff = filter_length // 2 + 1
    return torch.sqrt(ft[:, :cutoff] ** 2 + ft[:, cutoff:] ** 2)


def mel_spectrogram(audio: Tensor, mel_basis: Optional[Tensor] = None) -> Tensor:
    """TacotronSTFT.mel_spectrogram, layers.py:97-114 + audio_processing.py:78-84."""
    assert audio.min() >= -1 and audio.max() <= 1
    if mel_basis is None:
        mel_basis = slaney_mel_basis()
    mag = stft_magnitude(audio)
    return torch.log(torch.clamp(torch.matmul(mel_basis, mag), min=1e-5))


# ----------------------------------------------------------------------------- training steps
def init_params(hp, seed: int = 1234) -> (Dict[str, Tensor], Dict[str, Tensor]):
    """Parameter dicts (G, D) with the reference's shapes and initialisers
    (layers.py:14-16,35-36,52; model.py:603-605; torch defaults for LSTM/BN/Conv bias)."""
    g = torch.Generator().manual_seed(seed)

    def xavier(shape, gain):
        fan_out, fan_in = shape[0], shape[1]
        rf = 1
        for s in shape[2:]:
            rf *= s
        a = gain * math.sqrt(6.0 / ((fan_in + fan_out) * rf))
        return (torch.rand(shape, generator=g) * 2 - 1) * a

    def unif(shape, a):
        return (torch.rand(shape, generator=g) * 2 - 1) * a

    G = {}
    gain_relu, gain_tanh, gain_lin, gain_sig = math.sqrt(2.0), 5.0 / 3, 1.0, 1.0
    E = hp.encoder_embedding_dim
    val = math.sqrt(3.0) * math.sqrt(2.0 / (hp.n_symbols + hp.symbols_embedding_dim))
    G['embedding.weight'] = unif((hp.n_symbols, hp.symbols_embedding_dim), val)
    ks = hp.encoder_kernel_size

    def bn(prefix, c):
        G[prefix + '.weight'] = torch.ones(c); G[prefix + '.bias'] = torch.zeros(c)
        G[prefix + '.running_mean'] = torch.zeros(c); G[prefix + '.running_var'] = torch.ones(c)

    noise = hp.noise_size if hp.use_noise else 0
    labels = 5 if (hp.use_labels and hp.vesus_path) else 0
    enc_in = bool(getattr(hp, 'encoder_inputs', False))                      # model.py:162-170, 240-246
    for i in range(hp.encoder_n_convolutions):
        cin = E + (((hp.n_labels if labels else 0) + noise) if (enc_in and i == 0) else 0)
        G[f'encoder.convolutions.{i}.0.conv.weight'] = xavier((E, cin, ks), gain_relu)
        G[f'encoder.convolutions.{i}.0.conv.bias'] = unif((E,), 1 / math.sqrt(cin * ks))
        bn(f'encoder.convolutions.{i}.1', E)
    H = E // 2
    for sfx in ('', '_reverse'):
        G['encoder.lstm.weight_ih_l0' + sfx] = unif((4 * H, E), 1 / math.sqrt(H))
        G['encoder.lstm.weight_hh_l0' + sfx] = unif((4 * H, H), 1 / math.sqrt(H))
        G['encoder.lstm.bias_ih_l0' + sfx] = unif((4 * H,), 1 / math.sqrt(H))
        G['encoder.lstm.bias_hh_l0' + sfx] = unif((4 * H,), 1 / math.sqrt(H))
    D_in = E + (0 if enc_in else noise)
    if hp.vesus_path:
        D_in += hp.speakers_embedding + (0 if enc_in else labels)
        G['speaker_embedding.weight'] = unif((123, hp.speakers_embedding), val)
    n_mel, pd = hp.n_mel_channels, hp.prenet_dim
    G['decoder.prenet.layers.0.linear_layer.weight'] = xavier((pd, n_mel), gain_lin)
    G['decoder.prenet.layers.1.linear_layer.weight'] = xavier((pd, pd), gain_lin)
    Ha, Hd, A = hp.attention_rnn_dim, hp.decoder_rnn_dim, hp.attention_dim
    for name, inp, hid in (('attention_rnn', pd + D_in, Ha), ('decoder_rnn', Ha + D_in, Hd)):
        G[f'decoder.{name}.weight_ih'] = unif((4 * hid, inp), 1 / math.sqrt(hid))
        G[f'decoder.{name}.weight_hh'] = unif((4 * hid, hid), 1 / math.sqrt(hid))
        G[f'decoder.{name}.bias_ih'] = unif((4 * hid,), 1 / math.sqrt(hid))
        G[f'decoder.{name}.bias_hh'] = unif((4 * hid,), 1 / math.sqrt(hid))
    al = 'decoder.attention_layer.'
    G[al + 'query_layer.linear_layer.weight'] = xavier((A, Ha), gain_tanh)
    G[al + 'memory_layer.linear_layer.weight'] = xavier((A, D_in), gain_tanh)
    G[al + 'v.linear_layer.weight'] = xavier((1, A), gain_lin)
    nf, lk = hp.attention_location_n_filters, hp.attention_location_kernel_size
    G[al + 'location_layer.location_conv.conv.weight'] = xavier((nf, 2, lk), gain_lin)
    G[al + 'location_layer.location_dense.linear_layer.weight'] = xavier((A, nf), gain_tanh)
    G['decoder.linear_projection.linear_layer.weight'] = xavier((n_mel, Hd + D_in), gain_lin)
    G['decoder.linear_projection.linear_layer.bias'] = unif((n_mel,), 1 / math.sqrt(Hd + D_in))
    G['decoder.gate_layer.linear_layer.weight'] = xavier((1, Hd + D_in), gain_sig)
    G['decoder.gate_layer.linear_layer.bias'] = unif((1,), 1 / math.sqrt(Hd + D_in))
    pe, pk = hp.postnet_embedding_dim, hp.postnet_kernel_size
    chans = [n_mel] + [pe] * (hp.postnet_n_convolutions - 1) + [n_mel]
    for i in range(hp.postnet_n_convolutions):
        gain = gain_tanh if i < hp.postnet_n_convolutions - 1 else gain_lin
        G[f'postnet.convolutions.{i}.0.conv.weight'] = xavier((chans[i + 1], chans[i], pk), gain)
        G[f'postnet.convolutions.{i}.0.conv.bias'] = unif((chans[i + 1],), 1 / math.sqrt(chans[i] * pk))
        bn(f'postnet.convolutions.{i}.1', chans[i + 1])

    Dp = {}
    in_dim = hp.discriminator_window * n_mel
    dd = hp.discriminator_dim
    first = min((in_dim // dd + 1) * dd, 1024)
    dch = [in_dim, first, dd, dd, n_mel]
    for i in range(4):
        Dp[f'discriminator.{i}.module.0.weight'] = xavier((dch[i + 1], dch[i], 5), gain_tanh)
        Dp[f'discriminator.{i}.module.0.bias'] = unif((dch[i + 1],), 1 / math.sqrt(dch[i] * 5))
    Dp['discriminator.4.weight'] = unif((1, n_mel, 1), 1 / math.sqrt(n_mel))
    Dp['discriminator.4.bias'] = unif((1,), 1 / math.sqrt(n_mel))
    return G, Dp


BUFFER_SUFFIXES = ('running_mean', 'running_var', 'num_batches_tracked')


def trainable(P: Dict[str, Tensor]) -> List[str]:
    return [k for k in P if not k.endswith(BUFFER_SUFFIXES)]


def adam_step(P, grads, state, lr, step, wd=1e-6, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam(weight_decay=wd) (train.py:242-243): L2 added to the gradient."""
    for k in grads:
        g = grads[k] + wd * P[k]
        m, v = state.setdefault(k, (torch.zeros_like(g), torch.zeros_like(g)))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
        P[k] = P[k] - (lr / (1 - b1 ** step)) * m / denom


def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float) -> Tensor:
    """torch.nn.utils.clip_grad_norm_ (train.py:404-405)."""
    total = torch.sqrt(sum((g.detach() ** 2).sum() for g in grads.values()))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return total


def _with_hp_precision(fn):
    """Run a step function under fp16_run(hp.fp16_run); hp is the positional / keyword argument named `hp`."""
    import functools
    import inspect
    pos = list(inspect.signature(fn).parameters).index('hp')

    @functools.wraps(fn)
    def wrapped(*a, **k):
        hp = k['hp'] if 'hp' in k else a[pos]
        with fp16_run(getattr(hp, 'fp16_run', False)):
            return fn(*a, **k)
    return wrapped


@_with_hp_precision
def generator_step_grads(G, Dp, batch, rng, hp, use_attn_loss=True):
    """Loss and parameter gradients of one generator step, train.py:361-397.
    batch = (text, in_len, mel, gate, out_len[, speaker_ids, emotions]);
    rng additionally holds 'disc_keep' for the adversarial D forward."""
    text, in_len, mel_t, gate_t, out_len = batch[:5]
    spk = batch[5] if len(batch) > 5 else None
    emo = batch[6] if len(batch) > 6 else None
    Pg = {k: (v.clone().requires_grad_(True) if k in trainable(G) else v) for k, v in G.items()}
    new_stats = {}
    out = generator_forward(Pg, text, in_len, mel_t, out_len, rng, spk, emo,
                            use_labels=bool(hp.use_labels and hp.vesus_path), new_stats=new_stats,
                            p_att=hp.p_attention_dropout, p_dec=hp.p_decoder_dropout,
                            encoder_inputs=bool(getattr(hp, 'encoder_inputs', False)),
                            bf16_products=bool(getattr(hp, 'fp16_run', False)))
    mel_loss, gate_loss, attn_loss = tacotron_loss(out, mel_t, gate_t, in_len, out_len)
    adv = adversarial_loss(Dp, out[1], out_len, hp.discriminator_window, rng.get('disc_keep'))
    total = mel_loss + gate_loss + adv
    if use_attn_loss:
        total = total + 10 * attn_loss
    names = trainable(Pg)
    grads = torch.autograd.grad(total, [Pg[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(Pg[k])) for k, g in zip(names, grads)}
    losses = dict(mel=mel_loss.detach(), gate=gate_loss.detach(), attn=attn_loss.detach(),
                  adv=adv.detach(), total=total.detach())
    return [o.detach() for o in out], losses, grads, new_stats


@_with_hp_precision
def discriminator_step_grads(Dp, real_mel, real_len, fake_mel, fake_len, keep_real, keep_fake, hp):
    """Loss and D-parameter gradients of one discriminator step, train.py:302-344:
    loss = (D(real) - D(fake)) / 2 (real=+1, fake=-1, train.py:509-510)."""
    Pd = {k: v.clone().requires_grad_(True) for k, v in Dp.items()}
    real = adversarial_loss(Pd, real_mel, real_len, hp.discriminator_window, keep_real)
    fake = -adversarial_loss(Pd, fake_mel.detach(), fake_len, hp.discriminator_window, keep_fake)
    loss = (real + fake) / 2
    names = list(Pd)
    grads = dict(zip(names, torch.autograd.grad(loss, [Pd[k] for k in names])))
    return dict(real=real.detach(), fake=fake.detach(), total=loss.detach()), grads
