"""`Tacotron2Loss` with the reference's call signature (reference loss_function.py:6-52).

criterion(model_output, targets, input_lengths, output_lengths) -> (mel_loss, gate_loss, attn_loss)

One autograd node: the HIP kernels compute the three losses and their gradients w.r.t. the four
generator outputs in the forward pass (gt_taco_loss, gt_guided_attn_loss); backward scales those
unit gradients by the upstream scalars on the device."""
import torch
from torch import nn

from . import ops


class _TacoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mel, post, gate, align, mel_t, gate_t, in_len, out_len):
        dev = mel.device
        mel, post, gate, align = mel.contiguous(), post.contiguous(), gate.contiguous(), align.contiguous()
        sums = torch.empty(3, device=dev, dtype=torch.float32)
        dmel, dpost, dgate = torch.empty_like(mel), torch.empty_like(post), torch.empty_like(gate)
        ops.call('gt_taco_loss', mel, post, gate, mel_t.contiguous(), gate_t.contiguous(), mel.numel(), gate.numel(),
                 sums, dmel, dpost, dgate)
        attn = torch.empty(1, device=dev, dtype=torch.float32)
        dalign = torch.empty_like(align)
        B, To, Ti = align.shape
        ops.call('gt_guided_attn_loss', align, in_len.contiguous().long(), out_len.contiguous().long(), B, To, Ti, attn,
                 dalign)
        mel_loss = torch.empty(1, device=dev, dtype=torch.float32)
        ops.call('gt_axpby', sums[0:1], 1.0, sums[1:2], 1.0, mel_loss, 1)
        ctx.save_for_backward(dmel, dpost, dgate, dalign)
        return mel_loss.view(()), sums[2:3].clone().view(()), attn.view(())

    @staticmethod
    def backward(ctx, g_mel, g_gate, g_attn):
        dmel, dpost, dgate, dalign = ctx.saved_tensors
        outs = []
        for unit, g in ((dmel, g_mel), (dpost, g_mel), (dgate, g_gate), (dalign, g_attn)):
            out = torch.empty_like(unit)
            ops.call('gt_scale_dev', unit, g.contiguous().view(1), 1.0, out, unit.numel(), 0)
            outs.append(out)
        return tuple(outs) + (None, None, None, None)


class Tacotron2Loss(nn.Module):
    def forward(self, model_output, targets, input_lengths, output_lengths):
        mel_out, mel_out_postnet, gate_out, alignments = model_output
        return _TacoLossFn.apply(mel_out, mel_out_postnet, gate_out, alignments, targets[0], targets[1],
                                 input_lengths, output_lengths)
