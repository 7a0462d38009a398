"""The generator-step and discriminator-step bodies of the reference training loop
(reference train.py:361-423 and :302-360), factored out so that train.py, bench.py and the tests
run the identical sequence of calls.  Sign convention: real = +1, fake = -1 (train.py:509-510)."""
import torch

REAL, FAKE = 1, -1


def generator_step(generator, discriminator, g_optimizer, criterion, batch, hparams, iteration=0, lr=None):
    """zero_grad -> parse_batch -> G forward -> Tacotron2Loss (+ guided attention) -> adversarial
    loss through the frozen D -> backward (-> gradient all-reduce) -> clip(1.0) -> Adam.
    Returns (losses dict of device scalars, [generated mel_postnet, output_lengths])."""
    if lr is not None:
        for group in g_optimizer.param_groups:
            group['lr'] = lr
    for p in discriminator.parameters():
        p.requires_grad = False
    g_optimizer.zero_grad()
    x, y = generator.parse_batch(batch)
    y_pred = generator(x)
    mel_loss, gate_loss, attn_loss = criterion(y_pred, y, x[1], x[-1])
    total = mel_loss + gate_loss
    adv = None
    if hparams.d_freq > 0:
        adv = REAL * discriminator.adversarial_loss(y_pred[1], x[-1])
        total = total + adv
    if iteration < hparams.attn_steps:
        total = total + 10 * attn_loss
    total.backward()
    sumsq = g_optimizer.step(max_norm=hparams.grad_clip_thresh)
    losses = dict(total=total.detach(), mel=mel_loss.detach(), gate=gate_loss.detach(), attn=attn_loss.detach(),
                  adv=None if adv is None else adv.detach(), grad_sumsq=sumsq)
    return losses, [y_pred[1].detach(), x[-1]]


def gradient_penalty(discriminator, real_data, generated_data, real_lengths, generated_lengths):
    """reference train.py:32-69 (same signature); the conv Discriminator implements it on the device."""
    if not hasattr(discriminator, 'gradient_penalty'):
        raise NotImplementedError('gradient penalty is implemented for discriminator_type=conv')
    return discriminator.gradient_penalty(real_data, generated_data, real_lengths, generated_lengths)


def discriminator_step(generator, discriminator, d_optimizer, batch, generated, hparams, lr=None):
    """zero_grad -> D(real) - D(fake.detach()) over 2 -> backward (-> all-reduce) -> Adam.
    (The reference calls clip_grad_norm_ BEFORE backward, train.py:326-327: a no-op on zeroed
    gradients, so no clipping is applied here either.)"""
    if lr is not None:
        for group in d_optimizer.param_groups:
            group['lr'] = lr
    for p in discriminator.parameters():
        p.requires_grad = True
    d_optimizer.zero_grad()
    x, _ = generator.parse_batch(batch)
    real_mel, output_lengths = x[2], x[-1]
    generated_mel, generated_lengths = generated
    real_loss = REAL * discriminator.adversarial_loss(real_mel, output_lengths)
    fake_loss = FAKE * discriminator.adversarial_loss(generated_mel.detach(), generated_lengths)
    loss = (real_loss + fake_loss) / 2
    out = {}
    # train.py:326-333: clipping (a no-op before backward) takes precedence; the penalty only runs with clipping off
    if not hparams.clipping_value > 0 and hparams.gradient_penalty_lambda > 0:
        gp = gradient_penalty(discriminator, real_mel, generated_mel.detach(), output_lengths, generated_lengths)
        loss = loss + hparams.gradient_penalty_lambda * gp
        out['gradient_penalty'] = gp.detach()
    loss.backward()
    d_optimizer.step()
    out.update(total=loss.detach(), real=real_loss.detach(), fake=fake_loss.detach())
    return out
