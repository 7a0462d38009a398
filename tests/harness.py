"""Test-side helpers: layout conversion between the oracle (channels-first, B rows) and the HIP
engine (channels-last, Bp rows per decoder step), tolerance checks, numpy restatement of the
counter-based RNG."""
import numpy as np
import torch


def close(a, b, rtol=1e-3, atol=1e-5, what=''):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, f'{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}'
    assert torch.isfinite(a).all(), f'{what}: non-finite values'
    err = (a - b).abs().max().item() if a.numel() else 0.0
    scale = b.abs().max().item() if b.numel() else 0.0
    assert err <= atol + rtol * scale, f'{what}: max err {err:.3e} vs scale {scale:.3e} (rtol {rtol}, atol {atol})'


def roundup(x, m):
    return -(-x // m) * m


def pad_rows(t, Bp):
    """(T, B, C) -> (T, Bp, C) zero padded."""
    T, B, C = t.shape
    out = torch.zeros(T, Bp, C, dtype=t.dtype)
    out[:, :B] = t
    return out


def kernel_rng(rng, B, device):
    """Oracle-layout rng dict (tests/golden/common.draw_generator_rng) -> engine layout on `device`."""
    Bp = roundup(B, 16) if B <= 64 else roundup(B, 64)
    u8 = lambda t: t.to(torch.uint8).contiguous().to(device)
    out = {}
    if rng.get('enc_keep') is not None:
        out['enc_keep'] = [u8(k.permute(0, 2, 1)) for k in rng['enc_keep']]
    To = rng['att_keep'].shape[0]
    out['prenet_keep'] = [u8(pad_rows(k[:To], Bp)) for k in rng['prenet_keep']]
    out['att_keep'] = u8(pad_rows(rng['att_keep'], Bp))
    out['dec_keep'] = u8(pad_rows(rng['dec_keep'], Bp))
    out['post_keep'] = [u8(k.permute(0, 2, 1)) for k in rng['post_keep']]
    if rng.get('noise') is not None:
        out['noise'] = rng['noise'].contiguous().to(device)
    return out


def kernel_disc_keep(keeps, device):
    return [k.permute(0, 2, 1).to(torch.uint8).contiguous().to(device) for k in keeps]


# ---- numpy restatement of gt_hash32 / gt_u01 (gantron_amd/csrc/common.h)
def u01(seed, offset, n):
    with np.errstate(over='ignore'):
        ctr = np.arange(offset, offset + n, dtype=np.uint64)
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (ctr + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        h = (z >> np.uint64(32)).astype(np.uint32)
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
