"""`TacotronSTFT.mel_spectrogram` on the GPU (reference layers.py:76-114, stft.py:42-105,
audio_processing.py:78-84): reflect-pad, windowed DFT as an MFMA GEMM whose A operand is the
padded signal itself read with row stride = hop (overlapping frames, nothing materialised),
magnitude, mel filterbank GEMM, log(clamp(., 1e-5)).

The mel filterbank restates librosa 0.7.2 `filters.mel(sr, n_fft, n_mels, fmin, fmax)` (Slaney
scale, area normalised).  librosa is not part of the reference tree: parity for those coefficients
is unpinned."""
import math

import numpy as np
import torch
from torch import nn

from . import ops


def hann_periodic(n):
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def windowed_dft_basis(filter_length):
    """Rows 0..cutoff-1: real parts, rows cutoff..2*cutoff-1: imaginary parts of the DFT, times the
    periodic Hann window (reference stft.py:53-75)."""
    n = filter_length
    cutoff = n // 2 + 1
    k = np.arange(cutoff, dtype=np.float64)[:, None]
    t = np.arange(n, dtype=np.float64)[None, :]
    ang = 2.0 * np.pi * k * t / n
    basis = np.concatenate([np.cos(ang), -np.sin(ang)], axis=0).astype(np.float32)
    return basis * hann_periodic(n).astype(np.float32)[None, :]


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, sr / 2.0, n_fft // 2 + 1)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).astype(np.float32)


class TacotronSTFT(nn.Module):
    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, sampling_rate=22050,
                 mel_fmin=0.0, mel_fmax=8000.0):
        super().__init__()
        assert win_length == filter_length, 'the reference pads the window to filter_length; only equal sizes are used'
        self.filter_length, self.hop_length = filter_length, hop_length
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self.cutoff = filter_length // 2 + 1
        self.register_buffer('forward_basis', torch.from_numpy(windowed_dft_basis(filter_length)))
        self.register_buffer('mel_basis', torch.from_numpy(
            slaney_mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)))

    def _magnitude_rows(self, y):
        """y (B, n) cuda -> (B*frames, cutoff) magnitudes, frames = n // hop + 1."""
        if not y.is_cuda:
            # no CPU fallback by design: the windowed DFT is a HIP GEMM.  A forked DataLoader worker cannot own a HIP
            # context, so raw audio is turned into mels in the MAIN process (data_utils.TextAudioLoader -> DeviceBatches)
            raise RuntimeError('TacotronSTFT runs on the GPU only (no CPU path): call it from the main process with CUDA '
                               'tensors -- inside DataLoader workers use data_utils.TextAudioLoader + DeviceBatches (raw '
                               'audio out of the worker, one batched STFT-mel on the device) or load_mel_from_disk=True')
        assert y.dim() == 2
        B, n = y.shape
        N, hop, cut = self.filter_length, self.hop_length, self.cutoff
        frames = n // hop + 1
        padded = torch.empty(B, n + N, device=y.device, dtype=torch.float32)
        ops.call('gt_reflect_pad', y.contiguous().float(), B, n, N // 2, padded)
        ft = torch.empty(B * frames, 2 * cut, device=y.device, dtype=torch.float32)
        ops.sgemm(ops.Operand(padded, ops.KC, ld=hop, batch_stride=n + N),
                  ops.Operand(self.forward_basis, ops.KC, ld=N), ft, frames, 2 * cut, N, batch=B,
                  c_batch_stride=frames * 2 * cut)
        mag = torch.empty(B * frames, cut, device=y.device, dtype=torch.float32)
        ops.call('gt_stft_magnitude', ft, B * frames, cut, mag)
        return mag, frames

    def magnitudes(self, y):
        mag, frames = self._magnitude_rows(y)
        B = y.size(0)
        out = torch.empty(B, self.cutoff, frames, device=y.device, dtype=torch.float32)
        ops.call('gt_permute3', mag, frames * self.cutoff, 1, self.cutoff, B, self.cutoff, frames, out, 0, 0, 0, 1.0)
        return out

    def mel_spectrogram(self, y):
        """y (B, T) in [-1, 1] -> (B, n_mel, T // hop + 1) log-mel (reference layers.py:97-114)."""
        assert torch.min(y.data) >= -1
        assert torch.max(y.data) <= 1
        mag, frames = self._magnitude_rows(y)
        B, nm = y.size(0), self.n_mel_channels
        mel = ops.linear(mag, self.mel_basis, B * frames, self.cutoff, nm)
        ops.call('gt_log_clamp', mel, mel.numel(), 1e-5, mel)
        out = torch.empty(B, nm, frames, device=y.device, dtype=torch.float32)
        ops.call('gt_permute3', mel, frames * nm, 1, nm, B, nm, frames, out, 0, 0, 0, 1.0)
        return out

    def mel_spectrogram_batch(self, audio, lengths):
        """One collated batch of raw audio -> (mel_padded (B, n_mel, F), gate_padded (B, F), frames (B,) int64).
        audio (B, Nmax) cuda, zero beyond lengths (B,) int64 cuda.  Item b's frames [0, lengths[b]//hop + 1) equal
        `mel_spectrogram(audio[b:b+1, :lengths[b]])`: every row is reflect-padded around its own ends; later
        frames are zero and the gate target is 1 from the last valid frame on (data_utils.py:117-128)."""
        assert audio.is_cuda and audio.dim() == 2 and lengths.dtype == torch.int64 and lengths.is_cuda
        B, n = audio.shape
        N, hop, cut, nm = self.filter_length, self.hop_length, self.cutoff, self.n_mel_channels
        F = n // hop + 1
        padded = torch.empty(B, n + N, device=audio.device, dtype=torch.float32)
        ops.call('gt_reflect_pad_ragged', audio.contiguous().float(), B, n, lengths, N // 2, padded)
        ft = torch.empty(B * F, 2 * cut, device=audio.device, dtype=torch.float32)
        ops.sgemm(ops.Operand(padded, ops.KC, ld=hop, batch_stride=n + N),
                  ops.Operand(self.forward_basis, ops.KC, ld=N), ft, F, 2 * cut, N, batch=B, c_batch_stride=F * 2 * cut)
        mag = torch.empty(B * F, cut, device=audio.device, dtype=torch.float32)
        ops.call('gt_stft_magnitude', ft, B * F, cut, mag)
        mel = ops.linear(mag, self.mel_basis, B * F, cut, nm)
        ops.call('gt_log_clamp', mel, mel.numel(), 1e-5, mel)
        frames = lengths // hop + 1
        mel_padded = torch.empty(B, nm, F, device=audio.device, dtype=torch.float32)
        gate_padded = torch.empty(B, F, device=audio.device, dtype=torch.float32)
        ops.call('gt_collate_mel', mel, frames, B, nm, F, mel_padded, gate_padded)
        return mel_padded, gate_padded, frames
