"""`TextMelLoader` / `TextMelCollate` with the reference's behaviour (reference data_utils.py:13-131).

Differences that come with the MI355X design:
* the mel spectrogram is computed by the HIP STFT (stft.TacotronSTFT) in the MAIN process -- a forked
  DataLoader worker cannot touch the GPU -- so use `num_workers=0`, or precompute mels
  (`load_mel_from_disk=True`) when workers are wanted;
* wav files are decoded with scipy (16-bit PCM or float); files whose rate differs from
  `hparams.sampling_rate` are rejected instead of resampled (librosa/resampy are not in the image).
"""
import random

import numpy as np
import torch
import torch.utils.data

from .text import text_to_sequence


def load_filepaths_and_text(filename, wavs_path, split='|'):
    """reference utils.py:47-53"""
    with open(filename, encoding='utf-8') as f:
        return [[wavs_path + parts[0]] + parts[1:] for parts in (line.strip().split(split) for line in f)]


def load_vesus(filename, wavs_path, split='|', use_labels='one', use_text=True):
    """reference utils.py:81-122: (paths[, text]), speaker ids, 5-way emotion labels."""
    ids = {'Neutral': [1, 0, 0, 0, 0], 'Angry': [0, 1, 0, 0, 0], 'Happy': [0, 0, 1, 0, 0],
           'Sad': [0, 0, 0, 1, 0], 'Fearful': [0, 0, 0, 0, 1]}
    items, speakers, emotions = [], [], []
    with open(filename, encoding='utf-8') as f:
        for line in f:
            parts = line.strip().split(split)
            path = wavs_path + parts[0]
            items.append([path, parts[1]] if use_text else path)
            speakers.append(int(parts[2]))
            if use_labels == 'one':
                votes = [float(v) for v in parts[3].split(',')]
                onehot = np.zeros(len(votes))
                onehot[int(np.argmax(votes))] = 1
                emotions.append(onehot)
            elif use_labels == 'intended':
                emotions.append(ids[parts[0].split('/')[1]])
            else:
                emotions.append([float(v) for v in parts[3].split(',')])
    return items, speakers, emotions


def load_wav_to_torch(full_path, sampling_rate=22050):
    """reference utils.py:27-31 (librosa.load -> float32 in [-1, 1], peak-normalised only if it clips)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(full_path)
    if sr != sampling_rate:
        raise ValueError(f'{full_path}: sampling rate {sr} != {sampling_rate} (resampling is not available)')
    if data.dtype == np.int16:
        data = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        data = data.astype(np.float32) / 2147483648.0
    data = data.astype(np.float32)
    if data.ndim > 1:
        data = data.mean(axis=1)
    peak = max(abs(float(data.min())), abs(float(data.max())))
    if peak > 1:
        data = data / peak
    return torch.from_numpy(data)


class TextMelLoader(torch.utils.data.Dataset):
    def __init__(self, audiopaths_and_text, hparams, wavs_path):
        self.audiopaths_and_text = load_filepaths_and_text(audiopaths_and_text[0], wavs_path)
        self.vesus = bool(hparams.vesus_path)
        if self.vesus:
            extra, speakers, emotions = load_vesus(
                audiopaths_and_text[1], hparams.vesus_path,
                use_labels='intended' if hparams.use_intended_labels else 'multi')
            n = len(self.audiopaths_and_text)
            self.speakers = torch.IntTensor([0] * n + speakers)
            self.emotions = torch.FloatTensor([[0, 0, 0, 0, 0]] * n + [list(e) for e in emotions])
            self.audiopaths_and_text.extend(extra)
        self.text_cleaners = hparams.text_cleaners
        self.sampling_rate = hparams.sampling_rate
        self.load_mel_from_disk = hparams.load_mel_from_disk
        self.n_mel_channels = hparams.n_mel_channels
        self._stft_args = (hparams.filter_length, hparams.hop_length, hparams.win_length, hparams.n_mel_channels,
                           hparams.sampling_rate, hparams.mel_fmin, hparams.mel_fmax)
        self._stft = None
        self.idx = list(range(len(self.audiopaths_and_text)))
        random.seed(hparams.seed)
        random.shuffle(self.idx)

    def _stft_module(self):
        if self._stft is None:
            from .stft import TacotronSTFT
            self._stft = TacotronSTFT(*self._stft_args).cuda()
        return self._stft

    def get_mel(self, filename):
        if self.load_mel_from_disk:
            mel = torch.from_numpy(np.load(filename))
            assert mel.size(0) == self.n_mel_channels, \
                'Mel dimension mismatch: given {}, expected {}'.format(mel.size(0), self.n_mel_channels)
            return mel
        audio = load_wav_to_torch(filename, self.sampling_rate).unsqueeze(0)
        return self._stft_module().mel_spectrogram(audio.cuda()).squeeze(0).cpu()

    def get_text(self, text):
        return torch.IntTensor(text_to_sequence(text, self.text_cleaners))

    def __getitem__(self, index):
        i = self.idx[index]
        path, text = self.audiopaths_and_text[i][0], self.audiopaths_and_text[i][1]
        item = (self.get_text(text), self.get_mel(path))
        if self.vesus:
            item += (self.speakers[i], self.emotions[i])
        return item

    def __len__(self):
        return len(self.audiopaths_and_text)


class TextMelCollate:
    """Sort by text length (descending), zero-pad text and mel, gate = 1 from the last valid frame on
    (reference data_utils.py:88-131)."""

    def __init__(self, n_frames_per_step):
        self.n_frames_per_step = n_frames_per_step

    def __call__(self, batch):
        lengths = torch.LongTensor([len(x[0]) for x in batch])
        input_lengths, order = torch.sort(lengths, dim=0, descending=True)
        B = len(batch)
        text_padded = torch.zeros(B, int(input_lengths[0]), dtype=torch.long)
        speaker_ids = torch.zeros(B)          # the reference leaves these uninitialised without VESUS
        emotions = torch.zeros(B, 5)
        num_mels = batch[0][1].size(0)
        max_target_len = max(x[1].size(1) for x in batch)
        rem = max_target_len % self.n_frames_per_step
        if rem:
            max_target_len += self.n_frames_per_step - rem
        mel_padded = torch.zeros(B, num_mels, max_target_len)
        gate_padded = torch.zeros(B, max_target_len)
        output_lengths = torch.zeros(B, dtype=torch.long)
        for i, src in enumerate(order.tolist()):
            item = batch[src]
            text, mel = item[0], item[1]
            text_padded[i, :text.size(0)] = text
            if len(item) == 4:
                speaker_ids[i] = item[2]
                emotions[i] = item[3]
            mel_padded[i, :, :mel.size(1)] = mel
            gate_padded[i, mel.size(1) - 1:] = 1
            output_lengths[i] = mel.size(1)
        return text_padded, input_lengths, mel_padded, gate_padded, speaker_ids, emotions, output_lengths
