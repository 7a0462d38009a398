"""`TextMelLoader` / `TextMelCollate` with the reference's behaviour (reference data_utils.py:13-131).

Differences that come with the MI355X design:
* the training path is `TextAudioLoader` (workers decode wav + text on the CPU only) -> `TextAudioCollate`
  (raw audio padded into pinned host buffers) -> `DeviceBatches` (main process: asynchronous H2D on a
  copy stream, ONE batched STFT-mel per batch on the GPU, double-buffered so batch i+1 is staged while
  batch i trains).  `TextMelLoader` keeps the reference's per-item behaviour (mel per item through the
  HIP STFT in the main process, or `load_mel_from_disk=True`) and is what validation uses;
* wav files are decoded with scipy (8/16/32-bit PCM or float); files whose rate differs from
  `hparams.sampling_rate` are resampled by `resample()` (a Kaiser-windowed polyphase filter with resampy's
  kaiser_best parameters: the reference's librosa/resampy are not in the image, so this step is parity-unpinned).
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


def resample(x, sr_in, sr_out):
    """Band-limited rate conversion for files that are not at hparams.sampling_rate.  The reference gets this from
    librosa.load -> resampy 0.2.2 'kaiser_best' (utils.py:27-31), third-party code that is not in its tree: this is a
    polyphase Kaiser-windowed sinc filter with resampy's published kaiser_best parameters (beta 14.7697, roll-off
    0.9476, 64 zero crossings) through scipy.signal.resample_poly -- the same design, not bit-identical (parity
    unpinned); the parity and benchmark inputs are generated at the target rate and never come here."""
    from math import gcd
    from scipy.signal import firwin, resample_poly
    g = gcd(int(sr_in), int(sr_out))
    up, down = int(sr_out) // g, int(sr_in) // g
    zeros, beta, rolloff = 64, 14.769656459379492, 0.9475632
    half = zeros * max(up, down)
    h = firwin(2 * half + 1, rolloff / max(up, down), window=('kaiser', beta))
    return resample_poly(x.astype(np.float64), up, down, window=h).astype(np.float32)


def load_wav_to_torch(full_path, sampling_rate=22050):
    """reference utils.py:27-31 (librosa.load -> float32 in [-1, 1], peak-normalised only if it clips)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(full_path)
    if data.dtype == np.int16:
        data = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        data = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        data = (data.astype(np.float32) - 128.0) / 128.0
    data = data.astype(np.float32)
    if data.ndim > 1:
        data = data.mean(axis=1)
    if sr != sampling_rate:
        data = resample(data, sr, sampling_rate)
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
        if torch.utils.data.get_worker_info() is not None:
            # reference data_utils.py:51-58 runs its CPU STFT inside the (forked) worker; a HIP context does not survive
            # a fork, and this package has no CPU arithmetic path on purpose
            raise RuntimeError('TextMelLoader.get_mel on raw audio needs the GPU and cannot run inside a DataLoader worker: '
                               'use num_workers=0, or TextAudioLoader + TextAudioCollate + DeviceBatches (what train.py does: '
                               'workers decode wav + text, the main process runs one batched STFT-mel per batch on the '
                               'device), or precomputed mels with load_mel_from_disk=True')
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


class TextAudioLoader(TextMelLoader):
    """Items are (text ids, raw audio[, speaker, emotions]): nothing here touches the GPU, so DataLoader
    workers are allowed; the mel is computed per batch on the device (DeviceBatches)."""

    def __getitem__(self, index):
        i = self.idx[index]
        path, text = self.audiopaths_and_text[i][0], self.audiopaths_and_text[i][1]
        audio = load_wav_to_torch(path, self.sampling_rate)
        assert float(audio.min()) >= -1 and float(audio.max()) <= 1        # layers.py:107-108
        item = (self.get_text(text), audio)
        if self.vesus:
            item += (self.speakers[i], self.emotions[i])
        return item


class TextAudioCollate:
    """TextMelCollate's ordering and padding (data_utils.py:88-131) applied to raw audio: sort by text
    length (descending), zero-pad text and audio.  Returns
    (text_padded, input_lengths, audio_padded, audio_lengths, speaker_ids, emotions)."""

    def __call__(self, batch):
        lengths = torch.LongTensor([len(x[0]) for x in batch])
        input_lengths, order = torch.sort(lengths, dim=0, descending=True)
        B = len(batch)
        text_padded = torch.zeros(B, int(input_lengths[0]), dtype=torch.long)
        audio_lengths = torch.zeros(B, dtype=torch.long)
        audio_padded = torch.zeros(B, max(x[1].numel() for x in batch))
        speaker_ids = torch.zeros(B)
        emotions = torch.zeros(B, 5)
        for i, src in enumerate(order.tolist()):
            item = batch[src]
            text, audio = item[0], item[1]
            text_padded[i, :text.size(0)] = text
            audio_padded[i, :audio.numel()] = audio
            audio_lengths[i] = audio.numel()
            if len(item) == 4:
                speaker_ids[i] = item[2]
                emotions[i] = item[3]
        return text_padded, input_lengths, audio_padded, audio_lengths, speaker_ids, emotions


class DeviceBatches:
    """Wraps a DataLoader of TextAudioCollate batches and yields the 7-tuple `Tacotron2.parse_batch`
    expects, with text / mel / gate already on the device and both length vectors on the host (so the
    step never synchronises on them).  Batch i+1 is copied (pinned, non_blocking) and its mel computed
    on a side stream while batch i trains."""

    def __init__(self, loader, stft, device=None):
        self.loader, self.stft = loader, stft
        self.device = device or torch.device('cuda', torch.cuda.current_device())
        self.stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, host_batch):
        text, in_len, audio, audio_len, speakers, emotions = host_batch
        hop = self.stft.hop_length
        assert int(audio_len.min()) > self.stft.filter_length // 2, 'clip shorter than the STFT reflect padding'
        pin = lambda t: t if t.is_pinned() else t.pin_memory()
        with torch.cuda.stream(self.stream):
            dev = lambda t: pin(t).to(self.device, non_blocking=True)
            text_d, audio_d, len_d = dev(text), dev(audio), dev(audio_len)
            spk_d, emo_d = dev(speakers), dev(emotions)
            mel, gate, _ = self.stft.mel_spectrogram_batch(audio_d, len_d)
            ready = torch.cuda.Event()
            ready.record(self.stream)       # end of THIS batch's staging (the next batch is staged behind it)
        out_len = audio_len // hop + 1
        return (text_d, in_len, mel, gate, spk_d, emo_d, out_len), ready

    def __iter__(self):
        staged = None
        for host_batch in self.loader:
            nxt = self._stage(host_batch)
            if staged is not None:
                yield self._release(staged)
            staged = nxt
        if staged is not None:
            yield self._release(staged)

    def _release(self, staged):
        batch, ready = staged
        main = torch.cuda.current_stream(self.device)
        # wait for this batch's own event, not for the side stream: batch i+1 is already being copied and transformed
        # there, and a wait_stream() would put its H2D copy + STFT on the critical path of step i
        main.wait_event(ready)
        for t in batch:
            if t.is_cuda:
                t.record_stream(main)
        return batch
