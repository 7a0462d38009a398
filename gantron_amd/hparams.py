"""Hyper-parameter bag with the reference's names, defaults and override rules.

Mirror of `hparams.HParams` (reference hparams.py:7-149): plain attributes, a
`k=v,k=v` override string parsed with `ast.literal_eval` (values containing '/' stay
strings, unparsable values stay strings; hparams.py:115-128) and `add_params()` that
merges an argparse namespace / dict first and the `hparams` string last
(hparams.py:133-149).
"""
import argparse
import ast

from .text.symbols import symbols

_DEFAULTS = dict(
    version=0.6,
    # experiment
    epochs=100, iterations=None, iters_per_checkpoint=5000, seed=1234,
    dynamic_loss_scaling=True, fp16_run=False, distributed_run=False,
    dist_backend='nccl', dist_url='tcp://localhost:54321',
    cudnn_enabled=True, cudnn_benchmark=False,
    ignore_layers=['decoder.attention_rnn.weight_ih',
                   'decoder.attention_layer.memory_layer.linear_layer.weight',
                   'decoder.decoder_rnn.weight_ih',
                   'decoder.linear_projection.linear_layer.weight',
                   'decoder.gate_layer.linear_layer.weight'],
    attn_steps=5000, reduce_lr_steps_every=5e4, vesus_path=None, speakers_embedding=64,
    use_labels=True, use_noise=False, use_intended_labels=True,
    # data
    load_mel_from_disk=False,
    training_files=['filelists/ljs_audio_text_train_filelist.txt', 'filelists/vesus_train.txt'],
    validation_files=['filelists/ljs_audio_text_val_filelist.txt', 'filelists/vesus_val.txt'],
    text_cleaners=['english_cleaners'], n_labels=5,
    # audio
    max_wav_value=32768.0, sampling_rate=22050, filter_length=1024, hop_length=256,
    win_length=1024, n_ftt=1024, n_mel_channels=80, mel_fmin=0.0, mel_fmax=8000.0,
    # model
    n_symbols=len(symbols), symbols_embedding_dim=512,
    encoder_kernel_size=5, encoder_n_convolutions=3, encoder_embedding_dim=512,
    n_frames_per_step=1, decoder_rnn_dim=1024, prenet_dim=256, max_decoder_steps=500,
    gate_threshold=0.5, p_attention_dropout=0.1, p_decoder_dropout=0.1,
    attention_rnn_dim=1024, attention_dim=128,
    attention_location_n_filters=32, attention_location_kernel_size=31,
    postnet_embedding_dim=512, postnet_kernel_size=5, postnet_n_convolutions=5,
    # GAN
    discriminator_window=20, discriminator_dim=512, g_freq=2, d_freq=1,
    clipping_value=0.001, gradient_penalty_lambda=0, noise_size=512, disc_warmp_up=500,
    discriminator_type='conv', encoder_inputs=False,
    # optimisation
    use_saved_learning_rate=False, g_learning_rate=0.001, d_learning_rate=0.0007,
    weight_decay=1e-6, grad_clip_thresh=1.0, batch_size=32, mask_padding=True,
)


class HParams:
    def __init__(self, hparams_string=None):
        for key, value in _DEFAULTS.items():
            setattr(self, key, list(value) if isinstance(value, list) else value)
        if hparams_string:
            self.add_params_string(hparams_string)

    def add_param(self, param, value):
        setattr(self, param, value)

    def add_params_string(self, hparams_string):
        for item in hparams_string.split(','):
            key, value = item.split('=')
            if '/' not in value:
                try:
                    value = ast.literal_eval(value)
                except Exception:      # same catch-all as the reference: keep the raw string
                    pass
            self.add_param(key, value)

    def add_params(self, params):
        if type(params) is str and '=' in params:
            self.add_params_string(params)
            return
        if type(params) is argparse.Namespace:
            params = vars(params)
        override = None
        for key, value in params.items():
            if key == 'hparams':
                override = value
            elif value is not None:
                self.add_param(key, value)
        if override is not None:
            self.add_params_string(override)    # the --hparams string has the last word
