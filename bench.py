#!/usr/bin/env python3
"""Headline benchmark: mel-frames/sec over a (generator step + discriminator step) pair, vanilla
GANtron (use_labels=False, use_noise=True), LJSpeech-shaped synthetic batch, fp32.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pair: the body of the reference's generator step (train.py:361-423: parse_batch,
G forward, Tacotron2Loss incl. guided attention, adversarial loss through the frozen D, backward,
clip, Adam) followed by the body of its discriminator step (train.py:302-360) on the same batch
with the cached generator output.  Inputs are resident on the device before the timed region.
Multi-GPU: one process per GPU, each with its own batch (weak scaling), flat-buffer gradient
all-reduce over RCCL after every backward.  Rank 0 prints ONE JSON line.

    python bench.py --bf16        # BASELINE config 3 (hparams.fp16_run), a separate, explicitly labelled line
    python bench.py --vesus       # BASELINE config 4: VESUS labels + speakers + noise (E = 1093), 50/50 LJ / VESUS-shaped mix
    python bench.py --merged      # BASELINE config 5: merged-filelist length mix, batch 128, bf16 (implies both)
    python bench.py --batch 128   # any of the above at another per-GPU batch size
"""
import argparse
import ctypes
import json
import os
import sys
import time

# The training step drives three HIP streams of its own (main, auxiliary decoder chain, weight-gradient leaf)
# and RCCL adds more: with the runtime's default of 4 hardware queues per process they get time-sliced
# (measured +7 % per step as soon as RCCL is initialised).  Must be set before the HIP runtime starts.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HPARAMS = 'use_labels=False,use_noise=True'
VESUS_HPARAMS = 'use_labels=True,use_noise=True,vesus_path=/synthetic/vesus/'      # E = 512 + 64 + 5 + 512 = 1093
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (the headline figures with 2:1 sparsity are not used)
HBM_PEAK_GBS = 8000.0
# ('attn_context_bwd' = the backward attention step: since round 4 ONE launch per decoder step with the attention LSTM's cell backward
# inside, attn_bwd_step_kernel; 'attn_energies_fwd' = the one-launch forward attention step)
FAMILIES = ['rec_gemm', 'sgemm', 'attn_energies_fwd', 'attn_context_fwd', 'attn_context_bwd', 'attn_energies_bwd',
            'lstm_cell_fwd', 'lstm_cell_bwd', 'rec_cell', 'plstm']
NF = len(FAMILIES)                 # GT_PROF_FAMILIES of include/gantron_hip.h
# kernel symbol of each family as rocprofv3 / profiles/rNN_pmc_traffic.json name it
KERNEL_OF = {'rec_gemm': 'rec_gemm_kernel', 'sgemm': 'sgemm_kernel', 'attn_energies_fwd': 'attn_step2_fwd',
             'attn_context_fwd': 'attn_context_fwd_kernel', 'attn_context_bwd': 'attn_bwd_step_kernel',
             'attn_energies_bwd': 'attn_energies_bwd', 'lstm_cell_fwd': 'lstm_cell_fwd_kernel',
             'lstm_cell_bwd': 'lstm_cell_bwd_kernel', 'rec_cell': 'rec_cell_kernel', 'plstm': 'plstm_fwd_kernel'}
MFMA_BOUND = {'rec_gemm', 'sgemm', 'rec_cell', 'plstm'}
# Share of the chip a family's launches hold while they run.  Every loop kernel is launched to fill the chip except the
# persistent recurrences (plstm: H / 16 workgroups, one per CU, for a whole block of steps -- 64 of the 256 CUs for the decoder
# LSTM, 32 for the encoder BiLSTM), which run BESIDE the attention chain by design: a millisecond of theirs is a quarter of a
# millisecond of chip time, and their matrix-pipe ceiling is that of the CUs they hold.
CHIP_SHARE = {'plstm': 64 / 256}
PROF_STRIDE = 61              # prime: the families alternate between two or three shapes per decoder step, an even stride samples one of them


def _lj_lengths(rs, n):
    ti = np.rint(np.clip(rs.normal(100, 33, n), 12, 187)).astype(int)
    to = np.clip(np.rint(5.66 * ti + rs.normal(0, 25, n)), 96, 870).astype(int)
    return ti, to


def _vesus_lengths(rs, n):
    ti = np.rint(np.clip(rs.normal(15.7, 7, n), 3, 38)).astype(int)
    to = np.clip(np.rint(7 * ti + rs.normal(40, 15, n)), 40, 400).astype(int)
    return ti, to


def _short_lengths(rs, n):
    """CREMA-D / RAVDESS-shaped items of the merged mix (config 5): 2-4 s clips with one short sentence."""
    ti = np.rint(np.clip(rs.normal(25, 8, n), 8, 50)).astype(int)
    to = np.clip(np.rint(rs.normal(255, 45, n)), 150, 400).astype(int)
    return ti, to


def _assemble(hp, rs, ti, to, spk, emo):
    order = np.argsort(-ti, kind='stable')
    ti, to, spk, emo = ti[order], to[order], spk[order], emo[order]
    B, Ti, To = len(ti), int(ti.max()), int(to.max())
    text = torch.zeros(B, Ti, dtype=torch.long)
    mel = torch.zeros(B, hp.n_mel_channels, To)
    gate = torch.zeros(B, To)
    for b in range(B):
        text[b, :ti[b]] = torch.from_numpy(rs.integers(1, hp.n_symbols, ti[b]))
        m = np.clip(rs.normal(-5.5, 2.2, (hp.n_mel_channels, to[b])), -11.52, 2.0)
        mel[b, :, :to[b]] = torch.from_numpy(m).float()
        gate[b, to[b] - 1:] = 1
    return (text, torch.from_numpy(ti).long(), mel, gate, torch.from_numpy(spk).float(), torch.from_numpy(emo).float(),
            torch.from_numpy(to).long())


def ljspeech_shaped_batch(hp, B, seed, fixed=False):
    """Ti ~ round(clip(N(100,33),12,187)), To = clip(round(5.66 Ti + N(0,25)),96,870),
    sorted by Ti descending; ids uniform [1,147]; mel ~ N(-5.5,2.2) clipped to [-11.52, 2]."""
    rs = np.random.default_rng(seed)
    if fixed:
        ti, to = np.full(B, 150), np.full(B, 800)
    else:
        ti, to = _lj_lengths(rs, B)
    return _assemble(hp, rs, ti, to, np.zeros(B), np.zeros((B, 5)))


def vesus_shaped_batch(hp, B, seed):
    """50/50 mix of LJSpeech-shaped items (speaker 0, labels 0 -- what the reference's loader
    gives LJ items, data_utils.py:27-28) and VESUS-shaped items (Ti ~ clip(N(15.7,7),3,38), To = clip(round(7 Ti +
    N(40,15)),40,400), speaker 1..10, one-hot 'intended' label, utils.py:118-119)."""
    rs = np.random.default_rng(seed)
    n_lj = B // 2
    t1, o1 = _lj_lengths(rs, n_lj)
    t2, o2 = _vesus_lengths(rs, B - n_lj)
    spk = np.concatenate([np.zeros(n_lj), rs.integers(1, 11, B - n_lj)])
    emo = np.concatenate([np.zeros((n_lj, 5)), np.eye(5)[rs.integers(0, 5, B - n_lj)]])
    return _assemble(hp, rs, np.concatenate([t1, t2]), np.concatenate([o1, o2]), spk, emo)


def merged_shaped_batch(hp, B, seed):
    """Config 5 (LJSpeech + VESUS + CREMA-D + RAVDESS merged filelists).  The reference trainer only reads the LJ and the
    VESUS filelist (data_utils.py:21-30) and the CREMA-D / RAVDESS lists carry no transcripts, so this is a LENGTH mix
    defined here: 1/2 LJSpeech-shaped, 1/4 VESUS-shaped, 1/4 CREMA-D / RAVDESS-shaped short clips, all
    with the VESUS conditioning path (speaker ids up to 122 = the embedding's 123 rows, one-hot labels)."""
    rs = np.random.default_rng(seed)
    n1, n2 = B // 2, B // 4
    n3 = B - n1 - n2
    t1, o1 = _lj_lengths(rs, n1)
    t2, o2 = _vesus_lengths(rs, n2)
    t3, o3 = _short_lengths(rs, n3)
    spk = np.concatenate([np.zeros(n1), rs.integers(1, 11, n2), rs.integers(11, 123, n3)])
    emo = np.concatenate([np.zeros((n1, 5)), np.eye(5)[rs.integers(0, 5, n2)], np.eye(5)[rs.integers(0, 5, n3)]])
    return _assemble(hp, rs, np.concatenate([t1, t2, t3]), np.concatenate([o1, o2, o3]), spk, emo)


def pair_flops(hp, in_lens, out_lens):
    """Algorithmic FLOPs of one G+D pair on the padded shapes."""
    B, Ti, To = len(in_lens), max(in_lens), max(out_lens)
    E = hp.encoder_embedding_dim + (hp.noise_size if hp.use_noise else 0)
    if hp.vesus_path:
        E += hp.speakers_embedding + (5 if hp.use_labels else 0)
    enc = 3932160 * B * Ti + 1572864 * sum(in_lens)
    # decoder MACs per (item, step): two gate GEMMs + projection + query + per-position attention terms (SURVEY 8d:
    # 22 317 056 + 7 232 Ti at E = 1024, 22 888 392 + 7 301 Ti at E = 1093)
    per_step = {1024: 22317056, 1093: 22888392}.get(E, 4096 * (256 + E + 1024) + 4096 * (1024 + E + 1024) + 81 * (1024 + E) + 128 * 1024)
    dec = B * To * (per_step + (1984 + 4096 + 128 + E) * Ti) + 128 * E * B * Ti + 86016 * B * (To + 1)
    post = 4341760 * B * To
    W = To // 20 + (1 if To % 20 else 0)
    disc = 12329040 * B * W
    g = 2 * (3 * (enc + dec + post) + 2 * disc)
    d = 2 * (3 * 2 * disc)
    return g + d


def host_threads():
    """CPU threads this process may really use: the affinity mask, capped at the 16-core share a
    one-GPU box grants (os.cpu_count() reports the whole host and would oversubscribe)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def cpu_sample_batch(hp, B=8):
    """A bounded LJSpeech-shaped sample (<= 200 frames / 50 characters per item): tests/golden/time_reference.py times
    the real reference against the oracle on it in the build container (the oracle must not be slower than ~1.1x)."""
    text, il, mel, gate, _, _, ol = ljspeech_shaped_batch(hp, B, seed=99)
    ol = torch.clamp(ol, max=200)
    il = torch.clamp(il, max=50)
    text, mel, gate = text[:, :int(il.max())], mel[:, :, :int(ol.max())].contiguous(), gate[:, :int(ol.max())].contiguous()
    for b in range(B):
        gate[b, int(ol[b]) - 1:] = 1
        mel[b, :, int(ol[b]):] = 0
        text[b, int(il[b]):] = 0
    return text.contiguous(), il, mel, gate, ol


def cpu_baseline(seconds_budget=45.0):
    """The CPU oracle (oracle/gantron_ref.py, a restatement of the reference pinned by its golden vectors) timed on the
    host cores for G+D pairs of BASELINE config 1's shape: vanilla GANtron, an LJSpeech-shaped batch of 8, UNCLIPPED
    (Ti up to 187 characters, To up to 870 frames), fp32.  One warm-up pair, then as many timed pairs as fit the budget
    (at least one); the sample string says how many."""
    from gantron_amd.hparams import HParams
    from oracle import gantron_ref as R
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import common
    hp = HParams(HPARAMS)
    B = 8
    text, il, mel, gate, _, _, ol = ljspeech_shaped_batch(hp, B, seed=99)
    Ti, To = text.shape[1], mel.shape[2]
    torch.set_num_threads(host_threads())
    Gp, Dp = R.init_params(hp, seed=1)
    stream = common.MaskStream(5)
    rng = common.draw_generator_rng(stream, hp, B, Ti, To)
    rng['disc_keep'] = common.draw_disc_keep(stream, hp, B, To)
    kr, kf = common.draw_disc_keep(stream, hp, B, To), common.draw_disc_keep(stream, hp, B, To)

    def pair():
        out, _, grads, _ = R.generator_step_grads(Gp, Dp, (text, il, mel, gate, ol), rng, hp)
        R.clip_grad_norm(grads, 1.0)
        R.discriminator_step_grads(Dp, mel, ol, out[1], ol, kr, kf, hp)
    t0 = time.perf_counter()
    pair()
    first = time.perf_counter() - t0
    log(f'cpu baseline: warm-up pair {first:.2f} s on {torch.get_num_threads()} threads')
    reps = max(1, min(3, int((seconds_budget - first) / max(first, 1e-3))))
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        pair()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    frames = int(ol.sum())
    return dict(value=frames / t, unit='mel-frames/s', cores=torch.get_num_threads(), kind='port',
                sample=f'BASELINE config 1 shape: oracle/gantron_ref.py G+D pair (incl. clip), vanilla, fp32, LJSpeech-shaped '
                       f'B={B} unclipped (Ti<={Ti}, To<={To}, {frames} valid frames); 1 warm-up + {len(times)} timed pair(s), '
                       f'median {t:.2f} s per pair')


def data_path_figures(hp, dev, out_lens, n_batches=6):
    """SURVEY 8(a) S1 / 8(f)-1 side figures, OUTSIDE the timed region: can the data path feed the step?
    * stft: TacotronSTFT.mel_spectrogram_batch (reflect pad, windowed DFT as a GEMM over overlapping frames, magnitude, mel
      filterbank GEMM, log, collate; reference layers.py:97-114, stft.py:77-105) on 64 clips whose lengths give exactly the
      bench batch's frame counts -- valid frames / s, and the algorithmic TFLOP/s of its two GEMMs;
    * loader: gantron_amd.data_utils.DeviceBatches over an in-memory stream of TextAudioCollate-shaped host batches (pinned
      H2D of 57 MB of raw audio per batch on a side stream + the batched STFT-mel) -- batches / s with nothing else running."""
    from gantron_amd.data_utils import DeviceBatches
    from gantron_amd.stft import TacotronSTFT
    stft = TacotronSTFT(hp.filter_length, hp.hop_length, hp.win_length, hp.n_mel_channels, hp.sampling_rate, hp.mel_fmin,
                        hp.mel_fmax).to(dev)
    hop, N, nm = hp.hop_length, hp.filter_length, hp.n_mel_channels
    B = len(out_lens)
    rs = np.random.default_rng(7)
    alen = torch.tensor([(f - 1) * hop + hop // 2 for f in out_lens], dtype=torch.int64)       # alen // hop + 1 == frames
    audio = torch.zeros(B, int(alen.max()))
    for b in range(B):
        audio[b, :alen[b]] = torch.from_numpy(rs.uniform(-0.5, 0.5, int(alen[b])).astype(np.float32))
    audio_d, alen_d = audio.to(dev), alen.to(dev)
    F = audio.shape[1] // hop + 1
    for _ in range(2):
        stft.mel_spectrogram_batch(audio_d, alen_d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        stft.mel_spectrogram_batch(audio_d, alen_d)
    e1.record()
    torch.cuda.synchronize()
    stft_ms = e0.elapsed_time(e1) / reps
    flops = B * F * (2.0 * (2 * (N // 2 + 1)) * N + 2.0 * (N // 2 + 1) * nm)
    text = torch.zeros(B, 187, dtype=torch.long)
    host = [(text, torch.full((B,), 187, dtype=torch.long), audio.pin_memory(), alen, torch.zeros(B), torch.zeros(B, 5))
            for _ in range(n_batches)]
    for _ in DeviceBatches(host[:2], stft, dev):
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in DeviceBatches(host, stft, dev):
        pass
    torch.cuda.synchronize()
    loader_s = (time.perf_counter() - t0) / n_batches
    return dict(stft_frames_per_s=sum(out_lens) / (stft_ms * 1e-3), stft_ms_per_batch=stft_ms, stft_tflops=flops / (stft_ms * 1e-3) / 1e12,
                stft_padded_frames=B * F, loader_batches_per_s=1.0 / loader_s, loader_ms_per_batch=loader_s * 1e3,
                note='outside the timed region: batched STFT-mel of 64 clips with the bench batch\'s frame counts; DeviceBatches '
                     'over in-memory raw-audio host batches (pinned H2D + STFT-mel on a side stream), nothing else running')


def self_launch(world):
    """`python bench.py --gpus N` without a launcher: one child process per GPU with the torchrun environment
    (gantron_amd.multiproc.worker_env: RANK / LOCAL_RANK / WORLD_SIZE, MASTER_ADDR=127.0.0.1, a free port, dmabuf IPC),
    the same command line, supervised (a failing rank stops the others instead of leaving them in a collective).  Rank 0
    inherits stdout, so its one JSON line is this command's; every rank shares stderr.  The counterpart of the
    reference's multiproc.py:13-23, which starts `train.py --rank=i` once per GPU."""
    import subprocess
    from gantron_amd.multiproc import _free_port, supervise, worker_env
    port = _free_port()
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for rank in range(world):
        procs.append(subprocess.Popen(cmd, env=worker_env(rank, world, port),
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    try:
        return supervise(procs, poll_s=0.2)
    except KeyboardInterrupt:
        for p in procs:
            p.terminate()
        return 130


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--fixed-shape', action='store_true', help='nominal Ti=150, To=800 for every item')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-data-path', action='store_true', help='skip the STFT-mel / loader side figures (config.data_path)')
    ap.add_argument('--bf16', action='store_true',
                    help='BASELINE config 3 (hparams.fp16_run): bf16 operands in the decoder LSTM products and the layer '
                         'GEMMs, fp32 accumulation and master weights; NOT the headline metric, the default line is fp32')
    ap.add_argument('--fp32-split', type=int, choices=(0, 2, 3, 4), default=None,
                    help='how the tiled fp32 GEMMs are computed (gt_sgemm_set_fp32_split): 3 = three bf16 planes per operand, six '
                         'exact products, fp32 accumulation (default, GANTRON_FP32_SPLIT); 0 = the fp32 matrix pipe; 2 = two '
                         'planes, 2^-16 accuracy class')
    ap.add_argument('--vesus', action='store_true',
                    help='BASELINE config 4: VESUS labels + speaker embedding + noise (E = 1093), 50/50 LJ / VESUS-shaped mix')
    ap.add_argument('--merged', action='store_true',
                    help='BASELINE config 5: merged-filelist length mix with the VESUS conditioning; implies --batch 128 --bf16 '
                         'unless given otherwise')
    args = ap.parse_args()
    if args.merged:
        args.vesus = True
        if '--batch' not in sys.argv:
            args.batch = 128
        args.bf16 = True

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # the plain command (`python bench.py --gpus N`, no launcher): start the N ranks from here.  Nothing in this
        # process has touched the GPU yet (importing torch does not), and it never will: it only supervises.
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; start it as `python bench.py --gpus N` '
                 f'(it launches its own ranks) or under torch.distributed.run with --nproc-per-node equal to --gpus')
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP extension has no CPU fallback)'
    # rehearsal on a 1-GPU box (tests/test_training_gpu.py): every rank on device 0 and gloo carrying the CUDA
    # tensors, so that the N > 1 control flow of this script runs before the driver's multi-GPU launch does
    dev_index = int(os.environ.get('GANTRON_BENCH_DEVICE', local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)

    import torch.distributed as dist
    from gantron_amd import _lib
    from gantron_amd.distributed import apply_gradient_allreduce
    from gantron_amd.hparams import HParams
    from gantron_amd.loss_function import Tacotron2Loss
    from gantron_amd.model import Discriminator, Tacotron2
    from gantron_amd.optim import FlatAdam
    from gantron_amd.train_steps import discriminator_step, generator_step

    hp = HParams((VESUS_HPARAMS if args.vesus else HPARAMS) + f',batch_size={args.batch}' + (',fp16_run=True' if args.bf16 else ''))
    from gantron_amd import ops as _ops
    if args.fp32_split is not None:
        _ops.FP32_SPLIT = args.fp32_split
    GEMM_ARITH = {0: 'fp32 MFMA (v_mfma_f32_32x32x2_f32)',
                  3: 'fp32 operands split exactly into three bf16 planes in the tile loader, six bf16 MFMA products (each exact), '
                     'fp32 accumulation: dropped terms <= 2^-24 |a||b|, measured error vs fp64 <= the fp32 MFMA\'s in every operand '
                     'mode (tests/test_kernels_gpu.py::test_sgemm_split_planes_accuracy, tools/split_accuracy.py)',
                  2: 'fp32 operands split into two bf16 planes, three bf16 MFMA products, fp32 accumulation: dropped terms '
                     '<= 2^-16 |a||b|',
                  4: 'fp32 operands scaled by a power of two per operand tensor (max |x| -> [2^14, 2^15), found by one pass in front '
                     'of the product) and split into two fp16 planes in the tile loader (22 of the 24 significand bits; exact '
                     'residuals), three fp16 MFMA products (each exact), fp32 accumulation, scales taken out exactly in the epilogue: '
                     'dropped terms <= 2^-22 |a||b| per product, measured error vs fp64 <= the fp32 MFMA\'s in every operand mode '
                     '(tests/test_kernels_gpu.py::test_sgemm_split_planes_accuracy, tools/split_accuracy.py)'}[_ops.FP32_SPLIT]
    # GANTRON_FORCE_DIST=1 exercises the RCCL gradient-averaging path on a single rank (1-GPU boxes)
    use_dist = world > 1 or os.environ.get('GANTRON_FORCE_DIST') == '1'
    if use_dist:
        hp.distributed_run = True
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        backend = os.environ.get('GANTRON_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', init_method='env://', world_size=world, rank=rank, device_id=dev)
        else:
            hp.dist_backend = backend
            dist.init_process_group(backend, init_method='env://', world_size=world, rank=rank)
    torch.manual_seed(hp.seed)
    G, D = Tacotron2(hp).to(dev).train(), Discriminator(hp).to(dev).train()
    g_opt = FlatAdam(G, lr=hp.g_learning_rate, weight_decay=hp.weight_decay)
    d_opt = FlatAdam(D, lr=hp.d_learning_rate, weight_decay=hp.weight_decay)
    if use_dist:
        G, D = apply_gradient_allreduce(G), apply_gradient_allreduce(D)
    crit = Tacotron2Loss()
    if args.merged:
        batch = merged_shaped_batch(hp, args.batch, seed=1234 + rank)
    elif args.vesus:
        batch = vesus_shaped_batch(hp, args.batch, seed=1234 + rank)
    else:
        batch = ljspeech_shaped_batch(hp, args.batch, seed=1234 + rank, fixed=args.fixed_shape)
    # lengths stay on the host, as they are when a DataLoader produces the batch (no per-step D2H sync)
    batch = tuple(t if i in (1, 6) else t.to(dev) for i, t in enumerate(batch))
    in_lens, out_lens = batch[1].tolist(), batch[6].tolist()
    frames = sum(out_lens)
    _lib.check(_lib.lib().gt_prof_set_context_rows(sum(in_lens)), 'gt_prof_set_context_rows')

    # GANTRON_EXPERIMENTAL=graph (hipGraph replay of the decoder loops) cannot capture on the legacy default stream: the step then
    # runs on a stream of its own
    side = torch.cuda.Stream() if 'graph' in os.environ.get('GANTRON_EXPERIMENTAL', '') else None

    def on_step_stream(fn):
        if side is None:
            return fn()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out = fn()
        torch.cuda.current_stream().wait_stream(side)
        return out

    def pair(it):
        def body():
            _, generated = generator_step(G, D, g_opt, crit, batch, hp, iteration=it)
            discriminator_step(G, D, d_opt, batch, generated, hp)
        on_step_stream(body)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    log(f'setup done: B={args.batch} Ti={max(in_lens)} To={max(out_lens)} frames={frames}')
    for i in range(args.warmup):
        pair(i)
        torch.cuda.synchronize()
        log(f'warmup pair {i} done')
    sync()
    # roofline leg: every PROF_STRIDE-th launch of each kernel family is bracketed by two HIP events on its
    # launch stream DURING the timed region (a sparse sample keeps the perturbation below 1 %)
    lib = _lib.lib()
    null_us = ctypes.c_double(0.0)
    _lib.check(lib.gt_prof_calibrate(200, ctypes.byref(null_us), torch.cuda.current_stream().cuda_stream),
               'gt_prof_calibrate')
    sync()
    lib.gt_prof_begin_sampled(PROF_STRIDE)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]       # one event per pair: spread of the pairs
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        pair(args.warmup + i)
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    pair_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    log('pairs (ms, by their own events): ' + ' '.join(f'{x:.1f}' for x in pair_ms))
    prof_out = (ctypes.c_double * (7 * NF + 2))()          # GT_PROF_DOUBLES
    _lib.check(lib.gt_prof_end(prof_out), 'gt_prof_end')
    log(f'timed region: {elapsed:.3f} s for {args.steps} pairs')
    from gantron_amd import engine as _engine
    _engine.check_handoffs()          # no in-launch hand-off of the loop kernels gave up during the timed region
    if world > 1:
        t = torch.tensor([elapsed, float(frames)], device=dev, dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank = [torch.zeros_like(t) for _ in range(world)]          # every rank's own clock: a straggler shows here
        dist.all_gather(per_rank, torch.tensor([elapsed, float(frames)], device=dev, dtype=torch.float64))
        rank_ms = [float(x[0]) / args.steps * 1e3 for x in per_rank]
        rank_frames = [int(x[1]) for x in per_rank]
        elapsed, total_frames = float(tmax[0]), float(t[1])
    else:
        total_frames = float(frames)
        rank_ms, rank_frames = [elapsed / args.steps * 1e3], [frames]

    # ---- SURVEY 8(d) side figures (outside the timed region): G step and D step alone, for the
    # schedule-faithful mix of g_freq=2 generator steps per d_freq=1 discriminator step (hparams.py:95-96)
    def alone(fn, n=2):
        sync()
        t1 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t1) / n
    gen_holder = {}

    def g_only():
        gen_holder['fake'] = on_step_stream(lambda: generator_step(G, D, g_opt, crit, batch, hp, iteration=args.warmup + args.steps)[1])
    g_s = alone(g_only)
    d_s = alone(lambda: on_step_stream(lambda: discriminator_step(G, D, d_opt, batch, gen_holder['fake'], hp)))
    if world > 1:
        t = torch.tensor([g_s, d_s], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        g_s, d_s = float(t[0]), float(t[1])

    # ---- data-parallel evidence: backend, world size, and whether every rank holds the SAME averaged generator gradient
    # after a step (sum and sum of squares of the flat gradient buffer, gathered from all ranks)
    dist_info = None
    if use_dist:
        fg = G._flat_grad.double()
        mine = torch.stack([fg.sum(), (fg * fg).sum()]).to(dev)
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        allc = torch.stack(allc).cpu()
        dist_info = dict(backend=dist.get_backend(), world_size=dist.get_world_size(),
                         ms_per_step_min=min(rank_ms), ms_per_step_max=max(rank_ms), ms_per_step_by_rank=[round(x, 3) for x in rank_ms],
                         valid_frames_by_rank=rank_frames,
                         flat_grad_mb=G._flat_grad.numel() * 4 / 1e6,
                         grad_checksum=[float(allc[0, 0]), float(allc[0, 1])],
                         grad_checksum_equal_on_all_ranks=bool((allc == allc[0:1]).all()))

    # ---- roofline of the dominant kernel family from the sampled brackets of the timed region
    roof = None
    cpu = None
    if rank == 0:
        out = prof_out
        # a bracket = kernel duration (dispatch begin to end, what rocprofv3 reports) + the latency of the two event
        # packets; that latency, in situ = bracket of an empty kernel on the same busy streams - the empty kernel's own
        # duration (gt_prof_calibrate: start / stop timestamps of its dispatch)
        null_bracket_ms = (out[5 * NF + 1] / out[5 * NF]) if out[5 * NF] > 0 else 0.0
        null_ms = max(null_bracket_ms - null_us.value * 1e-3, 0.0)
        # (the recurrent-product families are timed by their launches' own start / stop stamps -- out[5 NF + 2 + f] of
        # them -- which carry no event latency: hipExtLaunchKernelGGL, the duration rocprofv3 reports for the kernel)
        fam = [dict(name=FAMILIES[f], launches=int(out[4 * f]), exact=int(out[5 * NF + 2 + f]),
                    ms=max(out[4 * f + 1] - null_ms * (out[4 * f] - out[5 * NF + 2 + f]), 1e-9) if out[4 * f] else 0.0,
                    flops=out[4 * f + 2], bytes=out[4 * f + 3], seen=int(out[4 * NF + f])) for f in range(NF)]
        for r in fam:                            # estimated time of ALL launches of the family in the timed region
            r['est_total_ms'] = r['ms'] / r['launches'] * r['seen'] if r['launches'] else 0.0
        # the dominant KERNEL: every family but sgemm is one kernel symbol; the sgemm family is several template
        # instantiations (tile size x operand modes) none of which comes close to rec_gemm_kernel in rocprofv3's
        # per-symbol totals (profiles/), so it is listed under `families` but does not compete here
        # dominance = chip time: a family's total duration x the share of the chip its launches hold (CHIP_SHARE)
        dom = max((r for r in fam if r['name'] != 'sgemm'), key=lambda r: r['est_total_ms'] * CHIP_SHARE.get(r['name'], 1.0))
        avg_ms = dom['ms'] / max(dom['launches'], 1)
        if dom['name'] in MFMA_BOUND:
            ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12 if dom['ms'] > 0 else 0.0
            peak = BF16_MFMA_PEAK_TFLOPS if args.bf16 else FP32_MFMA_PEAK_TFLOPS     # --bf16: these families run bf16 MFMAs
            peak *= CHIP_SHARE.get(dom['name'], 1.0)
            roof = dict(bound='mfma', achieved=ach, peak=peak, unit='TFLOP/s', frac=ach / peak, traffic=None)
        else:
            ach = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9 if dom['ms'] > 0 else 0.0
            roof = dict(bound='hbm', achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s', frac=ach / HBM_PEAK_GBS, traffic=None)
        # HBM traffic per launch from the committed PMC passes of this same command (tools/pmc_traffic.py);
        # PMC counters cannot be read from inside the process, so the figure is the latest profiled one
        try:
            import glob
            pmc_path = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))[-1]
            pmc = json.load(open(pmc_path))
            roof['traffic'] = pmc['kernels'][KERNEL_OF[dom['name']]]['hbm_bytes_per_launch']
            roof['traffic_source'] = ('NOT measured in this run (PMC counters cannot be read in-process): copied from the committed '
                                      f'{os.path.relpath(pmc_path, ROOT)} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of this '
                                      'same command, tools/collect_profiles.sh), bytes per launch')
            roof['algorithmic_bytes_per_launch'] = dom['bytes'] / max(dom['launches'], 1)
        except (IndexError, KeyError, OSError, ValueError):
            pass
        # the same kernel in the committed rocprofv3 --kernel-trace --stats summary of this command (profiles/): under the profiler
        # the weight-gradient GEMMs do not run beside the loops, so that average is the kernel's UNCONTENDED duration
        try:
            import csv
            import glob
            stats_path = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_kernel_stats.csv')))[-1]
            calls, total_ns = 0, 0.0
            for row in csv.DictReader(open(stats_path)):
                if KERNEL_OF[dom['name']] in row['Name'] and 'bf16' not in row['Name']:
                    calls += int(row['Calls'])
                    total_ns += float(row['TotalDurationNs'])
            if calls:
                prof_us = total_ns / calls / 1e3
                roof['rocprofv3'] = dict(avg_launch_us=prof_us, calls=calls, source=os.path.relpath(stats_path, ROOT),
                                         frac_at_that_duration=(dom['flops'] / max(dom['launches'], 1)) / (prof_us * 1e-6) / 1e12 / roof['peak']
                                         if dom['name'] in MFMA_BOUND else
                                         (dom['bytes'] / max(dom['launches'], 1)) / (prof_us * 1e-6) / 1e9 / roof['peak'],
                                         note='committed kernel-trace summary of this command; under the profiler the weight-gradient GEMMs do '
                                              'not overlap the decoder loops, so this is the uncontended duration -- avg_launch_us above is '
                                              'measured live, inside the timed region, beside them')
        except (IndexError, KeyError, OSError, ValueError):
            pass
        roof.update(kernel=KERNEL_OF[dom['name']] + ('' if not args.bf16 else ' (bf16-operand instantiation)'), sampled_launches=dom['launches'], sample_stride=PROF_STRIDE,
                    avg_launch_us=avg_ms * 1e3,
                    timing=('start / stop events of the sampled launches themselves (hipExtLaunchKernelGGL: dispatch begin to end)'
                            if dom['exact'] == dom['launches'] else 'event brackets on the launch stream minus the in-situ event latency'),
                    event_latency_us=null_ms * 1e3, empty_kernel_bracket_us=null_bracket_ms * 1e3, empty_kernel_us=null_us.value,
                    families={r['name']: dict(launches=r['launches'], ms=round(r['ms'], 3), all_launches=r['seen'],
                                              est_ms_per_pair=round(r['est_total_ms'] / args.steps, 2),
                                              tflops=round(r['flops'] / max(r['ms'], 1e-9) / 1e9, 2),
                                              gbs=round(r['bytes'] / max(r['ms'], 1e-9) / 1e6, 1)) for r in fam})
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        # tile GEMMs run on the bf16 matrix pipe (three planes: 6 of the 16 bf16 MFMAs an fp32 product would cost, i.e. a
        # ceiling of 2500 / 6 TFLOP/s of fp32 work), the recurrent / attention products on the fp32 pipe: the pair is bound
        # by the SUM of the two pipes' times, not by all of its flops on the fp32 peak
        sg = next(r for r in fam if r['name'] == 'sgemm')
        gemm_flops_pair = out[6 * NF + 2 + FAMILIES.index('sgemm')] / args.steps          # every launch counted, not the sample
        gemm_peak = (BF16_MFMA_PEAK_TFLOPS if args.bf16 else
                     {0: FP32_MFMA_PEAK_TFLOPS, 3: BF16_MFMA_PEAK_TFLOPS / 6, 2: BF16_MFMA_PEAK_TFLOPS / 3,
                      4: BF16_MFMA_PEAK_TFLOPS / 3}[_ops.FP32_SPLIT])
        if roof['families']['plstm']['launches']:
            pl_peak = FP32_MFMA_PEAK_TFLOPS * CHIP_SHARE['plstm']
            roof['families']['plstm'].update(peak=round(pl_peak, 1), frac=round(roof['families']['plstm']['tflops'] / pl_peak, 3),
                                             chip_share=CHIP_SHARE['plstm'],
                                             note='persistent recurrences: 64 of the 256 CUs (32 for the encoder BiLSTM) for a whole block '
                                                  'of steps, beside the attention chain; peak = the fp32 matrix pipe of the CUs held')
        roof['families']['sgemm'].update(peak=round(gemm_peak, 1), frac=round(roof['families']['sgemm']['tflops'] / gemm_peak, 3),
                                         pipe=('fp16 MFMA' if (_ops.FP32_SPLIT == 4 and not args.bf16) else 'bf16 MFMA') if (args.bf16 or _ops.FP32_SPLIT) else 'fp32 MFMA',
                                         est_tflop_per_pair=round(gemm_flops_pair / 1e12, 3))
    if use_dist:
        dist.barrier()

    if rank == 0:
        Ti, To = max(in_lens), max(out_lens)
        ms = elapsed / args.steps * 1e3
        fl = pair_flops(hp, in_lens, out_lens)
        rec_peak = BF16_MFMA_PEAK_TFLOPS if args.bf16 else FP32_MFMA_PEAK_TFLOPS
        gemm_fl = min(gemm_flops_pair, fl)
        bound_s = gemm_fl / (gemm_peak * 1e12) + (fl - gemm_fl) / (rec_peak * 1e12)
        data_path = None
        if world == 1 and not args.no_data_path and not args.vesus:
            try:
                data_path = data_path_figures(hp, dev, out_lens)
                data_path['feeds_step'] = data_path['loader_ms_per_batch'] < ms
            except Exception as e:                                   # a side figure must never cost the headline line
                data_path = dict(error=repr(e))
        import hashlib
        line = dict(metric='mel-frames/sec (gen+disc step) LJSpeech bs=64', value=total_frames * args.steps / elapsed,
                    unit='mel-frames/s', n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms,
                    higher_is_better=True, scaling='weak', vs_baseline=None,
                    dtype='bf16 operands (LSTM products, layer GEMMs) / f32 accumulation and elsewhere' if args.bf16 else 'f32', data='synthetic',
                    config=dict(workload=('merged-filelist length mix (LJ + VESUS + CREMA-D/RAVDESS-shaped), VESUS conditioning E=1093, '
                                          if args.merged else 'VESUS labels+noise (E=1093), 50/50 LJSpeech/VESUS-shaped mix, ' if args.vesus
                                          else 'Vanilla GANtron (use_labels=False,use_noise=True) LJSpeech-shaped ') +
                                         'batch=%d %s, G step + D step pair incl. clip+Adam' % (args.batch, 'bf16' if args.bf16 else 'fp32'),
                                per_gpu_batch=args.batch, global_batch=args.batch * world, Ti_max=Ti, To_max=To,
                                valid_frames_per_rank=frames, padded_frames_per_rank=args.batch * To,
                                parallelism='dp%d' % world, pair_tflop=fl / 1e12,
                                tiled_gemm_arithmetic='bf16 operands, fp32 accumulation' if args.bf16 else GEMM_ARITH,
                                recurrent_product_arithmetic='bf16 operands, fp32 accumulation' if args.bf16 else
                                ('fp32 MFMA (v_mfma_f32_16x16x4_f32) in the persistent recurrences; the attention LSTM\'s forward step and backward product '
                                 '(bits 0 / 1 of engine.REC_H2 = %d) from two scaled fp16 planes per operand, three v_mfma_f32_16x16x32_f16, fp32 accumulation '
                                 '(22 significand bits per operand: tests/test_kernels_gpu.py::test_rec_gemm_h2)' % _engine.REC_H2
                                 if _engine.REC_H2 else 'fp32 MFMA (v_mfma_f32_16x16x4_f32)'),
                                padded_frames_per_s=args.batch * To * world * args.steps / elapsed,
                                pair_ms_min_median_max=[round(pair_ms[0], 2), round(pair_ms[len(pair_ms) // 2], 2), round(pair_ms[-1], 2)],
                                g_step_ms=g_s * 1e3, d_step_ms=d_s * 1e3,
                                schedule_2g1d_frames_per_s=2 * total_frames / (2 * g_s + d_s),
                                pair_mfma_frac=fl / (elapsed / args.steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                                pair_blended_frac=bound_s / (elapsed / args.steps),
                                pair_blended_note=f'{gemm_fl / 1e12:.2f} TFLOP of tile GEMMs on {gemm_peak:.0f} TFLOP/s + '
                                                  f'{(fl - gemm_fl) / 1e12:.2f} TFLOP of recurrent / attention products on '
                                                  f'{rec_peak:.1f} TFLOP/s = {bound_s * 1e3:.1f} ms of matrix-pipe time per pair',
                                data_path=data_path, cmd='python bench.py ' + ' '.join(sys.argv[1:]),
                                bench_py_sha16=hashlib.sha256(open(os.path.abspath(__file__), 'rb').read()).hexdigest()[:16]),
                    roofline=roof, cpu_baseline=cpu, dist=dist_info)
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
