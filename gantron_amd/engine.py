"""Forward/backward orchestration of the generator and the discriminator over the HIP kernels.

This is the host side of the hot path: it sequences C-ABI calls (ops.py / include/gantron_hip.h),
owns no arithmetic, and keeps the reference's tensor semantics:

* generator forward = Tacotron2.forward (reference model.py:644-673): embedding -> encoder
  (3x conv5+BN+ReLU+dropout, packed BiLSTM) -> teacher-forced decoder loop with location-sensitive
  attention -> PostNet -> autograd-invisible padding mask;
* generator backward = what autograd does for that graph, hand-scheduled: reverse-time decoder
  loop (weight gradients deferred to large GEMMs over all steps), PostNet/encoder conv+BN backward;
* discriminator forward/backward = Discriminator.forward / adversarial_loss (model.py:514-540).

Parameters are passed as a dict {state_dict name: CUDA tensor} in PyTorch layouts; gradients are
accumulated into a dict of same-shaped tensors.  Stochastic inputs (dropout keep-masks, style
noise) are explicit (`rng`), in kernel layout (channels-last, Bp rows per decoder step).
"""
import ctypes
import functools
import math
import os as _os
import time as _time

import torch

from . import ops
from . import switches as sw
from .ops import (ACT_NONE, ACT_RELU, ACT_TANH, IM2COL_K, IM2COL_R, KC, RC, Operand, call, ptr, roundup, sgemm)
from ._lib import GtBiLstm, GtDecoderBwd, GtDecoderFwd, GtDecoderInfer, GtPlstm, GtPlstmBwd, check, lib

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
# ---- switches (environment, read once at import; the defaults are the measured-best configuration)
REC_SPLITS_OVERRIDE = sw.debug('rec_splits')     # diagnostics: force the K-split count
# forward LSTM steps as ONE fused launch (gt_lstm_step_fused).  Stand-alone it beats product + cell (20.4 vs 23.2 us,
# 25.5 vs 28.1 us) but every workgroup stages the whole activation matrix (4x the L2->LDS traffic), which slows the
# kernels of the other chain in the two-stream loop: 69 ms per forward loop instead of 62.  Off by default.
FUSED_CELL = bool(sw.experimental('fused_cell'))
HOST_TIMING = bool(sw.debug('host_timing'))      # diagnostics: host time spent issuing each loop
CELL_QUERY = sw.schedule_on('cell_query')    # attention cell + query product in one launch
BWD_CHUNKS = 12
# hoisted attention backward (gt_attn_param_grads): keep tanh(.) of every step (B*Ti*A floats each) so that the loop
# only computes the recurrent terms of the energies backward
ATTN_HOIST = sw.schedule_on('attn_hoist')
DEC_RING = 16     # ring of hand-off buffers between the two decoder chains;
                                                             # cross-stream syncs happen every ring/2 steps (loops.hip)
# hoisted decoder-LSTM input product (GtDecoderFwd.W2hp / GtDecoderBwd.W2hTp): [h_att | ctx] . W2ih of a block of steps
# is one tiled GEMM, only h_dec . W2hh stays in the per-step loops
HOIST2 = sw.schedule_on('hoist2')
# one launch per backward attention step (gt_attn_bwd_step) instead of context + energies backward; needs ATTN_HOIST, HOIST2
# context + energies backward of a step as ONE launch (gt_attn_bwd_step: the softmax row sums are applied by the consumers).
# Round 2: neutral (132.3 against 132.5 ms per pair); round 3, with the rest of the loop shorter: 127.3-127.8 against
# 128.1-128.4 in three alternating runs.  On (GANTRON_SCHEDULE=no_attn_bwd_fused: the two launches).
ATTN_BWD_FUSED = sw.schedule_on('attn_bwd_fused')
# ... with the attention LSTM's cell backward of the step inside the same launch, done by each batch row's last workgroup
# (gt_attn_bwd_step_cell): two dependent launches per backward step instead of three.  Round 4, one box, 24 pairs per run,
# alternating: 113.12-113.22 against 113.63-113.78 ms (medians of the pairs).  On (GANTRON_SCHEDULE=no_attn_bwd_cell: three).
ATTN_BWD_CELL = sw.schedule_on('attn_bwd_cell')
# gt_attn_bwd_wterms (the per-position terms of that launch for ALL steps, forward quantities only) issued by the training
# forward right behind the decoder loop instead of at the top of the backward pass (round 4: the launch took 7.6 ms beside the
# PostNet backward's 5.5 ms and the backward loop waited 2.1 ms for it)
WTERMS_IN_FORWARD = sw.schedule_on('wterms_in_forward')
# fused product + cell (+ query) launches of the forward loops (gt_rec_cell_fwd): bit 0 = attention LSTM, bit 1 = decoder LSTM,
# bit 2 = encoder BiLSTM; bits 3 / 4 = attention / decoder LSTM in bf16 mode (gt_rec_cell_fwd_bf16); bit 5 = the decoder
# LSTM's backward product + cell backward (gt_rec_cellbwd; measured 0.3 ms slower per pair than two launches: off)
REC_CELL = (31 if sw.schedule_on('rec_cell') else 0) | (32 if sw.experimental('rec_cellbwd') else 0)
ENC_SPLITS = sw.debug('enc_splits')   # K-splits of the fused BiLSTM step (0: cost model)
DEC_BLK_BWD = 24  # steps per block of the backward loop (round 3, with the forward's persistent recurrences on: 16: 130.4-130.7 ms per
                  # pair, 20: 130.4, 24: 128.7-129.6, 28: 130.1, 32: 129.3-131.0; chunk boundaries on whole blocks: no difference)
# K-splits of the decoder LSTM's small recurrent products in the hoisted mode (0: cost model).  That chain has slack
# (it only follows / feeds the attention chain): fewer, longer workgroups leave the CUs to the critical chain.
S2_FWD = sw.debug('s2_fwd')
S2_BWD = sw.debug('s2_bwd')
# Persistent, weight-resident LSTM recurrences (gt_plstm_fwd / gt_plstm_bwd, csrc/plstm.hip): bit 0 = the encoder BiLSTM
# forward (ONE launch for all steps of both directions), bit 1 = the decoder LSTM's h_dec recurrence of a block of steps
# (hoisted mode, fp32), bit 2 = the encoder BiLSTM backward (one launch), bit 3 = the decoder LSTM's backward recurrence of a
# block.  Measured on MI355X (round 3): the encoder launch takes 9.8 us per step against ~10 for the
# per-step fused launches it replaces; the decoder block runs at 33 us per step on 64 CUs -- fast enough for the chain it
# feeds.  On their own bits 0-1 changed nothing (the attention chain's 256-workgroup kernels then ran on 192 CUs:
# 134.0-134.6 vs 133.3-134.2 ms per pair); with the attention LSTM re-split for the 192 CUs that are left (3 K-splits
# instead of 4, done below) and the one-launch attention step: 131.0-132.1 against 133.0-133.4.  Bit 2: the BiLSTM backward
# phase 2.9 -> 2.35 ms (pair 130.4 against 130.9).  Bit 3 lost for two rounds (132.0 against 130.4 in round 3, 118.2-118.5
# against 114.4-114.9 early in round 4: the 64 CUs it holds for a whole block were missed by the attention chain's three
# launches per step, the one-wave cell backward most of all); with the cell backward inside the attention step's launch
# (ATTN_BWD_CELL) it pays: 112.2 / 112.4 / 112.6 against 113.5 / 113.1 / 112.7 ms (medians of 24 pairs, one box, alternating;
# with ATTN_BWD_CELL off it still loses: 118.2), and 1 740 launches and 14 GB of weight stream per pair are gone.
# Default 15 (GANTRON_SCHEDULE=no_plstm or GANTRON_EXPERIMENTAL=plstm=<bits> to change it).
PLSTM = sw.experimental('plstm', 15 if sw.schedule_on('plstm') else 0)
# The attention LSTM's per-step products from two scaled fp16 planes per operand (gt_rec_cell_fwd_h2 forward, gt_rec_gemm_h2
# backward: gt_sgemm's mode-4 arithmetic, 4 bytes per weight as fp32, 3/16 of the fp32 pipe's matrix time; bit 0 = forward,
# bit 1 = backward).  Stand-alone (tools/rec_h2_probe.py, M = 64): backward product 9.3-10.2 against 12.5 us, forward product 8.8
# against 11.6, error against float64 2-3e-8 of max sum |x||w| against the fp32 pipe's 3-5e-8.  In the step (one box, 16 pairs
# per run): forward only 110.5-110.9 against 113.0-113.1 ms (forward loop 35.3 instead of 37.6 ms); + backward 106.9 against
# 108.9-109.1 with the product in the fp32 kernel's geometry (88 registers, 18 KB of LDS; 8 K-splits: 10 / 12 / 16 give
# 107.2 / 107.0 / 108.8).  recurrent_bf16.hip's own form of the backward product (256-k chunks, 68 KB of LDS, 156-210
# registers: GANTRON_DEBUG=rec_h2_big) is faster alone (8.7 us) and LOSES in the loop (115.9; 111.0 at best) -- its workgroups
# wait for room beside the weight-gradient GEMMs'.  Default 3 (GANTRON_SCHEDULE=no_rec_h2: the fp32 MFMA kernels in both
# directions).  The backward form needs the step's max |dG1_t|, which the attention step's cell tail leaves (ATTN_BWD_CELL).
REC_H2 = 3 if sw.schedule_on('rec_h2') else 0
# The attention LSTM's per-step products as exact fp32 products on the bf16 matrix pipe (three bf16 planes, csrc/recurrent_bf16.hip
# NP = 3): bit 0 = the forward fused product + cell (gt_rec_cell_fwd_x3), bit 1 = the backward product (gt_rec_gemm_x3).
# Measured (round 3) and not adopted: 31.1 us per forward launch against 26.7, 25.6 per backward product against 19.6, pair
# 141.0-141.4 ms against 127.9-128.4 -- the launches follow their weight bytes (6 per weight instead of 4), not their MFMAs.
REC_X3 = sw.experimental('rec_x3', 0)
S1_FWD = sw.debug('s1_fwd')       # diagnostics: K-splits of the attention LSTM's forward product
S1_BWD = sw.debug('s1_bwd')       # diagnostics: K-splits of its backward product when the decoder LSTM's backward is persistent
DEC_BLK = 16       # steps per block in that mode (measured: 8: 160.6 ms per pair,
                                                              # 12: 157.9, 16: 157.9, 32: 160.5, 48: 160.9, 64: 165.7)


# energies + softmax + context of a forward step as one launch (gt_attn_step_fwd: the row's workgroups meet inside it).
# Measured slower than the two launches it replaces (20.8 vs 11.1 + 6.5 us stand-alone, forward loop 49.5 vs 45.7 ms): the
# in-launch wait costs what the kernel boundary cost and the context phase loses its prefetch.  Off by default.
ATTN_STEP = bool(sw.experimental('attn_step'))
# the same step as one launch WITHOUT a wait: per-chunk partial contexts, the row's last arriver combines (gt_attn_step2_fwd)
ATTN_STEP2 = sw.schedule_on('attn_step2')
# The h_att half of the attention LSTM's forward product rides in the attention step in front of it (GtRecTail, csrc/rec_tail.h):
# the step's workgroups that have nothing (more) to do pick up (column group, K-slice) items of h_att_t . W1[:, E:]^T, and LSTM
# step t + 1 multiplies the ctx half only.  REC_TAIL_FWD = K-slices of that half (GANTRON_EXPERIMENTAL=rec_tail[=slices], 0: off),
# REC_TAIL_KB = how many of its 16-wide k-blocks ride (GANTRON_DEBUG=rec_tail_kb=<n>, 0: all).  Measured (round 4, DESIGN.md
# section 3) and NOT adopted: the LSTM launch gets 5-6 us shorter and the attention step 5.6-6.6 us longer, whatever the item
# size (2 / 3 / 4 / 6 / 8 slices), the prefetch depth of the tail's product, its issue priority and the share of the half
# that rides (24 / 36 / 44 / 64 k-blocks: the two launches always sum to 41.6-42.3 us) -- pair 123.8-125.4 against
# 123.4-123.8 ms.  The bytes cost the same wherever they are moved.
REC_TAIL_FWD = sw.experimental('rec_tail', 0)
REC_TAIL_FWD = 3 if REC_TAIL_FWD == 1 else REC_TAIL_FWD
REC_TAIL_KB = sw.debug('rec_tail_kb', 0)
_HANDOFF_FLAGS = {}


def handoff_flag(device):
    """Per-device int32 word that kernels with an in-launch wait raise when a bounded wait gives up (the step's result is
    then undefined).  Nothing on the step's path reads it (that would synchronise); check_handoffs() does."""
    key = torch.device(device).index
    if key not in _HANDOFF_FLAGS:
        _HANDOFF_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _HANDOFF_FLAGS[key]


def check_handoffs():
    """Raise if any in-launch hand-off timed out since the last check (synchronises: call it where the loop already
    does -- logging, validation, the end of a benchmark)."""
    for key, flag in _HANDOFF_FLAGS.items():
        if int(flag.item()) != 0:
            flag.zero_()
            raise RuntimeError(f'an in-launch hand-off of a decoder-loop kernel timed out on cuda:{key}: results since the '
                               'last check are invalid (the attn_step experiment is off by default: two launches)')


# GANTRON_POISON=1 (tests): buffers that are NOT zero-initialised start as NaN, so that a read of anything the kernels have
# not written shows up in the results
POISON = _os.environ.get('GANTRON_POISON', '0') == '1'


# Tests and tools/enc_grad_probe.py: when a dict, named intermediates of the next forward / backward are cloned into it
# (encoder activations whose signs are the ReLU decisions, d(memory), the gradient entering the encoder convolutions).
TAPS = None


def _tap(name, value):
    if TAPS is not None:
        TAPS[name] = [v.detach().clone() for v in value] if isinstance(value, (list, tuple)) else value.detach().clone()


def _f32(*shape, device, zero=True):
    if not zero and POISON:
        return torch.full(shape, float('nan'), device=device, dtype=torch.float32)
    return (torch.zeros if zero else torch.empty)(*shape, device=device, dtype=torch.float32)


def _state(T1, Bp, H, device, zero):
    """(T1, Bp, H) cell-state history whose slice 0 is the zero initial state; the rest is written step by step (zero = False:
    only slice 0 is cleared)."""
    t = _f32(T1, Bp, H, device=device, zero=zero)
    if not zero:
        t[0].zero_()
    return t


def _run(name, struct):
    check(getattr(lib(), name)(ctypes.byref(struct), ops._stream()), name)


class PhaseTimer:
    """Optional wall-clock phase breakdown (GANTRON_DEBUG=phase_timing): synchronises at every mark, so it
    serialises the streams -- a development aid, never on in a measured run."""

    def __init__(self):
        self.on = bool(sw.debug('phase_timing'))
        self.t = None
        self.acc = {}

    def mark(self, name):
        if not self.on:
            return
        import time
        torch.cuda.synchronize()
        now = time.perf_counter()
        if self.t is not None and name is not None:
            self.acc[name] = self.acc.get(name, 0.0) + (now - self.t) * 1e3
        self.t = now

    def report(self):
        if self.on and self.acc:
            import sys
            print('[phases ms] ' + '  '.join(f'{k}={v:.1f}' for k, v in self.acc.items()), file=sys.stderr, flush=True)
            self.acc = {}


PHASES = PhaseTimer()


class LeafStream:
    """Side stream for weight-gradient GEMMs.  They are leaves of the backward graph (nothing on the
    critical path consumes them), so they run concurrently with the latency-bound recurrent loops and
    fill the CUs those leave idle.  Inputs produced on the main stream are kept alive until join()."""

    def __init__(self):
        self.stream = None
        self.keep = []
        self.enabled = sw.schedule_on('leaf_stream')
        # experimental leaf_defer: queue the leaves and run them on the main stream at flush() / join(), after the loop they
        # would otherwise share the chip with
        self.defer = bool(sw.experimental('leaf_defer'))
        self.queue = []

    def flush(self):
        q, self.queue = self.queue, []
        for fn in q:
            fn()

    def run(self, fn, *tensors):
        if not self.enabled:
            fn()
            return
        if self.defer:
            self.keep.extend(tensors)
            self.queue.append(fn)
            return
        if self.stream is None:
            self.stream = torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[0])
        self.stream.wait_stream(torch.cuda.current_stream())
        self.keep.extend(tensors)
        with torch.cuda.stream(self.stream):
            fn()

    def join(self):
        self.flush()
        if self.stream is not None and self.enabled:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.keep.clear()


LEAF = LeafStream()


def acc_add(dst, src):
    """dst += src through the axpby kernel (dst, src contiguous, same numel)."""
    assert dst.is_contiguous() and src.is_contiguous() and dst.numel() == src.numel()
    call('gt_axpby', dst, 1.0, src, 1.0, dst, dst.numel())


def rec_splits(N, K, nz=1):
    """K-splits of a per-step recurrent GEMM, from a cost model of the weight-streaming kernel fitted to
    rocprofv3 timings on MI355X: a workgroup (64 columns x one K-slice) costs ~4 us of fixed latency plus ~1.7 us
    per 8-block LDS chunk of its slice, workgroups beyond the 256 CUs run as further rounds, and every extra slab
    costs the consuming cell kernel a little.  (N=3072, K=4096 is the case a plain 'fill the CUs' rule got wrong:
    6 splits = 288 workgroups = two rounds, 33 us; 5 splits = 240 workgroups in one round.)"""
    kb = -(-K // 16)
    groups = -(-N // 64) * nz
    best, best_cost = 1, None
    for s in range(1, min(16, kb) + 1):
        kbs = -(-kb // s)
        if -(-kb // kbs) != s:                 # would leave an empty slab (gt_rec_gemm rejects it)
            continue
        rounds = -(-groups * s // 256)
        cost = rounds * (-(-kbs // 8) * 1.7 + 4.0) + 0.15 * s
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = s, cost
    if REC_SPLITS_OVERRIDE:
        return ops.pick_splits(K, REC_SPLITS_OVERRIDE)
    return best


# chunks per K-slice of the bf16 product.  4 is allowed by the kernel but its 210 VGPRs x 512 threads fill a CU's
# register file, so nothing of the other chain can share the CU; measured per pair (B=64): 1 chunk 154.5 ms,
# 2: 150.2, 3: 149.4, 4: 153.1
BF16_MAX_CHUNKS = 3


def rec_splits_bf16(N, K):
    """K-splits of gt_rec_gemm_bf16 (slices of at most BF16_MAX_CHUNKS chunks of 256 columns, every load of a slice
    issued up front).  Cost model fitted to rocprofv3 durations at the decoder's shapes: a workgroup round costs
    ~2.0 us + 1.3 us per chunk; slices of <= 2 chunks need <= 128 VGPRs, so two workgroups share a CU (512 per round,
    the second 256 at ~0.6 of a round's cost), longer slices run one workgroup per CU (256 per round); 1.4 us of launch
    + epilogue and 0.15 us per slab for the consumer.  (N=3072, K=4096: 6 splits = 288 workgroups of 3 chunks = two
    rounds, 12.4 us; 8 splits = 384 workgroups of 2 chunks in one round, 8.0 us; 16 splits = 768 workgroups, 13.9 us.)"""
    kb = -(-K // 32)
    groups = -(-N // 64)
    best, best_cost = None, None
    for s in range(int(lib().gt_rec_bf16_min_splits(K)), min(16, kb) + 1):
        kbs = -(-kb // s)
        nch = -(-kbs // 8)
        if -(-kb // kbs) != s or nch > BF16_MAX_CHUNKS:
            continue
        wgs, per_round = groups * s, 2.0 + 1.3 * nch
        cap = 512 if nch <= 2 else 256
        cost = 1.4 + 0.15 * s
        while wgs > 0:
            w = min(wgs, cap)
            cost += per_round * (1.0 + 0.6 * max(0, w - 256) / 256.0)
            wgs -= w
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = s, cost
    return best


def pack_rec(W2d, N, n_valid, K, transposed=False):
    """Pack a (n_valid x K) [or (K x n_valid) when transposed] matrix for gt_rec_gemm."""
    out = torch.empty(int(lib().gt_rec_weight_elems(N, K)), device=W2d.device, dtype=torch.float32)
    call('gt_pack_rec_weight', W2d, W2d.stride(0), int(transposed), N, n_valid, K, out)
    return out


def pack_rec_bf16(W2d, N, n_valid, K, transposed=False):
    """Pack a (n_valid x K) [or (K x n_valid) when transposed] matrix as bf16 MFMA fragments for gt_rec_gemm_bf16."""
    out = torch.empty(int(lib().gt_rec_weight_bf16_elems(N, K)), device=W2d.device, dtype=torch.bfloat16)
    call('gt_pack_rec_weight_bf16', W2d, W2d.stride(0), int(transposed), N, n_valid, K, out)
    return out


def pack_rec_x3(W2d, N, n_valid, K, transposed=False):
    """Pack a (n_valid x K) [or (K x n_valid) when transposed] matrix as three bf16 planes of MFMA fragments for gt_rec_gemm_x3."""
    out = torch.empty(int(lib().gt_rec_weight_x3_elems(N, K)), device=W2d.device, dtype=torch.bfloat16)
    call('gt_pack_rec_weight_x3', W2d, W2d.stride(0), int(transposed), N, n_valid, K, out)
    return out


def pack_rec_h2(W2d, N, n_valid, K, transposed=False):
    """Pack a (n_valid x K) [or (K x n_valid) when transposed] matrix as two scaled fp16 planes of MFMA fragments for
    gt_rec_gemm_h2 (the scale's source, max |W|, is found here and kept in the image's trailer)."""
    out = torch.empty(int(lib().gt_rec_weight_h2_elems(N, K)), device=W2d.device, dtype=torch.float16)
    call('gt_pack_rec_weight_h2', W2d, W2d.stride(0), int(transposed), N, n_valid, K, out)
    return out


def rec_splits_x3(N, K, cus=256):
    """K-splits of the three-plane products (K-slices of at most 1024, gt_rec_x3_min_splits): the most splits whose
    workgroups (64 columns x one slice, one per CU: 101 KB of LDS) still fit `cus` CUs in one round; at least the minimum."""
    kb = -(-K // 32)
    groups = -(-N // 64)
    lo = int(lib().gt_rec_x3_min_splits(K))
    ok = [s for s in range(lo, min(16, kb) + 1) if -(-kb // -(-kb // s)) == s]
    fit = [s for s in ok if groups * s <= cus]
    return max(fit) if fit else ok[0]


def pack_rec_cell(W2d, H, K, bf16=False):
    """Pack a (4H x K) LSTM weight for gt_rec_cell_fwd / gt_rec_cell_fwd_bf16 / gt_rec_cell_fwd_x3 (bf16 = 3) (a
    workgroup's 64 columns = the four gates of 16 hidden units)."""
    if bf16 == 3 and bf16 is not True:
        out = torch.empty(int(lib().gt_rec_weight_x3_elems(4 * H, K)), device=W2d.device, dtype=torch.bfloat16)
        call('gt_pack_rec_cell_weight_x3', W2d, W2d.stride(0), H, K, out)
        return out
    if bf16 == 2 and bf16 is not True:          # two scaled fp16 planes (gt_rec_cell_fwd_h2)
        out = torch.empty(int(lib().gt_rec_weight_h2_elems(4 * H, K)), device=W2d.device, dtype=torch.float16)
        call('gt_pack_rec_cell_weight_h2', W2d, W2d.stride(0), H, K, out)
        return out
    if bf16:
        out = torch.empty(int(lib().gt_rec_weight_bf16_elems(4 * H, K)), device=W2d.device, dtype=torch.bfloat16)
        call('gt_pack_rec_cell_weight_bf16', W2d, W2d.stride(0), H, K, out)
        return out
    out = torch.empty(int(lib().gt_rec_weight_elems(4 * H, K)), device=W2d.device, dtype=torch.float32)
    call('gt_pack_rec_cell_weight', W2d, W2d.stride(0), H, K, out)
    return out


def pack_gates(W2d, H, K):
    """Pack a (4H x K) LSTM weight gate-interleaved (row 4*unit + gate) for gt_lstm_step_fused."""
    out = torch.empty(4 * H * int(lib().gt_fused_k_padded(K)), device=W2d.device, dtype=torch.float32)
    call('gt_pack_gates', W2d, W2d.stride(0), H, K, out)
    return out


def _gemm_precision(fn):
    """Engine entry point: layer GEMMs inside run with bf16 operands when the engine was built with hparams.fp16_run."""
    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        with ops.bf16_gemms(getattr(self, 'bf16', False)):
            return fn(self, *a, **k)
    return wrapped


class ConvBNStack:
    """[Conv1d(k) -> BatchNorm1d(train) -> act -> dropout] x n over channels-last rows; the encoder
    convolutions (model.py:165-178,196-197) and the PostNet (model.py:113-151)."""

    def __init__(self, prefix, n_layers, acts):
        self.prefix, self.n, self.acts = prefix, n_layers, acts

    def forward(self, P, x, B, T, keeps, training, save):
        for i in range(self.n):
            W = P[f'{self.prefix}.{i}.0.conv.weight']
            cout, cin, k = W.shape
            bn = f'{self.prefix}.{i}.1'
            Wf = ops.conv_weight_fwd(W)
            cin_x = x.shape[1]
            if cin_x != cin:        # input rows carry zero pad columns (the im2col operand needs channels % 4 == 0)
                Wp = _f32(cout, k, cin_x, device=x.device)
                Wp[:, :, :cin] = Wf
                Wf = Wp
            y = ops.conv1d_fwd(x, Wf, P[f'{self.prefix}.{i}.0.conv.bias'], B, T, cin_x, cout, k, 1)
            rows = B * T
            stats = mi = None
            if training:
                stats = _f32(2 * cout, device=x.device, zero=False)
                call('gt_col_stats', y, y, rows, cout, stats, ops.col_ws(rows, cout, x.device))
                mi = _f32(2 * cout, device=x.device, zero=False)
            keep = keeps[i] if (keeps is not None and training) else None
            out = _f32(rows, cout, device=x.device, zero=False)
            call('gt_bn_act_drop_fwd', y, stats, rows, cout, P[bn + '.weight'], P[bn + '.bias'],
                 P[bn + '.running_mean'], P[bn + '.running_var'], BN_MOMENTUM, BN_EPS, self.acts[i], keep, 0.5, out, mi,
                 int(training))
            if training and (bn + '.num_batches_tracked') in P:
                P[bn + '.num_batches_tracked'] += 1
            save.append(dict(x=x, y=y, mi=mi, keep=keep))
            x = out
        return x

    def backward(self, P, G, dy, B, T, save, need_dx=True, first_input_override=None):
        rows = B * T
        for i in reversed(range(self.n)):
            W = P[f'{self.prefix}.{i}.0.conv.weight']
            cout, cin, k = W.shape
            bn = f'{self.prefix}.{i}.1'
            s = save[i]
            dz = _f32(rows, cout, device=dy.device, zero=False)
            sums = _f32(2 * cout, device=dy.device, zero=False)
            call('gt_bn_act_drop_bwd1', dy, s['y'], s['mi'], P[bn + '.weight'], P[bn + '.bias'], rows, cout,
                 self.acts[i], s['keep'], 0.5, dz, sums, ops.col_ws(rows, cout, dy.device))
            dconv = _f32(rows, cout, device=dy.device, zero=False)
            call('gt_bn_act_drop_bwd2', dz, s['y'], s['mi'], P[bn + '.weight'], sums, rows, cout, dconv,
                 G[bn + '.weight'], G[bn + '.bias'])
            x_in = first_input_override if (i == 0 and first_input_override is not None) else s['x']

            def leaf(dconv=dconv, x_in=x_in, cin=cin, cout=cout, k=k, i=i):
                gw = G[f'{self.prefix}.{i}.0.conv.weight']
                if x_in.shape[1] != cin:                # zero-padded input channels: drop their (zero-input) rows
                    dWp = _f32(cout, x_in.shape[1], k, device=dconv.device)
                    ops.conv1d_dw(dconv, x_in, B, T, x_in.shape[1], cout, k, 1, dWp)
                    acc_add(gw, dWp[:, :cin, :].contiguous())
                else:
                    ops.conv1d_dw(dconv, x_in, B, T, cin, cout, k, 1, gw)
                acc_add(G[f'{self.prefix}.{i}.0.conv.bias'], ops.col_sums(dconv, rows, cout))
            LEAF.run(leaf, dconv, x_in)
            if i > 0 or need_dx:
                dy = ops.conv1d_dx(dconv, ops.conv_weight_bwd(W), B, T, cin, cout, k, 1)
        return dy


class GeneratorEngine:
    def __init__(self, hp):
        self.hp = hp
        self.n_mel = hp.n_mel_channels
        self.enc_dim = hp.encoder_embedding_dim
        self.H = self.enc_dim // 2
        self.P, self.Ha, self.Hd, self.A = hp.prenet_dim, hp.attention_rnn_dim, hp.decoder_rnn_dim, hp.attention_dim
        self.F, self.Kc = hp.attention_location_n_filters, hp.attention_location_kernel_size
        noise = hp.noise_size if hp.use_noise else 0
        self.vesus = bool(hp.vesus_path)
        self.use_labels = bool(hp.use_labels and hp.vesus_path)
        # encoder_inputs (hparams.py:102): labels and style noise are extra channels of the first encoder convolution
        # (model.py:162-170, 649-651) instead of extra columns of the decoder memory (model.py:240-246, 659)
        self.enc_in = bool(hp.encoder_inputs)
        self.enc_labels = hp.n_labels if (self.enc_in and self.use_labels) else 0
        self.enc_noise = noise if self.enc_in else 0
        self.noise = 0 if self.enc_in else noise
        self.mem_labels = 5 if (self.use_labels and not self.enc_in) else 0
        self.E = self.enc_dim + self.noise
        if self.vesus:
            self.E += hp.speakers_embedding + self.mem_labels
        # The decoder memory is stored Ep columns wide (E rounded up to 4, the extra columns zero): with VESUS conditioning
        # E = 512 + 64 + 5 + 512 = 1093, and every float4 path of the loop kernels (context, cells, fused product + cell)
        # wants rows that are multiples of 16 bytes.  The padding lives in activations and packed weight images only --
        # parameters, gradients and the state_dict keep the reference's shapes.
        self.Ep = roundup(self.E, 4)
        # hparams.fp16_run (the reference's apex path, train.py:246-250): on MI355X the two decoder LSTM products run
        # with bf16 operands and fp32 accumulation (gt_rec_gemm_bf16); weights, state and every other op stay fp32
        self.bf16 = bool(getattr(hp, 'fp16_run', False))
        self.enc_stack = ConvBNStack('encoder.convolutions', hp.encoder_n_convolutions,
                                     [ACT_RELU] * hp.encoder_n_convolutions)
        self.post_stack = ConvBNStack('postnet.convolutions', hp.postnet_n_convolutions,
                                      [ACT_TANH] * (hp.postnet_n_convolutions - 1) + [ACT_NONE])

    # ------------------------------------------------------------------------------ helpers
    @staticmethod
    def step_major_index(lengths, T, Bp):
        """Row maps of the packed BiLSTM.  fwd_idx/rev_idx: (T*Bp,) source row b*T+t of step-major row
        (s,b) (or -1); inverse maps (B*T,) give the step-major row of (b,t) for each direction."""
        import numpy as np
        B = len(lengths)
        L = np.asarray(lengths, dtype=np.int64)[None, :]                 # (1, B)
        s = np.arange(T, dtype=np.int64)[:, None]                        # (T, 1)
        b = np.arange(B, dtype=np.int64)[None, :]
        live = s < L                                                     # (T, B): step s of item b exists
        fwd = np.full((T, Bp), -1, dtype=np.int32)
        rev = np.full((T, Bp), -1, dtype=np.int32)
        active = np.zeros((T, Bp), dtype=np.int32)
        fwd[:, :B] = np.where(live, b * T + s, -1)
        rev[:, :B] = np.where(live, b * T + (L - 1 - s), -1)
        active[:, :B] = live
        # inverse maps, (B, T): step-major row of position t of item b in each direction
        t = np.arange(T, dtype=np.int64)[None, :]
        Lc = np.asarray(lengths, dtype=np.int64)[:, None]
        bb = np.arange(B, dtype=np.int64)[:, None]
        ifwd = np.where(t < Lc, t * Bp + bb, -1).astype(np.int32)
        irev = np.where(t < Lc, (Lc - 1 - t) * Bp + bb, -1).astype(np.int32)
        fwd, rev, ifwd, irev, active = (torch.from_numpy(a) for a in (fwd, rev, ifwd, irev, active))
        return fwd.reshape(-1), rev.reshape(-1), ifwd.reshape(-1), irev.reshape(-1), active

    def _encode(self, P, text, in_lens, rng, speaker_ids, emotions, training, sv):
        """Embedding, encoder convolutions, packed BiLSTM and the decoder memory
        [encoder outputs | speaker | labels | style noise] with its attention projection
        (model.py:648-663, 193-212, 273-279, 440).  rng['noise'] is (B, noise) or, for inference with a
        caller-supplied style, (B, Ti, noise)."""
        hp, dev = self.hp, text.device
        B, Ti = text.shape
        Bp = sv['Bp']
        H, E, Ep, A = self.H, self.E, self.Ep, self.A
        # ---- embedding + encoder convolutions (channels-last rows = (b, t))
        D0 = hp.symbols_embedding_dim
        cin0 = roundup(D0 + self.enc_labels + self.enc_noise, 4)     # channels % 4 == 0 for the im2col operand
        x = _f32(B * Ti, cin0, device=dev, zero=cin0 > D0)
        call('gt_embedding_fwd', text, P['embedding.weight'], B * Ti, D0, x, cin0)
        if cin0 > D0:                                   # [embedded text | labels | style noise | zero pad] per position
            x3 = x.view(B, Ti, cin0)
            if self.enc_labels:
                x3[:, :, D0:D0 + self.enc_labels] = emotions[:, None, :]
            if self.enc_noise:
                noise = rng['noise']                    # (B, noise), or (B, Ti, noise) from an inference caller
                c0 = D0 + self.enc_labels
                x3[:, :, c0:c0 + self.enc_noise] = noise if noise.dim() == 3 else noise[:, None, :]
        sv['enc_conv'] = []
        x = self.enc_stack.forward(P, x, B, Ti, rng.get('enc_keep'), training, sv['enc_conv'])
        sv['enc_x3'] = x
        _tap('enc_act', [s_['x'] for s_ in sv['enc_conv'][1:]] + [x])     # block outputs: (B*Ti, C) after BN, ReLU, dropout

        PHASES.mark('f.enc_conv')
        # ---- packed BiLSTM, step-major
        fwd_i, rev_i, ifwd, irev, active = self.step_major_index(in_lens, Ti, Bp)
        # one non-blocking copy for the five maps (ops.h2d: no host stall)
        maps = dict(fwd=fwd_i, rev=rev_i, ifwd=ifwd, irev=irev, active=active.reshape(-1))
        flat = ops.h2d(torch.cat([v.reshape(-1) for v in maps.values()]), torch.int32, dev)
        idx, o = {}, 0
        for k, v in maps.items():
            idx[k] = flat[o:o + v.numel()]
            o += v.numel()
        idx['active'] = idx['active'].view(Ti, Bp)
        sv['idx'] = idx
        C = self.enc_dim
        Xs = _f32(2, Ti * Bp, C, device=dev, zero=False)
        call('gt_gather_rows', x, C, idx['fwd'], Ti * Bp, C, Xs[0], C, 0)
        call('gt_gather_rows', x, C, idx['rev'], Ti * Bp, C, Xs[1], C, 0)
        pre = _f32(2, Ti * Bp, 4 * H, device=dev, zero=False)
        Whp = []
        for d, sfx in enumerate(('', '_reverse')):
            bias = _f32(4 * H, device=dev, zero=False)
            call('gt_axpby', P['encoder.lstm.bias_ih_l0' + sfx], 1.0, P['encoder.lstm.bias_hh_l0' + sfx], 1.0, bias, 4 * H)
            ops.linear(Xs[d], P['encoder.lstm.weight_ih_l0' + sfx], Ti * Bp, C, 4 * H, bias=bias, out=pre[d])
            Whp.append(pack_rec(P['encoder.lstm.weight_hh_l0' + sfx], 4 * H, 4 * H, H))
        Whp = torch.stack(Whp)
        gates = _f32(2, Ti, Bp, 4 * H, device=dev, zero=False)
        cst = _f32(2, Ti + 1, Bp, H, device=dev)
        hs = _f32(2, Ti, Bp, H, device=dev, zero=False)
        if (PLSTM & 1) and lib().gt_plstm_supported(Bp, H):
            # the whole recurrence of both directions as ONE persistent launch (W_hh resident in registers, 2 x H/16 workgroups)
            Whh = torch.stack([P['encoder.lstm.weight_hh_l0' + sfx] for sfx in ('', '_reverse')]).contiguous()
            q = GtPlstm()
            q.B, q.Bp, q.H, q.T, q.ndir, q.s_lo, q.s_hi, q.hx_ring = B, Bp, H, Ti, 2, 0, Ti, 2
            hx = _f32(2, 2, Bp, H, device=dev, zero=False)
            counters = torch.zeros(2, dtype=torch.int32, device=dev)
            q.Whh, q.w_dstride, q.pre, q.pre_dstride = ptr(Whh), Whh.stride(0), ptr(pre), pre.stride(0)
            q.gates, q.gates_dstride, q.c, q.c_dstride, q.hs, q.hs_dstride = ptr(gates), gates.stride(0), ptr(cst), cst.stride(0), \
                ptr(hs), hs.stride(0)
            q.hx, q.hx_dstride, q.hx_step, q.hx_ld = ptr(hx), hx.stride(0), hx.stride(1), H
            q.active, q.counters, q.timeout_flag = ptr(idx['active']), ptr(counters), ptr(handoff_flag(dev))
            _run('gt_plstm_fwd', q)
            sv.update(Xs=Xs, enc_gates=gates, enc_c=cst, enc_hs=hs)
            return self._encode_memory(P, sv, hs, idx, rng, speaker_ids, emotions, B, Ti, Bp, dev)
        ldx = roundup(H, 16)
        S = rec_splits(4 * H, H, nz=2)
        a = GtBiLstm()
        a.B, a.Bp, a.T, a.H, a.S, a.backward = B, Bp, Ti, H, S, 0
        fused_enc = bool(REC_CELL & 4) and H % 16 == 0         # product + cell of a BiLSTM step as one launch
        if fused_enc and ENC_SPLITS:
            S = a.S = ops.pick_splits(H, ENC_SPLITS)
        Xst = _f32(2, 2, Bp, ldx, device=dev) if fused_enc else _f32(2, Bp, ldx, device=dev)
        part = _f32(2, S, Bp, 4 * H, device=dev, zero=False)
        if fused_enc:
            Whc = torch.stack([pack_rec_cell(P['encoder.lstm.weight_hh_l0' + sfx], H, H) for sfx in ('', '_reverse')])
            assert Whc.stride(0) == Whp.stride(0)
            enc_counters = torch.zeros(2 * (Bp // 64 if Bp > 64 else 1) * (H // 16), dtype=torch.int32, device=dev)
            a.Whc, a.counters = ptr(Whc), ptr(enc_counters)
        a.pre, a.Whp, a.w_zstride, a.active = ptr(pre), ptr(Whp), Whp.stride(0), ptr(idx['active'])
        a.X, a.ldx, a.part, a.gates, a.c, a.hs = ptr(Xst), ldx, ptr(part), ptr(gates), ptr(cst), ptr(hs)
        _run('gt_bilstm_run', a)
        sv.update(Xs=Xs, enc_gates=gates, enc_c=cst, enc_hs=hs)
        return self._encode_memory(P, sv, hs, idx, rng, speaker_ids, emotions, B, Ti, Bp, dev)

    def _encode_memory(self, P, sv, hs, idx, rng, speaker_ids, emotions, B, Ti, Bp, dev):
        """The decoder memory [encoder outputs | speaker | labels | style noise] and its attention projection."""
        hp = self.hp
        H, E, Ep, A = self.H, self.E, self.Ep, self.A
        PHASES.mark('f.bilstm')
        # ---- decoder memory = [encoder outputs | speaker | labels | style noise]
        memory = _f32(B, Ti, Ep, device=dev)                 # columns [E, Ep) stay zero
        mem2 = memory.view(B * Ti, Ep)
        call('gt_gather_rows', hs[0], H, idx['ifwd'], B * Ti, H, mem2, Ep, 0)
        check(lib().gt_gather_rows(ptr(hs[1]), H, ptr(idx['irev']), B * Ti, H, ptr(memory, H), Ep, 0, ops._stream()),
              'gt_gather_rows')
        col = 2 * H
        if self.vesus:
            se = hp.speakers_embedding
            spk = _f32(B, se, device=dev, zero=False)
            call('gt_embedding_fwd', speaker_ids, P['speaker_embedding.weight'], B, se, spk, se)
            memory[:, :, col:col + se] = spk[:, None, :]
            col += se
            if self.mem_labels:
                memory[:, :, col:col + 5] = emotions[:, None, :]
                col += 5
            sv['speaker_ids'] = speaker_ids
        if self.noise:
            noise = rng['noise']
            memory[:, :, col:col + self.noise] = noise if noise.dim() == 3 else noise[:, None, :]
        sv['memory'] = memory
        Wmem = P['decoder.attention_layer.memory_layer.linear_layer.weight']
        pm = ops.linear(mem2, Wmem, B * Ti, E, A, ldx=Ep)
        sv['pm'] = pm

        return memory, mem2, pm

    def _decoder_operands(self, P):
        """Weights of the two decoder LSTMs as the loops read them, with the ctx columns padded from E to Ep (zeros):
        W1cat (4Ha, Ep + Ha) for the operand [ctx | h_att] (the Prenet columns of weight_ih are handled outside the loop),
        W2ihp (4Hd, Ha + Ep) for [h_att | ctx], and the projection [mel rows; gate row] (n_mel + 1, Hd + Ep) for [h_dec | ctx]."""
        E, Ep, Pd = self.E, self.Ep, self.P
        W1ih, W1hh = P['decoder.attention_rnn.weight_ih'], P['decoder.attention_rnn.weight_hh']
        W2ih = P['decoder.decoder_rnn.weight_ih']
        Wp, Wg = P['decoder.linear_projection.linear_layer.weight'], P['decoder.gate_layer.linear_layer.weight']
        pad = (lambda rows: [W1ih.new_zeros(rows, Ep - E)]) if Ep > E else (lambda rows: [])
        W1cat = torch.cat([W1ih[:, Pd:]] + pad(W1ih.shape[0]) + [W1hh], dim=1).contiguous()
        W2ihp = torch.cat([W2ih] + pad(W2ih.shape[0]), dim=1).contiguous() if Ep > E else W2ih.contiguous()
        Wpg = torch.cat((torch.cat([Wp] + pad(Wp.shape[0]), dim=1), torch.cat([Wg] + pad(Wg.shape[0]), dim=1)), dim=0).contiguous()
        return W1cat, W2ihp, Wpg

    # ------------------------------------------------------------------------------ forward
    @_gemm_precision
    def forward(self, P, text, in_lens, mels, out_lens, rng, speaker_ids=None, emotions=None, training=True,
                mask_padding=True):
        """text (B,Ti) int64 cuda; in_lens/out_lens python lists; mels (B,n_mel,To) cuda.
        Returns (mel (B,n_mel,To), mel_post, gate (B,To), align (B,To,Ti)), saved."""
        hp, dev = self.hp, text.device
        B, Ti = text.shape
        To = mels.shape[2]
        Bp = roundup(B, 16) if B <= 64 else roundup(B, 64)      # rows per decoder step (pad rows stay zero)
        n_mel, H, E, Ep, A = self.n_mel, self.H, self.E, self.Ep, self.A
        sv = dict(B=B, Bp=Bp, Ti=Ti, To=To, in_lens=list(in_lens), out_lens=list(out_lens), training=training,
                  text=text)

        PHASES.mark(None)
        # ---- prenet over all teacher-forcing frames (go frame + mel[:, :, :To-1]), step-major (To, Bp, .), and its
        # projection into the attention LSTM
        Pd = self.P
        Ha, Hd = self.Ha, self.Hd
        W1ih, W1hh = P['decoder.attention_rnn.weight_ih'], P['decoder.attention_rnn.weight_hh']
        W2ih, W2hh = P['decoder.decoder_rnn.weight_ih'], P['decoder.decoder_rnn.weight_hh']
        pk = rng['prenet_keep']

        def decoder_inputs():
            x0 = _f32(To, Bp, n_mel, device=dev)
            if To > 1:
                check(lib().gt_permute3(ptr(mels), 1, n_mel * To, To, To - 1, B, n_mel, ptr(x0, Bp * n_mel), Bp * n_mel,
                                        n_mel, 0, 1.0, ops._stream()), 'gt_permute3')
            W0, W1 = P['decoder.prenet.layers.0.linear_layer.weight'], P['decoder.prenet.layers.1.linear_layer.weight']
            p1 = ops.linear(x0, W0, To * Bp, n_mel, Pd)
            call('gt_bias_act_drop_fwd', p1, None, To * Bp, Pd, ACT_RELU, 0, pk[0], 0.5, p1)
            p2 = ops.linear(p1, W1, To * Bp, Pd, Pd)
            call('gt_bias_act_drop_fwd', p2, None, To * Bp, Pd, ACT_RELU, 0, pk[1], 0.5, p2)
            bias1 = _f32(4 * Ha, device=dev, zero=False)
            call('gt_axpby', P['decoder.attention_rnn.bias_ih'], 1.0, P['decoder.attention_rnn.bias_hh'], 1.0, bias1, 4 * Ha)
            bias2 = _f32(4 * Hd, device=dev, zero=False)
            call('gt_axpby', P['decoder.decoder_rnn.bias_ih'], 1.0, P['decoder.decoder_rnn.bias_hh'], 1.0, bias2, 4 * Hd)
            pre1 = ops.linear(p2, W1ih, To * Bp, Pd, 4 * Ha, bias=bias1, ldw=Pd + E)
            return x0, p1, p2, bias2, pre1

        # (Measured and not adopted: these launches on a side stream beside the encoder -- 133.1-133.5 against 133.0-133.8 ms
        # per pair; the host is far enough ahead that the encoder's small launches are already queued back to back.)
        memory, mem2, pm = self._encode(P, text, in_lens, rng, speaker_ids, emotions, training, sv)
        idx = sv['idx']
        x0, p1, p2, bias2, pre1 = decoder_inputs()
        sv.update(x0=x0, p1=p1, p2=p2)

        # ---- decoder loop
        K1, K2 = Ep + Ha, Ha + Ep + Hd             # loop operands are Ep wide in their ctx part (zero columns beyond E)
        W1cat, W2ihp, Wpg = self._decoder_operands(P)
        W2cat = None if (HOIST2 and not FUSED_CELL) else torch.cat((W2ihp, W2hh), dim=1).contiguous()
        Apad = roundup(A, 16)
        d = GtDecoderFwd()
        d.B, d.Bp, d.Ti, d.To, d.E, d.ldE, d.P, d.Ha, d.Hd, d.A, d.Apad, d.F, d.Kc = \
            B, Bp, Ti, To, Ep, Ep, Pd, Ha, Hd, A, Apad, self.F, self.Kc
        d.S1, d.S2, d.Sq = rec_splits(4 * Ha, K1), rec_splits(4 * Hd, K2), ops.pick_splits(Ha, 16)
        d.p_att = hp.p_attention_dropout if training else 0.0
        d.p_dec = hp.p_decoder_dropout if training else 0.0
        fused = FUSED_CELL and Ha % 4 == 0 and Hd % 4 == 0 and not self.bf16
        hoist2 = HOIST2 and not FUSED_CELL
        keep_t = dict(Wqp=pack_rec(P['decoder.attention_layer.query_layer.linear_layer.weight'], Apad, A, Ha))
        if hoist2:
            pack = pack_rec_bf16 if self.bf16 else pack_rec
            split = rec_splits_bf16 if self.bf16 else rec_splits
            d.bf16 = int(self.bf16)
            d.S1, d.S2 = split(4 * Ha, K1), split(4 * Hd, Hd)
            if S2_FWD and not self.bf16:
                d.S2 = ops.pick_splits(Hd, S2_FWD)
            if S1_FWD and not self.bf16:
                d.S1 = ops.pick_splits(K1, S1_FWD)
            elif (PLSTM & 2) and not self.bf16 and lib().gt_plstm_supported(Bp, Hd) and 4 * Ha == 4096 and d.S1 == 4:
                d.S1 = 3        # the persistent decoder-LSTM launch holds 64 CUs: 64 column groups x 3 K-splits = the other 192
            # fused product + cell launches (gt_rec_cell_fwd) take their own packing
            use_c1 = bool(REC_CELL & (8 if self.bf16 else 1)) and CELL_QUERY and Ha % 32 == 0
            use_c2 = bool(REC_CELL & (16 if self.bf16 else 2)) and Hd % 16 == 0 and Hd % 4 == 0
            keep_t['W2ih'] = W2ihp
            x3 = bool(REC_X3 & 1) and use_c1 and not self.bf16 and Bp <= 64
            if x3:      # the attention LSTM's product on the bf16 pipe, exactly (three planes): K-slices of at most 512
                d.rec_x3 = 1
                d.S1 = rec_splits_x3(4 * Ha, K1, 192 if ((PLSTM & 2) and lib().gt_plstm_supported(Bp, Hd)) else 256)
            h2f = bool(REC_H2 & 1) and use_c1 and not self.bf16 and not x3 and not REC_TAIL_FWD     # (rec_tail rides on the fp32 kernel)
            if h2f:     # ... from two scaled fp16 planes: [ctx | h_att] is bounded by max |memory| and 1 / (1 - p)
                d.rec_x3 = 2
                d.S1 = rec_splits_x3(4 * Ha, K1, 192 if ((PLSTM & 2) and lib().gt_plstm_supported(Bp, Hd)) else 256)
                keep_t['x1max'] = ops.absmax(memory)
                d.x1max, d.x1floor = ptr(keep_t['x1max']), 1.0625 / (1.0 - d.p_att)
            keep_t['W1c' if use_c1 else 'W1p'] = pack_rec_cell(W1cat, Ha, K1, 3 if x3 else 2 if h2f else self.bf16) if use_c1 else \
                pack(W1cat, 4 * Ha, 4 * Ha, K1)
            keep_t['W2hc' if use_c2 else 'W2hp'] = pack_rec_cell(W2hh, Hd, Hd, self.bf16) if use_c2 else \
                pack(W2hh, 4 * Hd, 4 * Hd, Hd)
        elif self.bf16:
            d.bf16 = 1
            d.S1, d.S2 = rec_splits_bf16(4 * Ha, K1), rec_splits_bf16(4 * Hd, K2)
            keep_t.update(W1p=pack_rec_bf16(W1cat, 4 * Ha, 4 * Ha, K1), W2p=pack_rec_bf16(W2cat, 4 * Hd, 4 * Hd, K2))
        elif fused:
            keep_t.update(W1g=pack_gates(W1cat, Ha, K1), W2g=pack_gates(W2cat, Hd, K2))
        else:
            keep_t.update(W1p=pack_rec(W1cat, 4 * Ha, 4 * Ha, K1), W2p=pack_rec(W2cat, 4 * Hd, 4 * Hd, K2))
        # (the one-launch attention steps finish a batch row in its last VALID tile's workgroup: a row without a symbol has none)
        assert min(int(n) for n in in_lens) >= 1, 'every text needs at least one symbol'
        lens32 = ops.h2d(in_lens, torch.int32, dev)
        ld1, ld2 = (int(lib().gt_fused_k_padded(K1)), int(lib().gt_fused_k_padded(K2))) if fused else \
            (roundup(K1, 16), roundup(K2, 16))
        if hoist2:
            ld2 = roundup(Ha + Ep, 16)          # X2 is the (To, Bp, ld2) history of [h_att_t | ctx_t]
        # The per-step histories (5.2 GB at the benchmark batch, forward + backward) are written element for element by the
        # loop kernels for every row < B; they only have to start as zeros where the loops leave something unwritten that a
        # later GEMM reads: padded rows (Bp > B), padded context columns (Ep > E), padded leading dimensions.
        zh = not (Bp == B and Ep == E)
        dims_f = ops.attn_dims(B, Ti, Ep, A, self.F, self.Kc)
        # the h_att half of the attention LSTM's product as a tail of the attention step (REC_TAIL_FWD K-slices): fp32, <= 64 rows
        tail1 = REC_TAIL_FWD if (hoist2 and use_c1 and not self.bf16 and not x3 and Bp <= 64 and Ep % 16 == 0 and ATTN_STEP2 and
                                 lib().gt_attn_step2_supported(dims_f) and REC_TAIL_FWD <= Ha // 16) else 0
        if tail1:
            # K-slices of what stays in the LSTM launch (step 0 multiplies all of K with the same count)
            tkb = REC_TAIL_KB if 1 <= REC_TAIL_KB <= Ha // 16 else Ha // 16
            d.S1 = next(s_ for s_ in range(S1_FWD or d.S1, 0, -1)
                        if ops.pick_splits(K1 - 16 * tkb, s_) == s_ and ops.pick_splits(K1, s_) == s_)
            tail1 = ops.pick_splits(16 * tkb, min(tail1, 16 - d.S1))
            d.tail_kb = tkb
        ws = dict(X1=_f32(2, Bp, ld1, device=dev),
                  X2=_f32(To if hoist2 else DEC_RING, Bp, ld2, device=dev, zero=zh or not hoist2 or ld2 != Ha + Ep),
                  part=_f32(max((d.S1 + tail1) * Bp * 4 * Ha, d.S2 * Bp * 4 * Hd, d.Sq * Bp * Apad), device=dev, zero=False),
                  part2=_f32(d.S2 * Bp * 4 * Hd, device=dev, zero=False),
                  energies=_f32(B, Ti, device=dev, zero=False))
        hist = dict(gates1=_f32(To, Bp, 4 * Ha, device=dev, zero=zh), c1=_state(To + 1, Bp, Ha, dev, zh),
                    gates2=_f32(To, Bp, 4 * Hd, device=dev, zero=zh), c2=_state(To + 1, Bp, Hd, dev, zh),
                    hatt=_f32(To, Bp, Ha, device=dev, zero=zh), HC=_f32(To, Bp, Hd + Ep, device=dev, zero=zh),
                    w_hist=_f32(To + 1, B, Ti, device=dev), cum_hist=_f32(To + 1, B, Ti, device=dev),
                    q_hist=_f32(To, B, A, device=dev, zero=False))
        al = 'decoder.attention_layer.'
        d.memory, d.pm, d.lengths, d.pre1, d.bias2 = ptr(memory), ptr(pm), ptr(lens32), ptr(pre1), ptr(bias2)
        d.W1p, d.Wqp, d.W2p = ptr(keep_t.get('W1p')), ptr(keep_t['Wqp']), ptr(keep_t.get('W2p'))
        d.W1g, d.W2g = ptr(keep_t.get('W1g')), ptr(keep_t.get('W2g'))
        n_wpack = int(lib().gt_attn_wpack_elems(ops.attn_dims(B, Ti, Ep, A, self.F, self.Kc)))
        if n_wpack:
            ws['attn_wpack'] = _f32(n_wpack, device=dev, zero=False)
            d.attn_wpack = ptr(ws['attn_wpack'])
        if CELL_QUERY and Ha % 32 == 0:
            Wq = P['decoder.attention_layer.query_layer.linear_layer.weight']
            ws['partq'] = _f32(Ha // 16, Bp, Apad, device=dev, zero=False)      # Ha/32 slabs, Ha/16 with gt_rec_cell_fwd
            d.Wq, d.partq = ptr(Wq), ptr(ws['partq'])
        d.Wc = ptr(P[al + 'location_layer.location_conv.conv.weight'])
        d.Wd = ptr(P[al + 'location_layer.location_dense.linear_layer.weight'])
        d.v = ptr(P[al + 'v.linear_layer.weight'])
        att_keep = rng.get('att_keep') if training else None
        dec_keep = rng.get('dec_keep') if training else None
        d.att_keep, d.dec_keep = ptr(att_keep), ptr(dec_keep)
        d.X1, d.ld1, d.X2, d.ld2, d.part, d.energies = ptr(ws['X1']), ld1, ptr(ws['X2']), ld2, ptr(ws['part']), ptr(ws['energies'])
        d.part2, d.ring = ptr(ws['part2']), DEC_RING
        for k in ('gates1', 'c1', 'gates2', 'c2', 'hatt', 'HC', 'w_hist', 'cum_hist', 'q_hist'):
            setattr(d, k, ptr(hist[k]))
        if hoist2:
            ldh = roundup(Hd, 16)
            ws['Xh'] = _f32(2, Bp, ldh, device=dev)
            # W2hp only tells the loop to run in hoisted mode when the fused launch carries the product
            d.W2hp, d.W2ih, d.ldw2 = ptr(keep_t.get('W2hp', keep_t.get('W2hc'))), ptr(keep_t['W2ih']), keep_t['W2ih'].stride(0)
            d.pre2, d.Xh, d.ldh, d.blk = ptr(hist['gates2']), ptr(ws['Xh']), ldh, DEC_BLK
            nzb = Bp // 64 if Bp > 64 else 1
            if use_c1:
                ws['counters1'] = torch.zeros(nzb * (Ha // 16), dtype=torch.int32, device=dev)
                d.W1c, d.counters1 = ptr(keep_t['W1c']), ptr(ws['counters1'])
            if use_c2:
                ws['counters2'] = torch.zeros(nzb * (Hd // 16), dtype=torch.int32, device=dev)
                d.W2hc, d.counters2 = ptr(keep_t['W2hc']), ptr(ws['counters2'])
        if hoist2 and (PLSTM & 2) and not self.bf16 and lib().gt_plstm_supported(Bp, Hd):
            keep_t['W2hh_raw'] = W2hh.contiguous()
            ws['pl_counters'] = torch.zeros(1, dtype=torch.int32, device=dev)
            d.W2hh, d.pl_counters, d.pl_timeout = ptr(keep_t['W2hh_raw']), ptr(ws['pl_counters']), ptr(handoff_flag(dev))
        if tail1:
            ws['tail_work'] = torch.zeros(To, dtype=torch.int32, device=dev)
            d.tail_work, d.Sx1 = ptr(ws['tail_work']), tail1
        if hoist2 and ATTN_STEP2 and lib().gt_attn_step2_supported(dims_f):
            nch = int(lib().gt_attn_chunks(Ti))
            ws['row_cnt'] = torch.zeros(B, dtype=torch.int32, device=dev)
            ws['attn_xbuf'] = _f32(B * nch * int(lib().gt_attn_step2_xstride(dims_f)), device=dev, zero=False)
            d.row_cnt, d.attn_xbuf = ptr(ws['row_cnt']), ptr(ws['attn_xbuf'])
        elif hoist2 and ATTN_STEP and lib().gt_attn_step_supported(ops.attn_dims(B, Ti, Ep, A, self.F, self.Kc)):
            ws['row_cnt'] = torch.zeros(B, dtype=torch.int32, device=dev)
            d.row_cnt, d.attn_timeout = ptr(ws['row_cnt']), ptr(handoff_flag(dev))
        if ATTN_HOIST and training and lib().gt_attn_hoist_supported(ops.attn_dims(B, Ti, Ep, A, self.F, self.Kc)):
            hist['u_hist'] = _f32(To, B, Ti, A, device=dev, zero=False)
            d.u_hist = ptr(hist['u_hist'])
        PHASES.mark('f.pre_loop')
        _t0 = _time.perf_counter()
        _run('gt_decoder_fwd', d)
        if HOST_TIMING:
            print(f'[host] gt_decoder_fwd issued in {1e3 * (_time.perf_counter() - _t0):.1f} ms (To={To})', flush=True)
        PHASES.mark('f.dec_loop')
        sv.update(hist=hist, pre1=pre1, lens32=lens32, att_keep=att_keep, dec_keep=dec_keep,
                  prenet_keep=pk)
        if training and WTERMS_IN_FORWARD:
            early = self._attn_wterms(P, sv, hist, in_forward=True)
            if early is not None:
                sv['attn_early'] = early

        # ---- mel / gate projection of all steps, output layouts
        hc = Hd + Ep
        bpg = torch.cat((P['decoder.linear_projection.linear_layer.bias'], P['decoder.gate_layer.linear_layer.bias']))
        nout = n_mel + 1
        out81 = ops.linear(hist['HC'], Wpg, To * Bp, hc, nout, bias=bpg)
        mel_cl = _f32(B * To, n_mel, device=dev, zero=False)        # (B, To, n_mel)
        call('gt_permute3', out81, nout, Bp * nout, 1, B, To, n_mel, mel_cl, 0, 0, 0, 1.0)
        mel = _f32(B, n_mel, To, device=dev, zero=False)
        call('gt_permute3', out81, nout, 1, Bp * nout, B, n_mel, To, mel, 0, 0, 0, 1.0)
        gate = _f32(B, To, device=dev, zero=False)
        check(lib().gt_permute3(ptr(out81, n_mel), 0, nout, Bp * nout, 1, B, To, ptr(gate), 0, 0, 0, 1.0,
                                ops._stream()), 'gt_permute3')
        align = _f32(B, To, Ti, device=dev, zero=False)
        check(lib().gt_permute3(ptr(hist['w_hist'], B * Ti), Ti, B * Ti, 1, B, To, Ti, ptr(align), 0, 0, 0, 1.0,
                                ops._stream()), 'gt_permute3')
        sv['Wpg'] = Wpg

        PHASES.mark('f.proj')
        # ---- PostNet + residual + padding mask
        sv['post_conv'] = []
        post_cl = self.post_stack.forward(P, mel_cl, B, To, rng.get('post_keep'), training, sv['post_conv'])
        mel_post = mel.clone()
        call('gt_permute3', post_cl, To * n_mel, 1, n_mel, B, n_mel, To, mel_post, 0, 0, 1, 1.0)
        if mask_padding:
            ol = ops.h2d(out_lens, torch.int64, dev)
            call('gt_mask_outputs', mel, mel_post, gate, ol, B, n_mel, To)
            # the reference masks `mel` in place after the PostNet consumed it: the first PostNet
            # conv's weight gradient therefore sees the MASKED frames (model.py:638-640)
            mel_cl_masked = _f32(B * To, n_mel, device=dev, zero=False)
            call('gt_permute3', mel, n_mel * To, 1, To, B, To, n_mel, mel_cl_masked, 0, 0, 0, 1.0)
            sv['mel_cl_masked'] = mel_cl_masked
        PHASES.mark('f.postnet')
        return (mel, mel_post, gate, align), sv

    # ------------------------------------------------------------------------------ inference
    @_gemm_precision
    def inference(self, P, text, style, prenet_keep, speaker_ids=None, emotions=None, gate_threshold=0.5,
                  max_decoder_steps=500, chunk=32):
        """Tacotron2.inference (model.py:675-723): evaluation-mode encoder (no packing: every row is Ti long),
        free-running decoder (gt_decoder_infer) and PostNet.  style: (B, noise) or (B, Ti, noise) or None when
        the model has no noise input; prenet_keep: two (max_decoder_steps, Bp, P) uint8 keep masks.  The host
        reads the stop flags every `chunk` steps; the reference decides after every step (:486-491) and keeps
        the stopping frame, so the returned tensors are cut at the first stop + 1.  For B > 1 (the reference
        only supports B = 1 here) decoding ends when every item has crossed the threshold.
        Returns mel (B,n_mel,T), mel_post (B,n_mel,T), gate (B,T,1), align (B,T,Ti)."""
        hp, dev = self.hp, text.device
        B, Ti = text.shape
        Bp = roundup(B, 16) if B <= 64 else roundup(B, 64)
        n_mel, E, Ep, A, Pd, Ha, Hd = self.n_mel, self.E, self.Ep, self.A, self.P, self.Ha, self.Hd
        To = int(max_decoder_steps)
        sv = dict(B=B, Bp=Bp, Ti=Ti, To=To, in_lens=[Ti] * B, training=False, text=text)
        memory, mem2, pm = self._encode(P, text, [Ti] * B, {'noise': style}, speaker_ids, emotions, False, sv)

        W1ih, W1hh = P['decoder.attention_rnn.weight_ih'], P['decoder.attention_rnn.weight_hh']
        W2ih, W2hh = P['decoder.decoder_rnn.weight_ih'], P['decoder.decoder_rnn.weight_hh']
        bias1 = P['decoder.attention_rnn.bias_ih'] + P['decoder.attention_rnn.bias_hh']
        bias2 = P['decoder.decoder_rnn.bias_ih'] + P['decoder.decoder_rnn.bias_hh']
        K1, K2 = Ep + Ha, Ha + Ep + Hd
        W1cat, W2ihp, Wpg = self._decoder_operands(P)
        W2cat = torch.cat((W2ihp, W2hh), dim=1).contiguous()
        Apad = roundup(A, 16)
        q = GtDecoderInfer()
        d = q.step
        d.B, d.Bp, d.Ti, d.To, d.E, d.ldE, d.P, d.Ha, d.Hd, d.A, d.Apad, d.F, d.Kc = \
            B, Bp, Ti, To, Ep, Ep, Pd, Ha, Hd, A, Apad, self.F, self.Kc
        d.S1, d.S2, d.Sq = rec_splits(4 * Ha, K1), rec_splits(4 * Hd, K2), ops.pick_splits(Ha, 16)
        d.p_att = d.p_dec = 0.0
        packed = dict(Wqp=pack_rec(P['decoder.attention_layer.query_layer.linear_layer.weight'], Apad, A, Ha))
        if self.bf16:
            d.bf16 = 1
            d.S1, d.S2 = rec_splits_bf16(4 * Ha, K1), rec_splits_bf16(4 * Hd, K2)
            packed.update(W1p=pack_rec_bf16(W1cat, 4 * Ha, 4 * Ha, K1), W2p=pack_rec_bf16(W2cat, 4 * Hd, 4 * Hd, K2))
        else:
            packed.update(W1p=pack_rec(W1cat, 4 * Ha, 4 * Ha, K1), W2p=pack_rec(W2cat, 4 * Hd, 4 * Hd, K2))
        lens32 = torch.full((B,), Ti, dtype=torch.int32, device=dev)
        ld1, ld2 = roundup(K1, 16), roundup(K2, 16)
        nout, hc = n_mel + 1, Hd + Ep
        ws = dict(X1=_f32(Bp, ld1, device=dev), X2=_f32(Bp, ld2, device=dev),
                  part=_f32(max(d.S1 * Bp * 4 * Ha, d.S2 * Bp * 4 * Hd, d.Sq * Bp * Apad), device=dev, zero=False),
                  energies=_f32(B, Ti, device=dev, zero=False), pre1=_f32(Bp, 4 * Ha, device=dev),
                  p1=_f32(Bp, Pd, device=dev), p2=_f32(Bp, Pd, device=dev),
                  gates1=_f32(To, Bp, 4 * Ha, device=dev, zero=False), c1=_f32(To + 1, Bp, Ha, device=dev),
                  gates2=_f32(To, Bp, 4 * Hd, device=dev, zero=False), c2=_f32(To + 1, Bp, Hd, device=dev),
                  hatt=_f32(To, Bp, Ha, device=dev, zero=False), HC=_f32(To, Bp, hc, device=dev),
                  w_hist=_f32(To + 1, B, Ti, device=dev), cum_hist=_f32(To + 1, B, Ti, device=dev),
                  q_hist=_f32(To, B, A, device=dev, zero=False), frames=_f32(To + 1, Bp, nout, device=dev),
                  stop=torch.full((B,), 2 ** 31 - 1, dtype=torch.int32, device=dev))
        bpg = torch.cat((P['decoder.linear_projection.linear_layer.bias'], P['decoder.gate_layer.linear_layer.bias']))
        W0 = P['decoder.prenet.layers.0.linear_layer.weight'].contiguous()
        W1 = P['decoder.prenet.layers.1.linear_layer.weight'].contiguous()
        k0, k1 = prenet_keep
        assert tuple(k0.shape) == tuple(k1.shape) == (To, Bp, Pd) and k0.dtype == torch.uint8, \
            f'prenet keep masks must be ({To},{Bp},{Pd}) uint8'
        al = 'decoder.attention_layer.'
        d.memory, d.pm, d.lengths, d.pre1, d.bias2 = ptr(memory), ptr(pm), ptr(lens32), ptr(ws['pre1']), ptr(bias2)
        d.W1p, d.Wqp, d.W2p = ptr(packed['W1p']), ptr(packed['Wqp']), ptr(packed['W2p'])
        d.Wc = ptr(P[al + 'location_layer.location_conv.conv.weight'])
        d.Wd = ptr(P[al + 'location_layer.location_dense.linear_layer.weight'])
        d.v = ptr(P[al + 'v.linear_layer.weight'])
        d.att_keep = d.dec_keep = None
        d.X1, d.ld1, d.X2, d.ld2, d.part, d.energies = ptr(ws['X1']), ld1, ptr(ws['X2']), ld2, ptr(ws['part']), ptr(ws['energies'])
        d.part2, d.ring = None, 1
        for k in ('gates1', 'c1', 'gates2', 'c2', 'hatt', 'HC', 'w_hist', 'cum_hist', 'q_hist'):
            setattr(d, k, ptr(ws[k]))
        q.W0, q.W1, q.W1x, q.ldw1x, q.bias1 = ptr(W0), ptr(W1), ptr(W1ih), W1ih.stride(0), ptr(bias1)
        q.keep0, q.keep1, q.Wpg, q.bpg = ptr(k0), ptr(k1), ptr(Wpg), ptr(bpg)
        q.p1, q.p2, q.frames, q.stop_step = ptr(ws['p1']), ptr(ws['p2']), ptr(ws['frames']), ptr(ws['stop'])
        q.n_mel, q.gate_threshold = n_mel, float(gate_threshold)
        t, T = 0, To
        while t < To:
            q.t_lo, q.t_hi = t, min(t + chunk, To)
            _run('gt_decoder_infer', q)
            t = q.t_hi
            stop = ws['stop'].tolist()                      # device sync: once per chunk
            if max(stop) < t:
                T = max(stop) + 1
                break
        else:
            print('Warning! Reached max decoder steps')     # reference model.py:489
        frames = ws['frames'][1:T + 1, :B]                   # (T, B, n_mel+1)
        mel_cl = frames[:, :, :n_mel].permute(1, 0, 2).contiguous()          # (B, T, n_mel)
        mel = mel_cl.transpose(1, 2).contiguous()
        gate = frames[:, :, n_mel:].permute(1, 0, 2).contiguous()            # (B, T, 1)
        align = ws['w_hist'][1:T + 1].permute(1, 0, 2).contiguous()          # (B, T, Ti)
        post_cl = self.post_stack.forward(P, mel_cl.view(B * T, n_mel), B, T, None, False, [])
        mel_post = mel + post_cl.view(B, T, n_mel).transpose(1, 2)
        return mel, mel_post, gate, align

    # ------------------------------------------------------------------------------ backward
    @_gemm_precision
    def _attn_wterms(self, P, sv, hist, in_forward=False):
        """gt_attn_bwd_wterms for every step of the decoder loop in one launch on the leaf stream (2.7 GB read, 2.7 GB written at
        the benchmark batch): P rows and the s-terms of the one-launch backward attention step.  Returns the buffers + the
        event the backward loop waits for, or None when that form of the backward step is not in use.
        in_forward: issued by the training forward behind the decoder loop, so that the launch (7.6 ms beside the PostNet
        backward's GEMMs, of which 2.1 ms used to be exposed in front of the backward loop: profiles/r04_trace_wterms.txt)
        runs under the projection, the PostNet forward, the losses and the PostNet backward instead."""
        B, Bp, Ti, To = sv['B'], sv['Bp'], sv['Ti'], sv['To']
        A, Ha, Ep = self.A, self.Ha, self.Ep
        if not (ATTN_BWD_FUSED and 'u_hist' in hist and HOIST2 and not FUSED_CELL and CELL_QUERY and Ha % 16 == 0):
            return None
        if in_forward and (not LEAF.enabled or LEAF.defer):
            return None
        dims_e = ops.attn_dims(B, Ti, Ep, A, self.F, self.Kc)
        if not lib().gt_attn_bwd_step_supported(dims_e):
            return None
        dev = hist['u_hist'].device
        al_e = 'decoder.attention_layer.'
        Wc_e, Wd_e, v_e = (P[al_e + 'location_layer.location_conv.conv.weight'],
                           P[al_e + 'location_layer.location_dense.linear_layer.weight'], P[al_e + 'v.linear_layer.weight'])
        Apad_e, nch_e, BT_e = roundup(A, 16), int(lib().gt_attn_chunks(Ti)), B * Ti
        wpack_e = _f32(int(lib().gt_attn_wpack_elems(dims_e)), device=dev, zero=False)
        fz_e = dict(sum_hist=_f32(To, B, nch_e, device=dev, zero=False), dq2_hist=_f32(To, Bp, Apad_e, device=dev),
                    dwcat2_hist=_f32(To, 2, B, Ti, device=dev), p_hist=_f32(To, B, Ti, 64, device=dev, zero=False))

        def wterms():
            call('gt_attn_pack_weights', dims_e, Wc_e, Wd_e, wpack_e)
            check(lib().gt_attn_bwd_wterms(
                dims_e, To, ptr(hist['w_hist'], BT_e), BT_e, ptr(hist['u_hist']), ptr(Wc_e), ptr(Wd_e), ptr(v_e),
                ptr(sv['lens32']), ptr(fz_e['dq2_hist']), Bp * Apad_e, Apad_e, ptr(fz_e['dwcat2_hist']), ptr(fz_e['p_hist']),
                ptr(wpack_e), ops._stream()), 'gt_attn_bwd_wterms')
        if in_forward:
            # not LEAF.run: nothing joins the leaf stream at the end of a forward, so the buffers are handed to the
            # allocator's stream bookkeeping instead of LEAF.keep (a forward that is never followed by a backward frees them
            # with `sv`)
            if LEAF.stream is None:
                LEAF.stream = torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[0])
            LEAF.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(LEAF.stream):
                wterms()
            for t_ in (wpack_e, *fz_e.values(), hist['u_hist'], hist['w_hist']):
                t_.record_stream(LEAF.stream)
        else:
            LEAF.run(wterms, wpack_e)
            if LEAF.defer:
                LEAF.flush()
        ready = torch.cuda.Event()
        ready.record(LEAF.stream if (LEAF.enabled and LEAF.stream is not None and not LEAF.defer) else torch.cuda.current_stream())
        return dict(fz=fz_e, wpack=wpack_e, ready=ready)

    def backward(self, P, G, sv, dmel, dmel_post, dgate, dalign, on_ready=None):
        """Generator backward (see _backward).  Whatever happens inside -- a GantronHipError from a loop call included --
        the split-K target of the overlapped region is restored and the leaf stream is joined before the buffers it
        reads can be recycled."""
        target = ops.SPLITK_TARGET
        try:
            return self._backward(P, G, sv, dmel, dmel_post, dgate, dalign, on_ready)
        finally:
            ops.SPLITK_TARGET = target
            LEAF.join()

    def _backward(self, P, G, sv, dmel, dmel_post, dgate, dalign, on_ready=None):
        """Accumulate parameter gradients into G (dict name -> tensor).  d* are gradients of the four
        outputs (any may be None).  on_ready(prefix, event): called when the gradients of every parameter under
        `prefix` are final, `event` being recorded behind the last kernel that writes them (the data-parallel wrapper
        starts that bucket's all-reduce while the decoder loop still runs).  (Running the whole backward on a high-priority stream so that the loop kernels win
        the CUs over the leaf stream's weight-gradient GEMMs was measured: no gain on MI355X.)"""
        hp = self.hp
        B, Bp, Ti, To = sv['B'], sv['Bp'], sv['Ti'], sv['To']
        n_mel, H, E, Ep, A, Pd, Ha, Hd = self.n_mel, self.H, self.E, self.Ep, self.A, self.P, self.Ha, self.Hd
        dev = sv['memory'].device
        hist = sv['hist']
        hc, nout = Hd + Ep, n_mel + 1          # [h_dec | ctx] rows are Ep wide in their ctx part (engine.Ep)

        PHASES.mark(None)
        # ---- the one-launch backward attention step's per-position terms of ALL steps (gt_attn_bwd_wterms) need forward
        # quantities only.  Round 4: the training forward issues them itself, right behind the decoder loop (`_attn_wterms`,
        # sv['attn_early']); a forward that did not (leaf stream off / deferred) gets them here, beside the PostNet backward.
        # The loop later waits for their own event, not for the whole leaf stream.
        early = sv.pop('attn_early', None)
        if early is None:
            early = self._attn_wterms(P, sv, hist)
        # ---- PostNet backward (channels-last)
        dmel_cl = _f32(B * To, n_mel, device=dev)
        if dmel is not None:
            call('gt_permute3', dmel.contiguous(), n_mel * To, 1, To, B, To, n_mel, dmel_cl, 0, 0, 1, 1.0)
        if dmel_post is not None:
            dpost_cl = _f32(B * To, n_mel, device=dev, zero=False)
            call('gt_permute3', dmel_post.contiguous(), n_mel * To, 1, To, B, To, n_mel, dpost_cl, 0, 0, 0, 1.0)
            call('gt_axpby', dmel_cl, 1.0, dpost_cl, 1.0, dmel_cl, B * To * n_mel)      # residual branch
            dx = self.post_stack.backward(P, G, dpost_cl, B, To, sv['post_conv'],
                                          first_input_override=sv.get('mel_cl_masked'))
            call('gt_axpby', dmel_cl, 1.0, dx, 1.0, dmel_cl, B * To * n_mel)

        PHASES.mark('b.postnet')
        if on_ready is not None and dmel_post is not None:
            # the PostNet's gradients are complete long before the decoder loop ends: weight gradients come off the leaf
            # stream, BatchNorm gradients off this one
            ev = torch.cuda.Event()
            LEAF.flush()
            if LEAF.enabled and LEAF.stream is not None:
                LEAF.stream.wait_stream(torch.cuda.current_stream())
                ev.record(LEAF.stream)
            else:
                ev.record(torch.cuda.current_stream())
            on_ready('postnet.', ev)
        # weight-gradient GEMMs issued from here to the end of the decoder loop run UNDER that loop (leaf stream)
        gentle = ops.splitk_target(ops.SPLITK_OVERLAPPED)
        gentle.__enter__()
        # ---- projection backward: gradient w.r.t. [h_dec | ctx] of every step
        dout = _f32(To, Bp, nout, device=dev)
        call('gt_permute3', dmel_cl, n_mel, To * n_mel, 1, To, B, n_mel, dout, Bp * nout, nout, 0, 1.0)
        if dgate is not None:
            check(lib().gt_permute3(ptr(dgate.contiguous()), 1, To, 0, To, B, 1, ptr(dout, n_mel), Bp * nout, nout, 0, 1.0,
                                    ops._stream()), 'gt_permute3')
        def leaf_proj():
            dWpg = _f32(nout, hc, device=dev)
            ops.linear_dw(dout, hist['HC'], To * Bp, nout, hc, dWpg)
            if Ep > E:                           # drop the (zero-input) pad columns
                dWpg = dWpg[:, :Hd + E].contiguous()
            acc_add(G['decoder.linear_projection.linear_layer.weight'], dWpg[:n_mel])
            acc_add(G['decoder.gate_layer.linear_layer.weight'], dWpg[n_mel:])
            bsum = ops.col_sums(dout, To * Bp, nout)
            acc_add(G['decoder.linear_projection.linear_layer.bias'], bsum[:n_mel])
            acc_add(G['decoder.gate_layer.linear_layer.bias'], bsum[n_mel:])
        LEAF.run(leaf_proj, dout)
        dHC = ops.linear_dx(dout, sv['Wpg'], To * Bp, nout, hc)

        # ---- decoder loop backward
        dalign_tm = None
        if dalign is not None:
            dalign_tm = _f32(To, B, Ti, device=dev, zero=False)
            call('gt_permute3', dalign.contiguous(), Ti, To * Ti, 1, To, B, Ti, dalign_tm, 0, 0, 0, 1.0)
        K1, K2 = Ep + Ha, Ha + Ep + Hd
        N1pad, N2pad, Apad = roundup(K1, 16), roundup(K2, 16), roundup(A, 16)
        W1ih, W1hh = P['decoder.attention_rnn.weight_ih'], P['decoder.attention_rnn.weight_hh']
        W2ih, W2hh = P['decoder.decoder_rnn.weight_ih'], P['decoder.decoder_rnn.weight_hh']
        W1cat, W2ihp, _ = self._decoder_operands(P)
        W2cat = None if (HOIST2 and not FUSED_CELL) else torch.cat((W2ihp, W2hh), dim=1).contiguous()
        Wq = P['decoder.attention_layer.query_layer.linear_layer.weight']
        packT = pack_rec_bf16 if self.bf16 else pack_rec
        hoist2 = HOIST2 and not FUSED_CELL
        Hdpad = roundup(Hd, 16)
        x3b = bool(REC_X3 & 2) and not self.bf16 and Bp <= 64 and N1pad % 64 == 0
        # two scaled fp16 planes: only with the attention step that carries the cell backward (it leaves max |dG1_t|) -- the
        # conditions of `fused_attn` + ATTN_BWD_CELL below
        h2b = (bool(REC_H2 & 2) and not self.bf16 and not x3b and ATTN_BWD_CELL and early is not None and 'u_hist' in hist and hoist2 and
               N1pad % 16 == 0 and Ha % 16 == 0 and Ep % 4 == 0 and CELL_QUERY)
        W1Tp = (pack_rec_x3 if x3b else pack_rec_h2 if h2b else packT)(W1cat, N1pad, K1, 4 * Ha, transposed=True)
        W2Tp = None if hoist2 else packT(W2cat, N2pad, K2, 4 * Hd, transposed=True)
        W2hTp = packT(W2hh, Hdpad, Hd, 4 * Hd, transposed=True) if hoist2 else None
        WqTp = pack_rec(Wq, Ha, Ha, A, transposed=True)
        b = GtDecoderBwd()
        b.B, b.Bp, b.Ti, b.To, b.E, b.ldE, b.P, b.Ha, b.Hd, b.A, b.Apad, b.F, b.Kc = \
            B, Bp, Ti, To, Ep, Ep, Pd, Ha, Hd, A, Apad, self.F, self.Kc
        b.S1, b.S2, b.Sq = rec_splits(N1pad, 4 * Ha), rec_splits(N2pad, 4 * Hd), ops.pick_splits(A, 2)
        if self.bf16:
            b.bf16 = 1
            b.S1, b.S2 = rec_splits_bf16(N1pad, 4 * Ha), rec_splits_bf16(N2pad, 4 * Hd)
        if hoist2:
            b.S2 = (rec_splits_bf16 if self.bf16 else rec_splits)(Hdpad, 4 * Hd)
            if S2_BWD and not self.bf16:
                b.S2 = ops.pick_splits(4 * Hd, S2_BWD)
        if x3b:
            b.rec_x3 = 1
            b.S1 = rec_splits_x3(N1pad, 4 * Ha)
        if h2b:
            b.rec_x3 = 2
            b.S1 = rec_splits_x3(N1pad, 4 * Ha, 256)        # (8 splits at N1pad = 2048: measured best of 4 / 6 / 8 / 10 / 12 / 16)
        if S1_BWD and not self.bf16:
            b.S1 = S1_BWD
        b.N1pad, b.N2pad = N1pad, N2pad
        if CELL_QUERY and Ha % 16 == 0:
            b.Wq = ptr(P['decoder.attention_layer.query_layer.linear_layer.weight'])
        n_wpack = int(lib().gt_attn_wpack_elems(ops.attn_dims(B, Ti, Ep, A, self.F, self.Kc)))
        wpack_b = _f32(n_wpack, device=dev, zero=False) if n_wpack else None
        b.attn_wpack = ptr(wpack_b)
        training = sv['training']
        b.p_att = hp.p_attention_dropout if training else 0.0
        b.p_dec = hp.p_decoder_dropout if training else 0.0
        al = 'decoder.attention_layer.'
        Wc, Wd, v = (P[al + 'location_layer.location_conv.conv.weight'],
                     P[al + 'location_layer.location_dense.linear_layer.weight'], P[al + 'v.linear_layer.weight'])
        dims = ops.attn_dims(B, Ti, Ep, A, self.F, self.Kc)
        zh = not (Bp == B)          # as in forward(): the gate-gradient histories are written for every row < B
        buf = dict(dG1=_f32(To, Bp, 4 * Ha, device=dev, zero=zh), dG2=_f32(To, Bp, 4 * Hd, device=dev, zero=zh),
                   dctx_hist=_f32(To, B, Ep, device=dev, zero=False), dpm=_f32(B, Ti, A, device=dev),
                   acc=_f32(int(lib().gt_attn_acc_elems(dims)), device=dev),
                   dc1=_f32(Bp, Ha, device=dev), dc2=_f32(Bp, Hd, device=dev), dwcat=_f32(2, 2, B, Ti, device=dev),
                   dcum_acc=_f32(B, Ti, device=dev), dq_hist=_f32(To, Bp, Apad, device=dev),
                   dw=_f32(B, Ti, device=dev, zero=False),
                   partX1=_f32(b.S1, Bp, N1pad, device=dev),
                   partX2=None if hoist2 else _f32(DEC_RING, b.S2, Bp, N2pad, device=dev),
                   partQ=_f32(b.Sq, Bp, Ha, device=dev))
        if h2b:
            buf['amaxG1'] = torch.zeros(To, dtype=torch.int64, device=dev)
        if hoist2:
            ldxa = roundup(Ha + Ep, 4)
            W2ih_c = W2ihp
            buf.update(dXA=_f32(To, Bp, ldxa, device=dev, zero=False), partXh=_f32(b.S2, Bp, Hdpad, device=dev))
            b.W2hTp, b.W2ih, b.ldw2, b.ldxa, b.blk = ptr(W2hTp), ptr(W2ih_c), W2ih_c.stride(0), ldxa, DEC_BLK_BWD
            if (REC_CELL & 32) and not self.bf16 and Hd % 64 == 0 and b.S2 <= 16:
                buf['countersB'] = torch.zeros((Bp // 64 if Bp > 64 else 1) * (Hd // 64), dtype=torch.int32, device=dev)
            if (PLSTM & 8) and not self.bf16 and lib().gt_plstm_supported(Bp, Hd) and hc % 4 == 0:
                # the decoder LSTM's backward recurrence of a block of steps as ONE persistent launch (gt_plstm_bwd)
                buf.update(W2hh=W2hh.contiguous(), pl_xbuf=_f32(int(lib().gt_plstm_bwd_xbuf_elems(Hd)), device=dev, zero=False),
                           pl_counters=torch.zeros(1, dtype=torch.int32, device=dev))
                b.pl_timeout = ptr(handoff_flag(dev))
        b.memory, b.pm, b.Wc, b.Wd, b.v = ptr(sv['memory']), ptr(sv['pm']), ptr(Wc), ptr(Wd), ptr(v)
        b.lengths = ptr(sv['lens32'])
        b.att_keep, b.dec_keep = ptr(sv['att_keep']), ptr(sv['dec_keep'])
        for k in ('gates1', 'c1', 'gates2', 'c2', 'w_hist', 'cum_hist', 'q_hist'):
            setattr(b, k, ptr(hist[k]))
        b.W1Tp, b.W2Tp, b.WqTp, b.dHC, b.dalign = ptr(W1Tp), ptr(W2Tp), ptr(WqTp), ptr(dHC), ptr(dalign_tm)
        for k in buf:
            setattr(b, k, ptr(buf[k]))
        b.ring = DEC_RING
        hoist = 'u_hist' in hist
        if hoist:
            # the loop computes dq / d[w_prev; w_cum] / de only; dv, dW_loc and dpm of a finished chunk of steps are
            # summed on the leaf stream by gt_attn_param_grads (needs the packed [Wc | Wd] image: made here, once)
            de_hist = _f32(To, B, Ti, device=dev, zero=False)
            b.u_hist, b.de_hist = ptr(hist['u_hist']), ptr(de_hist)
            if early is not None:
                wpack_p = early['wpack']
            else:
                wpack_p = _f32(n_wpack, device=dev, zero=False)
                call('gt_attn_pack_weights', dims, Wc, Wd, wpack_p)
            BT = B * Ti
        fused_attn = early is not None and hoist and hoist2 and N1pad % 4 == 0
        if fused_attn:
            # the terms of the energies backward that are linear in the step's own weights, for every step at once
            # (gt_attn_bwd_wterms, issued at the top of this function): what the per-step launch subtracts s times from
            nch = int(lib().gt_attn_chunks(Ti))
            fz = early['fz']
            torch.cuda.current_stream().wait_event(early['ready'])
            b.sum_hist, b.dq2_hist, b.dwcat2_hist, b.p_hist = (ptr(fz['sum_hist']), ptr(fz['dq2_hist']), ptr(fz['dwcat2_hist']),
                                                               ptr(fz['p_hist']))
            if ATTN_BWD_CELL and Ha % 4 == 0:
                buf['rowcntA'] = torch.zeros(B, dtype=torch.int32, device=dev)
                b.rowcntA = ptr(buf['rowcntA'])
        dG1, dG2 = buf['dG1'], buf['dG2']
        g1ih, g1hh = G['decoder.attention_rnn.weight_ih'], G['decoder.attention_rnn.weight_hh']
        g2ih, g2hh = G['decoder.decoder_rnn.weight_ih'], G['decoder.decoder_rnn.weight_hh']

        # Operands of the deferred weight-gradient GEMMs as bf16 planes (ops.Planes): the forward histories are split once,
        # on the leaf stream, under the PostNet / projection backward; a chunk's gate gradients when the chunk's loop
        # steps are done.  The GEMMs then run the kernel that has no split left to do (pgemm_kernel).
        pl = {}
        if ops.PLANE_GEMMS and not self.bf16 and ops.FP32_SPLIT == 3 and Bp % 8 == 0:
            def split_histories():
                for k, t in (('p2', sv['p2']), ('hatt', hist['hatt']), ('HC', hist['HC'])):
                    pl[k] = ops.Planes(t).split()
            LEAF.run(split_histories, sv['p2'], hist['hatt'], hist['HC'])
            pl['dG1'], pl['dG2'] = ops.Planes(dG1), ops.Planes(dG2)

        dG1_words = []                # ops.absmax words of the chunks of dG1 (fp16-plane products): the prenet's backward reads all of it

        def leaf_decoder(lo, hi):
            """Weight gradients contributed by decoder steps lo..hi (dW = dGates^T . inputs, K = rows)."""
            n, o = (hi - lo + 1) * Bp, lo * Bp
            if pl:
                pl['dG1'].split(o * 4 * Ha, (o + n) * 4 * Ha)
                pl['dG2'].split(o * 4 * Hd, (o + n) * 4 * Hd)
            g1p, g2p, p2p, hap, hcp = (pl.get(k) for k in ('dG1', 'dG2', 'p2', 'hatt', 'HC'))
            # fp16-plane products: the chunk's gate gradients enter three products each, its [h_dec | ctx] and h_att rows up
            # to three -- one pass over each for the maximum instead of one per product (rows of the step in front of the
            # chunk included: the recurrent inputs' products read those; a superset's maximum bounds the operand as well)
            a1 = a2 = ahc = aha = None
            if ops.h2_active():
                lo_ = max(lo - 1, 0)
                a1 = ops.absmax(dG1, n, 4 * Ha, 4 * Ha, offset=o * 4 * Ha)
                a2 = ops.absmax(dG2, n, 4 * Hd, 4 * Hd, offset=o * 4 * Hd)
                ahc = ops.absmax(hist['HC'], (hi - lo_ + 1) * Bp, hc, hc, offset=lo_ * Bp * hc)
                aha = ops.absmax(hist['hatt'], (hi - lo_ + 1) * Bp, Ha, Ha, offset=lo_ * Bp * Ha)
                dG1_words.append(a1)
            ops.linear_dw(dG1, sv['p2'], n, 4 * Ha, Pd, g1ih, dy_offset=o * 4 * Ha, x_offset=o * Pd, ldc=Pd + E,
                          dy_planes=g1p, x_planes=p2p, dy_amax=a1)
            ops.linear_dw(dG2, hist['hatt'], n, 4 * Hd, Ha, g2ih, dy_offset=o * 4 * Hd, x_offset=o * Ha, ldc=Ha + E,
                          dy_planes=g2p, x_planes=hap, dy_amax=a2, x_amax=aha)
            ops.linear_dw(dG2, hist['HC'], n, 4 * Hd, E, g2ih, dy_offset=o * 4 * Hd, ldx=hc, x_offset=o * hc + Hd,
                          ldc=Ha + E, c_offset=Ha, dy_planes=g2p, x_planes=hcp, dy_amax=a2, x_amax=ahc)
            l1 = max(lo, 1)                      # pairs (t, t-1): gradients of the recurrent inputs
            if hi >= l1:
                n1, o1, p1_ = (hi - l1 + 1) * Bp, l1 * Bp, (l1 - 1) * Bp
                ops.linear_dw(dG1, hist['HC'], n1, 4 * Ha, E, g1ih, dy_offset=o1 * 4 * Ha, ldx=hc,
                              x_offset=p1_ * hc + Hd, ldc=Pd + E, c_offset=Pd, dy_planes=g1p, x_planes=hcp, dy_amax=a1, x_amax=ahc)
                ops.linear_dw(dG1, hist['hatt'], n1, 4 * Ha, Ha, g1hh, dy_offset=o1 * 4 * Ha, x_offset=p1_ * Ha,
                              dy_planes=g1p, x_planes=hap, dy_amax=a1, x_amax=aha)
                ops.linear_dw(dG2, hist['HC'], n1, 4 * Hd, Hd, g2hh, dy_offset=o1 * 4 * Hd, ldx=hc, x_offset=p1_ * hc,
                              dy_planes=g2p, x_planes=hcp, dy_amax=a2, x_amax=ahc)
            s1 = ops.col_sums(dG1[lo:hi + 1], n, 4 * Ha)
            acc_add(G['decoder.attention_rnn.bias_ih'], s1)
            acc_add(G['decoder.attention_rnn.bias_hh'], s1)
            s2 = ops.col_sums(dG2[lo:hi + 1], n, 4 * Hd)
            acc_add(G['decoder.decoder_rnn.bias_ih'], s2)
            acc_add(G['decoder.decoder_rnn.bias_hh'], s2)
            ops.linear_dw(buf['dq_hist'], hist['hatt'], n, A, Ha, G[al + 'query_layer.linear_layer.weight'], lddy=Apad,
                          dy_offset=o * Apad, x_offset=o * Ha)

        PHASES.mark('b.pre_loop')
        # the loop is issued in chunks; the weight-gradient GEMMs of a finished chunk run on the leaf stream
        # while the (latency-bound) loop continues with earlier steps
        n_chunks = max(1, min(BWD_CHUNKS, To // 32))
        bounds = [round(i * To / n_chunks) for i in range(n_chunks + 1)]
        for ci in reversed(range(n_chunks)):
            lo, hi = bounds[ci], bounds[ci + 1] - 1
            b.t_hi, b.t_lo = hi, lo
            _t0 = _time.perf_counter()
            _run('gt_decoder_bwd', b)
            if HOST_TIMING:
                print(f'[host] gt_decoder_bwd steps {hi}..{lo} issued in {1e3 * (_time.perf_counter() - _t0):.1f} ms', flush=True)
            if fused_attn:      # de / dq of the chunk's steps still lack their row-sum terms
                LEAF.run(lambda lo=lo, hi=hi: check(lib().gt_attn_bwd_fixup(
                    dims, hi - lo + 1, ptr(fz['sum_hist'], lo * B * nch), ptr(hist['w_hist'], (lo + 1) * BT), BT,
                    ptr(de_hist, lo * BT), ptr(buf['dq_hist'], lo * Bp * Apad), Bp * Apad, Apad,
                    ptr(fz['dq2_hist'], lo * Bp * Apad), ops._stream()), 'gt_attn_bwd_fixup'), de_hist, fz['sum_hist'], fz['dq2_hist'])
            if hoist:
                LEAF.run(lambda lo=lo, hi=hi: check(lib().gt_attn_param_grads(
                    dims, hi - lo + 1, ptr(hist['u_hist'], lo * BT * A), ptr(de_hist, lo * BT), ptr(hist['w_hist'], lo * BT),
                    ptr(hist['cum_hist'], lo * BT), BT, ptr(Wc), ptr(Wd), ptr(v), ptr(sv['lens32']), ptr(buf['dpm']),
                    ptr(buf['acc']), ptr(wpack_p), ops._stream()), 'gt_attn_param_grads'), de_hist, wpack_p)
                if ci == 0:                       # dpm is complete: the main stream reads it after the loop
                    dpm_ready = torch.cuda.Event()
                    dpm_ready.record(LEAF.stream if (LEAF.enabled and LEAF.stream is not None) else torch.cuda.current_stream())
            LEAF.run(lambda lo=lo, hi=hi: leaf_decoder(lo, hi), dHC)
        if LEAF.defer:
            LEAF.flush()                          # (dpm below comes out of the queued leaves)
        if hoist:
            torch.cuda.current_stream().wait_event(dpm_ready)
        LEAF.run(lambda: call('gt_attn_acc_reduce', dims, buf['acc'], G[al + 'v.linear_layer.weight'],
                              G[al + 'location_layer.location_dense.linear_layer.weight'],
                              G[al + 'location_layer.location_conv.conv.weight']), buf['acc'])
        rows = To * Bp
        gentle.__exit__()
        PHASES.mark('b.dec_loop+dW')

        # ---- prenet backward: a leaf of the graph as a whole (nothing below depends on it) -- it runs on the leaf stream
        # beside d(memory), the BiLSTM backward and the encoder convolutions, which are latency-bound launch chains
        def leaf_prenet():
            # (the maximum of ALL gate gradients = the largest of the chunks' words: they carry the same tag, so the integer
            # maximum of the words is the word of the maximum -- no second pass over 912 MB at the tail of the step)
            a_all = torch.stack(dG1_words).max().reshape(1) if dG1_words and ops.h2_active() else None
            dp2 = ops.linear_dx(dG1, W1ih, rows, 4 * Ha, Pd, ldw=Pd + E, dy_amax=a_all)
            pk = sv['prenet_keep']
            call('gt_bias_act_drop_bwd', dp2, sv['p2'], rows, Pd, ACT_RELU, 0, pk[1], 0.5, dp2)
            ops.linear_dw(dp2, sv['p1'], rows, Pd, Pd, G['decoder.prenet.layers.1.linear_layer.weight'])
            dp1 = ops.linear_dx(dp2, P['decoder.prenet.layers.1.linear_layer.weight'], rows, Pd, Pd)
            call('gt_bias_act_drop_bwd', dp1, sv['p1'], rows, Pd, ACT_RELU, 0, pk[0], 0.5, dp1)
            ops.linear_dw(dp1, sv['x0'], rows, Pd, n_mel, G['decoder.prenet.layers.0.linear_layer.weight'])
        LEAF.run(leaf_prenet, dG1)

        # ---- d(memory) = sum_t w_t^T dctx_t (batched over b) + dpm @ Wmem ;  dWmem = dpm^T memory
        Wmem = P[al + 'memory_layer.linear_layer.weight']
        dmem = _f32(B, Ti, Ep, device=dev, zero=False)
        sgemm(Operand(hist['w_hist'], RC, ld=B * Ti, offset=B * Ti, batch_stride=Ti),
              Operand(buf['dctx_hist'], RC, ld=B * Ep, batch_stride=Ep), dmem, Ti, Ep, To, batch=B, c_batch_stride=Ti * Ep)
        sgemm(Operand(buf['dpm'], KC, ld=A), Operand(Wmem, RC, ld=E), dmem, B * Ti, E, A, ldc=Ep, accumulate=True,
              bf16=ops.BF16_GEMMS)          # input gradient of the memory layer (a layer GEMM)
        LEAF.run(lambda: ops.linear_dw(buf['dpm'], sv['memory'], B * Ti, A, E, G[al + 'memory_layer.linear_layer.weight'],
                                       ldx=Ep), buf['dpm'])

        _tap('dmem', dmem)
        col = 2 * H
        if self.vesus:
            se = hp.speakers_embedding
            # d speaker embedding = sum over Ti of dmem[:, :, col:col+se]  (rows = Ti per item)
            tmp = _f32(Ti, B, se, device=dev, zero=False)
            check(lib().gt_permute3(ptr(dmem, col), Ep, Ti * Ep, 1, Ti, B, se, ptr(tmp), 0, 0, 0, 1.0, ops._stream()),
                  'gt_permute3')
            dspk = ops.col_sums(tmp, Ti, B * se).contiguous()
            call('gt_embedding_bwd', sv['speaker_ids'], dspk, se, B, se, G['speaker_embedding.weight'])

        PHASES.mark('b.prenet_dmem')
        # ---- BiLSTM backward
        idx = sv['idx']
        C = self.enc_dim
        dhs = _f32(2, Ti * Bp, H, device=dev, zero=False)
        dm2 = dmem.view(B * Ti, Ep)
        call('gt_gather_rows', dm2, Ep, idx['fwd'], Ti * Bp, H, dhs[0], H, 0)
        check(lib().gt_gather_rows(ptr(dmem, H), Ep, ptr(idx['rev']), Ti * Bp, H, ptr(dhs[1]), H, 0, ops._stream()),
              'gt_gather_rows')
        dc = _f32(2, Bp, H, device=dev)
        if (PLSTM & 4) and lib().gt_plstm_supported(Bp, H):
            # the whole backward recurrence of both directions as ONE persistent launch (csrc/plstm.hip)
            Whh = torch.stack([P['encoder.lstm.weight_hh_l0' + sfx] for sfx in ('', '_reverse')]).contiguous()
            dG = _f32(2, Ti * Bp, 4 * H, device=dev, zero=False)
            xbuf = _f32(2, lib().gt_plstm_bwd_xbuf_elems(H), device=dev, zero=False)
            counters = torch.zeros(2, dtype=torch.int32, device=dev)
            q = GtPlstmBwd()
            q.B, q.Bp, q.H, q.T, q.ndir, q.s_lo, q.s_hi, q.recurrent_in = B, Bp, H, Ti, 2, 0, Ti, 0
            q.Whh, q.w_dstride, q.gates, q.gates_dstride = ptr(Whh), Whh.stride(0), ptr(sv['enc_gates']), sv['enc_gates'].stride(0)
            q.c, q.c_dstride, q.dh_direct, q.dd_dstride, q.dd_step, q.dd_ld = ptr(sv['enc_c']), sv['enc_c'].stride(0), ptr(dhs), \
                dhs.stride(0), Bp * H, H
            q.dG, q.dG_dstride, q.dc, q.dc_dstride, q.xbuf, q.x_dstride = ptr(dG), dG.stride(0), ptr(dc), dc.stride(0), ptr(xbuf), \
                xbuf.stride(0)
            q.active, q.counters, q.timeout_flag = ptr(idx['active']), ptr(counters), ptr(handoff_flag(dev))
            _run('gt_plstm_bwd', q)
        else:
            Hpad = roundup(H, 16)
            WhT = torch.stack([pack_rec(P['encoder.lstm.weight_hh_l0' + sfx], Hpad, H, 4 * H, transposed=True)
                               for sfx in ('', '_reverse')])
            S = rec_splits(Hpad, 4 * H, nz=2)
            a = GtBiLstm()
            a.B, a.Bp, a.T, a.H, a.S, a.backward, a.Hpad = B, Bp, Ti, H, S, 1, Hpad
            dG = _f32(2, Ti * Bp, 4 * H, device=dev)
            part = _f32(2, S, Bp, Hpad, device=dev)
            a.Whp, a.w_zstride, a.active = ptr(WhT), WhT.stride(0), ptr(idx['active'])
            a.part, a.gates, a.c, a.dhs, a.dG, a.dc = ptr(part), ptr(sv['enc_gates']), ptr(sv['enc_c']), ptr(dhs), ptr(dG), ptr(dc)
            _run('gt_bilstm_run', a)
        hs, Xs = sv['enc_hs'], sv['Xs']
        dXs = _f32(2, Ti * Bp, C, device=dev, zero=False)
        for d_, sfx in enumerate(('', '_reverse')):
            ops.linear_dw(dG[d_], Xs[d_], Ti * Bp, 4 * H, C, G['encoder.lstm.weight_ih_l0' + sfx])
            if Ti > 1:
                ops.linear_dw(dG[d_], hs[d_], (Ti - 1) * Bp, 4 * H, H, G['encoder.lstm.weight_hh_l0' + sfx],
                              dy_offset=Bp * 4 * H)
            bs = ops.col_sums(dG[d_], Ti * Bp, 4 * H)
            acc_add(G['encoder.lstm.bias_ih_l0' + sfx], bs)
            acc_add(G['encoder.lstm.bias_hh_l0' + sfx], bs)
            ops.linear_dx(dG[d_], P['encoder.lstm.weight_ih_l0' + sfx], Ti * Bp, 4 * H, C, out=dXs[d_])
        dx3 = _f32(B * Ti, C, device=dev, zero=False)
        call('gt_gather_rows', dXs[0], C, idx['ifwd'], B * Ti, C, dx3, C, 0)
        call('gt_gather_rows', dXs[1], C, idx['irev'], B * Ti, C, dx3, C, 1)

        _tap('dx3', dx3)
        PHASES.mark('b.bilstm')
        # ---- encoder convolutions + embedding
        demb = self.enc_stack.backward(P, G, dx3, B, Ti, sv['enc_conv'])
        # demb rows are [d embedded text | d labels | d noise]: only the text columns have a parameter behind them
        call('gt_embedding_bwd', sv['text'], demb, demb.shape[1], B * Ti, hp.symbols_embedding_dim,
             G['embedding.weight'])
        LEAF.join()
        PHASES.mark('b.enc_conv')
        PHASES.report()


class DiscriminatorEngine:
    """Discriminator.forward + adversarial_loss (reference model.py:500-540)."""

    DIL = (1, 2, 2, 2)

    def __init__(self, hp):
        self.hp = hp
        self.window, self.n_mel = hp.discriminator_window, hp.n_mel_channels
        self.bf16 = bool(getattr(hp, 'fp16_run', False))

    @_gemm_precision
    def forward(self, P, mel, lengths, keeps, training=True):
        """mel (B, n_mel, T) cuda; lengths int64 cuda tensor (B).  Returns loss (1,) and saved state."""
        dev = mel.device
        B, n_mel, T = mel.shape
        win = self.window
        W = T // win + (1 if T % win else 0)
        cin = win * n_mel
        x = _f32(B * W, cin, device=dev, zero=False)
        call('gt_disc_gather_fwd', mel.contiguous(), B, n_mel, T, win, x)
        sv = dict(B=B, T=T, W=W, layers=[], lengths=lengths)
        for i in range(4):
            Wt = P[f'discriminator.{i}.module.0.weight']
            cout, ci, k = Wt.shape
            y = ops.conv1d_fwd(x, ops.conv_weight_fwd(Wt), None, B, W, ci, cout, k, self.DIL[i])
            keep = keeps[i] if (keeps is not None and training) else None
            call('gt_bias_act_drop_fwd', y, P[f'discriminator.{i}.module.0.bias'], B * W, cout, ACT_TANH, 1, keep, 0.5, y)
            sv['layers'].append(dict(x=x, y=y, keep=keep))
            x = y
        W4 = P['discriminator.4.weight']
        out = ops.linear(x, W4.view(1, -1), B * W, W4.shape[1], 1, bias=P['discriminator.4.bias'])
        sv['h4'] = x
        loss = _f32(1, device=dev, zero=False)
        call('gt_adv_mean_fwd', out, lengths, B, W, win, loss)
        sv['out'] = out
        return loss, sv

    @_gemm_precision
    def backward(self, P, G, sv, dloss, scale=1.0, need_dinput=True, need_dparams=True, dout=None):
        """dloss: (1,) cuda tensor (upstream gradient of the adversarial loss) or None for 1; alternatively dout
        (B*W, 1): the gradient of the per-window outputs themselves (Discriminator.forward).  Returns d mel (B, n_mel, T)."""
        B, T, W = sv['B'], sv['T'], sv['W']
        dev = sv['out'].device
        win, n_mel = self.window, self.n_mel
        if dout is None:
            dout = _f32(B * W, 1, device=dev, zero=False)
            call('gt_adv_mean_bwd', dloss, sv['lengths'], B, W, win, float(scale), dout)
        W4 = P['discriminator.4.weight']
        c4 = W4.shape[1]
        if need_dparams:
            ops.linear_dw(dout, sv['h4'], B * W, 1, c4, G['discriminator.4.weight'].view(1, -1))
            acc_add(G['discriminator.4.bias'], ops.col_sums(dout, B * W, 1))
        dy = ops.linear_dx(dout, W4.view(1, -1), B * W, 1, c4)
        for i in reversed(range(4)):
            Wt = P[f'discriminator.{i}.module.0.weight']
            cout, ci, k = Wt.shape
            L = sv['layers'][i]
            dz = _f32(B * W, cout, device=dev, zero=False)
            call('gt_bias_act_drop_bwd', dy, L['y'], B * W, cout, ACT_TANH, 1, L['keep'], 0.5, dz)
            if need_dparams:
                ops.conv1d_dw(dz, L['x'], B, W, ci, cout, k, self.DIL[i], G[f'discriminator.{i}.module.0.weight'])
                acc_add(G[f'discriminator.{i}.module.0.bias'], ops.col_sums(dz, B * W, cout))
            if i > 0 or need_dinput:
                dy = ops.conv1d_dx(dz, ops.conv_weight_bwd(Wt), B, W, ci, cout, k, self.DIL[i])
        if not need_dinput:
            return None
        dmel = _f32(B, n_mel, T, device=dev, zero=False)
        call('gt_disc_gather_bwd', dy, B, n_mel, T, win, dmel, 0)
        return dmel

    @_gemm_precision
    def gradient_penalty(self, P, G, real, fake, lengths, alpha, keeps, training=True):
        """gradient_penalty of train.py:32-69 and its gradient with respect to the discriminator parameters.
        real, fake (B, n_mel, T) cuda (already cropped to the common T); lengths int64 cuda (B) of the tensor that
        sets the mask; alpha (B,) cuda U[0,1).  Returns gp (1,) and ADDS d gp / d theta to G.

        The penalty is a function of g = d sum(D(x)) / d x, itself the result of a backward pass, so its parameter
        gradient is a double backward.  With a_i = tanh(m_i z_i), z_i = conv_i(r_i) + b_i, t_i = m_i (1 - a_i^2):
          input-gradient pass   dz_i = da_i * t_i,  da_{i-1} = conv_i^T(dz_i),  g = scatter(da_{-1})
          penalty               v = d gp / d g                                       (gt_gp_norm)
          up pass   (i = 0..3)  u_0 = gather(v);  p_i = conv_i(u_i);  dW_i += corr(dz_i, u_i);
                                u_{i+1} = p_i * t_i;  e_i = p_i * da_i * dt_i/dz_i   (gt_tanh_drop_bwd2)
                                dW_4 += column sums of u_4        (da_3 = rows of W_4)
          down pass (i = 3..0)  h_3 = e_3;  dW_i += corr(h_i, r_i);  db_i += sum h_i;
                                h_{i-1} = conv_i^T(h_i) * t_{i-1} + e_{i-1}
        -- every product is one of the step's own GEMM-backed entries (conv1d_fwd / conv1d_dx / conv1d_dw)."""
        dev = real.device
        B, n_mel, T = real.shape
        win = self.window
        x = _f32(B, n_mel, T, device=dev, zero=False)
        call('gt_gp_interpolate', real.contiguous(), fake.contiguous(), alpha, lengths, B, n_mel, T, x)
        _, sv = self.forward(P, x, lengths, keeps, training)
        W = sv['W']
        rows = B * W
        layers = sv['layers']
        Wt = [P[f'discriminator.{i}.module.0.weight'] for i in range(4)]
        Wb = [ops.conv_weight_bwd(w) for w in Wt]
        W4 = P['discriminator.4.weight']
        c4 = W4.shape[1]
        # ---- input-gradient pass with unit upstream on every window (train.py:55-58)
        ones = torch.ones(rows, 1, device=dev, dtype=torch.float32)
        da, dz = [None] * 4, [None] * 4
        da[3] = ops.linear_dx(ones, W4.view(1, -1), rows, 1, c4)
        for i in reversed(range(4)):
            cout, ci, k = Wt[i].shape
            dz[i] = _f32(rows, cout, device=dev, zero=False)
            call('gt_bias_act_drop_bwd', da[i], layers[i]['y'], rows, cout, ACT_TANH, 1, layers[i]['keep'], 0.5, dz[i])
            dr = ops.conv1d_dx(dz[i], Wb[i], B, W, ci, cout, k, self.DIL[i])
            if i > 0:
                da[i - 1] = dr
        g = _f32(B, n_mel, T, device=dev, zero=False)
        call('gt_disc_gather_bwd', dr, B, n_mel, T, win, g, 0)
        gp = _f32(1, device=dev)
        call('gt_gp_norm', g, lengths, B, n_mel, T, gp)               # g <- d gp / d g
        # ---- up pass
        u = _f32(rows, win * n_mel, device=dev, zero=False)
        call('gt_disc_gather_fwd', g, B, n_mel, T, win, u)
        e = [None] * 4
        for i in range(4):
            cout, ci, k = Wt[i].shape
            gw = G[f'discriminator.{i}.module.0.weight']
            p = ops.conv1d_fwd(u, ops.conv_weight_fwd(Wt[i]), None, B, W, ci, cout, k, self.DIL[i])
            ops.conv1d_dw(dz[i], u, B, W, ci, cout, k, self.DIL[i], gw)
            u = _f32(rows, cout, device=dev, zero=False)
            e[i] = _f32(rows, cout, device=dev, zero=False)
            call('gt_tanh_drop_bwd2', p, da[i], layers[i]['y'], rows, cout, layers[i]['keep'], 0.5, u, e[i])
        acc_add(G['discriminator.4.weight'].view(-1), ops.col_sums(u, rows, c4).contiguous())
        # ---- down pass
        h = e[3]
        for i in reversed(range(4)):
            cout, ci, k = Wt[i].shape
            ops.conv1d_dw(h, layers[i]['x'], B, W, ci, cout, k, self.DIL[i], G[f'discriminator.{i}.module.0.weight'])
            acc_add(G[f'discriminator.{i}.module.0.bias'], ops.col_sums(h, rows, cout).contiguous())
            if i > 0:
                hr = ops.conv1d_dx(h, Wb[i], B, W, ci, cout, k, self.DIL[i])
                h = _f32(rows, ci, device=dev, zero=False)
                call('gt_bias_act_drop_bwd', hr, layers[i - 1]['y'], rows, ci, ACT_TANH, 1, layers[i - 1]['keep'], 0.5, h)
                acc_add(h, e[i - 1])
        return gp


class LinearDiscriminatorEngine:
    """LinearDiscriminator.forward + adversarial_loss (reference model.py:543-583): every item is cut into windows of
    `discriminator_window` frames whose starts advance by window - randint(0, 6) (host RNG, drawn by the caller in
    the reference's order) plus one final window ending at the item's length; each flattened window (frame-major)
    goes through 3 x [Linear -> Dropout(.5) -> Tanh] -> Linear(1); the loss is the batch mean of the per-item means."""

    def __init__(self, hp):
        self.window, self.n_mel = hp.discriminator_window, hp.n_mel_channels
        self.bf16 = bool(getattr(hp, 'fp16_run', False))

    def layers_forward(self, P, x, keeps, training):
        """The dense chain on already-cut windows x (NW, window*n_mel) -> (NW, 1) (LinearDiscriminator.forward)."""
        NW = x.shape[0]
        layers = []
        for i in range(3):
            Wt = P[f'discriminator.{i}.module.0.weight']
            cout, cin = Wt.shape
            y = ops.linear(x, Wt, NW, cin, cout)
            keep = keeps[i] if (keeps is not None and training) else None
            call('gt_bias_act_drop_fwd', y, P[f'discriminator.{i}.module.0.bias'], NW, cout, ACT_TANH, 1, keep, 0.5, y)
            layers.append(dict(x=x, y=y, keep=keep))
            x = y
        W3 = P['discriminator.3.weight']
        out = ops.linear(x, W3, NW, W3.shape[1], 1, bias=P['discriminator.3.bias'])
        return out, dict(NW=NW, layers=layers, h3=x, out=out)

    def layers_backward(self, P, G, sv, dout, need_dinput=True, need_dparams=True):
        """dout (NW, 1) -> gradient of the windows (NW, window*n_mel) (or None); parameter gradients are added to G."""
        NW = sv['NW']
        W3 = P['discriminator.3.weight']
        c3 = W3.shape[1]
        if need_dparams:
            ops.linear_dw(dout, sv['h3'], NW, 1, c3, G['discriminator.3.weight'])
            acc_add(G['discriminator.3.bias'], ops.col_sums(dout, NW, 1))
        dy = ops.linear_dx(dout, W3, NW, 1, c3)
        for i in reversed(range(3)):
            Wt = P[f'discriminator.{i}.module.0.weight']
            cout, cin = Wt.shape
            L = sv['layers'][i]
            dz = _f32(NW, cout, device=dout.device, zero=False)
            call('gt_bias_act_drop_bwd', dy, L['y'], NW, cout, ACT_TANH, 1, L['keep'], 0.5, dz)
            if need_dparams:
                ops.linear_dw(dz, L['x'], NW, cout, cin, G[f'discriminator.{i}.module.0.weight'])
                acc_add(G[f'discriminator.{i}.module.0.bias'], ops.col_sums(dz, NW, cout))
            if i > 0 or need_dinput:
                dy = ops.linear_dx(dz, Wt, NW, cout, cin)
        return dy if need_dinput else None

    @_gemm_precision
    def forward(self, P, mel, starts, keeps, training=True):
        """mel (B, n_mel, T) cuda; starts: per item the list of window start frames.  Returns loss (1,), saved."""
        dev = mel.device
        B, n_mel, T = mel.shape
        win = self.window
        rows, weights = [], []
        for b, st in enumerate(starts):
            for s0 in st:
                rows.extend(range(b * T + s0, b * T + s0 + win))
            weights.extend([1.0 / (B * len(st))] * len(st))
        NW = len(weights)
        idx = ops.h2d(rows, torch.int32, dev)
        wgt = ops.h2d(weights, torch.float32, dev)
        frames = _f32(B * T, n_mel, device=dev, zero=False)                 # (b, t) rows of n_mel
        call('gt_permute3', mel.contiguous(), n_mel * T, 1, T, B, T, n_mel, frames, 0, 0, 0, 1.0)
        x = _f32(NW, win * n_mel, device=dev, zero=False)
        call('gt_gather_rows', frames, n_mel, idx, NW * win, n_mel, x, n_mel, 0)
        out, sv = self.layers_forward(P, x, keeps, training)
        sv.update(B=B, T=T, idx=idx, wgt=wgt)
        loss = _f32(1, device=dev)
        sgemm(Operand(wgt, KC, ld=NW), Operand(out, RC, ld=1), loss, 1, 1, NW)           # sum_w wgt[w] * out[w]
        return loss, sv

    @_gemm_precision
    def backward(self, P, G, sv, dloss, scale=1.0, need_dinput=True, need_dparams=True):
        B, T, NW = sv['B'], sv['T'], sv['NW']
        dev = sv['out'].device
        win, n_mel = self.window, self.n_mel
        dout = _f32(NW, 1, device=dev, zero=False)
        call('gt_scale_dev', sv['wgt'], dloss, float(scale), dout, NW, 0)                  # dout[w] = dloss * wgt[w]
        dy = self.layers_backward(P, G, sv, dout, need_dinput, need_dparams)
        if not need_dinput:
            return None
        # scatter-add the window gradients back to their frames (windows overlap), then to (B, n_mel, T)
        dframes = _f32(B * T, n_mel, device=dev)
        call('gt_embedding_bwd', sv['idx'].long(), dy, n_mel, NW * win, n_mel, dframes)
        dmel = _f32(B, n_mel, T, device=dev, zero=False)
        call('gt_permute3', dframes, T * n_mel, 1, n_mel, B, n_mel, T, dmel, 0, 0, 0, 1.0)
        return dmel
