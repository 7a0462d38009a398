"""Every run-time switch of the package, in one place.  Five environment variables, read once at import:

GANTRON_FP32_SPLIT     0 / 2 / 3 / 4 (default 4): how the tiled fp32 GEMMs are computed -- the fp32 matrix pipe, two / three
                       bf16 planes per operand on the bf16 pipe (3 = exact operands), or two scaled fp16 planes (4: 22 significand
                       bits per operand, three MFMAs).
GANTRON_POISON=1       tests: buffers that are not zero-initialised start as NaN.
GANTRON_EXPERIMENTAL   comma list `name[=int]` of paths that were built, are parity-tested and measured SLOWER or no faster
                       than the default ("measured and not adopted"); nothing here is needed for
                       correctness and the default path never runs them:
                         fused_cell, attn_step, leaf_defer, rec_cellbwd (the decoder LSTM's fused backward step), plane_gemms
                         (pre-split GEMM operands), plstm=<bits> (persistent LSTM recurrences, default 15: 1 encoder forward, 2
                         decoder forward blocks, 4 encoder backward, 8 decoder backward blocks; 0 = per-step launches),
                         rec_x3=<bits> (attention-LSTM products from three bf16 planes: 1 forward, 2 backward), graph[=2] (hipGraph replay of the decoder loops; 2 also reports capture / replay),
                         no_pgemm2 (keep the plane GEMMs on 128 x 128 tiles), rec_tail[=slices] (the h_att half of the
                         attention LSTM's forward product rides in the attention step in front of it: csrc/rec_tail.h).
GANTRON_SCHEDULE       comma list switching OFF parts of the default schedule (what the schedule-invariance test toggles
                       through the module attributes instead): no_cell_query, no_attn_hoist, no_hoist2, no_leaf_stream,
                       no_fwd_splitk, no_rec_cell (per-step product + cell launches instead of the fused ones), no_attn_step2 (energies
                       and softmax + context of a forward step as two launches instead of gt_attn_step2_fwd), no_plstm (per-step
                       launches instead of the persistent LSTM recurrences), no_attn_bwd_fused (context and energies backward
                       of a step as two launches), no_attn_bwd_cell (the attention LSTM's cell backward as a launch of its own
                       instead of inside the backward attention step's: gt_attn_bwd_step_cell),
                       no_blk_gemm_aside (the forward loop's block GEMMs on the decoder-LSTM chain's own stream instead of a third one),
                       no_rec_h2 (the attention LSTM's per-step products on the fp32 matrix pipe instead of from two scaled fp16
                       planes: gt_rec_cell_fwd_h2 / gt_rec_gemm_h2),
                       no_wterms_in_forward (gt_attn_bwd_wterms at the top of the backward pass instead of behind the forward
                       loop), no_dist_overlap.
GANTRON_DEBUG          comma list of diagnostics: phase_timing, host_timing, block_timing, dist_timing, attn_valu,
                       rec=<bits> / fused=<bits> / pgemm=<bits> (kernel knock-outs), rec_splits=<n>, s1_fwd=<n>, s2_fwd=<n>,
                       s2_bwd=<n>, s1_bwd=<n>, enc_splits=<n>, sync_every=<n>, rec_lds_pad=<n>, rec_stamps / attn_stamps (phase
                       stamps of the attention LSTM's launch and of the attention step: tools/rec_stamps.py), attn2=<bits>
                       (knock-outs of the attention step), rec_tail_kb=<n>, amax (prints the operand maxima every fp16-plane product is
                       scaled by; synchronises), h2_skip=<bits> / h2_wide=<0..4> (fp16-plane products: pass knock-outs / tile-shape rule,
                       csrc/gemm.hip), amax_wrap=<n> (tests: the tags of the maxima ring start n calls before their 32-bit wrap),
                       bwd_skip=<bits> (knock-outs of the backward loop's decoder-LSTM chain: results invalid),
                       rec_h2_form=<bits> (1: gt_rec_cell_fwd_h2 in recurrent.hip's 64-k-chunk form instead of recurrent_bf16.hip's 256-k-chunk
                       one, 2: gt_rec_gemm_h2 the other way round).
An unknown name in any of the three lists raises at import with the valid names (a mistyped switch would otherwise run
the default schedule silently); values are decimal integers.  (The C side parses the same variables with the same
trimming rules: csrc/api.hip, gt_env_item.)  bench.py / the tests additionally use
GANTRON_FORCE_DIST, GANTRON_BENCH_DEVICE and GANTRON_BENCH_BACKEND for the one-GPU rehearsals of the multi-rank paths.
Tuned constants (block sizes, split-K targets, ring depth ...) are plain module attributes of engine.py / ops.py."""
import os as _os

# every name the package (Python or C side) looks up; keep in step with the docstring above
_KNOWN = {
    'GANTRON_EXPERIMENTAL': {'fused_cell', 'attn_step', 'leaf_defer', 'rec_cellbwd', 'plane_gemms', 'plstm', 'rec_x3', 'graph',
                             'no_pgemm2', 'rec_tail'},
    'GANTRON_SCHEDULE': {'no_cell_query', 'no_attn_hoist', 'no_hoist2', 'no_leaf_stream', 'no_fwd_splitk', 'no_rec_cell',
                         'no_attn_step2', 'no_plstm', 'no_attn_bwd_fused', 'no_attn_bwd_cell', 'no_rec_h2', 'no_blk_gemm_aside', 'no_wterms_in_forward',
                         'no_dist_overlap'},
    'GANTRON_DEBUG': {'phase_timing', 'host_timing', 'block_timing', 'dist_timing', 'attn_valu', 'rec', 'fused', 'pgemm',
                      'rec_splits', 's1_fwd', 's2_fwd', 's2_bwd', 's1_bwd', 'enc_splits', 'sync_every', 'rec_lds_pad',
                      'rec_stamps', 'attn_stamps', 'attn2', 'rec_tail_kb', 'amax', 'h2_skip', 'h2_wide', 'amax_wrap', 'bwd_skip', 'rec_h2_form'},
}


def _parse(var, text=None):
    """`name[=int]` items separated by ',' or ';', blanks around names, '=' and values ignored (the C parser's rules)."""
    out = {}
    text = _os.environ.get(var, '') if text is None else text
    for item in text.replace(';', ',').split(','):
        item = item.strip()
        if not item:
            continue
        k, eq, v = item.partition('=')
        k, v = k.strip(), v.strip()
        if k not in _KNOWN[var]:
            raise ValueError(f'{var}: unknown switch {k!r}; valid names: {", ".join(sorted(_KNOWN[var]))}')
        if eq and not v.lstrip('+-').isdigit():
            raise ValueError(f'{var}: {k}={v!r} is not a decimal integer')
        out[k] = int(v) if eq else 1
    return out


EXPERIMENTAL, SCHEDULE, DEBUG = _parse('GANTRON_EXPERIMENTAL'), _parse('GANTRON_SCHEDULE'), _parse('GANTRON_DEBUG')


def experimental(name, default=0):
    assert name in _KNOWN['GANTRON_EXPERIMENTAL'], name
    return EXPERIMENTAL.get(name, default)


def schedule_on(name):
    """True unless GANTRON_SCHEDULE lists no_<name>."""
    assert 'no_' + name in _KNOWN['GANTRON_SCHEDULE'], name
    return not SCHEDULE.get('no_' + name, 0)


def debug(name, default=0):
    assert name in _KNOWN['GANTRON_DEBUG'], name
    return DEBUG.get(name, default)
