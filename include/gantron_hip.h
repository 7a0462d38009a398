/*
 * gantron_hip.h — C ABI of libgantron_hip.so: hand-written HIP kernels (gfx950 / MI355X) for the
 * GANtron generator+discriminator training step.
 *
 * The reference (Brechard/GANtron) has no FFI/plugin layer: its hot path is a chain of stock
 * PyTorch ops inside nn.Module.forward and autograd.  This library replaces those chains; each
 * entry cites the reference code it stands in for.  The host side (gantron_amd/ *.py) keeps the
 * reference's Python API (hparams.HParams, model.Tacotron2, model.Discriminator, ...) and calls
 * these entries with raw device pointers (tensor.data_ptr()), explicit sizes, explicit dropout
 * keep-masks (u8, 1 = keep) and the caller's hipStream_t.
 *
 * Conventions
 *  - every function returns 0 on success, a negative GT_E_* code for invalid arguments, or a
 *    positive hipError_t; gt_last_error_string() describes the last failure of the calling thread;
 *  - nothing is allocated, freed or synchronised here: all buffers (outputs, workspaces, saved-for-
 *    backward state) are owned by the caller; every launch is asynchronous on `stream`;
 *  - all tensors are fp32 unless stated; activations are "channels-last": (rows = batch*time, C);
 *  - weights are read in PyTorch layout unless the argument is documented as packed
 *    (packed copies are produced per call by gt_pack_* into caller-owned scratch).
 */
#ifndef GANTRON_HIP_H
#define GANTRON_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GT_E_ARG   (-1)   /* bad argument / shape */
#define GT_E_ALIGN (-2)   /* pointer or leading dimension violates an alignment requirement */
#define GT_E_LIMIT (-3)   /* size exceeds a compiled-in limit */

#define GT_ABI_VERSION 403      /* bumped whenever a declared signature, struct layout or buffer size changes */
int gt_version(void);           /* GT_ABI_VERSION of the library that was built: compare it with the header's at load */
const char* gt_last_error_string(void);
/* sizeof of ABI structs: 0 GtOperand, 1 GtGemm, 2 GtSlabs, 3 GtAttnDims, 4 GtDecoderFwd,
 * 5 GtDecoderBwd, 6 GtBiLstm (lets a binding verify its layout at load time) */
int64_t gt_abi_sizeof(int32_t which);

/* Optional per-launch timing with HIP events recorded on the launch stream (bench.py's roofline
 * leg).  Between gt_prof_begin() and gt_prof_end() every launch of a profiled kernel family is
 * bracketed by two events; gt_prof_end waits for them and fills, per family f of the GT_PROF_FAMILIES = 10 families
 * (0 rec_gemm, 1 sgemm, 2 attn_energies_fwd, 3 attn_context_fwd, 4 attn_context_bwd, 5 attn_energies_bwd,
 * 6 lstm_cell_fwd, 7 lstm_cell_bwd, 8 rec_cell = the fused product + cell launches, 9 plstm = the persistent multi-step
 * recurrences): out[4f] bracketed launches,
 * out[4f+1] their total ms, out[4f+2] their algorithmic flops, out[4f+3] their algorithmic bytes, and
 * out[4*GT_PROF_FAMILIES + f] ALL launches of the family issued while profiling (bracketed or not); with sampling,
 * every 4th bracket is followed by the same bracket around an EMPTY kernel on the same stream:
 * out[5*GT_PROF_FAMILIES] their count, out[5*GT_PROF_FAMILIES + 1] their total ms.  `out` is a HOST array of
 * GT_PROF_DOUBLES = 72 doubles.  Not thread-safe.  A bracket = kernel duration (dispatch begin to end) + event latency
 * (+ what the dispatch waited for), so the duration of a sampled kernel is bracket - (empty-kernel bracket -
 * empty-kernel duration), the latter from gt_prof_calibrate.  The two recurrent-product families (0 and 8) are not
 * bracketed: their sampled launches carry their own start / stop events (hipExtLaunchKernelGGL), stamped with the
 * dispatch's begin and end -- the duration a profiler reports; out[5*GT_PROF_FAMILIES + 2 + f] counts, per family, the
 * timed launches of that kind (no latency to subtract for them); out[6*GT_PROF_FAMILIES + 2 + f] the algorithmic flops of ALL
 * launches of the family. */
#define GT_PROF_FAMILIES 10
#define GT_PROF_DOUBLES (7 * GT_PROF_FAMILIES + 2)
int gt_prof_begin(void);
/* the attention context kernels skip padded memory rows: tell the profiler how many rows of the current batch are valid
 * (sum of the memory lengths) so that their algorithmic bytes are charged for what they read; 0 = charge B*Ti */
int gt_prof_set_context_rows(int64_t valid_rows);
/* bracket only every stride-th launch of each family (the millisecond-scale sgemm family every
 * max(1, stride/16)-th: its few, very unequal launches need a denser sample and do not feel the bracket) */
int gt_prof_begin_sampled(int32_t stride);
int gt_prof_end(double* out);
/* duration (us) of the empty calibration kernel itself, from the start / stop timestamps of its own dispatch on the
 * idle `stream` (HOST pointer out_us) */
int gt_prof_calibrate(int32_t n, double* out_us, void* stream);

/* ------------------------------------------------------------------------------------------ GEMM
 * C[m,n] (=|+=) sum_k A(m,k) * B(n,k) (+ bias[n]),   fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Operand addressing modes (element (r,k) of an operand with rows r and reduction index k):
 *   GT_OP_KC        p[r*ld + k]                                   (k contiguous)
 *   GT_OP_RC        p[k*ld + r]                                   (r contiguous)
 *   GT_OP_IM2COL_K  rows are positions r=(b,t), k=(tap,c):  p[((b*T + t + tap*dil - pad)*C + c)]
 *   GT_OP_IM2COL_R  rows are r=(tap,c), k is a position (b,t): same address, 0 outside [0,T)
 * The im2col modes make Conv1d (model.py:117-144,165-178,506-511; layers.py:22-60) and its
 * input/weight gradients plain GEMMs over channels-last activations.
 */
enum { GT_OP_KC = 0, GT_OP_RC = 1, GT_OP_IM2COL_K = 2, GT_OP_IM2COL_R = 3 };

typedef struct GtOperand {
    const float* p;
    int32_t mode;
    int32_t T, C, dil, pad;     /* im2col modes only */
    int64_t ld;                 /* KC / RC modes */
    int64_t batch_stride;       /* elements between batches (grid batching) */
    const void* planes;         /* optional: the operand as three bf16 planes x = x0 + x1 + x2 (gt_split3): plane i of the
                                   element that `p` addresses as p[e] is ((uint16_t*)planes)[i * plane_stride + e].  When
                                   BOTH operands carry planes, the product is the exact three-plane one and the shapes
                                   allow it (K % 32 == 0, ld % 8 == 0, im2col C % 32 / % 8 == 0), gt_sgemm runs the kernel
                                   that has no split left to do; otherwise it uses p (p may be NULL only if planes suffice) */
    int64_t plane_stride;
    const uint64_t* amax;       /* optional (fp16-plane products, GtGemm.bf16 = 4 / gt_sgemm_set_fp32_split(4)): device word left by
                                   gt_absmax over everything this operand addresses; NULL: gt_sgemm makes the pass itself */
} GtOperand;

typedef struct GtGemm {
    GtOperand A, B;
    float* C;
    int64_t ldc, c_batch_stride;
    const float* bias;          /* per-n, may be NULL */
    int32_t M, N, K;
    int32_t batch;              /* >= 1 */
    int32_t accumulate;         /* 0: C = ..., 1: C += ... */
    int32_t split_k;            /* >= 1; > 1 uses fp32 atomics and therefore always accumulates */
    float alpha;
    int32_t bf16;               /* 0: fp32 product (computed as gt_sgemm_set_fp32_split selects);
                                   1: operands rounded to bf16 (nearest even) in the tile loader, bf16 MFMA, fp32
                                   accumulation and output (hparams.fp16_run);
                                   2 / 3: fp32 product from 2 / 3 bf16 planes per operand, regardless of the setting;
                                   4: from two scaled fp16 planes per operand (below), regardless of the setting */
} GtGemm;

int gt_sgemm(const GtGemm* g, void* stream);

/* How products with GtGemm.bf16 == 0 are computed (process-wide; default 4, the same as the Python package):
 *   0  v_mfma_f32_32x32x2_f32, the fp32 matrix pipe (157 TFLOP/s peak on MI355X);
 *   3  every fp32 operand x is staged as three bf16 planes x0 + x1 + x2 = x (exact: 3 x 8 significand bits) and
 *      a . b = a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0) runs as six bf16 MFMAs with fp32 accumulation: each
 *      product is exact, the dropped terms are <= 2^-24 |a||b| -- the rounding unit of the fp32 pipe itself -- and the
 *      bf16 pipe is 16x faster, so the product costs 6/16 of the fp32 MFMA time;
 *   2  two planes, three MFMAs: dropped terms <= 2^-16 |a||b| (the accuracy class of 3xTF32);
 *   4  (round 4) every fp32 operand TENSOR is multiplied by a power of two s that puts its largest finite magnitude into
 *      [2^14, 2^15) -- found by one pass over the operand in front of the product, or taken from GtOperand.amax -- and each
 *      element is staged as two fp16 planes h0 = f16(s x), h1 = f16(s x - h0) (nearest even; the residual is exact): 22 of the
 *      24 significand bits for elements down to 2^-18 of the tensor's maximum, absolute error <= 2^-39 of that maximum below;
 *      a . b = (a0 b0 + a0 b1 + a1 b0) / (sa sb) runs as three fp16 MFMAs with fp32 accumulation (every product exact, the
 *      scales leave exactly: v_ldexp_f32).  Dropped terms <= 2^-22 |a||b| per product, one seventh of what the fp32 pipe's own
 *      accumulation leaves at the reduction lengths of the training step: measured against float64 it errs less than mode 0 in
 *      every operand mode (tests/test_kernels_gpu.py::test_sgemm_split_planes_accuracy).  3/16 of the fp32 MFMA time.
 * Inf / NaN operands produce NaN in modes 2 / 3 / 4 (in the rows they belong to). */
int gt_sgemm_set_fp32_split(int32_t planes);
/* max |x| over the finite elements of a rows x cols matrix (row stride ld), as mode 4 wants it in GtOperand.amax: the word is
 * reset and filled on `stream`; it stays valid for as long as the operand's contents do */
int gt_absmax(const float* x, int64_t rows, int64_t cols, int64_t ld, uint64_t* out, void* stream);
int32_t gt_sgemm_get_fp32_split(void);

/* ------------------------------------------------------------------- recurrent (small-M) GEMM
 * Per-time-step product of the LSTM recurrences (model.py:391-392,406-407 and the BiLSTM of
 * model.py:207): part[s][m][n] = sum_{k in K-slice s} X[m][k] * W[n][k], m < 16*mtiles <= 64.
 * W is pre-packed by gt_pack_rec_weight into [N/16][KB][64 lanes][4] (KB = ceil(K/16) 16-wide
 * k-blocks, zero padded) so that every wave streams 1 KiB lines; X is row-major with
 * ldx >= 16*KB, rows beyond the batch and columns beyond K must be zero (or finite).
 * `part` receives `splits` slabs of [16*mtiles][N] spaced part_sstride elements apart; the
 * consumer (cell kernels) sums them.  nz > 1 runs independent problems (BiLSTM directions).
 */
int gt_pack_rec_weight(const float* W, int64_t ldw, int32_t transposed, int32_t N, int32_t n_valid,
                       int32_t K, float* Wp, void* stream);   /* rows n >= n_valid are zero; transposed: 0 W is (N,K),
                       1 W is (K,N) */
int64_t gt_rec_weight_elems(int32_t N, int32_t K);
int gt_rec_gemm(const float* X, int64_t ldx, const float* Wp, float* part, int32_t mtiles,
                int32_t N, int32_t K, int32_t splits, int64_t part_sstride, int32_t nz,
                int64_t x_zstride, int64_t w_zstride, int64_t part_zstride, void* stream);

/* Fused recurrent product + LSTM cell (model.py:391-394 / 406-409) (+ the attention query projection, model.py:56):
 * ONE launch for  gates = X . W^T (+ pre);  i,f,g,o;  c' = f c + i g;  h = dropout(o tanh c')  [;  q_slab = h . Wq^T].
 * Wp comes from gt_pack_rec_cell_weight (the columns of a workgroup are the four gates of 16 hidden units, H % 16 == 0,
 * gt_rec_weight_elems(4H, K) floats).  K is split over `splits` workgroups per column group; they exchange partial
 * tiles through `part` (splits x Bp x 4H floats, any content) with write-through stores and a ticket per column group
 * in `counters` ((Bp > 64 ? Bp/64 : 1) x H/16 uint32, zero-initialised ONCE; the kernel keeps them consistent from
 * launch to launch as long as `splits` does not change); the workgroup drawing a group's last ticket runs the cell for
 * its 16 units.  pre (Bp, 4H) holds the hoisted part of the pre-activations in PyTorch gate order (may be NULL);
 * gates (Bp, 4H, PyTorch gate order) may alias pre; h goes to h_raw (Bp, H) and to up to three strided destinations;
 * rows >= B are neither read nor written.  Wq (A, H) != NULL: every column group also writes its share of the query
 * projection as slab cg of qpart (H/16 slabs of (Bp, Apad), q_sstride apart) for gt_attn_energies_fwd*.
 * prio: wave issue priority 0..3 (s_setprio). */
int gt_pack_rec_cell_weight(const float* W, int64_t ldw, int32_t H, int32_t K, float* Wp, void* stream);
int gt_rec_cell_fwd(const float* X, int64_t ldx, const float* Wp, float* part, uint32_t* counters, int32_t Bp,
                    int32_t B, int32_t H, int32_t K, int32_t splits, const float* pre, int64_t ld_pre,
                    const float* c_prev, const uint8_t* keep, float p_drop, float* gates, float* c_out,
                    float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2,
                    const float* Wq, int64_t ldwq, float* qpart, int64_t q_sstride, int32_t A, int32_t Apad,
                    int32_t prio, void* stream);


/* bf16-operand variant (hparams.fp16_run, the reference's apex path train.py:246-250 -> bf16 MFMA with fp32
 * accumulation on MI355X): the weights are packed as bf16 fragments of v_mfma_f32_16x16x32_bf16
 * ([N/16][ceil(K/32)][64 lanes][8 bf16], gt_rec_weight_bf16_elems 16-bit elements), the fp32 activations X are
 * rounded to bf16 (nearest even) inside the kernel, slabs are fp32 as above.  A K-slice may not exceed 1024 columns:
 * splits >= gt_rec_bf16_min_splits(K).  Strides as in gt_rec_gemm (w_zstride in 16-bit elements, multiple of 8). */
int64_t gt_rec_weight_bf16_elems(int32_t N, int32_t K);
int32_t gt_rec_bf16_min_splits(int32_t K);
int gt_pack_rec_weight_bf16(const float* W, int64_t ldw, int32_t transposed, int32_t N, int32_t n_valid,
                            int32_t K, void* Wp, void* stream);
int gt_rec_gemm_bf16(const float* X, int64_t ldx, const void* Wp, float* part, int32_t mtiles, int32_t N,
                     int32_t K, int32_t splits, int64_t part_sstride, int32_t nz, int64_t x_zstride,
                     int64_t w_zstride, int64_t part_zstride, void* stream);
/* gt_rec_cell_fwd for a product whose tail was multiplied by an EARLIER launch (GtRecTail): only the first k_own (a
 * multiple of 16, <= K) reduction columns are this launch's, cut into `splits` K-slices; the last arriver of a column group
 * also adds `extra_slabs` slabs behind the K-slices' own (part + (splits + j) * Bp * 4H, j < extra_slabs; splits +
 * extra_slabs <= 16).  k_own = 0: all of K. */
int gt_rec_cell_fwd_part(const float* X, int64_t ldx, const float* Wp, float* part, uint32_t* counters, int32_t Bp,
                         int32_t B, int32_t H, int32_t K, int32_t splits, const float* pre, int64_t ld_pre,
                         const float* c_prev, const uint8_t* keep, float p_drop, float* gates, float* c_out,
                         float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2,
                         const float* Wq, int64_t ldwq, float* qpart, int64_t q_sstride, int32_t A, int32_t Apad,
                         int32_t k_own, int32_t extra_slabs, int32_t prio, void* stream);
/* gt_rec_cell_fwd with bf16-packed weights (gt_pack_rec_cell_weight_bf16: gt_rec_weight_bf16_elems(4H, K) 16-bit
 * elements, gate-grouped columns): bf16 MFMA product with fp32 accumulation, the cell, dropout and the query slabs in
 * fp32 exactly as in gt_rec_cell_fwd.  splits >= gt_rec_bf16_min_splits(K), <= 16. */
int gt_pack_rec_cell_weight_bf16(const float* W, int64_t ldw, int32_t H, int32_t K, void* Wp, void* stream);
int gt_rec_cell_fwd_bf16(const float* X, int64_t ldx, const void* Wp, float* part, uint32_t* counters, int32_t Bp,
                         int32_t B, int32_t H, int32_t K, int32_t splits, const float* pre, int64_t ld_pre,
                         const float* c_prev, const uint8_t* keep, float p_drop, float* gates, float* c_out,
                         float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2,
                         const float* Wq, int64_t ldwq, float* qpart, int64_t q_sstride, int32_t A, int32_t Apad,
                         int32_t prio, void* stream);




/* ------------------------------------------------------------------------------------ LSTM cell
 * nn.LSTMCell pointwise part + dropout (model.py:391-394,406-409); gate order i,f,g,o.
 * gates_pre = sum_s part[s] + pre (pre holds the non-recurrent input projection and both biases).
 * `active` (per row, may be NULL) implements pack_padded_sequence (model.py:203-210): inactive rows
 * keep their state and emit zeros.  h (after dropout) is written to up to three destinations.
 */
typedef struct GtSlabs { const float* p; int32_t n; int64_t slab_stride; int64_t ld; } GtSlabs;

int gt_lstm_cell_fwd(GtSlabs part, const float* pre, int64_t ld_pre, const float* c_prev,
                     const float* h_prev, int64_t ld_hprev, const uint8_t* keep, float p_drop,
                     const int32_t* active, float* gates, float* c_out, float* h_raw,
                     float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2,
                     int32_t B, int32_t H, void* stream);

/* gt_lstm_cell_fwd fused with the attention query projection (model.py:56 query_layer): the cell of 32 hidden
 * units per workgroup, then qpart[g] = h[:, 32g..32g+32) . Wq[:, 32g..]^T as H/32 slabs of (Bp, Apad) that
 * gt_attn_energies_fwd sums.  Wq is the (A, Ha) weight in PyTorch layout.  H % 32 == 0, float4-aligned buffers. */
int gt_cell_query_fwd(GtSlabs part, const float* pre, int64_t ld_pre, const float* c_prev, const uint8_t* keep,
                      float p_drop, float* gates, float* c_out, float* h_raw, float* h0, int64_t ld0, float* h1,
                      int64_t ld1, const float* Wq, int64_t ldwq, float* qpart, int64_t q_sstride, int32_t B,
                      int32_t Bp, int32_t H, int32_t A, int32_t Apad, void* stream);

/* backward counterpart: gt_lstm_cell_bwd of the attention LSTM with the query-projection input gradient
 * dh += dq . Wq computed inside (dq (Bp, ld_dq) zero padded to a multiple of 16 columns), instead of a
 * gt_rec_gemm for dq . Wq whose slabs the cell kernel would sum.  H % 16 == 0, float4-aligned buffers. */
int gt_cell_bwd_query(GtSlabs dh_a, GtSlabs dh_c, const float* dq, int64_t ld_dq, const float* Wq, int64_t ldwq,
                      const float* dc_next, const float* gates, const float* c_new, const float* c_prev,
                      const uint8_t* keep, float p_drop, float* dgates, float* dc_prev, int32_t B, int32_t Bp,
                      int32_t H, int32_t A, void* stream);


/* dh = (sum of up to three slab sources + direct) ; returns pre-activation gate grads + dc_prev. */
int gt_lstm_cell_bwd(GtSlabs dh_a, GtSlabs dh_b, GtSlabs dh_c, const float* dh_direct,
                     int64_t ld_direct, const float* dc_next, const float* gates,
                     const float* c_new, const float* c_prev, const uint8_t* keep, float p_drop,
                     const int32_t* active, float* dgates, float* dc_prev, float* dh_pass,
                     int32_t B, int32_t H, void* stream);

/* --------------------------------------------------------------- location-sensitive attention
 * One decoder step of Attention.forward (model.py:70-91) with LocationLayer (model.py:27-31):
 *   e[t] = v . tanh(q + Wd conv31([w_prev; w_cum])[t] + pm[t]);  pad -> -inf; w = softmax(e);
 *   ctx = w @ memory;  w_cum += w.
 * gt_attn_energies: grid (B, chunks of Ti).  q = sum of `qpart` slabs (query_layer product).
 * gt_attn_context : softmax over Ti + context; ctx written to up to three destinations.
 * `lengths` (may be NULL) lets every attention kernel skip the padded tail of a memory row: those
 * positions have weight exactly 0 and zero gradient, so results are unchanged.
 */
typedef struct GtAttnDims { int32_t B, Ti, E, A, F, Kc; } GtAttnDims;

int gt_attn_energies_fwd(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum,
                         const float* pm, const float* Wc, const float* Wd, const float* v,
                         const int32_t* lengths, float* q_out, float* energies, void* stream);
/* the same with `wpack` (may be NULL): the weight image of gt_attn_pack_weights (gt_attn_wpack_elems(d) floats, 0 when
 * the dimensions have no packed path), prepared once per decoder loop so that every launch stages it with float4 copies */
int gt_attn_energies_fwd_w(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum,
                         const float* pm, const float* Wc, const float* Wd, const float* v,
                         const int32_t* lengths, float* q_out, float* energies, const float* wpack, void* stream);
int64_t gt_attn_wpack_elems(GtAttnDims d);
int gt_attn_pack_weights(GtAttnDims d, const float* Wc, const float* Wd, float* wpack, void* stream);
/* One launch for a whole forward attention step (model.py:70-91) WITHOUT an in-launch wait: every (row, 32-position chunk)
 * workgroup requests its tile of `memory` at once, computes its energies (as gt_attn_energies_fwd_u), softmax numerators
 * relative to the chunk's own maximum and the chunk's partial context, publishes them in `xbuf` (write-through) and draws a
 * ticket on row_cnt[b]; the row's LAST arriver combines the partials (online softmax), normalises and writes w_out, cum_out
 * and the context to up to three destinations.  xbuf: B * ceil(Ti / 32) rows of gt_attn_step2_xstride(d) floats; row_cnt: B
 * words zeroed once per loop; step1 = 1, 2, ... counts the calls since then.  Attention dims (128, 32, 31), Ti <= 256,
 * E % 4 == 0, E <= 1024 (gt_attn_step2_supported). */
int32_t gt_attn_step2_supported(GtAttnDims d);
int64_t gt_attn_step2_xstride(GtAttnDims d);
int gt_attn_step2_fwd(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum, const float* pm,
                      const float* Wc, const float* Wd, const float* v, const int32_t* lengths, float* q_out,
                      const float* wpack, float* u_out, float* xbuf, uint32_t* row_cnt, uint32_t step1,
                      const float* memory, int64_t ld_mem, float* w_out, const float* cum_in, float* cum_out,
                      float* c0, int64_t ld0, float* c1, int64_t ld1, float* c2, int64_t ld2, void* stream);
/* Recurrent-product work that rides in another launch (csrc/rec_tail.h).  Half of what an LSTM step of the decoder's
 * attention chain multiplies exists one launch earlier than the launch that consumes it (forward: the h_att half of
 * [ctx | h_att] . W1; backward: the d h_att columns of dG1 . W1^T); that half is cut into (64-column group, K-slice) items
 * which the workgroups of the attention step in between pick up from a work counter when they have nothing (more) of their
 * own to do.  Every workgroup drains the counter before it exits: all items are done when the launch ends.
 * Item (g, s), g < groups, s < ceil((kb1 - kb0) / kbs): out[s * out_sstride + row * N + 64 (cg0 + g) + c] =
 * sum over the k-blocks [kb0 + s kbs, min(kb1, kb0 + (s + 1) kbs)) of X[row][k] Wp(column 64 (cg0 + g) + c, k), rows
 * 0 .. 16 mtiles - 1.  work == NULL: no tail. */
typedef struct GtRecTail {
    const float* X; int64_t ldx;      /* (16 mtiles rows, ldx >= 16 kb1) activations, 16-byte aligned, ldx % 4 == 0 */
    const float* Wp;                  /* gt_pack_rec_weight / gt_pack_rec_cell_weight image: [N/16][KB][64][4] */
    float* out; int64_t out_sstride;  /* slab of K-slice s at out + s * out_sstride, rows of N floats */
    int32_t N, KB;                    /* columns (N % 64 == 0) and 16-wide k-blocks of the packed image */
    int32_t kb0, kb1, kbs;            /* the items' K range [kb0, kb1) in k-blocks, kbs k-blocks per K-slice */
    int32_t cg0, groups;              /* 64-column groups [cg0, cg0 + groups) */
    int32_t mtiles;                   /* row tiles of 16 (1..4) */
    uint32_t* work;                   /* ONE word, zero before the launch: the launch's work counter */
} GtRecTail;
/* gt_attn_step2_fwd carrying a tail (NULL or tail->work == NULL: exactly gt_attn_step2_fwd) */
int gt_attn_step2_fwd_tail(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum, const float* pm,
                           const float* Wc, const float* Wd, const float* v, const int32_t* lengths, float* q_out,
                           const float* wpack, float* u_out, float* xbuf, uint32_t* row_cnt, uint32_t step1,
                           const float* memory, int64_t ld_mem, float* w_out, const float* cum_in, float* cum_out,
                           float* c0, int64_t ld0, float* c1, int64_t ld1, float* c2, int64_t ld2, const GtRecTail* tail,
                           void* stream);
int gt_attn_context_fwd(GtAttnDims d, const float* energies, const float* memory, int64_t ld_mem,
                        const int32_t* lengths, float* w_out, const float* cum_in, float* cum_out,
                        float* c0, int64_t ld0, float* c1, int64_t ld1, float* c2, int64_t ld2,
                        void* stream);

/* backward of the two kernels above (recomputes tanh arguments instead of saving them).
 * context_bwd: dctx = slabs a + slabs b + direct; dw[t'] = memory[b][t'].dctx + carried alignment
 *   gradients (dwcat_in[0] = d w_prev of step t+1, dcum_acc += dwcat_in[1]; both planes are
 *   consumed and zeroed) + dw_extern (gradient of the returned alignments).  Also stores dctx
 *   (for the deferred d memory GEMM) and zeroes this step's dq accumulator.
 * energies_bwd: softmax backward (needs w and dw over all Ti), then d tanh / dense / conv:
 *   dq += (atomics), dpm += , dwcat_out += (atomics, [2][B][Ti]), parameter gradients of
 *   v / location_dense / location_conv accumulate in per-workgroup slabs `acc`. */
int gt_attn_context_bwd(GtAttnDims d, GtSlabs dctx_a, GtSlabs dctx_b, const float* dctx_direct,
                        int64_t ld_direct, const float* memory, int64_t ld_mem,
                        const int32_t* lengths, float* dwcat_in, float* dcum_acc, const float* dw_extern, float* dctx_out,
                        int64_t ld_dctx_out, float* dq_zero, int64_t ld_dq, float* dw_out,
                        void* stream);
int gt_attn_energies_bwd(GtAttnDims d, const float* dw, const float* w, const float* q,
                         const float* w_prev, const float* w_cum, const float* pm, const float* Wc,
                         const float* Wd, const float* v, const int32_t* lengths, float* dq,
                         int64_t ld_dq, float* dpm, float* dwcat_out, float* acc, void* stream);
int gt_attn_energies_bwd_w(GtAttnDims d, const float* dw, const float* w, const float* q,
                         const float* w_prev, const float* w_cum, const float* pm, const float* Wc,
                         const float* Wd, const float* v, const int32_t* lengths, float* dq,
                         int64_t ld_dq, float* dpm, float* dwcat_out, float* acc, const float* wpack, void* stream);
/* Hoisted form of the energies backward (dimensions 128/32/31 only, gt_attn_hoist_supported): the forward keeps
 * u = tanh(.) of every step (gt_attn_energies_fwd_u, u_out (B,Ti,A)), the backward loop computes only the recurrent
 * terms -- dq, d[w_prev; w_cum] -- and the softmax-backward vector de_out (B,Ti) (gt_attn_energies_bwd_rec), and the
 * sums over steps that nothing inside the loop consumes -- dv, dWd, dWc into `acc`, d(processed memory) into dpm --
 * are added by gt_attn_param_grads for `nsteps` consecutive steps at once (u, de advance by B*Ti*A / B*Ti per step,
 * w_prev / w_cum by step_stride).  Results equal gt_attn_energies_bwd's up to summation order. */
int32_t gt_attn_hoist_supported(GtAttnDims d);
int gt_attn_energies_fwd_u(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum,
                           const float* pm, const float* Wc, const float* Wd, const float* v,
                           const int32_t* lengths, float* q_out, float* energies, const float* wpack,
                           float* u_out, void* stream);
int gt_attn_energies_bwd_rec(GtAttnDims d, const float* dw, const float* w, const float* u, const float* Wc,
                             const float* Wd, const float* v, const int32_t* lengths, float* dq, int64_t ld_dq,
                             float* dwcat_out, float* de_out, const float* wpack, void* stream);
int gt_attn_param_grads(GtAttnDims d, int32_t nsteps, const float* u, const float* de, const float* w_prev,
                        const float* w_cum, int64_t step_stride, const float* Wc, const float* Wd,
                        const float* v, const int32_t* lengths, float* dpm, float* acc, const float* wpack,
                        void* stream);
int64_t gt_attn_acc_elems(GtAttnDims d);     /* size of `acc` (dv | dWd | dWc per workgroup) */
int gt_attn_acc_reduce(GtAttnDims d, const float* acc, float* dv, float* dWd, float* dWc,
                       void* stream);
int32_t gt_attn_chunks(int32_t Ti);

/* ------------------------------------------------------------------------- sequential loops
 * gt_decoder_fwd / gt_decoder_bwd: the teacher-forced decoder loop (Decoder.forward + decode,
 * model.py:378-454) and its backward-through-time, To steps per call, 7 launches per step.
 * gt_bilstm_run: the packed encoder BiLSTM (model.py:202-210), both directions, forward or
 * backward.  Per-step histories use Bp = 16*ceil(B/16) rows per step (pad rows zero) so that they
 * can be fed to gt_rec_gemm / gt_sgemm directly; keep masks and attention tensors use B rows.
 */
typedef struct GtDecoderFwd {
    int32_t B, Bp, Ti, To, E, ldE, P, Ha, Hd, A, Apad, F, Kc;
    int32_t S1, S2, Sq;
    float p_att, p_dec;
    const float* memory;       /* (B, Ti, ldE) */
    const float* pm;           /* (B, Ti, A) processed memory */
    const int32_t* lengths;    /* (B) */
    const float* pre1;         /* (To, Bp, 4Ha) prenet projection + both biases of attention_rnn */
    const float* bias2;        /* (4Hd) b_ih + b_hh of decoder_rnn */
    const float* W1p;          /* packed [4Ha] x [ctx(E) | h_att(Ha)] */
    const float* Wqp;          /* packed [Apad] x [Ha] */
    const float* W2p;          /* packed [4Hd] x [h_att(Ha) | ctx(E) | h_dec(Hd)] */
    const float* Wq;           /* query_layer weight (A, Ha), PyTorch layout: when non-NULL (and Ha % 32 == 0) the
                                  attention cell and the query product run as ONE launch (gt_cell_query_fwd)
                                  writing Ha/32 query slabs to partq */
    float* partq;              /* (Ha/32, Bp, Apad) */
    const float* W1g;          /* the same two weights packed gate-interleaved: when both are non-NULL the LSTM */
    const float* W2g;          /*   steps run as gt_lstm_step_fused (W1p / W2p, part / part2 are then unused)   */
    const float* Wc; const float* Wd; const float* v;
    float* attn_wpack;         /* gt_attn_wpack_elems floats of scratch (or NULL): packed [Wc | Wd] image, filled by the loop */
    const uint8_t* att_keep;   /* (To, Bp, Ha) or NULL */
    const uint8_t* dec_keep;   /* (To, Bp, Hd) or NULL */
    float* X1; int64_t ld1;    /* (2, Bp, ld1) zero-initialised (the fused steps ping-pong between the two slots) */
    float* X2; int64_t ld2;    /* (ring, Bp, ld2) zero-initialised */
    float* part;               /* slab scratch of the attention chain */
    float* part2;              /* slab scratch of the decoder-LSTM chain (ring > 1) */
    int32_t ring;              /* 1: one stream; 2..32: two-chain pipelining over a ring of X2 buffers, cross-stream
                                  hand-offs every ring/2 steps */
    float* energies;           /* (B, Ti) */
    float* gates1; float* c1;  /* (To, Bp, 4Ha), (To+1, Bp, Ha) c1[0] = 0 */
    float* gates2; float* c2;  /* (To, Bp, 4Hd), (To+1, Bp, Hd) */
    float* hatt;               /* (To, Bp, Ha) */
    float* HC;                 /* (To, Bp, Hd+E): [h_dec | ctx] */
    float* w_hist;             /* (To+1, B, Ti), w_hist[0] = 0 */
    float* cum_hist;           /* (To+1, B, Ti), cum_hist[0] = 0 */
    float* q_hist;             /* (To, B, A) */
    int32_t bf16;              /* 1: W1p / W2p are gt_pack_rec_weight_bf16 images and the two LSTM products run as
                                  gt_rec_gemm_bf16 (hparams.fp16_run); needs the unfused path (W1g = W2g = NULL) */
    float* u_hist;             /* NULL or (To, B, Ti, A): tanh outputs of every step for the hoisted backward */
    /* Hoisted decoder-LSTM input product (W2hp != NULL).  Of [h_att_t | ctx_t | h_dec_{t-1}] . W2 only the h_dec
     * part is on the decoder LSTM's recurrence; the attention chain never reads the decoder LSTM.  So the attention
     * chain runs `blk` steps ahead, [h_att | ctx] . W2ih^T + bias2 of those steps is ONE tiled GEMM (M = blk*Bp rows)
     * into pre2, and the per-step product shrinks to h_dec . W2hh^T (a third of the weights and flops).  In this mode
     * X2 is a HISTORY (To, Bp, ld2) of [h_att_t | ctx_t] rows (ld2 >= Ha+E, zero-initialised), W2p / ring slots are
     * unused, S2 counts the K-splits of the h_dec product and part2 holds S2 slabs of (Bp, 4Hd). */
    const float* W2hp;         /* packed [4Hd] x [Hd] (gt_pack_rec_weight, or the bf16 image when bf16 = 1) */
    const float* W2ih;         /* decoder_rnn.weight_ih (4Hd, Ha+E), PyTorch layout, rows ldw2 apart */
    int64_t ldw2;
    float* pre2;               /* (To, Bp, 4Hd); may alias gates2 (the cell reads its pre-activations before it
                                  writes the gate values of the same elements) */
    float* Xh; int64_t ldh;    /* (Bp, ldh >= 16*ceil(Hd/16)) zero-initialised: h_dec operand of the per-step product */
    int32_t blk;               /* steps per block (hand-off granularity between the two chains); <= 0: 32 */
    /* Fused product + cell launches (gt_rec_cell_fwd / gt_rec_cell_fwd_bf16 when bf16 = 1) inside the hoisted mode.  W1c != NULL: the attention
     * LSTM step is ONE launch (product, cell, dropout, h scatter, query slabs): X1 must hold two (Bp, ld1) slots (the
     * cell writes the next step's operand while other workgroups still read this step's), partq Ha/16 slabs.
     * W2hc != NULL: the same for the decoder LSTM's h_dec product; Xh then holds two (Bp, ldh) slots. */
    const float* W1c;          /* gt_pack_rec_cell_weight (bf16 = 1: ..._bf16 image) of [4Ha] x [ctx(E) | h_att(Ha)] */
    const float* W2hc;         /* gt_pack_rec_cell_weight of [4Hd] x [Hd] */
    uint32_t* counters1;       /* (Bp > 64 ? Bp/64 : 1) x Ha/16 tickets, zero-initialised */
    uint32_t* counters2;       /* ... x Hd/16 */
    /* row_cnt != NULL (and gt_attn_step_supported): energies + softmax + context of a step are ONE launch
     * (gt_attn_step_fwd); `energies` is the exchange buffer, row_cnt (B) zero-initialised arrival counters,
     * attn_timeout (int32, may be NULL) is raised if an in-launch wait gives up. */
    uint32_t* row_cnt;
    int32_t* attn_timeout;
    /* W2hh != NULL (hoisted mode, fp32, gt_plstm_supported(Bp, Hd)): the decoder LSTM's recurrence of a whole block of
     * steps is ONE persistent launch (gt_plstm_fwd: W_hh resident in registers on Hd / 16 workgroups, h exchanged through
     * the HC history itself) instead of one product + cell launch per step.  W2hh = decoder_rnn.weight_hh (4Hd, Hd) in
     * PyTorch layout, pl_counters one word, pl_timeout as attn_timeout (not NULL). */
    const float* W2hh;
    uint32_t* pl_counters;
    int32_t* pl_timeout;
    /* attn_xbuf != NULL (with row_cnt, and gt_attn_step2_supported): energies + softmax + context of a step are ONE launch
     * with NO in-launch wait (gt_attn_step2_fwd): B * ceil(Ti / 32) rows of gt_attn_step2_xstride(d) floats. */
    float* attn_xbuf;
    /* fp32 mode only (bf16 = 0).  1: W1c is a gt_pack_rec_cell_weight_x3 image and the attention LSTM's fused product + cell
     * runs as gt_rec_cell_fwd_x3 (three bf16 planes).  2 (round 4, the engine's default): W1c is a gt_pack_rec_cell_weight_h2
     * image and the step runs as gt_rec_cell_fwd_h2 (two scaled fp16 planes); its activations [ctx | h_att] are bounded by
     * max(x1floor, the value in *x1max): x1max = gt_absmax word of `memory` (every context is a convex combination of its
     * rows), x1floor = 1 / (1 - p_att) (|h_att| = |o tanh c| / (1 - p)) */
    int32_t rec_x3;
    const uint64_t* x1max;
    float x1floor;
    /* tail_work != NULL (with W1c, attn_xbuf, fp32, Bp <= 64, E % 16 == 0, Sx1 >= 1): the h_att half of the attention LSTM's
     * product leaves its launch (GtRecTail): the workgroups of attention step t that have nothing (more) to do compute
     * h_att_t . W1[:, E:]^T in Sx1 K-slices into slabs S1 .. S1 + Sx1 - 1 of `part` (which then holds S1 + Sx1 slabs), and
     * LSTM step t + 1 multiplies the ctx half only (S1 K-slices of E) and adds those slabs.  tail_work: To words, zeroed. */
    uint32_t* tail_work;
    int32_t Sx1;
    int32_t tail_kb;           /* 16-wide k-blocks of the h_att half that ride in the attention step, counted from the END of
                                  K (1 .. Ha / 16; 0 = all of them): the LSTM step keeps [ctx | first h_att columns] */
} GtDecoderFwd;

typedef struct GtDecoderBwd {
    int32_t B, Bp, Ti, To, E, ldE, P, Ha, Hd, A, Apad, F, Kc;
    int32_t S1, S2, Sq;
    int32_t N1pad, N2pad;      /* roundup16(E+Ha), roundup16(Ha+E+Hd) */
    float p_att, p_dec;
    const float* memory; const float* pm;
    const int32_t* lengths;    /* (B) memory lengths */
    const float* Wc; const float* Wd; const float* v;
    float* attn_wpack;         /* as in GtDecoderFwd */
    const uint8_t* att_keep; const uint8_t* dec_keep;
    const float* gates1; const float* c1; const float* gates2; const float* c2;
    const float* w_hist; const float* cum_hist; const float* q_hist;
    const float* W1Tp;         /* packed [N1pad] x [4Ha]  (transpose of the forward operand) */
    const float* W2Tp;         /* packed [N2pad] x [4Hd] */
    const float* WqTp;         /* packed [Ha] x [Apad] */
    const float* Wq;           /* query_layer weight (A, Ha), PyTorch layout: when non-NULL (and Ha % 16 == 0) the query
                                  input gradient dq . Wq is computed inside the attention cell backward
                                  (gt_cell_bwd_query) instead of a gt_rec_gemm into partQ */
    const float* dHC;          /* (To, Bp, Hd+E) gradient of the projection input */
    const float* dalign;       /* (To, B, Ti) gradient of the returned alignments, or NULL */
    float* dG1; float* dG2;    /* (To, Bp, 4Ha), (To, Bp, 4Hd) pre-activation gate gradients */
    float* dctx_hist;          /* (To, B, E) */
    float* dpm;                /* (B, Ti, A) zero-initialised, accumulated */
    float* acc;                /* gt_attn_acc_elems floats, zero-initialised */
    float* dc1; float* dc2;    /* (B, H) zero-initialised */
    float* dwcat;              /* (2, 2, B, Ti) zero-initialised ping-pong */
    float* dcum_acc;           /* (B, Ti) zero-initialised */
    float* dq_hist;            /* (To, Bp, Apad) zero-initialised: per-step query gradients */
    float* dw;                 /* (B, Ti) */
    float* partX1; float* partX2; float* partQ;   /* partX2: (ring, S2, Bp, N2pad) */
    int32_t ring;              /* as in GtDecoderFwd */
    int32_t t_hi, t_lo;        /* steps t_hi..t_lo (descending) of this call; t_hi < t_lo = all steps.  State
                                  persists in the buffers, so a loop may be issued in several calls. */
    int32_t bf16;              /* 1: W1Tp / W2Tp are gt_pack_rec_weight_bf16 images (products run as gt_rec_gemm_bf16) */
    const float* u_hist;       /* both non-NULL: hoisted energies backward (gt_attn_energies_bwd_rec per step; the   */
    float* de_hist;            /*   caller runs gt_attn_param_grads over the steps afterwards); de_hist (To, B, Ti)  */
    /* Hoisted decoder-LSTM input gradient (W2hTp != NULL), the mirror of GtDecoderFwd's hoisted mode: the decoder-LSTM
     * chain leads and only carries d h_dec (per-step product dG2_t . W2hh, N = Hd); d[h_att_t | ctx_t] = dG2_t . W2ih
     * of a block of steps is ONE tiled GEMM into dXA before the attention chain runs those steps.  W2Tp, partX2 and
     * the ring are unused; S2 counts the K-splits of the d h_dec product. */
    const float* W2hTp;        /* packed [Hdpad = 16*ceil(Hd/16)] x [4Hd] (transpose of W2hh; bf16 image when bf16 = 1) */
    const float* W2ih;         /* decoder_rnn.weight_ih (4Hd, Ha+E), rows ldw2 apart */
    int64_t ldw2;
    float* dXA; int64_t ldxa;  /* (To, Bp, ldxa >= Ha+E, ldxa % 4 == 0): [d h_att_t | d ctx_t] from the decoder LSTM */
    float* partXh;             /* (S2, Bp, Hdpad) zero-initialised */
    int32_t blk;               /* steps per block; <= 0: 32 */
    /* Fused backward attention step (all three non-NULL, together with u_hist / de_hist and the hoisted mode above): ONE
     * launch per step (gt_attn_bwd_step) instead of gt_attn_context_bwd + gt_attn_energies_bwd_rec.  dq2_hist / dwcat2_hist /
     * p_hist come from gt_attn_bwd_wterms over all steps BEFORE the loop; sum_hist receives the per-tile row sums.  After the
     * loop de_hist holds w dw (not yet minus s w) and dq_hist the uncorrected dq: gt_attn_bwd_fixup finishes both before
     * anything outside the loop reads them.  dq_hist must be zero-initialised (nothing in the loop zeroes it). */
    float* sum_hist;           /* (To, B, gt_attn_chunks(Ti)) */
    const float* dq2_hist;     /* (To, Bp, Apad), the layout of dq_hist */
    const float* dwcat2_hist;  /* (To, 2, B, Ti) */
    const float* p_hist;       /* (To, B, Ti, 64) */
    /* hoisted mode, fp32, Hd % 64 == 0: the decoder LSTM's per-step product and the cell backward of the step before it as
     * ONE launch (gt_rec_cellbwd); (Bp > 64 ? Bp/64 : 1) x Hd/64 tickets, zero-initialised; NULL: two launches */
    uint32_t* countersB;
    /* W2hh != NULL (hoisted mode, fp32, gt_plstm_supported(Bp, Hd)): the decoder LSTM's backward recurrence of a whole block
     * of steps is ONE persistent launch (gt_plstm_bwd) instead of one product + cell-backward launch per step.  W2hh =
     * decoder_rnn.weight_hh (4Hd, Hd) in PyTorch layout, pl_xbuf gt_plstm_bwd_xbuf_elems(Hd) floats (persists between the
     * calls of a chunked loop), pl_counters one word, pl_timeout as GtDecoderFwd's. */
    const float* W2hh;
    float* pl_xbuf;
    uint32_t* pl_counters;
    int32_t* pl_timeout;
    /* 1 (fp32 mode only, bf16 = 0): W1Tp is a gt_pack_rec_weight_x3 image and the attention LSTM's backward product runs as
     * gt_rec_gemm_x3 */
    int32_t rec_x3;
    /* rec_x3 == 2 (round 4; needs rowcntA): W1Tp is a gt_pack_rec_weight_h2 image and the attention LSTM's backward product
     * runs as gt_rec_gemm_h2; amaxG1 = To zero-initialised words, word t receives max |dG1_t| from the attention step's
     * cell tail (GtCellBwdTail.dg_amax) one launch in front of the product that reads it */
    uint64_t* amaxG1;
    /* != NULL (with the fused backward attention step above and Wq): B zero-initialised words -- the attention LSTM's cell
     * backward rides in the attention step's launch (gt_attn_bwd_step_cell), two dependent launches per step instead of
     * three.  The words are shared by ALL calls of one backward pass (the step's launch number is To - t). */
    uint32_t* rowcntA;
} GtDecoderBwd;

typedef struct GtBiLstm {
    int32_t B, Bp, T, H, S, backward;
    const float* pre;          /* (2, T, Bp, 4H) step-major input projections + biases */
    const float* Whp;          /* (2) packed [4H] x [H]; for backward packed [Hpad] x [4H] */
    int64_t w_zstride;
    const int32_t* active;     /* (T, Bp) 1 while step < length */
    float* X; int64_t ldx;     /* (2, Bp, ldx) zero-initialised (forward only) */
    float* part;               /* (2, S, Bp, N) */
    float* gates;              /* (2, T, Bp, 4H) */
    float* c;                  /* (2, T+1, Bp, H), c[:,0] = 0 */
    float* hs;                 /* (2, T, Bp, H) emitted outputs (zero when inactive) */
    /* backward only */
    const float* dhs;          /* (2, T, Bp, H) gradient of the emitted outputs */
    float* dG;                 /* (2, T, Bp, 4H) */
    float* dc;                 /* (2, Bp, H) zero-initialised */
    int32_t Hpad;
    /* forward only, optional (H % 16 == 0): product + cell of a step as ONE launch (gt_rec_cell_fwd, both directions).
     * Whc = the two gt_pack_rec_cell_weight images (w_zstride apart), counters = 2 x (Bp > 64 ? Bp/64 : 1) x H/16 zeroed
     * tickets; X then holds TWO slots of (2, Bp, ldx) (the cell writes the next step's operand while other workgroups
     * still read this step's). */
    const float* Whc;
    uint32_t* counters;
} GtBiLstm;


/* Free-running decoder of Decoder.inference (model.py:456-497): every step feeds the previous mel
 * frame back through the Prenet (dropout always on, model.py:104), runs the decode step of
 * gt_decoder_fwd (same kernels, no LSTM dropout, lengths = Ti: inference passes mask=None) and the
 * mel/gate projection, and records per item the first step whose sigmoid(gate) exceeds the threshold.
 * The host issues steps [t_lo, t_hi) per call and reads stop_step between calls (model.py:486-491
 * tests the gate every step; frames computed past the stop are simply not returned). */
typedef struct GtDecoderInfer {
    GtDecoderFwd step;         /* To = capacity in frames; att_keep = dec_keep = NULL; ring = 1;
                                  pre1 = ONE step's (Bp, 4Ha) buffer, rewritten every step */
    const float* W0;           /* Prenet layer 0 (P, n_mel), PyTorch layout */
    const float* W1;           /* Prenet layer 1 (P, P) */
    const float* W1x;          /* attention_rnn.weight_ih: prenet columns [0,P) of rows of stride ldw1x */
    int64_t ldw1x;
    const float* bias1;        /* (4Ha) b_ih + b_hh of attention_rnn */
    const uint8_t* keep0;      /* (To, Bp, P) Prenet dropout keep masks */
    const uint8_t* keep1;
    const float* Wpg;          /* (n_mel+1, Hd+E): linear_projection rows then the gate_layer row */
    const float* bpg;          /* (n_mel+1) */
    float* p1; float* p2;      /* (Bp, P) scratch */
    float* frames;             /* (To+1, Bp, n_mel+1): block 0 = go frame (zeros), block t+1 = [mel | gate] of step t */
    int32_t* stop_step;        /* (B) initialised to INT32_MAX by the caller; min over steps with sigmoid(gate) > threshold */
    int32_t n_mel;
    float gate_threshold;
    int32_t t_lo, t_hi;
} GtDecoderInfer;

/* gt_cell_bwd_query with a corrected query gradient dq - s dq2, s[m] = sum of row m's n_sum entries of dq_sum (B, n_sum): the
 * consumer side of the fused backward attention step (gt_attn_bwd_step); dq2 laid out like dq.  dq_sum = dq2 = NULL:
 * gt_cell_bwd_query. */
int gt_cell_bwd_query_c(GtSlabs dh_a, GtSlabs dh_c, const float* dq, int64_t ld_dq, const float* dq_sum, int32_t n_sum,
                        const float* dq2, const float* Wq, int64_t ldwq, const float* dc_next, const float* gates,
                        const float* c_new, const float* c_prev, const uint8_t* keep, float p_drop, float* dgates,
                        float* dc_prev, int32_t B, int32_t Bp, int32_t H, int32_t A, void* stream);

/* Backward attention step as ONE launch (the default where gt_attn_bwd_step_supported: (128, 32, 31) dimensions, Ti <= 512;
 * model.py:70-91 through autograd).  The softmax backward needs one scalar per batch
 * row, s = sum_t w dw, but everything after it is linear in de = w dw - s w.  With M[t][d] = v[d] (1 - u[t][d]^2) and
 * P = (M . Wd) . Wc (62 columns per position, stored 64 wide), functions of forward quantities only:
 *     dq[d] = sum_t de[t] M[t][d],   d[w_prev; w_cum][c][t0 + p - 15] = sum_k de[p - k] P[p - k][(c, k)].
 *  gt_attn_bwd_wterms : for `nsteps` steps in ONE launch: p_hist[t] = P ((B, Ti, 64) per step), dq2[t] = Lq(w_t) (step
 *                       stride dq2_step_stride, rows ld_dq2 apart) and dwcat2[t] = Lw(w_t) ((2, B, Ti) per step); dq2 /
 *                       dwcat2 zero-initialised by the caller; w advances by w_step_stride per step, u by B*Ti*A.
 *  gt_attn_bwd_step   : gt_attn_context_bwd's inputs (dctx sources, memory, carried planes dwcat_in / dcum_acc, dw_extern,
 *                       dctx_out) + the step's w, u and P rows.  Writes e1 = w dw to e1_out (B, Ti), the tile sums of e1 to
 *                       sum_out (B, gt_attn_chunks(Ti)), and ADDS Lq(e1) to dq (zeroed by the caller) and Lw(e1) to
 *                       dwcat_out.  sum_prev / dwcat2_prev (both or neither): the previous call's sum_out and that step's
 *                       dwcat2 -- the carried planes are read as dwcat_in - s_prev dwcat2_prev.
 *  gt_attn_bwd_fixup  : de[t] = e1[t] - s_t w_t and dq[t] -= s_t dq2[t] for `nsteps` steps (sum: (nsteps, B, chunks)).
 * The attention cell backward takes the correction on the fly (gt_cell_bwd_query_c). */
int32_t gt_attn_bwd_step_supported(GtAttnDims d);
int gt_attn_bwd_wterms(GtAttnDims d, int32_t nsteps, const float* w, int64_t w_step_stride, const float* u,
                       const float* Wc, const float* Wd, const float* v, const int32_t* lengths, float* dq2,
                       int64_t dq2_step_stride, int64_t ld_dq2, float* dwcat2, float* p_hist, const float* wpack,
                       void* stream);
int gt_attn_bwd_step(GtAttnDims d, GtSlabs dctx_a, GtSlabs dctx_b, const float* dctx_direct, int64_t ld_direct,
                     const float* memory, int64_t ld_mem, const int32_t* lengths, float* dwcat_in, float* dcum_acc,
                     const float* dw_extern, float* dctx_out, int64_t ld_dctx_out, const float* w, const float* u,
                     const float* v, const float* p, const float* sum_prev, const float* dwcat2_prev, float* sum_out,
                     float* dq, int64_t ld_dq, float* dwcat_out, float* e1_out, void* stream);
int gt_attn_bwd_fixup(GtAttnDims d, int32_t nsteps, const float* sum, const float* w, int64_t w_step_stride,
                      float* de, float* dq, int64_t dq_step_stride, int64_t ld_dq, const float* dq2, void* stream);
/* gt_attn_bwd_step with the attention LSTM's cell backward of the same step inside the launch (round 4): each batch row's
 * LAST workgroup -- the one whose ticket on row_cnt[b] completes the row -- computes what gt_cell_bwd_query_c would in a
 * launch of its own: dh = dh_a + dh_c + (dq - s dq2) . Wq for the row, dropout, cell backward, dgates / dc_prev of the row
 * (model.py:391-394 through autograd).  dq2 is this step's row block of the dq2 history (laid out like dq), Wq (A, H) with rows
 * ldwq apart; the other fields as gt_cell_bwd_query_c's arguments; H % 4 == 0, every pointer 16-byte aligned.  row_cnt: B
 * words, zero-initialised once; `launch` = 1, 2, 3 ... for consecutive calls that share them (the counters only grow: no reset
 * between the steps of a backward pass).  No workgroup waits for another.  Rows m >= B of dgates / dc_prev are not written; a
 * row with lengths[b] == 0 has no workgroup that could finish it (the loops' text lengths are >= 1: data_utils.py:38-49). */
typedef struct GtCellBwdTail {
    GtSlabs dh_a, dh_c;
    const float* dq2;
    const float* Wq; int64_t ldwq;
    const float* dc_next;
    const float* gates;
    const float* c_new;
    const float* c_prev;
    const uint8_t* keep; float p_drop;
    float* dgates;
    float* dc_prev;
    int32_t H;
    uint32_t* row_cnt; uint32_t launch;
    uint64_t* dg_amax;      /* optional: zero-initialised word that receives max |dgates| of the step in gt_absmax's format (for
                               gt_rec_gemm_h2 behind this launch) */
} GtCellBwdTail;
int gt_attn_bwd_step_cell(GtAttnDims d, GtSlabs dctx_a, GtSlabs dctx_b, const float* dctx_direct, int64_t ld_direct,
                          const float* memory, int64_t ld_mem, const int32_t* lengths, float* dwcat_in, float* dcum_acc,
                          const float* dw_extern, float* dctx_out, int64_t ld_dctx_out, const float* w, const float* u,
                          const float* v, const float* p, const float* sum_prev, const float* dwcat2_prev, float* sum_out,
                          float* dq, int64_t ld_dq, float* dwcat_out, float* e1_out, const GtCellBwdTail* cell, void* stream);

/* Persistent, weight-resident LSTM recurrence: ONE launch walks steps [s_lo, s_hi) of
 *   G_s = pre_s + h_{s-1} . W_hh^T,  cell,  h_s = dropout(o tanh c_s)
 * for ndir independent problems, each on H / 16 workgroups that keep W_hh in registers and exchange h through `hx`
 * (csrc/plstm.hip).  Used for the encoder BiLSTM (ndir = 2, packed-sequence semantics through `active`: model.py:202-210)
 * and for the decoder LSTM's h_dec recurrence inside a block of steps (ndir = 1, hx = the HC history itself: model.py:406-409);
 * both are what the default schedule runs at 64 rows (GANTRON_SCHEDULE=no_plstm: the per-step launches).
 * Bp must be 64 and H 256 or 1024, and the CURRENT device must have at least twice as many CUs as the launch's workgroups
 * hold -- every workgroup waits for its peers, so all of them must be resident at once beside the rest of the step
 * (gt_plstm_supported: 128 CUs for H = 1024, 64 for H = 256; a CPX partition gets the per-step launches).  The waits are
 * bounded: a launch that could not become resident raises timeout_flag instead of hanging; gt_adam_step_guarded then
 * skips the optimizer step.  Direction d of every per-direction buffer lies *_dstride elements behind direction 0. */
typedef struct GtPlstm {
    int32_t B, Bp, H, T, ndir, s_lo, s_hi, hx_ring;
    float p_drop;              /* dropout probability of `keep` (0: none) */
    const float* Whh;          /* (4H, H) PyTorch layout, gate order i f g o */
    int64_t w_dstride;
    const float* pre;          /* (T, Bp, 4H) input projections + both biases */
    int64_t pre_dstride;
    float* gates;              /* (T, Bp, 4H) gate values kept for the backward pass; may alias pre */
    int64_t gates_dstride;
    float* c;                  /* (T + 1, Bp, H); slice s_lo is read, slices s_lo + 1 .. s_hi are written */
    int64_t c_dstride;
    float* hs;                 /* (T, Bp, H) emitted h (zero for inactive rows), or NULL */
    int64_t hs_dstride;
    float* hx;                 /* exchange: h_s is published at slot (hx_ring ? s % hx_ring : s) * hx_step, rows of hx_ld;
                                  h_{s_lo - 1} is read from there when s_lo > 0.  hx_ring = 0 or >= 2 */
    int64_t hx_dstride, hx_step, hx_ld;
    const uint8_t* keep;       /* (T, Bp, H) keep mask of the dropout on h, or NULL */
    const int32_t* active;     /* (T, Bp): rows with 0 pass their state through and emit zeros, or NULL */
    uint32_t* counters;        /* ndir words (zeroed by the call) */
    int32_t* timeout_flag;     /* raised when a bounded wait gives up (results invalid) */
} GtPlstm;
int32_t gt_plstm_supported(int32_t Bp, int32_t H);
int gt_plstm_fwd(const GtPlstm* a, void* stream);

/* Backward counterpart: ONE launch walks steps s_hi - 1 .. s_lo of  dh_s = dh_direct_s + dG_{s+1} . W_hh,  cell backward ->
 * dG_s, dc carried on (csrc/plstm.hip).  The recurrent term travels between the H / 16 workgroups of a problem as a
 * reduce-scatter of (64 x 16) tiles through `xbuf` (gt_plstm_bwd_xbuf_elems(H) floats per direction, two step slots);
 * recurrent_in = 1: the tiles of step s_hi were published by an earlier call (a loop issued in blocks, highest steps
 * first).  dc (Bp, H) is read at the start and holds the carried gradient at the end. */
typedef struct GtPlstmBwd {
    int32_t B, Bp, H, T, ndir, s_lo, s_hi, recurrent_in;
    float p_drop;
    const float* Whh;          /* (4H, H) PyTorch layout */
    int64_t w_dstride;
    const float* gates;        /* (T, Bp, 4H) gate values of the forward pass */
    int64_t gates_dstride;
    const float* c;            /* (T + 1, Bp, H) */
    int64_t c_dstride;
    const float* dh_direct;    /* gradient of the h emitted by step s at dh_direct + s * dd_step, rows of dd_ld */
    int64_t dd_dstride, dd_step, dd_ld;
    float* dG;                 /* (T, Bp, 4H) gate pre-activation gradients (out) */
    int64_t dG_dstride;
    float* dc;                 /* (Bp, H) */
    int64_t dc_dstride;
    float* xbuf;
    int64_t x_dstride;
    const uint8_t* keep;       /* (T, Bp, H) dropout keep mask of the forward pass, or NULL */
    const int32_t* active;     /* (T, Bp), or NULL */
    uint32_t* counters;        /* ndir words (zeroed by the call) */
    int32_t* timeout_flag;
} GtPlstmBwd;
int64_t gt_plstm_bwd_xbuf_elems(int32_t H);
int gt_plstm_bwd(const GtPlstmBwd* a, void* stream);

int gt_decoder_fwd(const GtDecoderFwd* a, void* stream);
int gt_decoder_bwd(const GtDecoderBwd* a, void* stream);
int gt_bilstm_run(const GtBiLstm* a, void* stream);
int gt_decoder_infer(const GtDecoderInfer* a, void* stream);

/* ------------------------------------------------------------------------------------ collective
 * The data-parallel gradient average (distributed.py:137-161: ONE flat all-reduce(SUM) per backward, then / world size) for a
 * host that binds this library without PyTorch: buf[0..n) <- (sum over the ranks of `comm`) / ranks, in place, enqueued on
 * `stream`.  `comm` is the caller's RCCL communicator (ncclComm_t).  RCCL is never linked and never loaded here: its entry
 * points are taken at the first call from the RCCL the process ALREADY has -- global symbols, or librccl.so.1 / librccl.so
 * (RTLD_NOLOAD) -- because only the library that created `comm` may be handed it, and its ncclGetVersion must report an
 * NCCL 2.x API (the ncclFloat32 / ncclSum values used).  Nothing found: GT_E_ARG.  Returns the RCCL error code (> 0) if RCCL
 * fails.  The Python package averages through torch.distributed (backend "nccl" = RCCL) instead: gantron_amd/distributed.py.
 * Exercised on one rank on the test pool (one GPU per box); the two-rank test skips itself there. */
int gt_allreduce_avg(void* comm, float* buf, int64_t n, void* stream);
/* the same for a host whose RCCL is not visible that way (a framework's private copy): it passes that library's
 * ncclAllReduce and ncclCommCount itself */
int gt_allreduce_avg_with(void* comm, float* buf, int64_t n, void* nccl_all_reduce, void* nccl_comm_count, void* stream);

/* -------------------------------------------------------------------- elementwise / reductions */
/* per-column sums over rows: BatchNorm1d batch statistics incl. padding (model.py:123,176) and
 * bias gradients.  out[0..C) = sum x, out[C..2C) = sum x*y (sum x*x if y is NULL).  Passing
 * y == x selects the shifted form sum (x-K), sum (x-K)^2 with K = x[0][c] that
 * gt_bn_act_drop_fwd expects (no cancellation when |mean| >> std).
 * ws == NULL: row blocks meet in fp32 atomics (the last bits depend on dispatch order);
 * ws = gt_col_ws_elems(rows, C) floats of scratch: the last row block to arrive adds the others'
 * partial sums in row order -- bit-reproducible (what BatchNorm statistics are computed with). */
int64_t gt_col_ws_elems(int64_t rows, int32_t C);
int gt_col_stats(const float* x, const float* y, int64_t rows, int32_t C, float* out2C, float* ws,
                 void* stream);
/* BN(train) -> activation (0 none, 1 relu, 2 tanh) -> dropout; updates running stats.
 * stats = gt_col_stats output; saves mean/invstd (2C) for backward. */
int gt_bn_act_drop_fwd(const float* x, const float* stats, int64_t rows, int32_t C,
                       const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float momentum, float eps, int32_t act,
                       const uint8_t* keep, float p_drop, float* y, float* mean_invstd,
                       int32_t training, void* stream);
/* backward, two passes: (1) dz = dy*keep*act' and column sums of dz, dz*xhat;
 * (2) dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)).  dgamma/dbeta += . */
int gt_bn_act_drop_bwd1(const float* dy, const float* x, const float* mean_invstd,
                        const float* gamma, const float* beta, int64_t rows, int32_t C,
                        int32_t act, const uint8_t* keep, float p_drop, float* dz, float* sums2C,
                        float* ws, void* stream);   /* ws: as gt_col_stats */
int gt_bn_act_drop_bwd2(const float* dz, const float* x, const float* mean_invstd,
                        const float* gamma, const float* sums2C, int64_t rows, int32_t C,
                        float* dx, float* dgamma, float* dbeta, void* stream);
/* y = dropout(act(x + bias)); act 0/1/2 as above; order 0: act then dropout (prenet, model.py:104),
 * order 1: dropout then act (DiscConv1d, layers.py:54-57). */
int gt_bias_act_drop_fwd(const float* x, const float* bias, int64_t rows, int32_t C, int32_t act,
                         int32_t order, const uint8_t* keep, float p_drop, float* y, void* stream);
int gt_bias_act_drop_bwd(const float* dy, const float* y, int64_t rows, int32_t C, int32_t act,
                         int32_t order, const uint8_t* keep, float p_drop, float* dx, void* stream);
/* embedding (model.py:648) gather / scatter-add of its gradient */
int gt_embedding_fwd(const int64_t* ids, const float* table, int64_t n, int32_t D, float* out,
                     int64_t ld_out, void* stream);
int gt_embedding_bwd(const int64_t* ids, const float* dout, int64_t ld, int64_t n, int32_t D,
                     float* dtable, void* stream);
/* generic strided 3-d permute copy: out[i*o0 + j*o1 + k] (=|+=) alpha * in[i*s0 + j*s1 + k*s2];
 * o0 = o1 = 0 selects a contiguous output.  Strides may be negative (tap-flipped weights). */
int gt_permute3(const float* in, int64_t s0, int64_t s1, int64_t s2, int32_t n0, int32_t n1,
                int32_t n2, float* out, int64_t o0, int64_t o1, int32_t accumulate, float alpha,
                void* stream);
/* row gather: out[r] = idx[r] >= 0 ? src[idx[r]] : 0 (C floats per row).  Builds the step-major
 * views of the packed BiLSTM (pack_padded_sequence / pad_packed_sequence, model.py:203-210). */
int gt_gather_rows(const float* src, int64_t ld_src, const int32_t* idx, int64_t rows, int32_t C,
                   float* out, int64_t ld_out, int32_t accumulate, void* stream);
/* a*x + b*y -> out (any may alias) */
int gt_axpby(const float* x, float a, const float* y, float b, float* out, int64_t n, void* stream);
/* out (=|+=) x * (*s) * host_scale with the scalar s read on the device (upstream loss gradients
 * never visit the host); s may be NULL */
int gt_scale_dev(const float* x, const float* s, float host_scale, float* out, int64_t n,
                 int32_t accumulate, void* stream);
/* padding mask of Tacotron2.parse_output (model.py:632-642), channels-first outputs */
int gt_mask_outputs(float* mel, float* mel_post, float* gate, const int64_t* out_len, int32_t B,
                    int32_t n_mel, int32_t To, void* stream);
/* counter-based dropout keep-mask: keep[i] = u(seed, offset+i) >= p  (shared with the oracle) */
int gt_keep_mask(uint64_t seed, uint64_t offset, float p, int64_t n, uint8_t* keep, void* stream);
int gt_uniform(uint64_t seed, uint64_t offset, int64_t n, float* out, void* stream);

/* ------------------------------------------------------------------------------- discriminator
 * Discriminator.forward input shaping (model.py:523-526): mel (B, n_mel, T) channels-first ->
 * r (B, W, window*n_mel) channels-last with r[b][w][c] = X'[b].flat[c*W + w]. */
int gt_disc_gather_fwd(const float* mel, int32_t B, int32_t n_mel, int32_t T, int32_t window,
                       float* r, void* stream);
int gt_disc_gather_bwd(const float* dr, int32_t B, int32_t n_mel, int32_t T, int32_t window,
                       float* dmel, int32_t accumulate, void* stream);
/* adversarial_loss reduction (model.py:529-540) and its gradient */
int gt_adv_mean_fwd(const float* out, const int64_t* lengths, int32_t B, int32_t W, int32_t window,
                    float* loss, void* stream);
int gt_adv_mean_bwd(const float* dloss, const int64_t* lengths, int32_t B, int32_t W,
                    int32_t window, float scale, float* dout, void* stream);
/* gradient_penalty (train.py:32-69).  gt_gp_interpolate: out[b][c][t] = t < len_b ? alpha_b real + (1 - alpha_b) fake
 * : 0 (:36-49).  gt_gp_norm: g (B, n_mel, T) = d sum(D(x)) / d x; masks it (:60), adds mean_b (||g_b|| - 1)^2 with
 * the reference's 1e-12 inside the root (:67-70) to *gp and overwrites g with d gp / d g.  gt_tanh_drop_bwd2: the
 * second-order terms of a = tanh(dropout(z)) needed to differentiate the input-gradient pass itself
 * (u = p * t, e = p * da * dt/dz with t = da/dz). */
int gt_gp_interpolate(const float* real, const float* fake, const float* alpha, const int64_t* lengths,
                      int32_t B, int32_t n_mel, int32_t T, float* out, void* stream);
int gt_gp_norm(float* g, const int64_t* lengths, int32_t B, int32_t n_mel, int32_t T, float* gp /* += */,
               void* stream);
int gt_tanh_drop_bwd2(const float* p, const float* da, const float* y, int64_t rows, int32_t C,
                      const uint8_t* keep, float p_drop, float* u, float* e, void* stream);

/* -------------------------------------------------------------------------------------- losses
 * Tacotron2Loss (loss_function.py:10-22): sums[0]=MSE(mel), [1]=MSE(mel_post), [2]=BCE(gate);
 * gradients (unit upstream) written to dmel, dmel_post, dgate. */
int gt_taco_loss(const float* mel, const float* mel_post, const float* gate, const float* mel_t,
                 const float* gate_t, int64_t n_mel_elems, int64_t n_gate, float* sums3,
                 float* dmel, float* dmel_post, float* dgate, void* stream);
/* guided attention loss (loss_function.py:25-52); align (B, To, Ti); loss += ; dalign = grad */
int gt_guided_attn_loss(const float* align, const int64_t* in_len, const int64_t* out_len,
                        int32_t B, int32_t To, int32_t Ti, float* loss, float* dalign,
                        void* stream);

/* ----------------------------------------------------------------------------------- optimiser
 * clip_grad_norm_ + Adam with L2 weight decay (train.py:242-243,404-406) over one flat buffer. */
int gt_sumsq(const float* x, int64_t n, float* out /* += */, void* stream);
int gt_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr,
                 float beta1, float beta2, float eps, float weight_decay, int32_t step,
                 const float* sumsq /* NULL: no clipping */, float max_norm, void* stream);
/* the same, NOT applied when *skip_flag != 0 at the time the kernel runs (skip_flag = the word that the loop kernels with a
 * bounded in-launch wait -- gt_plstm_fwd / gt_plstm_bwd, gt_attn_step_fwd -- raise when a wait gives up: that step's
 * gradients are undefined); parameters and moments are left as they were.  skip_flag = NULL: gt_adam_step. */
int gt_adam_step_guarded(float* param, const float* grad, float* m, float* v, int64_t n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int32_t step,
                         const float* sumsq, float max_norm, const int32_t* skip_flag, void* stream);

/* -------------------------------------------------------------------------------------- STFT
 * TacotronSTFT.mel_spectrogram (layers.py:97-114, stft.py:77-105, audio_processing.py:78-84) */
int gt_reflect_pad(const float* audio, int32_t B, int64_t n, int32_t pad, float* out, void* stream);
/* ragged batch (one collated batch of raw audio, data_utils.py:51-64 + 88-131 done on the device): row b holds
 * lengths[b] samples, rows ld apart, each reflect-padded around its own ends into rows of ld + 2*pad */
int gt_reflect_pad_ragged(const float* audio, int32_t B, int64_t ld, const int64_t* lengths, int32_t pad,
                          float* out, void* stream);
/* log-mel rows (B*F, n_mel) -> mel_padded (B, n_mel, F), zero beyond frames[b]; gate_padded (B, F) = 1 from
 * frame frames[b]-1 on */
int gt_collate_mel(const float* rows, const int64_t* frames, int32_t B, int32_t n_mel, int32_t F, float* mel,
                   float* gate, void* stream);
int gt_stft_magnitude(const float* ft /* rows x 2*cutoff */, int64_t rows, int32_t cutoff,
                      float* mag, void* stream);
int gt_log_clamp(const float* x, int64_t n, float clip, float* out, void* stream);


/* ================================================================================================ GtExperimental
 * Everything below was built, is covered by kernel and parity tests, was MEASURED on MI355X and is NOT on the default path
 * ("measured and not adopted"; the switches are listed in gantron_amd/switches.py under
 * GANTRON_EXPERIMENTAL).  A binding that only wants the training step's path can ignore this section. */

/* The same two kernels as an EXACT fp32 product on the bf16 matrix pipe ("x3"): the weights are packed as three bf16 planes
 * w = w0 + w1 + w2 (all 24 significand bits; gt_rec_weight_x3_elems 16-bit elements, plane p of fragment block (nb, kb) at
 * ((nb KB + kb) 3 + p) 64 lanes x 8), the fp32 activations are split the same way on their way through LDS and
 * a . b = a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0) is accumulated in fp32, smallest terms first: every product
 * exact, dropped terms <= 2^-24 |a||b| (the arithmetic of gt_sgemm's three-plane mode).  K-slices of at most 1024:
 * splits >= gt_rec_x3_min_splits(K).
 * Measured (MI355X, round 3) and NOT adopted: the attention LSTM's fused step 31.1 us against 26.7 on the fp32 pipe, its backward
 * product 25.6 against 19.6 (pair 141 ms against 128): these launches are bound by their weight stream (6 bytes per weight
 * here instead of 4), not by the matrix pipe.  GANTRON_EXPERIMENTAL=rec_x3=<1 forward | 2 backward>. */
int64_t gt_rec_weight_x3_elems(int32_t N, int32_t K);
int32_t gt_rec_x3_min_splits(int32_t K);
int gt_pack_rec_weight_x3(const float* W, int64_t ldw, int32_t transposed, int32_t N, int32_t n_valid,
                          int32_t K, void* Wp, void* stream);
int gt_rec_gemm_x3(const float* X, int64_t ldx, const void* Wp, float* part, int32_t mtiles, int32_t N,
                   int32_t K, int32_t splits, int64_t part_sstride, int32_t nz, int64_t x_zstride,
                   int64_t w_zstride, int64_t part_zstride, void* stream);
int gt_pack_rec_cell_weight_x3(const float* W, int64_t ldw, int32_t H, int32_t K, void* Wp, void* stream);
int gt_rec_cell_fwd_x3(const float* X, int64_t ldx, const void* Wp, float* part, uint32_t* counters, int32_t Bp,
                         int32_t B, int32_t H, int32_t K, int32_t splits, const float* pre, int64_t ld_pre,
                         const float* c_prev, const uint8_t* keep, float p_drop, float* gates, float* c_out,
                         float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2,
                         const float* Wq, int64_t ldwq, float* qpart, int64_t q_sstride, int32_t A, int32_t Apad,
                         int32_t prio, void* stream);

/* The same two kernels from TWO SCALED fp16 planes per operand ("h2", round 4): gt_sgemm's mode-4 arithmetic for the recurrent
 * products -- x s = h0 + h1 with s a power of two that puts the operand tensor's largest magnitude into [2^14, 2^15), a . b =
 * (a0 b0 + a0 b1 + a1 b0) / (sa sb) as three fp16 MFMAs with fp32 accumulation, 22 significand bits per operand -- at 4 bytes
 * per weight (what the fp32 kernels stream) and 3/16 of their matrix time.  The pack finds max |W| itself (gt_absmax) and
 * keeps the word in a 16-byte trailer of the image (gt_rec_weight_h2_elems counts it); the activations' bound is `xmax` (a
 * gt_absmax-format device word of a superset of X, or NULL) and / or `xfloor` (a magnitude known a priori, 0: none).
 * One weight image per launch (no w_zstride, one direction). */
int64_t gt_rec_weight_h2_elems(int32_t N, int32_t K);
int gt_pack_rec_weight_h2(const float* W, int64_t ldw, int32_t transposed, int32_t N, int32_t n_valid,
                          int32_t K, void* Wp, void* stream);
int gt_rec_gemm_h2(const float* X, int64_t ldx, const void* Wp, float* part, int32_t mtiles, int32_t N,
                   int32_t K, int32_t splits, int64_t part_sstride, int32_t nz, int64_t x_zstride,
                   int64_t part_zstride, const uint64_t* xmax, float xfloor, void* stream);
int gt_pack_rec_cell_weight_h2(const float* W, int64_t ldw, int32_t H, int32_t K, void* Wp, void* stream);
int gt_rec_cell_fwd_h2(const float* X, int64_t ldx, const void* Wp, float* part, uint32_t* counters, int32_t Bp,
                       int32_t B, int32_t H, int32_t K, int32_t splits, const float* pre, int64_t ld_pre,
                       const float* c_prev, const uint8_t* keep, float p_drop, float* gates, float* c_out,
                       float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2,
                       const float* Wq, int64_t ldwq, float* qpart, int64_t q_sstride, int32_t A, int32_t Apad,
                       const uint64_t* xmax, float xfloor, int32_t prio, void* stream);

/* diagnostics (GANTRON_DEBUG=rec_stamps): wall-clock stamps (10 ns ticks) of the phases of the last attention-LSTM
 * gt_rec_cell_fwd launch, 8 per workgroup (0 start, 1 first operands in LDS, 2 MFMAs done, 3 tile published, 4 other slabs
 * added, 5 cell done, 6 query slab issued, 7 shader clocks (clock64) between stamps 1 and 2; 0 = not reached), into a HOST array of 8 * 4096; returns the workgroup count. */
int gt_debug_rec_stamps(int64_t* out);
/* the same for the last gt_attn_step2_fwd launch (GANTRON_DEBUG=attn_stamps): 0 start, 2 inputs staged, 3 location
 * convolution done, 4 energies done, 5 partial context done, 6 published, 7 last arriver done */
int gt_debug_attn_stamps(int64_t* out);
/* and for the tail that launch carried (GtRecTail), 4 per workgroup of the whole grid: 0 workgroup start, 1 first item
 * drawn, 2 counter ran dry, 3 items this workgroup computed; HOST array of 4 * 4096 */
int gt_debug_attn_tail_stamps(int64_t* out);

/* x[0..n) -> three bf16 planes, plane i at planes + i * plane_stride (16-bit elements): x0 = bf16(x), x1 = bf16(x - x0),
 * x2 = bf16(x - x0 - x1); x0 + x1 + x2 == x exactly (3 x 8 significand bits).  n, plane_stride multiples of 8. */
int gt_split3(const float* x, int64_t n, void* planes, int64_t plane_stride, void* stream);
/* which kernel the last gt_sgemm call of this thread's process launched (tests): 0 fp32 pipe, 1 bf16 operands, 2 in-kernel
 * plane split, 3 pre-split planes on 128 x 128 / 64 x 64 tiles, 4 pre-split planes on 256 x 128 tiles; -1 none since the
 * last query */
int32_t gt_sgemm_last_kernel(void);

/* Backward counterpart for a recurrence whose only per-step product is d h_{t-1} = dG_t . W_hh (the decoder LSTM in the
 * hoisted mode, model.py:406-409): ONE launch per step = that product (Wp: gt_pack_rec_weight of W_hh^T, i.e. [H] x [4H],
 * K = 4H; X = dG_t (Bp, ldx)) + the cell backward of step t - 1 in the last-arriving workgroup of every 64-unit column
 * group: d h = product + dh_direct (gradient of the emitted h_{t-1}), dropout, gate derivatives -> dgates = dG_{t-1}
 * (Bp, 4H), dc (Bp, H) replaced by the gradient carried on.  gates / c_new / c_prev / keep are step t - 1's.  H % 64 == 0,
 * splits <= 16; `part` (splits x Bp x H) and `counters` ((Bp > 64 ? Bp/64 : 1) x H/64, zeroed once) as in gt_rec_cell_fwd. */
int gt_rec_cellbwd(const float* X, int64_t ldx, const float* Wp, float* part, uint32_t* counters, int32_t Bp,
                   int32_t B, int32_t H, int32_t K, int32_t splits, const float* dh_direct, int64_t ld_direct, float* dc,
                   const float* gates, const float* c_new, const float* c_prev, const uint8_t* keep, float p_drop,
                   float* dgates, int32_t prio, void* stream);

/* Two products with their own shapes (both with 16*mtiles rows) in ONE launch: the decoder loops issue the
 * attention-LSTM and decoder-LSTM products of a step together so that they share the device as co-resident
 * workgroups without cross-stream synchronisation.  Arguments per problem as in gt_rec_gemm (nz = 1). */
int gt_rec_gemm2(const float* X0, int64_t ldx0, const float* Wp0, float* part0, int32_t N0, int32_t K0,
                 int32_t splits0, int64_t sstride0, const float* X1, int64_t ldx1, const float* Wp1,
                 float* part1, int32_t N1, int32_t K1, int32_t splits1, int64_t sstride1, int32_t mtiles,
                 void* stream);

/* One whole LSTMCell step forward in a single launch: gates = X.W^T + pre, pointwise cell, dropout, h
 * written to up to three operand buffers (model.py:391-394, 406-409).  Wg is the gate-interleaved packing
 * (gt_pack_rec_weight mode 2) of the (4H, K) weight matching X's column order; pre (row stride ld_pre, 0 = one
 * bias row) is added to the pre-activations; gates (Bp,4H), c_out, h_raw in the layouts of gt_lstm_cell_fwd.
 * One workgroup per 4 hidden units sees all of K: no K-split slabs, no second launch.
 * The loop is branch-free: Wg is packed by gt_pack_gates for Kp = gt_fused_k_padded(K) columns (K rounded up
 * to a multiple of 512, zeros beyond K) and X needs ldx >= Kp zero-padded columns. */
int64_t gt_fused_k_padded(int32_t K);
int gt_pack_gates(const float* W, int64_t ldw, int32_t H, int32_t K, float* Wg, void* stream);   /* Wg: 4H*Kp floats */
int gt_lstm_step_fused(const float* X, int64_t ldx, const float* Wg, int32_t K, const float* pre,
                       int64_t ld_pre, const float* c_prev, const uint8_t* keep, float p_drop, float* gates,
                       float* c_out, float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2,
                       int64_t ld2, int32_t B, int32_t Bp, int32_t H, void* stream);

/* One launch for a whole forward attention step = gt_attn_energies_fwd_u + gt_attn_context_fwd (model.py:70-91), for the
 * dimensions gt_attn_step_supported() accepts ((128, 32, 31), Ti <= 256, E % 4 == 0).  The workgroups of a batch row
 * exchange their energies INSIDE the launch: `erow` (B, Ti) receives them (write-through), `row_cnt` (B) uint32 arrival
 * counters -- zero-initialised ONCE per loop, monotonic -- tell a row's workgroups when the row is complete for step
 * number `step1` = t + 1 (1, 2, 3, ... in launch order); then every workgroup normalises the row and accumulates the
 * context for its slice of the E columns.  The wait is bounded: on time-out *timeout_flag (int32, may be NULL) is set
 * to 1 and the step's result is undefined.  Other arguments as in the two calls it replaces. */
int32_t gt_attn_step_supported(GtAttnDims d);
int gt_attn_step_fwd(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum, const float* pm,
                     const float* Wc, const float* Wd, const float* v, const int32_t* lengths, float* q_out,
                     const float* wpack, float* u_out, float* erow, uint32_t* row_cnt, uint32_t step1,
                     int32_t* timeout_flag, const float* memory, int64_t ld_mem, float* w_out, const float* cum_in,
                     float* cum_out, float* c0, int64_t ld0, float* c1, int64_t ld1, float* c2, int64_t ld2,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GANTRON_HIP_H */
