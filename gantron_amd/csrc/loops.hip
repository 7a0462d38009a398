// Host-side drivers of the two sequential loops of the generator: the teacher-forced decoder
// (reference Decoder.forward / decode, model.py:378-454) and the encoder BiLSTM (model.py:202-210),
// forward and backward-through-time.  A loop is one C call: launches are issued back-to-back on
// the caller's stream from a tight C++ loop (no per-step Python), 7 launches per decoder step.
//
// Work that does not depend on the recurrence is hoisted out of the loop by the caller and runs
// as large MFMA GEMMs: the prenet->attention-LSTM input projection for all steps (`pre1`), the
// mel/gate projection of all steps (reads HC), every weight gradient (dW = dGates^T . inputs with
// K = To*B) and d(memory) (batched W^T . dctx).  Inside the loop only the recurrent products
// ([ctx|h] . W) remain, as weight-streaming small-M MFMA kernels (recurrent.hip).
#include <stdlib.h>
#include <functional>
#include <map>
#include <mutex>
#include <vector>
#include "common.h"

extern "C" {

#define GT_TRY(call)                \
    do {                            \
        int rc__ = (call);          \
        if (rc__ != 0) return rc__; \
    } while (0)

static inline GtSlabs slabs(const float* p, int n, int64_t stride, int64_t ld) {
    GtSlabs s; s.p = p; s.n = n; s.slab_stride = stride; s.ld = ld; return s;
}
static const GtSlabs kNoSlabs = {nullptr, 0, 0, 0};

// Recurrent product for Bp activation rows: up to 64 rows are one launch with Bp/16 row tiles; larger
// batches (Bp a multiple of 64) use blockIdx.z over 64-row blocks that share the packed weights.
// prio: 1 for the products of the attention chain (the loops' serial critical path): raised wave priority.
static int rec_rows(const float* X, int64_t ldx, const float* Wp, float* part, int Bp, int N, int K, int S,
                    void* stream, int bf16 = 0, int prio = 0) {
    if (bf16) {         // Wp is a bf16 fragment image (gt_pack_rec_weight_bf16), three planes of them (bf16 == 3: _x3) or two
                        // scaled fp16 planes (bf16 == 2: _h2; the caller has named the activations' bound in g_rec_h2x)
        const int planes = bf16 == 3 ? 3 : bf16 == 2 ? 2 : 1;
        static const int h2_big = gt_env_item("GANTRON_DEBUG", "rec_h2_form", 0) & 2;
        if (planes == 2 && !h2_big) {       // the fp16-plane product in the fp32 kernel's geometry (recurrent.hip)
            const GtRecH2X hx = g_rec_h2x;
            g_rec_h2x = {nullptr, 0.f};
            if (Bp <= 64) return gt_rec_gemm_h2_lean(X, ldx, Wp, part, Bp / 16, N, K, S, (int64_t)Bp * N, 1, 0, 0, hx.xmax, hx.xfloor, prio, stream);
            return gt_rec_gemm_h2_lean(X, ldx, Wp, part, 4, N, K, S, (int64_t)Bp * N, Bp / 64, 64 * ldx, (int64_t)64 * N, hx.xmax, hx.xfloor, prio, stream);
        }
        if (Bp <= 64) return gt_rec_gemm_bf16_prio(X, ldx, Wp, part, Bp / 16, N, K, S, (int64_t)Bp * N, 1, 0, 0, 0, prio, stream, planes);
        return gt_rec_gemm_bf16_prio(X, ldx, Wp, part, 4, N, K, S, (int64_t)Bp * N, Bp / 64, 64 * ldx, 0, (int64_t)64 * N, prio,
                                     stream, planes);
    }
    if (Bp <= 64) return gt_rec_gemm_prio(X, ldx, Wp, part, Bp / 16, N, K, S, (int64_t)Bp * N, 1, 0, 0, 0, prio, stream);
    return gt_rec_gemm_prio(X, ldx, Wp, part, 4, N, K, S, (int64_t)Bp * N, Bp / 64, 64 * ldx, 0, (int64_t)64 * N, prio, stream);
}
static bool rows_ok(int B, int Bp) { return B > 0 && Bp >= B && ((Bp <= 64 && Bp % 16 == 0) || Bp % 64 == 0); }

// ---- two-chain pipelining -----------------------------------------------------------------------
// A decoder step is two dependent chains: the attention chain (attention LSTM, query, energies,
// softmax+context) and the decoder-LSTM chain.  Forward, the attention chain of step t+1 depends only
// on the attention chain of step t, so the decoder LSTM of step t runs concurrently with it on an
// auxiliary stream; backward the roles swap (the decoder-LSTM chain leads).  Every kernel of the loop
// is latency-bound at ~1 wave per SIMD, so two kernels in flight fill each other's bubbles.
// Hand-offs use a ring of `ring` buffers (X2 forward, partX2 backward) and HIP events.
// wave issue priorities (s_setprio) inside the two-chain loops: the attention chain is the serial critical path (3),
// the decoder-LSTM chain comes next (2), the weight-gradient GEMMs of the leaf stream keep the default (0)
static const int kPrioA = 3, kPrioB = 2;

// Auxiliary stream and hand-off events are per DEVICE (one process may drive several GPUs) and guarded by a mutex that
// a loop call holds while it issues its launches: the forward loop (Python main thread) and the backward loop
// (autograd's backward thread) of one device can therefore never interleave their event records.
struct LoopCtx {
    std::mutex mu;
    hipStream_t aux = nullptr;
    hipStream_t aux2 = nullptr;       // the forward loop's block GEMMs, beside the decoder-LSTM chain they feed
    std::vector<hipEvent_t> events;
};
static std::mutex g_ctx_mu;
static std::map<std::pair<int, void*>, LoopCtx*> g_ctx;

// one context (auxiliary stream, events, issue lock) per device AND caller stream: loops issued on different streams of a
// device are independent of each other
static int loop_ctx(LoopCtx** out, void* stream) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    GT_REQUIRE(e == hipSuccess, (int)e, "decoder loop: hipGetDevice failed");
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    const auto key = std::make_pair(dev, stream);
    auto it = g_ctx.find(key);
    if (it == g_ctx.end()) it = g_ctx.emplace(key, new LoopCtx()).first;
    *out = it->second;
    return 0;
}

// steps between cross-stream hand-offs for a ring of R buffers (GANTRON_DEBUG=sync_every=<n> overrides, clamped so that
// the follower's allowed lag R - k + 1 still covers a burst of k steps)
static int sync_every(int R) {
    static const int env = gt_env_item("GANTRON_DEBUG", "sync_every", 0);
    int k = env > 0 ? env : R / 2;
    if (k > (R + 1) / 2) k = (R + 1) / 2;
    return k < 1 ? 1 : k;
}

static int aux_stream(LoopCtx* c, hipStream_t* out) {
    if (c->aux == nullptr) {
        // highest priority: the loop kernels are the critical path, the weight-gradient GEMMs that
        // share the device (leaf stream, normal priority) should only fill what they leave idle
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipError_t e = hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, hi);
        GT_REQUIRE(e == hipSuccess, (int)e, "decoder loop: cannot create the auxiliary stream");
    }
    *out = c->aux;
    return 0;
}
static int aux2_stream(LoopCtx* c, hipStream_t* out) {
    if (c->aux2 == nullptr) {
        hipError_t e = hipStreamCreateWithFlags(&c->aux2, hipStreamNonBlocking);
        GT_REQUIRE(e == hipSuccess, (int)e, "decoder loop: cannot create the second auxiliary stream");
    }
    *out = c->aux2;
    return 0;
}
static int loop_event(LoopCtx* c, size_t i, hipEvent_t* out) {
    while (c->events.size() <= i) {
        hipEvent_t ev;
        hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        GT_REQUIRE(e == hipSuccess, (int)e, "decoder loop: cannot create an event");
        c->events.push_back(ev);
    }
    *out = c->events[i];
    return 0;
}
// On every exit of a two-stream loop -- the error returns included -- `stream` is made to wait for whatever was issued
// on the auxiliary stream, so that the caller never recycles buffers that auxiliary-stream kernels still use.
struct AuxJoin {
    LoopCtx* c; hipStream_t main, aux; bool on;
    ~AuxJoin() {
        if (!on) return;
        hipEvent_t ev;
        if (loop_event(c, 0, &ev) != 0) return;
        if (hipEventRecord(ev, aux) == hipSuccess) (void)hipStreamWaitEvent(main, ev, 0);
    }
};
#define GT_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        GT_REQUIRE(e__ == hipSuccess, (int)e__, "%s failed: %s", #call, hipGetErrorString(e__)); \
    } while (0)

// ---- hipGraph replay of a whole loop call (GANTRON_EXPERIMENTAL=graph; BASELINE config 5 names it) --------------------------------
// The launches of a loop call -- both streams, their hand-off events included -- are captured once into a graph and the
// graph is replayed whenever a call arrives with byte-identical arguments (same buffers, same sizes: the steady state of a
// training run whose allocator hands back the same blocks).  Host issue time drops from ~19 ms to one graph launch; what
// it does to the GPU-side duration is recorded in DESIGN.md (the loops are bound by dependent-kernel latency, not by issue).
static int graph_mode() {
    static const int v = gt_env_item("GANTRON_EXPERIMENTAL", "graph", 0);
    return v;
}
struct GraphEntry { std::vector<unsigned char> key; hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; };
static std::vector<GraphEntry> g_graphs;
static std::mutex g_graphs_mu;

static int run_graphed(const void* args, size_t nbytes, hipStream_t stream, const std::function<int()>& issue) {
    std::lock_guard<std::mutex> lk(g_graphs_mu);
    std::vector<unsigned char> key((const unsigned char*)args, (const unsigned char*)args + nbytes);
    key.insert(key.end(), (const unsigned char*)&stream, (const unsigned char*)&stream + sizeof(stream));
    for (auto& g : g_graphs)
        if (g.key == key) {
            GT_HIP(hipGraphLaunch(g.exec, stream));
            if (graph_mode() > 1) fprintf(stderr, "[graph] replay (%zu cached)\n", g_graphs.size());
            return 0;
        }
    if (graph_mode() > 1) fprintf(stderr, "[graph] capture (%zu cached)\n", g_graphs.size());
    if (g_graphs.size() >= 16) {               // new shapes / buffers: drop the old graphs (nothing of them may still run)
        GT_HIP(hipStreamSynchronize(stream));
        for (auto& g : g_graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
        g_graphs.clear();
    }
    GraphEntry e;
    e.key = key;
    GT_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed));
    g_gt_capturing = true;
    const int rc = issue();
    g_gt_capturing = false;
    hipError_t ce = hipStreamEndCapture(stream, &e.graph);
    GT_REQUIRE(rc == 0, rc, "decoder loop: a launch failed during graph capture (%s)", gt_last_error_string());
    GT_REQUIRE(ce == hipSuccess && e.graph != nullptr, (int)ce, "decoder loop: hipStreamEndCapture failed: %s", hipGetErrorString(ce));
    GT_HIP(hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0));
    GT_HIP(hipGraphLaunch(e.exec, stream));
    g_graphs.push_back(e);
    return 0;
}

static GtOperand op_kc(const float* p, int64_t ld) {
    GtOperand o = {}; o.p = p; o.mode = GT_OP_KC; o.T = o.C = o.dil = o.pad = 0; o.ld = ld; o.batch_stride = 0; return o;
}
static GtOperand op_rc(const float* p, int64_t ld) {
    GtOperand o = op_kc(p, ld); o.mode = GT_OP_RC; return o;
}
// C[M,N] (=|+=) A[M,K] . B(n,k) (+ bias) through the tiled MFMA GEMM (bf16 operands when the loop runs in that mode)
static int block_gemm(GtOperand A, GtOperand B, const float* bias, float* C, int64_t ldc, int M, int N, int K, int accumulate,
                      int bf16, void* stream, const uint64_t* b_amax = nullptr) {
    B.amax = b_amax;          // fp16-plane products: the weights' maximum is found once per loop call, not once per block
    GtGemm g = {}; g.A = A; g.B = B; g.C = C; g.ldc = ldc; g.c_batch_stride = 0; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.batch = 1; g.accumulate = accumulate; g.split_k = 1; g.alpha = 1.f; g.bf16 = bf16;
    return gt_sgemm(&g, stream);
}
static int block_steps(int blk) { return blk > 0 ? blk : 32; }

// ---- hoisted decoder-LSTM input product (GtDecoderFwd::W2hp) ---------------------------------------------------------
// The attention chain (stream A) never reads the decoder LSTM, so it simply runs ahead: per block of `blk` steps it
// leaves [h_att_t | ctx_t] rows in the X2 history, one tiled GEMM on stream B turns them into the decoder LSTM's input
// pre-activations of the whole block, and stream B then walks the block with the small recurrent product
// h_dec_{t-1} . W2hh^T (+ cell).  No ring, no back-pressure: A waits for B only at the very end.
static int decoder_fwd_hoisted(const GtDecoderFwd* a, void* stream) {
    GT_REQUIRE(a->W2ih && a->pre2 && a->Xh && a->X2 && a->part2 && a->ldw2 >= a->Ha + a->E && a->ld2 >= a->Ha + a->E &&
                   a->ldh >= 16 * cdiv(a->Hd, 16) && a->ldh % 4 == 0, GT_E_ARG, "gt_decoder_fwd: hoisted mode needs W2ih, pre2, Xh, X2 history, part2");
    const int N1 = 4 * a->Ha, N2 = 4 * a->Hd;
    const int K1 = a->E + a->Ha;
    const int64_t hc_ld = a->Hd + a->E;
    GtAttnDims d; d.B = a->B; d.Ti = a->Ti; d.E = a->E; d.A = a->A; d.F = a->F; d.Kc = a->Kc;
    const int64_t BT = (int64_t)a->B * a->Ti;
    const bool two = a->ring > 1;
    const bool cq = a->Wq != nullptr && a->partq != nullptr && a->Ha % 32 == 0 && a->ld1 % 4 == 0 && a->ld2 % 4 == 0 &&
                    a->E % 4 == 0;
    hipStream_t sA = (hipStream_t)stream, sB = sA;
    hipEvent_t ev;
    LoopCtx* ctx = nullptr;
    GT_TRY(loop_ctx(&ctx, stream));
    std::lock_guard<std::mutex> lk(ctx->mu);
    AuxJoin join{ctx, sA, sB, false};
    if (two) {
        GT_TRY(aux_stream(ctx, &sB));
        GT_TRY(loop_event(ctx, 0, &ev));
        GT_HIP(hipEventRecord(ev, sA));
        GT_HIP(hipStreamWaitEvent(sB, ev, 0));
        join.aux = sB; join.on = true;
    }
    const float* wpack = nullptr;
    if (a->attn_wpack != nullptr && gt_attn_wpack_elems(d) > 0) {
        GT_TRY(gt_attn_pack_weights(d, a->Wc, a->Wd, a->attn_wpack, sA));
        wpack = a->attn_wpack;
    }
    const int blk = block_steps(a->blk);
    const uint64_t* w2_amax = nullptr;
    if (!a->bf16 && gt_sgemm_get_fp32_split() == 4 && !g_gt_capturing)
        GT_TRY(gt_absmax_ring(a->W2ih, N2, a->Ha + a->E, a->ldw2, &w2_amax, sA));
    const int64_t xa_step = (int64_t)a->Bp * a->ld2;
    // fused product + cell (+ query) launches: float4-friendly sizes (bf16 mode: the bf16 images of the same packing)
    const bool fusedA = a->W1c != nullptr && a->counters1 != nullptr && cq && a->Ha % 16 == 0;
    const bool fusedB = a->W2hc != nullptr && a->counters2 != nullptr && a->Hd % 16 == 0 && (a->Hd + a->E) % 4 == 0;
    GT_REQUIRE((a->W1c == nullptr || fusedA) && (a->W2hc == nullptr || fusedB), GT_E_ARG,
               "gt_decoder_fwd: W1c / W2hc need Wq + partq, Ha %% 32 == 0, Hd %% 16 == 0, E %% 4 == 0 and their counters");
    const bool attn_step = a->row_cnt != nullptr && a->attn_xbuf == nullptr && gt_attn_step_supported(d) && wpack != nullptr && a->ld1 % 4 == 0 &&
                           a->ld2 % 4 == 0 && a->ldE % 4 == 0 && (a->Hd % 4 == 0) && a->To < (1 << 24);
    const bool persistB = a->W2hh != nullptr && a->pl_counters != nullptr && a->pl_timeout != nullptr && !a->bf16 &&
                          gt_plstm_supported(a->Bp, a->Hd) && hc_ld % 4 == 0 && a->pre2 == a->gates2;
    const bool attn_step2 = a->attn_xbuf != nullptr && a->row_cnt != nullptr && gt_attn_step2_supported(d) && wpack != nullptr &&
                            a->ld1 % 4 == 0 && a->ld2 % 4 == 0 && a->ldE % 4 == 0 && (a->Hd % 4 == 0) && a->To < (1 << 24);
    const int64_t x1_slot = fusedA ? (int64_t)a->Bp * a->ld1 : 0;      // operand ping-pong (see GtDecoderFwd::W1c)
    const int64_t xh_slot = fusedB ? (int64_t)a->Bp * a->ldh : 0;
    // the h_att half of the attention LSTM's product rides in the attention step in front of it (GtDecoderFwd::tail_work)
    const bool tailA = a->tail_work != nullptr && a->Sx1 >= 1 && fusedA && attn_step2 && !a->bf16 && !a->rec_x3 && a->Bp <= 64 &&
                       a->E % 16 == 0 && a->S1 + a->Sx1 <= 16 && a->Sx1 <= a->Ha / 16;
    GT_REQUIRE(a->tail_work == nullptr || tailA, GT_E_ARG,
               "gt_decoder_fwd: tail_work needs W1c + attn_xbuf, fp32, Bp <= 64, E %% 16 == 0, 1 <= Sx1 <= Ha / 16, S1 + Sx1 <= 16");
    const int tail_kb = (a->tail_kb >= 1 && a->tail_kb <= a->Ha / 16) ? a->tail_kb : a->Ha / 16;      // k-blocks of the tail
    const int k_own = K1 - 16 * tail_kb;                                                                 // what the LSTM launch keeps
    int nblock = 0;
    for (int t0 = 0; t0 < a->To;) {
        // the last `blk` steps go in quarter blocks: what stream B still has to do after A's last step is exposed
        const int len = (a->To - t0 <= blk) ? max(blk / 4, 1) : blk;
        const int t1 = min(a->To, t0 + len);
        // ---- attention chain (stream A), steps t0 .. t1-1
        for (int t = t0; t < t1; ++t) {
            const uint8_t* k1 = a->att_keep ? a->att_keep + (int64_t)t * a->Bp * a->Ha : nullptr;
            float* hc_t = a->HC + (int64_t)t * a->Bp * hc_ld;
            float* XAt = a->X2 + (int64_t)t * xa_step;         // [h_att_t | ctx_t]
            float* X1c = a->X1 + (t & 1) * x1_slot;            // [ctx_{t-1} | h_att_{t-1}]
            float* X1n = a->X1 + ((t + 1) & 1) * x1_slot;      // receives h_att_t (cell) and ctx_t (context kernel)
            if (fusedA) {       // product + cell + dropout + query slabs: one launch
                const int fmt1 = a->bf16 ? 1 : (a->rec_x3 == 2 ? 2 : a->rec_x3 ? 3 : 0);
                if (fmt1 == 2) g_rec_h2x = {a->x1max, a->x1floor};
                GT_TRY(gt_rec_cell_fwd_z(X1c, a->ld1, a->W1c, fmt1, a->part, a->counters1, a->Bp, a->B, a->Ha, K1, a->S1,
                                       a->pre1 + (int64_t)t * a->Bp * N1, N1, a->c1 + (int64_t)t * a->Bp * a->Ha, k1, a->p_att,
                                       a->gates1 + (int64_t)t * a->Bp * N1, a->c1 + (int64_t)(t + 1) * a->Bp * a->Ha,
                                       a->hatt + (int64_t)t * a->Bp * a->Ha, X1n + a->E, a->ld1, XAt, a->ld2, nullptr, 0, a->Wq,
                                       a->Ha, a->partq, (int64_t)a->Bp * a->Apad, a->A, a->Apad, nullptr, nullptr, 0, 1, nullptr,
                                       two ? kPrioA : 0, sA, (tailA && t > 0) ? k_own : 0, (tailA && t > 0) ? a->Sx1 : 0));
            } else {
                GT_TRY(rec_rows(X1c, a->ld1, a->W1p, a->part, a->Bp, N1, K1, a->S1, sA, a->bf16, two ? kPrioA : 0));
                if (cq)      // cell + query projection in one launch
                    GT_TRY(gt_cell_query_fwd(slabs(a->part, a->S1, (int64_t)a->Bp * N1, N1), a->pre1 + (int64_t)t * a->Bp * N1, N1,
                                             a->c1 + (int64_t)t * a->Bp * a->Ha, k1, a->p_att, a->gates1 + (int64_t)t * a->Bp * N1,
                                             a->c1 + (int64_t)(t + 1) * a->Bp * a->Ha, a->hatt + (int64_t)t * a->Bp * a->Ha,
                                             X1n + a->E, a->ld1, XAt, a->ld2, a->Wq, a->Ha, a->partq, (int64_t)a->Bp * a->Apad, a->B,
                                             a->Bp, a->Ha, a->A, a->Apad, sA));
                else {
                    GT_TRY(gt_lstm_cell_fwd(slabs(a->part, a->S1, (int64_t)a->Bp * N1, N1), a->pre1 + (int64_t)t * a->Bp * N1, N1,
                                            a->c1 + (int64_t)t * a->Bp * a->Ha, nullptr, 0, k1, a->p_att, nullptr,
                                            a->gates1 + (int64_t)t * a->Bp * N1, a->c1 + (int64_t)(t + 1) * a->Bp * a->Ha,
                                            a->hatt + (int64_t)t * a->Bp * a->Ha, X1n + a->E, a->ld1, XAt, a->ld2, nullptr, 0,
                                            a->B, a->Ha, sA));
                    GT_TRY(rec_rows(XAt, a->ld2, a->Wqp, a->part, a->Bp, a->Apad, a->Ha, a->Sq, sA));
                }
            }
            const GtSlabs qs = cq ? slabs(a->partq, a->Ha / (fusedA ? 16 : 32), (int64_t)a->Bp * a->Apad, a->Apad)
                                  : slabs(a->part, a->Sq, (int64_t)a->Bp * a->Apad, a->Apad);
            if (attn_step2) {       // energies + softmax + context: one launch, partial contexts, the row's last arriver combines
                GtRecTail tl = {};
                if (tailA && t + 1 < a->To) {       // h_att_t . W1[:, E:]^T for LSTM step t + 1: h_att_t is in X1n already
                    tl.X = X1n; tl.ldx = a->ld1; tl.Wp = a->W1c;
                    tl.out = a->part + (int64_t)a->S1 * a->Bp * N1; tl.out_sstride = (int64_t)a->Bp * N1;
                    tl.N = N1; tl.KB = cdiv(K1, 16); tl.kb0 = k_own / 16; tl.kb1 = tl.KB; tl.kbs = cdiv(tl.kb1 - tl.kb0, a->Sx1);
                    tl.cg0 = 0; tl.groups = N1 / 64; tl.mtiles = a->Bp / 16; tl.work = a->tail_work + t;
                }
                GT_TRY(gt_attn_step2_fwd_tail(d, qs, a->w_hist + (int64_t)t * BT, a->cum_hist + (int64_t)t * BT, a->pm, a->Wc, a->Wd,
                                              a->v, a->lengths, a->q_hist + (int64_t)t * a->B * a->A, wpack,
                                              a->u_hist ? a->u_hist + (int64_t)t * BT * a->A : nullptr, a->attn_xbuf, a->row_cnt,
                                              (uint32_t)(t + 1), a->memory, a->ldE, a->w_hist + (int64_t)(t + 1) * BT,
                                              a->cum_hist + (int64_t)t * BT, a->cum_hist + (int64_t)(t + 1) * BT, X1n, a->ld1,
                                              XAt + a->Ha, a->ld2, hc_t + a->Hd, hc_ld, &tl, sA));
                continue;
            }
            if (attn_step) {        // energies + softmax + context: one launch, the row's workgroups meet inside it
                GT_TRY(gt_attn_step_fwd(d, qs, a->w_hist + (int64_t)t * BT, a->cum_hist + (int64_t)t * BT, a->pm, a->Wc, a->Wd,
                                        a->v, a->lengths, a->q_hist + (int64_t)t * a->B * a->A, wpack,
                                        a->u_hist ? a->u_hist + (int64_t)t * BT * a->A : nullptr, a->energies, a->row_cnt,
                                        (uint32_t)(t + 1), a->attn_timeout, a->memory, a->ldE, a->w_hist + (int64_t)(t + 1) * BT,
                                        a->cum_hist + (int64_t)t * BT, a->cum_hist + (int64_t)(t + 1) * BT, X1n, a->ld1,
                                        XAt + a->Ha, a->ld2, hc_t + a->Hd, hc_ld, sA));
                continue;
            }
            GT_TRY(gt_attn_energies_fwd_u(d, qs, a->w_hist + (int64_t)t * BT, a->cum_hist + (int64_t)t * BT, a->pm, a->Wc, a->Wd,
                                          a->v, a->lengths, a->q_hist + (int64_t)t * a->B * a->A, a->energies, wpack,
                                          a->u_hist ? a->u_hist + (int64_t)t * BT * a->A : nullptr, sA));
            GT_TRY(gt_attn_context_fwd(d, a->energies, a->memory, a->ldE, a->lengths, a->w_hist + (int64_t)(t + 1) * BT,
                                       a->cum_hist + (int64_t)t * BT, a->cum_hist + (int64_t)(t + 1) * BT, X1n, a->ld1,
                                       XAt + a->Ha, a->ld2, hc_t + a->Hd, hc_ld, sA));
        }
        // ---- decoder LSTM (stream B): input pre-activations of the block, then its recurrence.  With the persistent recurrence
        // the block GEMM goes to a stream of its own (C): the GEMM of block n + 1 then runs beside the recurrence of block n
        // instead of in front of it -- 530 us per block on stream B instead of 700, which the attention chain (620) had overtaken.
        // (Round 4: whether this pays follows which chain is the longer one.  With the block GEMM on fp16 planes, ~120 us, and the
        // attention LSTM's step still on the fp32 pipe, stream B alone was enough: 111.7 against 112.2-112.4 ms per pair without the
        // third stream; with the attention chain itself shortened -- gt_rec_cell_fwd_h2 -- stream B is the longer chain again and
        // the third stream wins 109.3-109.4 against 110.2-110.5; the VESUS memory width, K = 2120: 122.9 against 124.7.)
        hipStream_t sG = sB;
        static const int gemm_aside = gt_env_item("GANTRON_SCHEDULE", "no_blk_gemm_aside", 0) == 0;
        if (two && persistB && gemm_aside && !g_gt_capturing) GT_TRY(aux2_stream(ctx, &sG));
        if (two) {
            GT_TRY(loop_event(ctx, 1 + (size_t)(nblock % 32), &ev));
            GT_HIP(hipEventRecord(ev, sA));
            GT_HIP(hipStreamWaitEvent(sG, ev, 0));
        }
        GT_TRY(block_gemm(op_kc(a->X2 + (int64_t)t0 * xa_step, a->ld2), op_kc(a->W2ih, a->ldw2), a->bias2,
                          a->pre2 + (int64_t)t0 * a->Bp * N2, N2, (t1 - t0) * a->Bp, N2, a->Ha + a->E, 0, a->bf16, sG, w2_amax));
        if (sG != sB) {
            GT_TRY(loop_event(ctx, 40 + (size_t)(nblock % 32), &ev));
            GT_HIP(hipEventRecord(ev, sG));
            GT_HIP(hipStreamWaitEvent(sB, ev, 0));
        }
        if (persistB) {
            // the block's recurrence as ONE persistent launch: h_dec_t is published into HC[t] and read from there
            GtPlstm q = {};
            q.B = a->B; q.Bp = a->Bp; q.H = a->Hd; q.T = a->To; q.ndir = 1; q.s_lo = t0; q.s_hi = t1; q.hx_ring = 0;
            q.p_drop = a->dec_keep ? a->p_dec : 0.f;
            q.Whh = a->W2hh; q.pre = a->pre2; q.gates = a->gates2; q.c = a->c2; q.hs = nullptr;
            q.hx = a->HC; q.hx_step = (int64_t)a->Bp * hc_ld; q.hx_ld = hc_ld;
            q.keep = a->dec_keep; q.active = nullptr; q.counters = a->pl_counters; q.timeout_flag = a->pl_timeout;
            GT_TRY(gt_plstm_fwd(&q, sB));
            t0 = t1;
            ++nblock;
            continue;
        }
        for (int t = t0; t < t1; ++t) {
            const uint8_t* k2 = a->dec_keep ? a->dec_keep + (int64_t)t * a->Bp * a->Hd : nullptr;
            float* hc_t = a->HC + (int64_t)t * a->Bp * hc_ld;
            float* Xhc = a->Xh + (t & 1) * xh_slot;
            float* Xhn = a->Xh + ((t + 1) & 1) * xh_slot;
            if (fusedB) {
                GT_TRY(gt_rec_cell_fwd_z(Xhc, a->ldh, a->W2hc, a->bf16, a->part2, a->counters2, a->Bp, a->B, a->Hd, a->Hd, a->S2,
                                       a->pre2 + (int64_t)t * a->Bp * N2, N2, a->c2 + (int64_t)t * a->Bp * a->Hd, k2, a->p_dec,
                                       a->gates2 + (int64_t)t * a->Bp * N2, a->c2 + (int64_t)(t + 1) * a->Bp * a->Hd, nullptr,
                                       Xhn, a->ldh, hc_t, hc_ld, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, nullptr, nullptr, 0, 1,
                                       nullptr, two ? kPrioB : 0, sB));
                continue;
            }
            GT_TRY(rec_rows(Xhc, a->ldh, a->W2hp, a->part2, a->Bp, N2, a->Hd, a->S2, sB, a->bf16, two ? kPrioB : 0));
            GT_TRY(gt_lstm_cell_fwd(slabs(a->part2, a->S2, (int64_t)a->Bp * N2, N2), a->pre2 + (int64_t)t * a->Bp * N2, N2,
                                    a->c2 + (int64_t)t * a->Bp * a->Hd, nullptr, 0, k2, a->p_dec, nullptr,
                                    a->gates2 + (int64_t)t * a->Bp * N2, a->c2 + (int64_t)(t + 1) * a->Bp * a->Hd, nullptr,
                                    Xhn, a->ldh, hc_t, hc_ld, nullptr, 0, a->B, a->Hd, sB));
        }
        t0 = t1;
        ++nblock;
    }
    return 0;       // AuxJoin: `stream` waits for stream B
}

int gt_decoder_fwd(const GtDecoderFwd* a, void* stream) {
    GT_REQUIRE(a != nullptr, GT_E_ARG, "gt_decoder_fwd: NULL");
    GT_REQUIRE(rows_ok(a->B, a->Bp), GT_E_LIMIT,
               "gt_decoder_fwd: batch %d must be padded to a multiple of 16 (<= 64) or of 64 (got %d)", a->B, a->Bp);
    GT_REQUIRE(a->To > 0 && a->Ti > 0 && a->Ha % 16 == 0 && a->Hd % 16 == 0 && a->Apad % 16 == 0, GT_E_ARG,
               "gt_decoder_fwd: bad dims");
    GT_REQUIRE(a->ring >= 1 && a->ring <= 32 && (a->ring == 1 || a->part2 != nullptr), GT_E_ARG, "gt_decoder_fwd: ring/part2");
    if (a->W2hp != nullptr) {
        if (graph_mode() && !g_gt_capturing)
            return run_graphed(a, sizeof(*a), (hipStream_t)stream, [&]() { return decoder_fwd_hoisted(a, stream); });
        return decoder_fwd_hoisted(a, stream);
    }
    const int N1 = 4 * a->Ha, N2 = 4 * a->Hd;
    const int K1 = a->E + a->Ha, K2 = a->Ha + a->E + a->Hd;
    const int64_t hc_ld = a->Hd + a->E;
    GtAttnDims d; d.B = a->B; d.Ti = a->Ti; d.E = a->E; d.A = a->A; d.F = a->F; d.Kc = a->Kc;
    const int64_t BT = (int64_t)a->B * a->Ti;
    const int R = a->ring;
    const bool two = R > 1;
    // fused LSTM steps write h into the operand buffer of the NEXT step while other workgroups may still read the
    // current one: X1 ping-pongs between two slots and X2 needs the ring (ring == 1 keeps the two-launch path)
    const bool fused = !a->bf16 && two && a->W1g != nullptr && a->W2g != nullptr && a->Ha % 4 == 0 && a->Hd % 4 == 0;
    const int64_t x1_slot = fused ? (int64_t)a->Bp * a->ld1 : 0;
    const bool fused1 = fused, fused2 = fused;
    // attention cell + query product as one launch (needs float4-friendly sizes; the fused step supersedes it)
    const bool cq = !fused1 && a->Wq != nullptr && a->partq != nullptr && a->Ha % 32 == 0 && a->ld1 % 4 == 0 && a->ld2 % 4 == 0 &&
                    a->E % 4 == 0;
    hipStream_t sA = (hipStream_t)stream, sB = sA;
    float* partB = two ? a->part2 : a->part;
    hipEvent_t ev;
    LoopCtx* ctx = nullptr;
    GT_TRY(loop_ctx(&ctx, stream));
    std::lock_guard<std::mutex> lk(ctx->mu);
    AuxJoin join{ctx, sA, sB, false};
    if (two) {
        GT_TRY(aux_stream(ctx, &sB));
        GT_TRY(loop_event(ctx, 0, &ev));
        GT_HIP(hipEventRecord(ev, sA));
        GT_HIP(hipStreamWaitEvent(sB, ev, 0));
        join.aux = sB; join.on = true;
    }
    // Cross-stream hand-offs are batched: an event record or wait costs the stream it is issued on ~6 us of
    // barrier-packet latency (measured in the kernel trace: the only gaps between back-to-back kernels), and the
    // attention chain is the critical path.  So chain A records its "context ready" event only every k-th step and
    // chain B's k steps are issued as one burst behind it; A's wait for ring-slot reuse likewise happens once per
    // k steps.  k = ring / 2 keeps B's allowed lag (ring - k + 1 steps) above a full burst.
    const int k = two ? sync_every(R) : 1;
    const int NE = R + 3;
    auto recorded = [&](int t) { return t % k == k - 1 || t == a->To - 1; };       // steps after which events are recorded
    auto cover = [&](int t) { return min((t / k) * k + k - 1, a->To - 1); };       // the recorded step that covers step t
    auto evA = [&](int t, hipEvent_t* e) { return loop_event(ctx, 1 + (size_t)((t / k) % NE), e); };
    auto evB = [&](int t, hipEvent_t* e) { return loop_event(ctx, 1 + NE + (size_t)((t / k) % NE), e); };
    const float* wpack = nullptr;
    if (a->attn_wpack != nullptr && gt_attn_wpack_elems(d) > 0) {
        GT_TRY(gt_attn_pack_weights(d, a->Wc, a->Wd, a->attn_wpack, sA));
        wpack = a->attn_wpack;
    }
    // ---- decoder LSTM chain of one step (stream B): [h_att_t | ctx_t | h_dec_{t-1}] . W2 + bias
    auto chain_b = [&](int t) -> int {
        const uint8_t* k2 = a->dec_keep ? a->dec_keep + (int64_t)t * a->Bp * a->Hd : nullptr;
        float* hc_t = a->HC + (int64_t)t * a->Bp * hc_ld;
        float* X2t = a->X2 + (int64_t)(t % R) * a->Bp * a->ld2;
        float* X2n = a->X2 + (int64_t)((t + 1) % R) * a->Bp * a->ld2;
        if (fused2) {
            GT_TRY(gt_lstm_step_fused(X2t, a->ld2, a->W2g, K2, a->bias2, 0, a->c2 + (int64_t)t * a->Bp * a->Hd, k2, a->p_dec,
                                      a->gates2 + (int64_t)t * a->Bp * N2, a->c2 + (int64_t)(t + 1) * a->Bp * a->Hd, nullptr,
                                      X2n + a->Ha + a->E, a->ld2, hc_t, hc_ld, nullptr, 0, a->B, a->Bp, a->Hd, sB));
        } else {
            GT_TRY(rec_rows(X2t, a->ld2, a->W2p, partB, a->Bp, N2, K2, a->S2, sB, a->bf16, two ? kPrioB : 0));
            GT_TRY(gt_lstm_cell_fwd(slabs(partB, a->S2, (int64_t)a->Bp * N2, N2), a->bias2, 0,
                                    a->c2 + (int64_t)t * a->Bp * a->Hd, nullptr, 0, k2, a->p_dec, nullptr,
                                    a->gates2 + (int64_t)t * a->Bp * N2, a->c2 + (int64_t)(t + 1) * a->Bp * a->Hd, nullptr,
                                    X2n + a->Ha + a->E, a->ld2, hc_t, hc_ld, nullptr, 0, a->B, a->Hd, sB));
        }
        return 0;
    };
    int last_wait_b = -1, batch_start = 0;
    for (int t = 0; t < a->To; ++t) {
        const uint8_t* k1 = a->att_keep ? a->att_keep + (int64_t)t * a->Bp * a->Ha : nullptr;
        float* hc_t = a->HC + (int64_t)t * a->Bp * hc_ld;
        float* X2t = a->X2 + (int64_t)(t % R) * a->Bp * a->ld2;
        float* X1c = a->X1 + (t & 1) * x1_slot;            // [ctx_{t-1} | h_att_{t-1}]
        float* X1n = a->X1 + ((t + 1) & 1) * x1_slot;      // receives h_att_t (cell) and ctx_t (context kernel)
        // ---- attention chain (stream A)
        if (!fused1) GT_TRY(rec_rows(X1c, a->ld1, a->W1p, a->part, a->Bp, N1, K1, a->S1, sA, a->bf16, two ? kPrioA : 0));
        if (two && t >= R) {     // X2 ring slot t%R was last read by the decoder chain of step t-R
            const int c = cover(t - R);
            if (c != last_wait_b) {
                GT_TRY(evB(c, &ev));
                GT_HIP(hipStreamWaitEvent(sA, ev, 0));
                last_wait_b = c;
            }
        }
        if (fused1)
            GT_TRY(gt_lstm_step_fused(X1c, a->ld1, a->W1g, K1, a->pre1 + (int64_t)t * a->Bp * N1, N1,
                                      a->c1 + (int64_t)t * a->Bp * a->Ha, k1, a->p_att, a->gates1 + (int64_t)t * a->Bp * N1,
                                      a->c1 + (int64_t)(t + 1) * a->Bp * a->Ha, a->hatt + (int64_t)t * a->Bp * a->Ha,
                                      X1n + a->E, a->ld1, X2t, a->ld2, nullptr, 0, a->B, a->Bp, a->Ha, sA));
        else if (cq)      // cell + query projection in one launch
            GT_TRY(gt_cell_query_fwd(slabs(a->part, a->S1, (int64_t)a->Bp * N1, N1), a->pre1 + (int64_t)t * a->Bp * N1, N1,
                                     a->c1 + (int64_t)t * a->Bp * a->Ha, k1, a->p_att, a->gates1 + (int64_t)t * a->Bp * N1,
                                     a->c1 + (int64_t)(t + 1) * a->Bp * a->Ha, a->hatt + (int64_t)t * a->Bp * a->Ha,
                                     X1n + a->E, a->ld1, X2t, a->ld2, a->Wq, a->Ha, a->partq, (int64_t)a->Bp * a->Apad, a->B,
                                     a->Bp, a->Ha, a->A, a->Apad, sA));
        else
            GT_TRY(gt_lstm_cell_fwd(slabs(a->part, a->S1, (int64_t)a->Bp * N1, N1), a->pre1 + (int64_t)t * a->Bp * N1, N1,
                                    a->c1 + (int64_t)t * a->Bp * a->Ha, nullptr, 0, k1, a->p_att, nullptr,
                                    a->gates1 + (int64_t)t * a->Bp * N1, a->c1 + (int64_t)(t + 1) * a->Bp * a->Ha,
                                    a->hatt + (int64_t)t * a->Bp * a->Ha, X1n + a->E, a->ld1, X2t, a->ld2, nullptr, 0,
                                    a->B, a->Ha, sA));
        if (!cq) GT_TRY(rec_rows(X2t, a->ld2, a->Wqp, a->part, a->Bp, a->Apad, a->Ha, a->Sq, sA));
        GT_TRY(gt_attn_energies_fwd_u(d, cq ? slabs(a->partq, a->Ha / 32, (int64_t)a->Bp * a->Apad, a->Apad)
                                            : slabs(a->part, a->Sq, (int64_t)a->Bp * a->Apad, a->Apad),
                                      a->w_hist + (int64_t)t * BT, a->cum_hist + (int64_t)t * BT, a->pm, a->Wc, a->Wd, a->v,
                                      a->lengths, a->q_hist + (int64_t)t * a->B * a->A, a->energies, wpack,
                                      a->u_hist ? a->u_hist + (int64_t)t * BT * a->A : nullptr, sA));
        GT_TRY(gt_attn_context_fwd(d, a->energies, a->memory, a->ldE, a->lengths, a->w_hist + (int64_t)(t + 1) * BT,
                                   a->cum_hist + (int64_t)t * BT, a->cum_hist + (int64_t)(t + 1) * BT, X1n, a->ld1,
                                   X2t + a->Ha, a->ld2, hc_t + a->Hd, hc_ld, sA));
        if (!two) {
            GT_TRY(chain_b(t));
        } else if (recorded(t)) {
            GT_TRY(evA(t, &ev));
            GT_HIP(hipEventRecord(ev, sA));
            GT_HIP(hipStreamWaitEvent(sB, ev, 0));
            for (int s2 = batch_start; s2 <= t; ++s2) GT_TRY(chain_b(s2));
            batch_start = t + 1;
            GT_TRY(evB(t, &ev));
            GT_HIP(hipEventRecord(ev, sB));
        }
    }
    // join (AuxJoin): everything issued on the auxiliary stream is ordered before later work on `stream`
    return 0;
}

// ---- free-running decode (Decoder.inference, model.py:456-497) ------------------------------------
__global__ void gate_stop_kernel(const float* gate, int64_t ld, int B, float threshold, int t, int* stop_step) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float s = 1.f / (1.f + __expf(-gate[(int64_t)b * ld]));
    if (s > threshold) atomicMin(stop_step + b, t);
}

static GtOperand kc(const float* p, int64_t ld) {
    GtOperand o = {}; o.p = p; o.mode = GT_OP_KC; o.T = o.C = o.dil = o.pad = 0; o.ld = ld; o.batch_stride = 0; return o;
}
static int small_linear(const float* X, int64_t ldx, const float* W, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                        int M, int N, int K, void* stream) {
    GtGemm g = {}; g.A = kc(X, ldx); g.B = kc(W, ldw); g.C = Y; g.ldc = ldy; g.c_batch_stride = 0; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.batch = 1; g.accumulate = 0; g.split_k = 1; g.alpha = 1.f;
    return gt_sgemm(&g, stream);
}

int gt_decoder_infer(const GtDecoderInfer* q, void* stream) {
    GT_REQUIRE(q != nullptr, GT_E_ARG, "gt_decoder_infer: NULL");
    const GtDecoderFwd* a = &q->step;
    GT_REQUIRE(rows_ok(a->B, a->Bp), GT_E_LIMIT, "gt_decoder_infer: batch %d must be padded to 16/64 rows (got %d)", a->B, a->Bp);
    GT_REQUIRE(a->To > 0 && a->Ti > 0 && a->Ha % 16 == 0 && a->Hd % 16 == 0 && a->Apad % 16 == 0 && q->n_mel > 0, GT_E_ARG,
               "gt_decoder_infer: bad dims");
    GT_REQUIRE(a->att_keep == nullptr && a->dec_keep == nullptr && a->ring == 1, GT_E_ARG,
               "gt_decoder_infer: evaluation mode only (no LSTM dropout, one stream)");
    GT_REQUIRE(q->t_lo >= 0 && q->t_lo <= q->t_hi && q->t_hi <= a->To, GT_E_ARG, "gt_decoder_infer: step range [%d,%d) of %d",
               q->t_lo, q->t_hi, a->To);
    GT_REQUIRE(q->W0 && q->W1 && q->W1x && q->bias1 && q->keep0 && q->keep1 && q->Wpg && q->bpg && q->p1 && q->p2 && q->frames &&
               q->stop_step && a->pre1, GT_E_ARG, "gt_decoder_infer: NULL pointer");
    const int N1 = 4 * a->Ha, N2 = 4 * a->Hd, P = a->P, nout = q->n_mel + 1;
    const int K1 = a->E + a->Ha, K2 = a->Ha + a->E + a->Hd;
    const int64_t hc_ld = a->Hd + a->E;
    GtAttnDims d; d.B = a->B; d.Ti = a->Ti; d.E = a->E; d.A = a->A; d.F = a->F; d.Kc = a->Kc;
    const int64_t BT = (int64_t)a->B * a->Ti;
    float* pre1 = const_cast<float*>(a->pre1);
    hipStream_t s = (hipStream_t)stream;
    for (int t = q->t_lo; t < q->t_hi; ++t) {
        const float* frame = q->frames + (int64_t)t * a->Bp * nout;
        float* out_t = q->frames + (int64_t)(t + 1) * a->Bp * nout;
        float* hc_t = a->HC + (int64_t)t * a->Bp * hc_ld;
        // Prenet on the previous frame, then its share of the attention-LSTM gate pre-activations
        GT_TRY(small_linear(frame, nout, q->W0, q->n_mel, nullptr, q->p1, P, a->Bp, P, q->n_mel, stream));
        GT_TRY(gt_bias_act_drop_fwd(q->p1, nullptr, a->Bp, P, 1, 0, q->keep0 + (int64_t)t * a->Bp * P, 0.5f, q->p1, stream));
        GT_TRY(small_linear(q->p1, P, q->W1, P, nullptr, q->p2, P, a->Bp, P, P, stream));
        GT_TRY(gt_bias_act_drop_fwd(q->p2, nullptr, a->Bp, P, 1, 0, q->keep1 + (int64_t)t * a->Bp * P, 0.5f, q->p2, stream));
        GT_TRY(small_linear(q->p2, P, q->W1x, q->ldw1x, q->bias1, pre1, N1, a->Bp, N1, P, stream));
        // decode step (same sequence as gt_decoder_fwd, single stream, single X2 buffer)
        GT_TRY(rec_rows(a->X1, a->ld1, a->W1p, a->part, a->Bp, N1, K1, a->S1, s, a->bf16));
        GT_TRY(gt_lstm_cell_fwd(slabs(a->part, a->S1, (int64_t)a->Bp * N1, N1), pre1, N1,
                                a->c1 + (int64_t)t * a->Bp * a->Ha, nullptr, 0, nullptr, 0.f, nullptr,
                                a->gates1 + (int64_t)t * a->Bp * N1, a->c1 + (int64_t)(t + 1) * a->Bp * a->Ha,
                                a->hatt + (int64_t)t * a->Bp * a->Ha, a->X1 + a->E, a->ld1, a->X2, a->ld2, nullptr, 0,
                                a->B, a->Ha, s));
        GT_TRY(rec_rows(a->X2, a->ld2, a->Wqp, a->part, a->Bp, a->Apad, a->Ha, a->Sq, s));
        GT_TRY(gt_attn_energies_fwd(d, slabs(a->part, a->Sq, (int64_t)a->Bp * a->Apad, a->Apad),
                                    a->w_hist + (int64_t)t * BT, a->cum_hist + (int64_t)t * BT, a->pm, a->Wc, a->Wd, a->v,
                                    a->lengths, a->q_hist + (int64_t)t * a->B * a->A, a->energies, s));
        GT_TRY(gt_attn_context_fwd(d, a->energies, a->memory, a->ldE, a->lengths, a->w_hist + (int64_t)(t + 1) * BT,
                                   a->cum_hist + (int64_t)t * BT, a->cum_hist + (int64_t)(t + 1) * BT, a->X1, a->ld1,
                                   a->X2 + a->Ha, a->ld2, hc_t + a->Hd, hc_ld, s));
        GT_TRY(rec_rows(a->X2, a->ld2, a->W2p, a->part, a->Bp, N2, K2, a->S2, s, a->bf16));
        GT_TRY(gt_lstm_cell_fwd(slabs(a->part, a->S2, (int64_t)a->Bp * N2, N2), a->bias2, 0,
                                a->c2 + (int64_t)t * a->Bp * a->Hd, nullptr, 0, nullptr, 0.f, nullptr,
                                a->gates2 + (int64_t)t * a->Bp * N2, a->c2 + (int64_t)(t + 1) * a->Bp * a->Hd, nullptr,
                                a->X2 + a->Ha + a->E, a->ld2, hc_t, hc_ld, nullptr, 0, a->B, a->Hd, s));
        // [mel | gate] of this step = next step's input frame, and the stop test
        GT_TRY(small_linear(hc_t, hc_ld, q->Wpg, hc_ld, q->bpg, out_t, nout, a->Bp, nout, (int)hc_ld, stream));
        hipLaunchKernelGGL(gate_stop_kernel, dim3((a->B + 63) / 64), dim3(64), 0, s, out_t + q->n_mel, (int64_t)nout, a->B,
                           q->gate_threshold, t, q->stop_step);
        GT_LAUNCH_CHECK();
    }
    return 0;
}

// ---- hoisted decoder-LSTM input gradient (GtDecoderBwd::W2hTp): the mirror of decoder_fwd_hoisted ---------------------
// The decoder-LSTM chain (stream B) leads and carries only d h_dec through its per-step product dG2_t . W2hh; after a
// block of steps ONE tiled GEMM produces d[h_att_t | ctx_t] = dG2_t . W2ih for the whole block (dXA), and the attention
// chain (stream A) then walks the block.  Across the calls of a chunked backward (t_hi / t_lo) stream B keeps running
// ahead: it is synchronised with `stream` once, at the first call.
static int decoder_bwd_hoisted(const GtDecoderBwd* a, void* stream) {
    const int N1 = 4 * a->Ha, N2 = 4 * a->Hd;
    const int Hdpad = 16 * cdiv(a->Hd, 16);
    GT_REQUIRE(a->W2ih && a->dXA && a->partXh && a->ldw2 >= a->Ha + a->E && a->ldxa >= a->Ha + a->E && a->ldxa % 4 == 0, GT_E_ARG,
               "gt_decoder_bwd: hoisted mode needs W2ih, dXA (ldxa %% 4 == 0) and partXh");
    GT_REQUIRE(a->rec_x3 != 2 || (a->amaxG1 && a->rowcntA && a->Wq && a->sum_hist && a->dq2_hist && a->dwcat2_hist && a->p_hist &&
                                  a->u_hist && a->de_hist && !a->bf16), GT_E_ARG,
               "gt_decoder_bwd: rec_x3 = 2 (fp16-plane backward product) needs amaxG1 and the attention step with the cell inside (rowcntA ...)");
    const int64_t hc_ld = a->Hd + a->E;
    GtAttnDims d; d.B = a->B; d.Ti = a->Ti; d.E = a->E; d.A = a->A; d.F = a->F; d.Kc = a->Kc;
    const int64_t BT = (int64_t)a->B * a->Ti;
    const int64_t sX1 = (int64_t)a->Bp * a->N1pad, sXh = (int64_t)a->Bp * Hdpad, sQ = (int64_t)a->Bp * a->Ha;
    const bool two = a->ring > 1;
    const bool bq = a->Wq != nullptr && a->Ha % 16 == 0 && a->E % 4 == 0 && a->N1pad % 4 == 0;
    const int t_hi = (a->t_hi >= a->t_lo && a->t_hi < a->To) ? a->t_hi : a->To - 1;
    const int t_lo = (a->t_hi >= a->t_lo && a->t_lo >= 0 && a->t_hi < a->To) ? a->t_lo : 0;
    hipStream_t sA = (hipStream_t)stream, sB = sA;
    hipEvent_t ev;
    LoopCtx* ctx = nullptr;
    GT_TRY(loop_ctx(&ctx, stream));
    std::lock_guard<std::mutex> lk(ctx->mu);
    AuxJoin join{ctx, sA, sB, false};
    if (two) {
        GT_TRY(aux_stream(ctx, &sB));
        if (t_hi == a->To - 1 || g_gt_capturing) {   // first call of a backward pass: the inputs (dHC ...) were produced on
                                                     // `stream`; under capture every call forks the auxiliary stream anew
            GT_TRY(loop_event(ctx, 0, &ev));
            GT_HIP(hipEventRecord(ev, sA));
            GT_HIP(hipStreamWaitEvent(sB, ev, 0));
        }
        join.aux = sB; join.on = true;
    }
    const float* wpack = nullptr;
    if (a->attn_wpack != nullptr && gt_attn_wpack_elems(d) > 0) {
        GT_TRY(gt_attn_pack_weights(d, a->Wc, a->Wd, a->attn_wpack, sA));
        wpack = a->attn_wpack;
    }
    const int blk = block_steps(a->blk);
    const uint64_t* w2_amax = nullptr;
    if (!a->bf16 && gt_sgemm_get_fp32_split() == 4 && !g_gt_capturing)       // (stream B issues the block GEMMs)
        GT_TRY(gt_absmax_ring(a->W2ih, N2, a->Ha + a->E, a->ldw2, &w2_amax, sB));
    int nblock = 0;
    // GANTRON_DEBUG=block_timing (diagnostic): events around the B-chain block, its block GEMM and the A-chain block; printed
    // at the end of the call (synchronises)
    static const bool block_timing = gt_env_item("GANTRON_DEBUG", "block_timing", 0) != 0;
    struct BT_ { hipEvent_t b0, b1, g1, a0, a1; int steps; };
    std::vector<BT_> bts;
    auto mark = [&](hipEvent_t* e, hipStream_t st) { if (block_timing) { (void)hipEventCreate(e); (void)hipEventRecord(*e, st); } };
    for (int b1 = t_hi; b1 >= t_lo;) {
        // the very first blocks of a backward pass are short: stream A idles until stream B has finished one
        // (short first blocks at the start of EVERY chunk call, round 3: 123.0-123.5 against 122.9-123.4 ms -- nothing)
        const int len = (t_hi == a->To - 1 && nblock < 2) ? max(blk / 4, 1) : blk;
        const int b0 = max(t_lo, b1 - len + 1);
        BT_ bt; bt.steps = b1 - b0 + 1;
        mark(&bt.b0, sB);
        // ---- decoder LSTM chain (stream B, leads), steps b1 .. b0
        // countersB != NULL: the product of step t + 1 and the cell backward of step t are ONE launch (gt_rec_cellbwd)
        const bool fusedB = a->countersB != nullptr && !a->bf16 && a->Hd % 64 == 0 && a->S2 <= 16;
        const bool persistB = a->W2hh != nullptr && a->pl_xbuf != nullptr && a->pl_counters != nullptr && a->pl_timeout != nullptr &&
                              !a->bf16 && gt_plstm_supported(a->Bp, a->Hd) && hc_ld % 4 == 0;
        if (persistB) {
            // the block's backward recurrence as ONE persistent launch; the exchange tiles of step b1 + 1 are in pl_xbuf
            GtPlstmBwd q = {};
            q.B = a->B; q.Bp = a->Bp; q.H = a->Hd; q.T = a->To; q.ndir = 1; q.s_lo = b0; q.s_hi = b1 + 1;
            q.recurrent_in = b1 < a->To - 1 ? 1 : 0;
            q.p_drop = a->dec_keep ? a->p_dec : 0.f;
            q.Whh = a->W2hh; q.gates = a->gates2; q.c = a->c2;
            q.dh_direct = a->dHC; q.dd_step = (int64_t)a->Bp * hc_ld; q.dd_ld = hc_ld;
            q.dG = a->dG2; q.dc = a->dc2; q.xbuf = a->pl_xbuf;
            q.keep = a->dec_keep; q.active = nullptr; q.counters = a->pl_counters; q.timeout_flag = a->pl_timeout;
            GT_TRY(gt_plstm_bwd(&q, sB));
        }
        // GANTRON_DEBUG=bwd_skip=<bits> (diagnostics, results invalid): 1 = no per-step launches of the decoder-LSTM chain,
        // 2 = no block GEMM behind them -- what do they cost the attention chain they run beside?
        static const int bwd_skip = gt_env_item("GANTRON_DEBUG", "bwd_skip", 0);
        for (int t = b1; t >= b0 && !persistB && !(bwd_skip & 1); --t) {
            const bool last = (t == a->To - 1);
            const uint8_t* k2 = a->dec_keep ? a->dec_keep + (int64_t)t * a->Bp * a->Hd : nullptr;
            const float* dhc_t = a->dHC + (int64_t)t * a->Bp * hc_ld;
            float* dG2_t = a->dG2 + (int64_t)t * a->Bp * N2;
            if (fusedB && !last) {
                GT_TRY(gt_rec_cellbwd(a->dG2 + (int64_t)(t + 1) * a->Bp * N2, N2, a->W2hTp, a->partXh, a->countersB, a->Bp, a->B,
                                      a->Hd, N2, a->S2, dhc_t, hc_ld, a->dc2, a->gates2 + (int64_t)t * a->Bp * N2,
                                      a->c2 + (int64_t)(t + 1) * a->Bp * a->Hd, a->c2 + (int64_t)t * a->Bp * a->Hd, k2, a->p_dec,
                                      dG2_t, two ? kPrioB : 0, sB));
                continue;
            }
            GT_TRY(gt_lstm_cell_bwd(last ? kNoSlabs : slabs(a->partXh, a->S2, sXh, Hdpad), kNoSlabs, kNoSlabs, dhc_t, hc_ld,
                                    a->dc2, a->gates2 + (int64_t)t * a->Bp * N2, a->c2 + (int64_t)(t + 1) * a->Bp * a->Hd,
                                    a->c2 + (int64_t)t * a->Bp * a->Hd, k2, a->p_dec, nullptr, dG2_t, a->dc2, nullptr, a->B,
                                    a->Hd, sB));
            if (t > 0 && !fusedB) GT_TRY(rec_rows(dG2_t, N2, a->W2hTp, a->partXh, a->Bp, Hdpad, N2, a->S2, sB, a->bf16, two ? kPrioB : 0));
        }
        mark(&bt.b1, sB);
        if (!(bwd_skip & 2))
        GT_TRY(block_gemm(op_kc(a->dG2 + (int64_t)b0 * a->Bp * N2, N2), op_rc(a->W2ih, a->ldw2), nullptr,
                          a->dXA + (int64_t)b0 * a->Bp * a->ldxa, a->ldxa, (b1 - b0 + 1) * a->Bp, a->Ha + a->E, N2, 0, a->bf16, sB, w2_amax));
        mark(&bt.g1, sB);
        if (two) {
            GT_TRY(loop_event(ctx, 1 + (size_t)(nblock % 32), &ev));
            GT_HIP(hipEventRecord(ev, sB));
            GT_HIP(hipStreamWaitEvent(sA, ev, 0));
        }
        mark(&bt.a0, sA);
        // ---- attention chain (stream A), steps b1 .. b0
        for (int t = b1; t >= b0; --t) {
            const bool last = (t == a->To - 1);
            const uint8_t* k1 = a->att_keep ? a->att_keep + (int64_t)t * a->Bp * a->Ha : nullptr;
            const float* dhc_t = a->dHC + (int64_t)t * a->Bp * hc_ld;
            const float* dXA_t = a->dXA + (int64_t)t * a->Bp * a->ldxa;
            float* dG1_t = a->dG1 + (int64_t)t * a->Bp * N1;
            float* dq_t = a->dq_hist + (int64_t)t * a->Bp * a->Apad;
            float* dwcat_in = a->dwcat + (int64_t)((t + 1) & 1) * 2 * BT;
            float* dwcat_out = a->dwcat + (int64_t)(t & 1) * 2 * BT;
            const bool fused_attn = a->sum_hist != nullptr && a->dq2_hist != nullptr && a->dwcat2_hist != nullptr &&
                                    a->p_hist != nullptr && a->u_hist != nullptr && a->de_hist != nullptr;
            const int nch = gt_attn_chunks(a->Ti);
            if (fused_attn && a->rowcntA != nullptr && bq) {
                // ... and the attention LSTM's cell backward inside the same launch (each row's last workgroup): two dependent
                // launches per step
                GtCellBwdTail cell = {};
                cell.dh_a = slabs(dXA_t, 1, 0, a->ldxa);
                cell.dh_c = last ? kNoSlabs : slabs(a->partX1 + a->E, a->S1, sX1, a->N1pad);
                cell.dq2 = a->dq2_hist + (int64_t)t * a->Bp * a->Apad; cell.Wq = a->Wq; cell.ldwq = a->Ha;
                cell.dc_next = a->dc1; cell.gates = a->gates1 + (int64_t)t * a->Bp * N1;
                cell.c_new = a->c1 + (int64_t)(t + 1) * a->Bp * a->Ha; cell.c_prev = a->c1 + (int64_t)t * a->Bp * a->Ha;
                cell.keep = k1; cell.p_drop = a->p_att; cell.dgates = dG1_t; cell.dc_prev = a->dc1; cell.H = a->Ha;
                cell.row_cnt = a->rowcntA; cell.launch = (uint32_t)(a->To - t);
                const bool h2b = !a->bf16 && a->rec_x3 == 2 && a->amaxG1 != nullptr;
                cell.dg_amax = h2b ? a->amaxG1 + t : nullptr;
                GT_TRY(gt_attn_bwd_step_cell(d, slabs(dXA_t + a->Ha, 1, 0, a->ldxa),
                                             last ? kNoSlabs : slabs(a->partX1, a->S1, sX1, a->N1pad), dhc_t + a->Hd, hc_ld,
                                             a->memory, a->ldE, a->lengths, dwcat_in, a->dcum_acc,
                                             a->dalign ? a->dalign + (int64_t)t * BT : nullptr,
                                             a->dctx_hist + (int64_t)t * a->B * a->E, a->E, a->w_hist + (int64_t)(t + 1) * BT,
                                             a->u_hist + (int64_t)t * BT * a->A, a->v, a->p_hist + (int64_t)t * BT * 64,
                                             last ? nullptr : a->sum_hist + (int64_t)(t + 1) * a->B * nch,
                                             last ? nullptr : a->dwcat2_hist + (int64_t)(t + 1) * 2 * BT,
                                             a->sum_hist + (int64_t)t * a->B * nch, dq_t, a->Apad, dwcat_out,
                                             a->de_hist + (int64_t)t * BT, &cell, sA));
                if (t > 0) {
                    if (h2b) g_rec_h2x = {a->amaxG1 + t, 0.f};
                    GT_TRY(rec_rows(dG1_t, N1, a->W1Tp, a->partX1, a->Bp, a->N1pad, N1, a->S1, sA,
                                    a->bf16 ? 1 : h2b ? 2 : (a->rec_x3 ? 3 : 0), two ? kPrioA : 0));
                }
                continue;
            }
            if (fused_attn) {
                // context + energies backward of the step in ONE launch (gt_attn_bwd_step); the row sums of the softmax
                // backward are applied by the consumers: the next call here (carried planes), the cell backward below
                GT_TRY(gt_attn_bwd_step(d, slabs(dXA_t + a->Ha, 1, 0, a->ldxa),
                                        last ? kNoSlabs : slabs(a->partX1, a->S1, sX1, a->N1pad), dhc_t + a->Hd, hc_ld,
                                        a->memory, a->ldE, a->lengths, dwcat_in, a->dcum_acc,
                                        a->dalign ? a->dalign + (int64_t)t * BT : nullptr,
                                        a->dctx_hist + (int64_t)t * a->B * a->E, a->E, a->w_hist + (int64_t)(t + 1) * BT,
                                        a->u_hist + (int64_t)t * BT * a->A, a->v, a->p_hist + (int64_t)t * BT * 64,
                                        last ? nullptr : a->sum_hist + (int64_t)(t + 1) * a->B * nch,
                                        last ? nullptr : a->dwcat2_hist + (int64_t)(t + 1) * 2 * BT,
                                        a->sum_hist + (int64_t)t * a->B * nch, dq_t, a->Apad, dwcat_out,
                                        a->de_hist + (int64_t)t * BT, sA));
                GT_REQUIRE(bq, GT_E_ARG, "gt_decoder_bwd: the fused attention step needs the query gradient inside the cell backward (Wq)");
                GT_TRY(gt_cell_bwd_query_c(slabs(dXA_t, 1, 0, a->ldxa),
                                           last ? kNoSlabs : slabs(a->partX1 + a->E, a->S1, sX1, a->N1pad), dq_t, a->Apad,
                                           a->sum_hist + (int64_t)t * a->B * nch, nch,
                                           a->dq2_hist + (int64_t)t * a->Bp * a->Apad, a->Wq, a->Ha, a->dc1,
                                           a->gates1 + (int64_t)t * a->Bp * N1, a->c1 + (int64_t)(t + 1) * a->Bp * a->Ha,
                                           a->c1 + (int64_t)t * a->Bp * a->Ha, k1, a->p_att, dG1_t, a->dc1, a->B, a->Bp, a->Ha,
                                           a->A, sA));
                if (t > 0) GT_TRY(rec_rows(dG1_t, N1, a->W1Tp, a->partX1, a->Bp, a->N1pad, N1, a->S1, sA, a->bf16 ? 1 : (a->rec_x3 ? 3 : 0), two ? kPrioA : 0));
                continue;
            }
            GT_TRY(gt_attn_context_bwd(d, slabs(dXA_t + a->Ha, 1, 0, a->ldxa),
                                       last ? kNoSlabs : slabs(a->partX1, a->S1, sX1, a->N1pad), dhc_t + a->Hd, hc_ld,
                                       a->memory, a->ldE, a->lengths, dwcat_in, a->dcum_acc,
                                       a->dalign ? a->dalign + (int64_t)t * BT : nullptr,
                                       a->dctx_hist + (int64_t)t * a->B * a->E, a->E, dq_t, a->Apad, a->dw, sA));
            if (a->u_hist != nullptr && a->de_hist != nullptr) {
                GT_TRY(gt_attn_energies_bwd_rec(d, a->dw, a->w_hist + (int64_t)(t + 1) * BT, a->u_hist + (int64_t)t * BT * a->A,
                                                a->Wc, a->Wd, a->v, a->lengths, dq_t, a->Apad, dwcat_out,
                                                a->de_hist + (int64_t)t * BT, wpack, sA));
            } else {
                GT_TRY(gt_attn_energies_bwd_w(d, a->dw, a->w_hist + (int64_t)(t + 1) * BT, a->q_hist + (int64_t)t * a->B * a->A,
                                              a->w_hist + (int64_t)t * BT, a->cum_hist + (int64_t)t * BT, a->pm, a->Wc, a->Wd, a->v,
                                              a->lengths, dq_t, a->Apad, a->dpm, dwcat_out, a->acc, wpack, sA));
            }
            if (bq) {
                GT_TRY(gt_cell_bwd_query(slabs(dXA_t, 1, 0, a->ldxa),
                                         last ? kNoSlabs : slabs(a->partX1 + a->E, a->S1, sX1, a->N1pad), dq_t, a->Apad, a->Wq,
                                         a->Ha, a->dc1, a->gates1 + (int64_t)t * a->Bp * N1, a->c1 + (int64_t)(t + 1) * a->Bp * a->Ha,
                                         a->c1 + (int64_t)t * a->Bp * a->Ha, k1, a->p_att, dG1_t, a->dc1, a->B, a->Bp, a->Ha, a->A,
                                         sA));
            } else {
                GT_TRY(rec_rows(dq_t, a->Apad, a->WqTp, a->partQ, a->Bp, a->Ha, a->A, a->Sq, sA));
                GT_TRY(gt_lstm_cell_bwd(slabs(dXA_t, 1, 0, a->ldxa), slabs(a->partQ, a->Sq, sQ, a->Ha),
                                        last ? kNoSlabs : slabs(a->partX1 + a->E, a->S1, sX1, a->N1pad), nullptr, 0, a->dc1,
                                        a->gates1 + (int64_t)t * a->Bp * N1, a->c1 + (int64_t)(t + 1) * a->Bp * a->Ha,
                                        a->c1 + (int64_t)t * a->Bp * a->Ha, k1, a->p_att, nullptr, dG1_t, a->dc1, nullptr, a->B,
                                        a->Ha, sA));
            }
            if (t > 0) GT_TRY(rec_rows(dG1_t, N1, a->W1Tp, a->partX1, a->Bp, a->N1pad, N1, a->S1, sA, a->bf16 ? 1 : (a->rec_x3 ? 3 : 0), two ? kPrioA : 0));
        }
        mark(&bt.a1, sA);
        if (block_timing) bts.push_back(bt);
        b1 = b0 - 1;
        ++nblock;
    }
    if (block_timing && !bts.empty()) {
        (void)hipStreamSynchronize(sA); (void)hipStreamSynchronize(sB);
        for (size_t i = 0; i < bts.size(); ++i) {
            float bms = 0, gms = 0, ams = 0, lag = 0, await = 0;
            (void)hipEventElapsedTime(&bms, bts[i].b0, bts[i].b1);
            (void)hipEventElapsedTime(&gms, bts[i].b1, bts[i].g1);
            (void)hipEventElapsedTime(&ams, bts[i].a0, bts[i].a1);
            (void)hipEventElapsedTime(&lag, bts[i].g1, bts[i].a0);           // > 0: A reached the block after B's GEMM was done
            if (i > 0) (void)hipEventElapsedTime(&await, bts[i - 1].a1, bts[i].a0);
            fprintf(stderr, "[block] steps %2d  B chain %7.1f us (%.1f/step)  dXA GEMM %6.1f us  A chain %7.1f us (%.1f/step)  "
                            "B-done -> A-start %8.1f us  A idle before block %6.1f us\n", bts[i].steps, bms * 1e3,
                    bms * 1e3 / bts[i].steps, gms * 1e3, ams * 1e3, ams * 1e3 / bts[i].steps, lag * 1e3, await * 1e3);
        }
    }
    // (Measured and not adopted: letting the decoder-LSTM chain keep its lead across the calls of a chunked backward instead
    // of joining it at the end of every call -- each call then starts without the attention chain waiting ~0.7 ms for the other
    // chain's first block -- 132.7 / 133.6 against 132.8 / 132.9 ms: that wait coincided with the previous chunk's
    // weight-gradient GEMMs, which simply ran faster while it lasted.)
    return 0;
}

int gt_decoder_bwd(const GtDecoderBwd* a, void* stream) {
    GT_REQUIRE(a != nullptr, GT_E_ARG, "gt_decoder_bwd: NULL");
    GT_REQUIRE(rows_ok(a->B, a->Bp), GT_E_LIMIT, "gt_decoder_bwd: batch padding");
    GT_REQUIRE(a->ring >= 1 && a->ring <= 32, GT_E_ARG, "gt_decoder_bwd: ring");
    if (a->W2hTp != nullptr) {
        if (graph_mode() && !g_gt_capturing)
            return run_graphed(a, sizeof(*a), (hipStream_t)stream, [&]() { return decoder_bwd_hoisted(a, stream); });
        return decoder_bwd_hoisted(a, stream);
    }
    GT_REQUIRE(a->rec_x3 != 2, GT_E_ARG, "gt_decoder_bwd: rec_x3 = 2 needs the hoisted mode");
    const int N1 = 4 * a->Ha, N2 = 4 * a->Hd;
    const int64_t hc_ld = a->Hd + a->E;
    GtAttnDims d; d.B = a->B; d.Ti = a->Ti; d.E = a->E; d.A = a->A; d.F = a->F; d.Kc = a->Kc;
    const int64_t BT = (int64_t)a->B * a->Ti;
    const int64_t sX1 = (int64_t)a->Bp * a->N1pad, sX2 = (int64_t)a->Bp * a->N2pad, sQ = (int64_t)a->Bp * a->Ha;
    const int R = a->ring;
    const bool two = R > 1;
    hipStream_t sA = (hipStream_t)stream, sB = sA;
    hipEvent_t ev;
    LoopCtx* ctx = nullptr;
    GT_TRY(loop_ctx(&ctx, stream));
    std::lock_guard<std::mutex> lk(ctx->mu);
    AuxJoin join{ctx, sA, sB, false};
    if (two) {
        GT_TRY(aux_stream(ctx, &sB));
        GT_TRY(loop_event(ctx, 0, &ev));
        GT_HIP(hipEventRecord(ev, sA));
        GT_HIP(hipStreamWaitEvent(sB, ev, 0));
        join.aux = sB; join.on = true;
    }
    // batched cross-stream hand-offs (see gt_decoder_fwd): chain B leads here and records "product ready" only at
    // steps t % k == 0 (and at the last step of the call); the k attention-chain steps behind it are then issued as
    // one burst after a single wait.  Chain A records "ring slot free" at steps t % k == 0.
    const int k = two ? sync_every(R) : 1;
    const int NE = R + 3;
    auto evA = [&](int t, hipEvent_t* e) { return loop_event(ctx, 1 + (size_t)((t / k) % NE), e); };
    auto evB = [&](int t, hipEvent_t* e) { return loop_event(ctx, 1 + NE + (size_t)((t / k) % NE), e); };
    auto ringX2 = [&](int t) { return a->partX2 + (int64_t)(t % R) * a->S2 * sX2; };
    const bool bq = a->Wq != nullptr && a->Ha % 16 == 0 && a->E % 4 == 0 && a->N1pad % 4 == 0 && a->N2pad % 4 == 0;
    const int t_hi = (a->t_hi >= a->t_lo && a->t_hi < a->To) ? a->t_hi : a->To - 1;
    const int t_lo = (a->t_hi >= a->t_lo && a->t_lo >= 0 && a->t_hi < a->To) ? a->t_lo : 0;
    const float* wpack = nullptr;
    if (a->attn_wpack != nullptr && gt_attn_wpack_elems(d) > 0) {
        GT_TRY(gt_attn_pack_weights(d, a->Wc, a->Wd, a->attn_wpack, sA));
        wpack = a->attn_wpack;
    }
    // ---- attention chain of one step (stream A)
    auto chain_a = [&](int t) -> int {
        const bool last = (t == a->To - 1);
        const uint8_t* k1 = a->att_keep ? a->att_keep + (int64_t)t * a->Bp * a->Ha : nullptr;
        const float* dhc_t = a->dHC + (int64_t)t * a->Bp * hc_ld;
        float* dG1_t = a->dG1 + (int64_t)t * a->Bp * N1;
        float* dq_t = a->dq_hist + (int64_t)t * a->Bp * a->Apad;
        float* pX2 = ringX2(t);
        float* dwcat_in = a->dwcat + (int64_t)((t + 1) & 1) * 2 * BT;
        float* dwcat_out = a->dwcat + (int64_t)(t & 1) * 2 * BT;
        GT_TRY(gt_attn_context_bwd(d, slabs(pX2 + a->Ha, a->S2, sX2, a->N2pad),
                                   last ? kNoSlabs : slabs(a->partX1, a->S1, sX1, a->N1pad), dhc_t + a->Hd, hc_ld,
                                   a->memory, a->ldE, a->lengths, dwcat_in, a->dcum_acc,
                                   a->dalign ? a->dalign + (int64_t)t * BT : nullptr,
                                   a->dctx_hist + (int64_t)t * a->B * a->E, a->E, dq_t, a->Apad, a->dw, sA));
        if (a->u_hist != nullptr && a->de_hist != nullptr) {
            // hoisted form: only the recurrent terms here, the sums over steps in gt_attn_param_grads (caller)
            GT_TRY(gt_attn_energies_bwd_rec(d, a->dw, a->w_hist + (int64_t)(t + 1) * BT, a->u_hist + (int64_t)t * BT * a->A,
                                            a->Wc, a->Wd, a->v, a->lengths, dq_t, a->Apad, dwcat_out,
                                            a->de_hist + (int64_t)t * BT, wpack, sA));
        } else {
            GT_TRY(gt_attn_energies_bwd_w(d, a->dw, a->w_hist + (int64_t)(t + 1) * BT, a->q_hist + (int64_t)t * a->B * a->A,
                                          a->w_hist + (int64_t)t * BT, a->cum_hist + (int64_t)t * BT, a->pm, a->Wc, a->Wd, a->v,
                                          a->lengths, dq_t, a->Apad, a->dpm, dwcat_out, a->acc, wpack, sA));
        }
        if (bq) {
            // attention LSTM cell backward with dq . Wq computed inside: one dependent launch less per step
            GT_TRY(gt_cell_bwd_query(slabs(pX2, a->S2, sX2, a->N2pad),
                                     last ? kNoSlabs : slabs(a->partX1 + a->E, a->S1, sX1, a->N1pad), dq_t, a->Apad, a->Wq,
                                     a->Ha, a->dc1, a->gates1 + (int64_t)t * a->Bp * N1, a->c1 + (int64_t)(t + 1) * a->Bp * a->Ha,
                                     a->c1 + (int64_t)t * a->Bp * a->Ha, k1, a->p_att, dG1_t, a->dc1, a->B, a->Bp, a->Ha, a->A,
                                     sA));
        } else {
            GT_TRY(rec_rows(dq_t, a->Apad, a->WqTp, a->partQ, a->Bp, a->Ha, a->A, a->Sq, sA));
            // attention LSTM cell: dh = d(h_att as decoder-LSTM input) + d(query) + recurrent grad from t+1
            GT_TRY(gt_lstm_cell_bwd(slabs(pX2, a->S2, sX2, a->N2pad), slabs(a->partQ, a->Sq, sQ, a->Ha),
                                    last ? kNoSlabs : slabs(a->partX1 + a->E, a->S1, sX1, a->N1pad), nullptr, 0, a->dc1,
                                    a->gates1 + (int64_t)t * a->Bp * N1, a->c1 + (int64_t)(t + 1) * a->Bp * a->Ha,
                                    a->c1 + (int64_t)t * a->Bp * a->Ha, k1, a->p_att, nullptr, dG1_t, a->dc1, nullptr, a->B,
                                    a->Ha, sA));
        }
        if (two && t % k == 0) {          // ring slot t%R (and those of the steps above) may be rewritten by chain B
            hipEvent_t e2;
            GT_TRY(evA(t, &e2));
            GT_HIP(hipEventRecord(e2, sA));
        }
        if (t > 0) GT_TRY(rec_rows(dG1_t, N1, a->W1Tp, a->partX1, a->Bp, a->N1pad, N1, a->S1, sA, a->bf16 ? 1 : (a->rec_x3 ? 3 : 0), two ? kPrioA : 0));
        return 0;
    };
    int a_next = t_hi, last_wait_a = -1;
    for (int t = t_hi; t >= t_lo; --t) {
        const bool last = (t == a->To - 1);
        const uint8_t* k2 = a->dec_keep ? a->dec_keep + (int64_t)t * a->Bp * a->Hd : nullptr;
        const float* dhc_t = a->dHC + (int64_t)t * a->Bp * hc_ld;
        float* dG2_t = a->dG2 + (int64_t)t * a->Bp * N2;
        float* pX2 = ringX2(t);
        const float* pX2next = ringX2(t + 1);
        // ---- decoder LSTM chain (stream B, leads): dh = projection grad + recurrent grad from step t+1
        GT_TRY(gt_lstm_cell_bwd(last ? kNoSlabs : slabs(pX2next + a->Ha + a->E, a->S2, sX2, a->N2pad), kNoSlabs,
                                kNoSlabs, dhc_t, hc_ld, a->dc2, a->gates2 + (int64_t)t * a->Bp * N2,
                                a->c2 + (int64_t)(t + 1) * a->Bp * a->Hd, a->c2 + (int64_t)t * a->Bp * a->Hd, k2,
                                a->p_dec, nullptr, dG2_t, a->dc2, nullptr, a->B, a->Hd, sB));
        if (two && t + R <= a->To - 1) {   // ring slot t%R was last read by the attention chain of step t+R
            const int c = ((t + R) / k) * k;          // the recorded attention-chain step at or below t+R
            if (c != last_wait_a) {
                GT_TRY(evA(c, &ev));
                GT_HIP(hipStreamWaitEvent(sB, ev, 0));
                last_wait_a = c;
            }
        }
        GT_TRY(rec_rows(dG2_t, N2, a->W2Tp, pX2, a->Bp, a->N2pad, N2, a->S2, sB, a->bf16, two ? kPrioB : 0));
        if (!two) {
            GT_TRY(chain_a(t));
        } else if (t % k == 0 || t == t_lo) {
            GT_TRY(evB(t, &ev));
            GT_HIP(hipEventRecord(ev, sB));
            GT_HIP(hipStreamWaitEvent(sA, ev, 0));
            for (int s2 = a_next; s2 >= t; --s2) GT_TRY(chain_a(s2));
            a_next = t - 1;
        }
    }
    return 0;
}

int gt_bilstm_run(const GtBiLstm* a, void* stream) {
    GT_REQUIRE(a != nullptr, GT_E_ARG, "gt_bilstm_run: NULL");
    GT_REQUIRE(rows_ok(a->B, a->Bp) && a->H % 4 == 0 && a->T > 0, GT_E_LIMIT,
               "gt_bilstm_run: bad dims B=%d Bp=%d H=%d T=%d", a->B, a->Bp, a->H, a->T);
    const int mt = a->Bp <= 64 ? a->Bp / 16 : 4;
    const int nblk = a->Bp <= 64 ? 1 : a->Bp / 64;      // 64-row blocks (extra launches for big batches)
    const int N = 4 * a->H;
    const int64_t TB = (int64_t)a->T * a->Bp;
    if (!a->backward) {
        const int64_t pz = (int64_t)a->S * a->Bp * N;
        const bool fused = a->Whc != nullptr && a->counters != nullptr && a->H % 16 == 0;
        const int64_t xdir = (int64_t)a->Bp * a->ldx, xslot = 2 * xdir;
        for (int s = 0; fused && s < a->T; ++s) {
            const float* Xc = a->X + (s & 1) * xslot;
            float* Xn = a->X + ((s + 1) & 1) * xslot;
            const int64_t dz[9] = {xdir, a->w_zstride, pz, TB * N, (int64_t)(a->T + 1) * a->Bp * a->H, TB * N, TB * a->H, xdir, xdir};
            GT_TRY(gt_rec_cell_fwd_z(Xc, a->ldx, a->Whc, 0, a->part, a->counters, a->Bp, a->Bp, a->H, a->H, a->S,
                                     a->pre + (int64_t)s * a->Bp * N, N, a->c + (int64_t)s * a->Bp * a->H, nullptr, 0.f,
                                     a->gates + (int64_t)s * a->Bp * N, a->c + (int64_t)(s + 1) * a->Bp * a->H,
                                     a->hs + (int64_t)s * a->Bp * a->H, Xn, a->ldx, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0,
                                     0, 0, a->active + (int64_t)s * a->Bp, Xc, a->ldx, 2, dz, 0, stream));
        }
        for (int s = 0; !fused && s < a->T; ++s) {
            for (int rb = 0; rb < nblk; ++rb)
                GT_TRY(gt_rec_gemm(a->X + (int64_t)rb * 64 * a->ldx, a->ldx, a->Whp, a->part + (int64_t)rb * 64 * N, mt, N,
                                   a->H, a->S, (int64_t)a->Bp * N, 2, (int64_t)a->Bp * a->ldx, a->w_zstride, pz, stream));
            // both directions in one launch: blockIdx.y = direction, per-pointer element offsets
            const int64_t zf[7] = {pz, TB * N, (int64_t)(a->T + 1) * a->Bp * a->H, (int64_t)a->Bp * a->ldx, TB * N,
                                   TB * a->H, (int64_t)a->Bp * a->ldx};
            GT_TRY(gt_cell_fwd_z(slabs(a->part, a->S, (int64_t)a->Bp * N, N), a->pre + (int64_t)s * a->Bp * N, N,
                                 a->c + (int64_t)s * a->Bp * a->H, a->X, a->ldx, nullptr, 0.f,
                                 a->active + (int64_t)s * a->Bp, a->gates + (int64_t)s * a->Bp * N,
                                 a->c + (int64_t)(s + 1) * a->Bp * a->H, a->hs + (int64_t)s * a->Bp * a->H, a->X, a->ldx,
                                 nullptr, 0, nullptr, 0, a->Bp, a->H, 2, zf, stream));
        }
    } else {
        GT_REQUIRE(a->Hpad % 16 == 0 && a->Hpad >= a->H, GT_E_ARG, "gt_bilstm_run: Hpad");
        const int64_t pz = (int64_t)a->S * a->Bp * a->Hpad;
        for (int s = a->T - 1; s >= 0; --s) {
            const int64_t zb[6] = {pz, TB * a->H, (int64_t)a->Bp * a->H, TB * N, (int64_t)(a->T + 1) * a->Bp * a->H, TB * N};
            GT_TRY(gt_cell_bwd_z(s == a->T - 1 ? kNoSlabs : slabs(a->part, a->S, (int64_t)a->Bp * a->Hpad, a->Hpad),
                                 kNoSlabs, kNoSlabs, a->dhs + (int64_t)s * a->Bp * a->H, a->H, a->dc,
                                 a->gates + (int64_t)s * a->Bp * N, a->c + (int64_t)(s + 1) * a->Bp * a->H,
                                 a->c + (int64_t)s * a->Bp * a->H, nullptr, 0.f, a->active + (int64_t)s * a->Bp,
                                 a->dG + (int64_t)s * a->Bp * N, a->dc, nullptr, a->Bp, a->H, 2, zb, stream));
            if (s > 0)
                for (int rb = 0; rb < nblk; ++rb)
                    GT_TRY(gt_rec_gemm(a->dG + ((int64_t)s * a->Bp + rb * 64) * N, N, a->Whp,
                                       a->part + (int64_t)rb * 64 * a->Hpad, mt, a->Hpad, N, a->S,
                                       (int64_t)a->Bp * a->Hpad, 2, TB * N, a->w_zstride, pz, stream));
        }
    }
    return 0;
}

}  // extern "C"
