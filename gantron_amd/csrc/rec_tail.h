// Recurrent-product work that rides in ANOTHER launch (GtRecTail, include/gantron_hip.h).
//
// A decoder step's attention chain is LSTM step -> attention step -> LSTM step ..., and half of what an LSTM step
// multiplies is available one launch EARLIER than the launch that consumes it: forward, the h_att half of
// [ctx_{t} | h_att_{t}] . W1 exists as soon as LSTM step t has ended, while ctx_t only comes out of attention step t;
// backward, the d h_att columns of dG1_t . W1^T are not read before the cell backward of step t - 1, which follows the
// attention backward of step t - 1.  So that half leaves the critical launch: it is cut into (64-column group, K-slice)
// items, and the workgroups of the launch in between -- the attention step -- pick the items up from a work counter
// when they have nothing (more) of their own to do: the (row, chunk) workgroups whose chunk lies in the row's padding at
// once, the others when they retire.  Every workgroup drains the counter before it exits, so all items are done when the
// launch ends, whatever the batch looks like; the consumer is the NEXT launch on the stream (plain stores, plain loads).
// The critical launch then multiplies half the K (forward) / half the columns (backward).
//
// One item is rec_gemm_tile's product (recurrent.hip) for 256-thread workgroups: wave w owns 16 of the item's 64 columns
// and every k-block of a chunk; activation chunks (rows x 64 k) are staged through three LDS buffers (one barrier per
// chunk), activations and the wave's weight fragments travel through registers two chunks ahead, v_mfma_f32_16x16x4_f32.
#pragma once
#include "common.h"

namespace {

constexpr int RT_CK = 4;                          // k-blocks (of 16) per LDS chunk
constexpr int RT_LD = RT_CK * 16 + 4;             // padded LDS row: conflict-free ds_read_b128 / ds_write_b128
constexpr int RT_EP_LD = 68;                      // epilogue tile row stride
constexpr int RT_THREADS = 256;
constexpr int RT_LDS_FLOATS = 3 * 64 * RT_LD + 4; // three staging buffers of 64 rows + the dequeued item (52.2 KB)

__device__ __forceinline__ int rec_tail_items(const GtRecTail& q) {
    return q.work != nullptr ? q.groups * cdiv(q.kb1 - q.kb0, q.kbs) : 0;
}

template <int MT>
__device__ __forceinline__ void rec_tail_item(const GtRecTail& q, int cg, int kb0, int kb1, float* __restrict__ out, float* lds) {
    constexpr int SLOT = 64 * RT_LD;
    constexpr int NV = MT * 16 * RT_CK * 4 / RT_THREADS;      // float4 of an activation chunk per thread (MT: 1..4)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = cg * 4 + wave;                             // this wave's 16-column block (N % 64 == 0: host check)
    const f32x4* wp = reinterpret_cast<const f32x4*>(q.Wp) + ((long)nb * q.KB + kb0) * 64 + lane;
    const float* X = q.X;
    const long ldx = q.ldx;

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A workgroup of the tail shares its CU with a workgroup of the step at higher issue priority and has one wave per SIMD:
    // what bounds it is the memory round trip, not bytes or MFMAs (one chunk ahead: 2.5 us per chunk of 4 k-blocks in the loop).
    // So the operands of TWO chunks are in flight while a third is multiplied: three register sets, bound statically.
    // (plain local arrays indexed by unrolled constants and macros instead of structs passed to lambdas: with the latter the
    // compiler left one set in scratch and moved another into LDS -- 80 KB per workgroup, one workgroup per CU.)
    // (native vectors, not HIP's float4 struct: its assignment is a memcpy that kept the arrays in memory)
    f32x4 x0[NV > 0 ? NV : 1], x1[NV > 0 ? NV : 1], x2[NV > 0 ? NV : 1];
    f32x4 w0[RT_CK], w1[RT_CK], w2[RT_CK];
    // Loads are UNCONDITIONAL, also those of k-blocks beyond the slice (addresses clamped to its last block; what they fetch
    // is never multiplied: the MFMAs of those k-blocks are skipped).  A branch around a load -- the compiler turns a select
    // on a uniform condition behind a load back into one -- makes it give up counting outstanding loads and wait for all
    // of them before every use, which would undo the ring.
#define RT_LOAD(XR, WR, KB_)                                                                                                  \
    do {                                                                                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < NV; ++i_) {                                                                   \
            const int f_ = tid + RT_THREADS * i_;                                                                             \
            const int row_ = f_ / (RT_CK * 4), c4_ = f_ % (RT_CK * 4);                                                        \
            XR[i_] = *reinterpret_cast<const f32x4*>(X + (long)row_ * ldx + (long)min((KB_) + (c4_ >> 2), kb1 - 1) * 16 +     \
                                                     4 * (c4_ & 3));                                                          \
        }                                                                                                                     \
        _Pragma("unroll") for (int u_ = 0; u_ < RT_CK; ++u_) WR[u_] = wp[(long)(min((KB_) + u_, kb1 - 1) - kb0) * 64];        \
    } while (0)
    // one chunk: publish its activations, request the chunk two ahead into the set the previous chunk has just freed, multiply
#define RT_CHUNK(XC, WC, XF, WF, KB_, SLOT_)                                                                                  \
    do {                                                                                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < NV; ++i_) {                                                                   \
            const int f_ = tid + RT_THREADS * i_;                                                                             \
            const int row_ = f_ / (RT_CK * 4), c4_ = f_ % (RT_CK * 4);                                                        \
            *reinterpret_cast<f32x4*>(&lds[(SLOT_) * SLOT + row_ * RT_LD + 4 * c4_]) = XC[i_];                                \
        }                                                                                                                     \
        __syncthreads();      /* also: every wave is done with the chunk three back, which lived in this slot */              \
        RT_LOAD(XF, WF, (KB_) + 2 * RT_CK);                                                                                   \
        const float* xs_ = lds + (SLOT_) * SLOT + (lane & 15) * RT_LD + 4 * (lane >> 4);                                      \
        _Pragma("unroll") for (int u_ = 0; u_ < RT_CK; ++u_) {                                                                \
            if ((KB_) + u_ >= kb1) break;     /* (uniform) k-blocks beyond the slice */                                       \
            f32x4 a_[MT];                                                                                                     \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) a_[mt] = *reinterpret_cast<const f32x4*>(xs_ + mt * 16 * RT_LD + u_ * 16); \
            const f32x4 b_ = WC[u_];                                                                                          \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[mt].x, b_.x, acc[mt], 0, 0, 0); \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[mt].y, b_.y, acc[mt], 0, 0, 0); \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[mt].z, b_.z, acc[mt], 0, 0, 0); \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_[mt].w, b_.w, acc[mt], 0, 0, 0); \
        }                                                                                                                     \
    } while (0)
    RT_LOAD(x0, w0, kb0);
    RT_LOAD(x1, w1, kb0 + RT_CK);
    // three chunks per trip, each with its register set and its LDS slot; a slice that is not a multiple of three chunks ends
    // with chunks of nothing (stores and barriers, no MFMAs): slices of 12 k-blocks (or 11, ...) waste nothing
    for (int kb = kb0; kb < kb1; kb += 3 * RT_CK) {
        RT_CHUNK(x0, w0, x2, w2, kb, 0);
        RT_CHUNK(x1, w1, x0, w0, kb + RT_CK, 1);
        RT_CHUNK(x2, w2, x1, w1, kb + 2 * RT_CK, 2);
    }
#undef RT_CHUNK
#undef RT_LOAD
    __syncthreads();                      // the staging buffers are free: the tile goes through the first one
    // MFMA C layout (column lane & 15, rows 4 (lane >> 4) + r) -> row-major tile -> full-line float4 stores
    const int ecol = wave * 16 + (lane & 15), erow = 4 * (lane >> 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[(mt * 16 + erow + r) * RT_EP_LD + ecol] = acc[mt][r];
    __syncthreads();
    float* o = out + (long)cg * 64;
#pragma unroll
    for (int i = 0; i < MT * 16 * 16 / RT_THREADS; ++i) {
        const int f = tid + RT_THREADS * i;
        const int row = f >> 4, c4 = f & 15;
        *reinterpret_cast<float4*>(o + (long)row * q.N + 4 * c4) = *reinterpret_cast<const float4*>(&lds[row * RT_EP_LD + 4 * c4]);
    }
}

// Called by EVERY thread of a 256-thread workgroup that is done with its own work (it contains barriers); `lds` has
// RT_LDS_FLOATS floats that nothing else uses any more.  Returns when the launch's work counter is exhausted.
// stamps (diagnostics, or NULL): [1] first item drawn, [2] counter ran dry, [3] items computed here (wall clock, 10 ns)
__device__ __forceinline__ void rec_tail_run(const GtRecTail& q, float* lds, long long* stamps = nullptr) {
    const int n_items = rec_tail_items(q);
    if (n_items <= 0) return;
    unsigned* slot = reinterpret_cast<unsigned*>(lds + RT_LDS_FLOATS - 4);
    int done = 0;
    for (;;) {
        __syncthreads();                  // the previous item's tile has left LDS, its index has been read by everyone
        if (threadIdx.x == 0) *slot = __hip_atomic_fetch_add(q.work, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned it = *slot;
        if (stamps != nullptr && threadIdx.x == 0) {
            if (done == 0) stamps[1] = (long long)__builtin_amdgcn_s_memrealtime();
            if (it >= (unsigned)n_items) { stamps[2] = (long long)__builtin_amdgcn_s_memrealtime(); stamps[3] = done; }
        }
        if (it >= (unsigned)n_items) return;
        ++done;
        // K-slices of a column group are dealt to different workgroups one after the other: item = slice * groups + group
        const int s = (int)it / q.groups, cg = q.cg0 + (int)it % q.groups;
        const int kb0 = q.kb0 + s * q.kbs, kb1 = min(q.kb1, kb0 + q.kbs);
        float* out = q.out + (long)s * q.out_sstride;
        switch (q.mtiles) {
            case 1: rec_tail_item<1>(q, cg, kb0, kb1, out, lds); break;
            case 2: rec_tail_item<2>(q, cg, kb0, kb1, out, lds); break;
            case 3: rec_tail_item<3>(q, cg, kb0, kb1, out, lds); break;
            default: rec_tail_item<4>(q, cg, kb0, kb1, out, lds); break;
        }
    }
}

}  // namespace

// host-side check shared by the launchers that carry a tail
static inline int gt_rec_tail_check(const GtRecTail* q, const char* who) {
    if (q == nullptr || q->work == nullptr) return 0;
    GT_REQUIRE(q->X && q->Wp && q->out, GT_E_ARG, "%s: tail with a NULL pointer", who);
    GT_REQUIRE(q->mtiles >= 1 && q->mtiles <= 4 && q->N > 0 && q->N % 64 == 0 && q->KB > 0 && q->kb0 >= 0 && q->kb0 < q->kb1 &&
                   q->kb1 <= q->KB && q->kbs >= 1 && q->cg0 >= 0 && q->groups >= 1 && (q->cg0 + q->groups) * 64 <= q->N,
               GT_E_ARG, "%s: tail sizes N=%d KB=%d k-blocks [%d, %d) by %d, groups [%d, +%d), %d row tiles", who, q->N, q->KB, q->kb0,
               q->kb1, q->kbs, q->cg0, q->groups, q->mtiles);
    GT_REQUIRE(q->ldx >= (int64_t)q->kb1 * 16 && q->ldx % 4 == 0 && aligned16(q->X) && aligned16(q->Wp) && aligned16(q->out) &&
                   q->out_sstride % 4 == 0 && q->out_sstride >= (int64_t)16 * q->mtiles * q->N,
               GT_E_ALIGN, "%s: tail operands must be 16-byte aligned, ldx >= 16 * kb1, slab stride >= rows * N", who);
    return 0;
}
