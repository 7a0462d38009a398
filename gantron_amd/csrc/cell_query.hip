// Attention-LSTM cell fused with the attention query projection (forward):
//   h_att = LSTMCell pointwise(sum of K-split slabs of the gate product + pre) with dropout  (model.py:391-394)
//   q partial = h_att[:, 32 units of this workgroup] . Wq[:, those units]^T                 (model.py:56, query_layer)
// One launch instead of gt_lstm_cell_fwd + gt_rec_gemm for the query: a dependent kernel less per decoder step.
// Workgroup (g, r) owns hidden units [32g, 32g+32) of batch rows [16r, 16r+16): 128 threads each take one float4
// of units of one row (all slab loads in flight together), write h to global and to an LDS tile, then the 2 waves
// multiply that [16 x 32] tile with the matching 32 columns of Wq (wave w: column blocks w, w+2, ...; fragments read
// straight from global, requested at kernel start) and store their part of query slab g.  Many small workgroups
// (128 for B=64) rather than few large ones: the kernel is latency bound and shares the CUs with the other chain.
// The energies kernel sums the Ha/32 slabs.
#include "common.h"

namespace {

constexpr int CQ_U = 32;            // hidden units per workgroup
constexpr int CQ_LD = CQ_U + 4;     // LDS row stride: conflict-free b128 fragment reads
constexpr int CQ_MAXCB = 16;        // attention_dim <= 256: 16 column blocks of 16, 8 per wave

struct CellQuery {
    GtSlabs part;
    const float* pre; long ld_pre;
    const float* c_prev;
    const uint8_t* keep; float scale;
    float* gates; float* c_out; float* h_raw;
    float* h0; long ld0; float* h1; long ld1;
    const float* Wq; long ldwq;       // (A, Ha) row-major
    float* qpart; long q_sstride;     // slab g at qpart + g*q_sstride, rows of Apad
    int B, H, A, Apad, mtiles;
};
static_assert(sizeof(CellQuery) <= 1024, "CellQuery is a by-value kernel argument: keep it well below the 4 KB kernel-argument segment");

__global__ __launch_bounds__(128) void cell_query_kernel(CellQuery a) {
    __builtin_amdgcn_s_setprio(3);      // the attention chain is the serial critical path of the decoder loops: its waves
                                        // issue ahead of the co-resident waves of the other chain and of the leaf GEMMs
    __shared__ __attribute__((aligned(16))) float Xh[16 * CQ_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;          // 2 waves
    const int g = blockIdx.x, row0 = blockIdx.y * 16;
    const int H = a.H;
    // ---- query weight fragments of this wave's column blocks (cb = wave, wave + 2, ...), requested first.
    // B operand of the 16x16x4 MFMA: lane supplies W[n = lane & 15][k = 4 * (lane >> 4) + j]
    const int ncb = a.Apad / 16;
    float4 wq[CQ_MAXCB / 2][2];      // [column block slot][k-block]
#pragma unroll
    for (int sl = 0; sl < CQ_MAXCB / 2; ++sl) {
        const int cb = wave + 2 * sl;
        const int ch = min(cb * 16 + (lane & 15), a.A - 1);           // clamped: rows >= A are masked below
        const float* wp = a.Wq + (long)ch * a.ldwq + g * CQ_U + 4 * (lane >> 4);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float4 v = *reinterpret_cast<const float4*>(wp + 16 * kb);
            if (cb >= ncb || cb * 16 + (lane & 15) >= a.A) v = make_float4(0.f, 0.f, 0.f, 0.f);
            wq[sl][kb] = v;
        }
    }
    // ---- cell: thread = (row, float4 of units)
    const int rl = tid >> 3, ug = tid & 7;
    const int m = row0 + rl;
    const int u = g * CQ_U + 4 * ug;
    float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < a.B) {
        float4 gi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            gi[q] = a.pre != nullptr ? *reinterpret_cast<const float4*>(a.pre + (long)m * a.ld_pre + q * H + u)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            const float* p = a.part.p + (long)m * a.part.ld + q * H + u;
#pragma unroll 4
            for (int s = 0; s < a.part.n; ++s) {
                const float4 t = *reinterpret_cast<const float4*>(p + (long)s * a.part.slab_stride);
                gi[q].x += t.x; gi[q].y += t.y; gi[q].z += t.z; gi[q].w += t.w;
            }
        }
        const long idx = (long)m * H + u;
        const float4 cp = *reinterpret_cast<const float4*>(a.c_prev + idx);
        uchar4 kp = make_uchar4(1, 1, 1, 1);
        if (a.keep != nullptr) kp = *reinterpret_cast<const uchar4*>(a.keep + idx);
        float4 gv[4], cn;
        const float* gif = reinterpret_cast<const float*>(&gi[0]);
        float* gvf = reinterpret_cast<float*>(&gv[0]);
        const float* cpf = reinterpret_cast<const float*>(&cp);
        float* cnf = reinterpret_cast<float*>(&cn);
        float* hvf = reinterpret_cast<float*>(&hv);
        const unsigned char* kpf = reinterpret_cast<const unsigned char*>(&kp);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float i = sigmoidf_(gif[0 * 4 + e]), f = sigmoidf_(gif[1 * 4 + e]);
            const float gg = tanhf_(gif[2 * 4 + e]), o = sigmoidf_(gif[3 * 4 + e]);
            gvf[0 * 4 + e] = i; gvf[1 * 4 + e] = f; gvf[2 * 4 + e] = gg; gvf[3 * 4 + e] = o;
            const float c = f * cpf[e] + i * gg;
            cnf[e] = c;
            float h = o * tanhf_(c);
            if (a.keep != nullptr) h *= kpf[e] ? a.scale : 0.f;
            hvf[e] = h;
        }
        if (a.gates != nullptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(a.gates + (long)m * 4 * H + q * H + u) = gv[q];
        }
        *reinterpret_cast<float4*>(a.c_out + idx) = cn;
        if (a.h_raw != nullptr) *reinterpret_cast<float4*>(a.h_raw + idx) = hv;
        if (a.h0 != nullptr) *reinterpret_cast<float4*>(a.h0 + (long)m * a.ld0 + u) = hv;
        if (a.h1 != nullptr) *reinterpret_cast<float4*>(a.h1 + (long)m * a.ld1 + u) = hv;
    }
    *reinterpret_cast<float4*>(&Xh[rl * CQ_LD + 4 * ug]) = hv;         // rows beyond the batch contribute zeros
    __syncthreads();
    // ---- query partial: [16 x 32] . [32 x Apad]
    float* out = a.qpart + (long)g * a.q_sstride;
    float4 x[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) x[kb] = *reinterpret_cast<const float4*>(&Xh[(lane & 15) * CQ_LD + 16 * kb + 4 * (lane >> 4)]);
#pragma unroll
    for (int sl = 0; sl < CQ_MAXCB / 2; ++sl) {
        const int cb = wave + 2 * sl;
        if (cb >= ncb) break;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float4 b = wq[sl][kb];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[kb].x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[kb].y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[kb].z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x[kb].w, b.w, acc, 0, 0, 0);
        }
        // C layout: column n = lane & 15, rows 4*(lane>>4) + r
#pragma unroll
        for (int r = 0; r < 4; ++r)
            out[(long)(row0 + 4 * (lane >> 4) + r) * a.Apad + cb * 16 + (lane & 15)] = acc[r];
    }
}

}  // namespace

extern "C" int gt_cell_query_fwd(GtSlabs part, const float* pre, int64_t ld_pre, const float* c_prev, const uint8_t* keep,
                                 float p_drop, float* gates, float* c_out, float* h_raw, float* h0, int64_t ld0, float* h1,
                                 int64_t ld1, const float* Wq, int64_t ldwq, float* qpart, int64_t q_sstride, int32_t B,
                                 int32_t Bp, int32_t H, int32_t A, int32_t Apad, void* stream) {
    GT_REQUIRE(part.p && c_prev && c_out && Wq && qpart && part.n >= 1, GT_E_ARG, "gt_cell_query_fwd: NULL pointer");
    GT_REQUIRE(B > 0 && Bp >= B && ((Bp <= 64 && Bp % 16 == 0) || Bp % 64 == 0) && H > 0 && H % CQ_U == 0, GT_E_ARG,
               "gt_cell_query_fwd: bad sizes B=%d Bp=%d H=%d (H must be a multiple of %d)", B, Bp, H, CQ_U);
    GT_REQUIRE(A > 0 && Apad >= A && Apad % 16 == 0 && Apad <= 256 && q_sstride >= (int64_t)Bp * Apad, GT_E_LIMIT,
               "gt_cell_query_fwd: attention_dim %d (padded %d) must fit 16 column blocks", A, Apad);
    GT_REQUIRE(p_drop >= 0.f && p_drop < 1.f, GT_E_ARG, "gt_cell_query_fwd: p_drop");
    auto ok4 = [](const void* p, int64_t ld) { return p == nullptr || (aligned16(p) && ld % 4 == 0); };
    GT_REQUIRE(ok4(part.p, part.ld) && part.slab_stride % 4 == 0 && ok4(pre, ld_pre) && aligned16(c_prev) &&
                   (keep == nullptr || (((uintptr_t)keep) & 3) == 0) && ok4(gates, 0) && aligned16(c_out) && ok4(h_raw, 0) &&
                   ok4(h0, ld0) && ok4(h1, ld1) && ok4(Wq, ldwq),
               GT_E_ALIGN, "gt_cell_query_fwd: every pointer must be 16-byte aligned with leading dimensions %% 4 == 0");
    CellQuery a;
    a.part = part; a.pre = pre; a.ld_pre = ld_pre; a.c_prev = c_prev; a.keep = keep; a.scale = 1.0f / (1.0f - p_drop);
    a.gates = gates; a.c_out = c_out; a.h_raw = h_raw; a.h0 = h0; a.ld0 = ld0; a.h1 = h1; a.ld1 = ld1;
    a.Wq = Wq; a.ldwq = ldwq; a.qpart = qpart; a.q_sstride = q_sstride; a.B = B; a.H = H; a.A = A; a.Apad = Apad;
    a.mtiles = 1;
    dim3 grid(H / CQ_U, Bp / 16);
    const long n = (long)B * H;
    GtProfScope prof(GT_FAM_CELL_FWD, 2.0 * Bp * (double)H * A, 4.0 * n * (4.0 * part.n + 4 + 1 + 4 + 1 + 3), (hipStream_t)stream);
    hipLaunchKernelGGL(cell_query_kernel, grid, dim3(128), 0, (hipStream_t)stream, a);
    GT_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Backward counterpart: the attention-LSTM cell backward with the query-projection input gradient
// dh_q = dq . Wq computed inside (K = attention_dim), instead of gt_rec_gemm (slabs) + gt_lstm_cell_bwd.
// Workgroup (g, r) = one wave owns hidden units [16g, 16g+16) of rows [16r, 16r+16): it computes that 16 x 16 tile
// of dq . Wq on the matrix cores, parks it in LDS, then its 64 threads (row, float4 of units) add it to the slab
// sources and run the cell backward (256 one-wave workgroups for B=64, H=1024, like gt_lstm_cell_bwd).
namespace {

constexpr int CB_U = 16;
constexpr int CB_LD = CB_U + 4;

struct CellBwdQuery {
    GtSlabs da, dc;                   // d h_att from the decoder-LSTM product and from the next step's product
    const float* dq; long ld_dq;      // (Bp, ld_dq) gradient of the query, columns [0, A)
    const float* dq_sum; int n_sum;   // optional (gt_cell_bwd_query_c): the query gradient is dq - s dq2 with s[m] the sum of
    const float* dq2;                 // row m's n_sum entries of dq_sum; dq2 laid out like dq
    const float* Wq; long ldwq;       // (A, Ha)
    const float* dc_next;
    const float* gates; const float* c_new; const float* c_prev;
    const uint8_t* keep; float scale;
    float* dgates; float* dc_prev;
    int B, H, A, mtiles;
};
static_assert(sizeof(CellBwdQuery) <= 1024, "CellBwdQuery is a by-value kernel argument: keep it well below the 4 KB kernel-argument segment");

__global__ __launch_bounds__(64) void cell_bwd_query_kernel(CellBwdQuery a) {
    __builtin_amdgcn_s_setprio(3);      // the attention chain is the serial critical path of the decoder loops: its waves
                                        // issue ahead of the co-resident waves of the other chain and of the leaf GEMMs
    __shared__ __attribute__((aligned(16))) float T[16 * CB_LD];
    const int tid = threadIdx.x, lane = tid;
    const int g = blockIdx.x, row0 = blockIdx.y * 16;
    const int H = a.H;
    const int u0 = g * CB_U;
    // ---- dh_q tile: C[m][n] = sum_a dq[m][a] * Wq[a][u0 + n]
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* xq = a.dq + (long)(row0 + (lane & 15)) * a.ld_dq + 4 * (lane >> 4);
        const float* xq2 = a.dq2 != nullptr ? a.dq2 + (long)(row0 + (lane & 15)) * a.ld_dq + 4 * (lane >> 4) : xq;
        float sm = 0.f;
        if (a.dq2 != nullptr) {
            // the row's <= 16 tile sums, all requested at once (clamped indices, masked afterwards)
            const int rowc = min(row0 + (lane & 15), a.B - 1);
            float sp[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) sp[c] = a.dq_sum[(long)rowc * a.n_sum + min(c, a.n_sum - 1)];
#pragma unroll
            for (int c = 0; c < 16; ++c) sm += c < a.n_sum ? sp[c] : 0.f;
            if (row0 + (lane & 15) >= a.B) sm = 0.f;
        }
        const float* wc = a.Wq + u0 + (lane & 15);
        const int nkb = (a.A + 15) / 16;
#pragma unroll 4
        for (int kb = 0; kb < nkb; ++kb) {
            // A operand: dq[m = lane & 15][k = 16 kb + 4 (lane >> 4) + j]; columns >= A of dq are zero (padded buffer)
            float4 x = *reinterpret_cast<const float4*>(xq + 16 * kb);
            if (a.dq2 != nullptr) {
                const float4 x2 = *reinterpret_cast<const float4*>(xq2 + 16 * kb);
                x.x -= sm * x2.x; x.y -= sm * x2.y; x.z -= sm * x2.z; x.w -= sm * x2.w;
            }
            const int k0 = 16 * kb + 4 * (lane >> 4);
            // B operand: Wq[k][n = lane & 15], rows clamped (the matching dq columns are zero)
            const float b0 = wc[(long)min(k0 + 0, a.A - 1) * a.ldwq];
            const float b1 = wc[(long)min(k0 + 1, a.A - 1) * a.ldwq];
            const float b2 = wc[(long)min(k0 + 2, a.A - 1) * a.ldwq];
            const float b3 = wc[(long)min(k0 + 3, a.A - 1) * a.ldwq];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, b3, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(4 * (lane >> 4) + r) * CB_LD + (lane & 15)] = acc[r];
    }
    // ---- cell backward inputs of this thread (requested before the barrier)
    const int rl = tid >> 2, ug = tid & 3;
    const int m = row0 + rl;
    const int u = u0 + 4 * ug;
    const bool live = m < a.B;
    const long idx = (long)m * H + u;
    float4 dh = make_float4(0.f, 0.f, 0.f, 0.f), dcn = dh, gi = dh, gf = dh, gg = dh, go = dh, cn = dh, cp = dh;
    uchar4 kp = make_uchar4(1, 1, 1, 1);
    if (live) {
        auto add_slabs = [&](const GtSlabs& sl) __attribute__((always_inline)) {
            if (sl.p == nullptr) return;
            const float* p = sl.p + (long)m * sl.ld + u;
#pragma unroll 4
            for (int s = 0; s < sl.n; ++s) {
                const float4 t = *reinterpret_cast<const float4*>(p + (long)s * sl.slab_stride);
                dh.x += t.x; dh.y += t.y; dh.z += t.z; dh.w += t.w;
            }
        };
        add_slabs(a.da);
        add_slabs(a.dc);
        if (a.dc_next != nullptr) dcn = *reinterpret_cast<const float4*>(a.dc_next + idx);
        const float* gt = a.gates + (long)m * 4 * H + u;
        gi = *reinterpret_cast<const float4*>(gt); gf = *reinterpret_cast<const float4*>(gt + H);
        gg = *reinterpret_cast<const float4*>(gt + 2 * H); go = *reinterpret_cast<const float4*>(gt + 3 * H);
        cn = *reinterpret_cast<const float4*>(a.c_new + idx);
        cp = *reinterpret_cast<const float4*>(a.c_prev + idx);
        if (a.keep != nullptr) kp = *reinterpret_cast<const uchar4*>(a.keep + idx);
    }
    __syncthreads();
    if (!live) return;
    const float4 dhq = *reinterpret_cast<const float4*>(&T[rl * CB_LD + 4 * ug]);
    const float dhv[4] = {dh.x + dhq.x, dh.y + dhq.y, dh.z + dhq.z, dh.w + dhq.w};
    const float dcv[4] = {dcn.x, dcn.y, dcn.z, dcn.w};
    const float giv[4] = {gi.x, gi.y, gi.z, gi.w}, gfv[4] = {gf.x, gf.y, gf.z, gf.w};
    const float ggv[4] = {gg.x, gg.y, gg.z, gg.w}, gov[4] = {go.x, go.y, go.z, go.w};
    const float cnv[4] = {cn.x, cn.y, cn.z, cn.w}, cpv[4] = {cp.x, cp.y, cp.z, cp.w};
    const unsigned char kpv[4] = {kp.x, kp.y, kp.z, kp.w};
    float dgi[4], dgf[4], dgg[4], dgo[4], dcp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float d = dhv[e];
        if (a.keep != nullptr) d *= (kpv[e] ? a.scale : 0.f);
        const float tc = tanhf_(cnv[e]);
        const float dcell = dcv[e] + d * gov[e] * (1.f - tc * tc);
        dgi[e] = dcell * ggv[e] * giv[e] * (1.f - giv[e]);
        dgf[e] = dcell * cpv[e] * gfv[e] * (1.f - gfv[e]);
        dgg[e] = dcell * giv[e] * (1.f - ggv[e] * ggv[e]);
        dgo[e] = d * tc * gov[e] * (1.f - gov[e]);
        dcp[e] = dcell * gfv[e];
    }
    float* dg = a.dgates + (long)m * 4 * H + u;
    *reinterpret_cast<float4*>(dg) = make_float4(dgi[0], dgi[1], dgi[2], dgi[3]);
    *reinterpret_cast<float4*>(dg + H) = make_float4(dgf[0], dgf[1], dgf[2], dgf[3]);
    *reinterpret_cast<float4*>(dg + 2 * H) = make_float4(dgg[0], dgg[1], dgg[2], dgg[3]);
    *reinterpret_cast<float4*>(dg + 3 * H) = make_float4(dgo[0], dgo[1], dgo[2], dgo[3]);
    *reinterpret_cast<float4*>(a.dc_prev + idx) = make_float4(dcp[0], dcp[1], dcp[2], dcp[3]);
}

}  // namespace

extern "C" int gt_cell_bwd_query(GtSlabs dh_a, GtSlabs dh_c, const float* dq, int64_t ld_dq, const float* Wq, int64_t ldwq,
                                 const float* dc_next, const float* gates, const float* c_new, const float* c_prev,
                                 const uint8_t* keep, float p_drop, float* dgates, float* dc_prev, int32_t B, int32_t Bp,
                                 int32_t H, int32_t A, void* stream) {
    return gt_cell_bwd_query_c(dh_a, dh_c, dq, ld_dq, nullptr, 0, nullptr, Wq, ldwq, dc_next, gates, c_new, c_prev, keep, p_drop,
                               dgates, dc_prev, B, Bp, H, A, stream);
}

extern "C" int gt_cell_bwd_query_c(GtSlabs dh_a, GtSlabs dh_c, const float* dq, int64_t ld_dq, const float* dq_sum,
                                   int32_t n_sum, const float* dq2, const float* Wq, int64_t ldwq, const float* dc_next,
                                   const float* gates, const float* c_new, const float* c_prev, const uint8_t* keep,
                                   float p_drop, float* dgates, float* dc_prev, int32_t B, int32_t Bp, int32_t H, int32_t A,
                                   void* stream) {
    GT_REQUIRE((dq2 == nullptr) == (dq_sum == nullptr) && (dq2 == nullptr || (n_sum > 0 && n_sum <= 16 && aligned16(dq2))), GT_E_ARG,
               "gt_cell_bwd_query_c: dq_sum (B x n_sum) and dq2 go together");
    GT_REQUIRE(dq && Wq && gates && c_new && c_prev && dgates && dc_prev, GT_E_ARG, "gt_cell_bwd_query: NULL pointer");
    GT_REQUIRE(B > 0 && Bp >= B && ((Bp <= 64 && Bp % 16 == 0) || Bp % 64 == 0) && H > 0 && H % CB_U == 0 && A > 0, GT_E_ARG,
               "gt_cell_bwd_query: bad sizes B=%d Bp=%d H=%d A=%d (H must be a multiple of %d)", B, Bp, H, A, CB_U);
    GT_REQUIRE(ld_dq >= (int64_t)((A + 15) / 16) * 16, GT_E_ARG, "gt_cell_bwd_query: dq needs ld >= A rounded up to 16 (zero padded)");
    auto sl4 = [](const GtSlabs& sl) { return sl.p == nullptr || (aligned16(sl.p) && sl.ld % 4 == 0 && sl.slab_stride % 4 == 0); };
    GT_REQUIRE(sl4(dh_a) && sl4(dh_c) && aligned16(dq) && ld_dq % 4 == 0 && (dc_next == nullptr || aligned16(dc_next)) &&
                   aligned16(gates) && aligned16(c_new) && aligned16(c_prev) && (keep == nullptr || (((uintptr_t)keep) & 3) == 0) &&
                   aligned16(dgates) && aligned16(dc_prev) && H % 4 == 0,
               GT_E_ALIGN, "gt_cell_bwd_query: every pointer must be 16-byte aligned with leading dimensions %% 4 == 0");
    CellBwdQuery a;
    a.dq_sum = dq_sum; a.n_sum = n_sum; a.dq2 = dq2;
    a.da = dh_a; a.dc = dh_c; a.dq = dq; a.ld_dq = ld_dq; a.Wq = Wq; a.ldwq = ldwq; a.dc_next = dc_next; a.gates = gates;
    a.c_new = c_new; a.c_prev = c_prev; a.keep = keep; a.scale = 1.0f / (1.0f - p_drop); a.dgates = dgates; a.dc_prev = dc_prev;
    a.B = B; a.H = H; a.A = A; a.mtiles = 1;
    dim3 grid(H / CB_U, Bp / 16);
    const long n = (long)B * H;
    GtProfScope prof(GT_FAM_CELL_BWD, 2.0 * Bp * (double)H * A, 4.0 * n * (dh_a.n + dh_c.n + 1 + 4 + 2 + 4 + 1), (hipStream_t)stream);
    hipLaunchKernelGGL(cell_bwd_query_kernel, grid, dim3(64), 0, (hipStream_t)stream, a);
    GT_LAUNCH_CHECK();
    return 0;
}
