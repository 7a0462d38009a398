// Attention-LSTM cell fused with the attention query projection (forward):
//   h_att = LSTMCell pointwise(sum of K-split slabs of the gate product + pre) with dropout  (model.py:391-394)
//   q partial = h_att[:, 32 units of this workgroup] . Wq[:, those units]^T                 (model.py:56, query_layer)
// One launch instead of gt_lstm_cell_fwd + gt_rec_gemm for the query: a dependent kernel less per decoder step.
// Workgroup g owns hidden units [32g, 32g+32) of all (<= 64) batch rows of its row block: 512 threads each take
// one float4 of units of one row (all slab loads in flight together), write h to global and to an LDS tile, then
// the 8 waves multiply that [64 x 32] tile with the matching 32 columns of Wq (wave w: attention channels
// [16w, 16w+16), fragments read straight from global, requested at kernel start) and store query slab g.
// The energies kernel sums the Ha/32 slabs.
#include "common.h"

namespace {

constexpr int CQ_U = 32;            // hidden units per workgroup
constexpr int CQ_LD = CQ_U + 4;     // LDS row stride: conflict-free b128 fragment reads

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

__global__ __launch_bounds__(512) void cell_query_kernel(CellQuery a) {
    __shared__ __attribute__((aligned(16))) float Xh[64 * CQ_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, row0 = blockIdx.y * 64;
    const int H = a.H;
    // ---- query weight fragments of this wave's column blocks, requested first (they only depend on g)
    // B operand of the 16x16x4 MFMA: lane supplies W[n = lane & 15][k = 4 * (lane >> 4) + j]
    const int ncb = a.Apad / 16;
    float4 wq[2][2];                 // [column block slot][k-block]
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int cb = wave + 8 * sl;
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
    if (rl < a.mtiles * 16 && m < a.B) {
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
            const float gg = tanhf(gif[2 * 4 + e]), o = sigmoidf_(gif[3 * 4 + e]);
            gvf[0 * 4 + e] = i; gvf[1 * 4 + e] = f; gvf[2 * 4 + e] = gg; gvf[3 * 4 + e] = o;
            const float c = f * cpf[e] + i * gg;
            cnf[e] = c;
            float h = o * tanhf(c);
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
    // ---- query partial: [16*mtiles x 32] . [32 x Apad]
    float* out = a.qpart + (long)g * a.q_sstride;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int cb = wave + 8 * sl;
        if (cb >= ncb) break;
        for (int mt = 0; mt < a.mtiles; ++mt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const float4 x = *reinterpret_cast<const float4*>(&Xh[(mt * 16 + (lane & 15)) * CQ_LD + 16 * kb + 4 * (lane >> 4)]);
                const float4 b = wq[sl][kb];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, b.w, acc, 0, 0, 0);
            }
            // C layout: column n = lane & 15, rows 4*(lane>>4) + r
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(long)(row0 + mt * 16 + 4 * (lane >> 4) + r) * a.Apad + cb * 16 + (lane & 15)] = acc[r];
        }
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
    a.mtiles = Bp <= 64 ? Bp / 16 : 4;
    dim3 grid(H / CQ_U, Bp <= 64 ? 1 : Bp / 64);
    const long n = (long)B * H;
    GtProfScope prof(GT_FAM_CELL_FWD, 2.0 * Bp * (double)H * A, 4.0 * n * (4.0 * part.n + 4 + 1 + 4 + 1 + 3), (hipStream_t)stream);
    hipLaunchKernelGGL(cell_query_kernel, grid, dim3(512), 0, (hipStream_t)stream, a);
    GT_LAUNCH_CHECK();
    return 0;
}
