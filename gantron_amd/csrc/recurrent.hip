// Per-time-step kernels of the LSTM recurrences (decoder attention/decoder LSTMs, encoder BiLSTM).
//
// The recurrent product has M = batch <= 64 rows against N = 4H gate columns: far too few rows
// for a tiled GEMM, and the weights (84 MiB fp32 for the two decoder cells) are re-read every step.
// gt_rec_gemm therefore runs as a weight-streaming kernel: every wave owns 16 gate columns, streams
// its pre-packed weight block (1 KiB per 16x16 k-block, lane-linear float4) straight into VGPRs,
// reads the small activation matrix (L2 resident) as MFMA A fragments and accumulates with
// v_mfma_f32_16x16x4_f32 (4 independent accumulators = 4 row tiles keep the pipe at issue rate).
// K is split over blockIdx.y into partial slabs that the pointwise cell kernel sums, so a step
// occupies (N/64) x splits workgroups with no atomics and no inter-workgroup hand-off.
#include "common.h"

namespace {

// W (N x K, row stride ldw; or K x N when transposed) -> Wp[N/16][KB][64][4], zero padded.
__global__ __launch_bounds__(256) void pack_rec_weight_kernel(const float* W, long ldw, int transposed,
                                                              int N, int n_valid, int K, int KB, float* Wp) {
    const long total = (long)(N / 16) * KB * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long blk = idx >> 6;
        const int kb = (int)(blk % KB);
        const int nb = (int)(blk / KB);
        const int n = nb * 16 + (lane & 15);
        const int k = kb * 16 + 4 * (lane >> 4);
        float4 v;
        float* vv = reinterpret_cast<float*>(&v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = k + j;
            vv[j] = (kk < K && n < n_valid) ? (transposed ? W[(long)kk * ldw + n] : W[(long)n * ldw + kk]) : 0.f;
        }
        reinterpret_cast<float4*>(Wp)[idx] = v;
    }
}

template <int MT>
__global__ __launch_bounds__(256) void rec_gemm_kernel(const float* __restrict__ X, long ldx,
                                                       const float* __restrict__ Wp,
                                                       float* __restrict__ part, int N, int KB, int KBS,
                                                       long xz, long wz, long pz, long ps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = blockIdx.x * 4 + wave;
    if (nb * 16 >= N) return;
    const int s = blockIdx.y;
    const int kb0 = s * KBS;
    const int kb1 = min(KB, kb0 + KBS);
    const int z = blockIdx.z;
    const float4* wp = reinterpret_cast<const float4*>(Wp + (long)z * wz) + ((long)nb * KB + kb0) * 64 + lane;
    const float* xp = X + (long)z * xz + (long)(lane & 15) * ldx + (long)kb0 * 16 + 4 * (lane >> 4);

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll 4
    for (int kb = kb0; kb < kb1; ++kb) {
        const float4 b = *wp;
        float4 a[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4*>(xp + (long)mt * 16 * ldx);
        wp += 64;
        xp += 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b.x, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b.y, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, b.z, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, b.w, acc[mt], 0, 0, 0);
        }
    }
    // C/D map of the 16x16 MFMA: n = lane&15, m = 4*(lane>>4) + reg
    float* out = part + (long)z * pz + (long)s * ps + nb * 16 + (lane & 15);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(long)(mt * 16 + 4 * (lane >> 4) + r) * N] = acc[mt][r];
}

struct CellFwd {
    GtSlabs part;
    const float* pre; long ld_pre;
    const float* c_prev;
    const float* h_prev; long ld_hprev;
    const uint8_t* keep; float scale;
    const int* active;
    float* gates; float* c_out; float* h_raw;
    float* h0; long ld0; float* h1; long ld1; float* h2; long ld2;
    int B, H;
};

__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(CellFwd a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)a.B * a.H) return;
    const int b = (int)(idx / a.H), j = (int)(idx - (long)b * a.H);
    const bool act = (a.active == nullptr) || (a.active[b] != 0);
    float gi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = (a.pre != nullptr) ? a.pre[(long)b * a.ld_pre + q * a.H + j] : 0.f;
        v += slab_sum(a.part.p, a.part.n, a.part.slab_stride, (long)b * a.part.ld + q * a.H + j);
        gi[q] = v;
    }
    const float i = sigmoidf_(gi[0]), f = sigmoidf_(gi[1]), g = tanhf(gi[2]), o = sigmoidf_(gi[3]);
    const float cp = a.c_prev[idx];
    float c = f * cp + i * g;
    float h = o * tanhf(c);
    float hout;
    if (act) {
        hout = h;
        if (a.keep != nullptr) hout = h * (a.keep[idx] ? a.scale : 0.f);
    } else {
        // packed-sequence semantics: state passes through, the emitted output is zero
        c = cp;
        hout = 0.f;
    }
    if (a.gates != nullptr) {
        a.gates[(long)b * 4 * a.H + 0 * a.H + j] = i;
        a.gates[(long)b * 4 * a.H + 1 * a.H + j] = f;
        a.gates[(long)b * 4 * a.H + 2 * a.H + j] = g;
        a.gates[(long)b * 4 * a.H + 3 * a.H + j] = o;
    }
    a.c_out[idx] = c;
    // value carried to the next step: for inactive rows the previous hidden state
    const float hcarry = act ? hout : (a.h_prev != nullptr ? a.h_prev[(long)b * a.ld_hprev + j] : 0.f);
    if (a.h_raw != nullptr) a.h_raw[idx] = hout;          // emitted output (zero when inactive)
    if (a.h0 != nullptr) a.h0[(long)b * a.ld0 + j] = hcarry;
    if (a.h1 != nullptr) a.h1[(long)b * a.ld1 + j] = hcarry;
    if (a.h2 != nullptr) a.h2[(long)b * a.ld2 + j] = hcarry;
}

struct CellBwd {
    GtSlabs da, db, dc;
    const float* dh_direct; long ld_direct;
    const float* dc_next;
    const float* gates; const float* c_new; const float* c_prev;
    const uint8_t* keep; float scale;
    const int* active;
    float* dgates; float* dc_prev; float* dh_pass;
    int B, H;
};

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(CellBwd a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)a.B * a.H) return;
    const int b = (int)(idx / a.H), j = (int)(idx - (long)b * a.H);
    // dh_carry: gradient w.r.t. the hidden state carried into later steps (slab sources),
    // dh_out  : gradient w.r.t. the emitted output of this step (direct source).
    float dh_carry = 0.f;
    if (a.da.p != nullptr) dh_carry += slab_sum(a.da.p, a.da.n, a.da.slab_stride, (long)b * a.da.ld + j);
    if (a.db.p != nullptr) dh_carry += slab_sum(a.db.p, a.db.n, a.db.slab_stride, (long)b * a.db.ld + j);
    if (a.dc.p != nullptr) dh_carry += slab_sum(a.dc.p, a.dc.n, a.dc.slab_stride, (long)b * a.dc.ld + j);
    const float dh_out = (a.dh_direct != nullptr) ? a.dh_direct[(long)b * a.ld_direct + j] : 0.f;
    const float dcn = (a.dc_next != nullptr) ? a.dc_next[idx] : 0.f;
    const bool act = (a.active == nullptr) || (a.active[b] != 0);
    float* dg = a.dgates + (long)b * 4 * a.H;
    if (!act) {
        // inactive packed-sequence row: state passed through untouched, output was zero
        dg[0 * a.H + j] = 0.f; dg[1 * a.H + j] = 0.f; dg[2 * a.H + j] = 0.f; dg[3 * a.H + j] = 0.f;
        a.dc_prev[idx] = dcn;
        if (a.dh_pass != nullptr) a.dh_pass[idx] = dh_carry;
        return;
    }
    float dh = dh_carry + dh_out;
    if (a.keep != nullptr) dh *= (a.keep[idx] ? a.scale : 0.f);
    const float* gt = a.gates + (long)b * 4 * a.H;
    const float i = gt[0 * a.H + j], f = gt[1 * a.H + j], g = gt[2 * a.H + j], o = gt[3 * a.H + j];
    const float tc = tanhf(a.c_new[idx]);
    const float cp = a.c_prev[idx];
    const float d_o = dh * tc;
    const float dcell = dcn + dh * o * (1.f - tc * tc);
    dg[0 * a.H + j] = dcell * g * i * (1.f - i);
    dg[1 * a.H + j] = dcell * cp * f * (1.f - f);
    dg[2 * a.H + j] = dcell * i * (1.f - g * g);
    dg[3 * a.H + j] = d_o * o * (1.f - o);
    a.dc_prev[idx] = dcell * f;
    if (a.dh_pass != nullptr) a.dh_pass[idx] = 0.f;
}

}  // namespace

extern "C" int64_t gt_rec_weight_elems(int32_t N, int32_t K) {
    return (int64_t)(N / 16) * cdiv(K, 16) * 256;
}

extern "C" int gt_pack_rec_weight(const float* W, int64_t ldw, int32_t transposed, int32_t N, int32_t n_valid,
                                  int32_t K, float* Wp, void* stream) {
    GT_REQUIRE(W && Wp && N > 0 && K > 0 && N % 16 == 0 && n_valid > 0 && n_valid <= N, GT_E_ARG,
               "gt_pack_rec_weight: N=%d must be a multiple of 16 and 0 < n_valid=%d <= N", N, n_valid);
    GT_REQUIRE(aligned16(Wp), GT_E_ALIGN, "gt_pack_rec_weight: Wp must be 16-byte aligned");
    const int KB = cdiv(K, 16);
    const long total = (long)(N / 16) * KB * 64;
    const int blocks = (int)std::min<long>(cdiv64(total, 256), 4096);
    hipLaunchKernelGGL(pack_rec_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, (long)ldw,
                       transposed, N, n_valid, K, KB, Wp);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_rec_gemm(const float* X, int64_t ldx, const float* Wp, float* part, int32_t mtiles,
                           int32_t N, int32_t K, int32_t splits, int64_t part_sstride, int32_t nz,
                           int64_t x_zstride, int64_t w_zstride, int64_t part_zstride, void* stream) {
    GT_REQUIRE(X && Wp && part, GT_E_ARG, "gt_rec_gemm: NULL pointer");
    GT_REQUIRE(mtiles >= 1 && mtiles <= 4 && N > 0 && N % 16 == 0 && K > 0 && splits >= 1 && nz >= 1, GT_E_ARG,
               "gt_rec_gemm: bad sizes mtiles=%d N=%d K=%d splits=%d", mtiles, N, K, splits);
    const int KB = cdiv(K, 16);
    GT_REQUIRE(ldx >= (int64_t)KB * 16 && ldx % 4 == 0 && aligned16(X) && aligned16(Wp) && x_zstride % 4 == 0 &&
                   w_zstride % 4 == 0,
               GT_E_ALIGN, "gt_rec_gemm: X needs ldx >= 16*ceil(K/16) (=%d), ldx%%4==0 and 16-byte alignment", KB * 16);
    GT_REQUIRE(part_sstride >= (int64_t)mtiles * 16 * N, GT_E_ARG, "gt_rec_gemm: part_sstride too small");
    const int KBS = cdiv(KB, splits);
    const int ns = cdiv(KB, KBS);
    GT_REQUIRE(ns == splits, GT_E_ARG, "gt_rec_gemm: splits=%d leaves empty slabs for K=%d (use <= %d)", splits, K, ns);
    dim3 grid(cdiv(N, 64), splits, nz);
    hipStream_t st = (hipStream_t)stream;
    GtProfScope prof(GT_FAM_REC_GEMM, 2.0 * 16 * mtiles * (double)N * K * nz,
                     4.0 * nz * ((double)N * K + 16.0 * mtiles * K + 16.0 * mtiles * N * splits), st);
#define GT_REC(MT)                                                                                    \
    hipLaunchKernelGGL((rec_gemm_kernel<MT>), grid, dim3(256), 0, st, X, (long)ldx, Wp, part, N, KB, \
                       KBS, (long)x_zstride, (long)w_zstride, (long)part_zstride, (long)part_sstride)
    switch (mtiles) {
        case 1: GT_REC(1); break;
        case 2: GT_REC(2); break;
        case 3: GT_REC(3); break;
        default: GT_REC(4); break;
    }
#undef GT_REC
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_lstm_cell_fwd(GtSlabs part, const float* pre, int64_t ld_pre, const float* c_prev,
                                const float* h_prev, int64_t ld_hprev, const uint8_t* keep, float p_drop,
                                const int32_t* active, float* gates, float* c_out, float* h_raw, float* h0,
                                int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2, int32_t B,
                                int32_t H, void* stream) {
    GT_REQUIRE(part.p && c_prev && c_out && B > 0 && H > 0 && part.n >= 1, GT_E_ARG, "gt_lstm_cell_fwd: bad argument");
    GT_REQUIRE(p_drop >= 0.f && p_drop < 1.f, GT_E_ARG, "gt_lstm_cell_fwd: p_drop");
    CellFwd a;
    a.part = part; a.pre = pre; a.ld_pre = ld_pre; a.c_prev = c_prev; a.h_prev = h_prev; a.ld_hprev = ld_hprev;
    a.keep = keep; a.scale = 1.0f / (1.0f - p_drop); a.active = active; a.gates = gates; a.c_out = c_out;
    a.h_raw = h_raw; a.h0 = h0; a.ld0 = ld0; a.h1 = h1; a.ld1 = ld1; a.h2 = h2; a.ld2 = ld2; a.B = B; a.H = H;
    const long n = (long)B * H;
    GtProfScope prof(GT_FAM_CELL_FWD, 0.0, 4.0 * n * (4.0 * part.n + 4 + 1 + 4 + 1 + 3), (hipStream_t)stream);
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_lstm_cell_bwd(GtSlabs dh_a, GtSlabs dh_b, GtSlabs dh_c, const float* dh_direct,
                                int64_t ld_direct, const float* dc_next, const float* gates, const float* c_new,
                                const float* c_prev, const uint8_t* keep, float p_drop, const int32_t* active,
                                float* dgates, float* dc_prev, float* dh_pass, int32_t B, int32_t H,
                                void* stream) {
    GT_REQUIRE(gates && c_new && c_prev && dgates && dc_prev && B > 0 && H > 0, GT_E_ARG, "gt_lstm_cell_bwd: bad argument");
    CellBwd a;
    a.da = dh_a; a.db = dh_b; a.dc = dh_c; a.dh_direct = dh_direct; a.ld_direct = ld_direct; a.dc_next = dc_next;
    a.gates = gates; a.c_new = c_new; a.c_prev = c_prev; a.keep = keep; a.scale = 1.0f / (1.0f - p_drop);
    a.active = active; a.dgates = dgates; a.dc_prev = dc_prev; a.dh_pass = dh_pass; a.B = B; a.H = H;
    const long n = (long)B * H;
    GtProfScope prof(GT_FAM_CELL_BWD, 0.0, 4.0 * n * (dh_a.n + dh_b.n + dh_c.n + 1 + 1 + 4 + 2 + 4 + 1), (hipStream_t)stream);
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, a);
    GT_LAUNCH_CHECK();
    return 0;
}
