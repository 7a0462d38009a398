// Fused recurrent product + LSTM cell of one decoder step (forward): gates = X . W^T (+ pre), cell
// pointwise math, dropout, and the scatter of h into the next kernels' operand buffers, in ONE launch
// (reference model.py:391-394 / 406-409: nn.LSTMCell + F.dropout).
//
// Compared with gt_rec_gemm + gt_lstm_cell_fwd (recurrent.hip) this removes a dependent launch per
// cell per step and the K-split slabs (4 x Bp x 4H floats written and read back per product).  The
// price is that a workgroup must see the whole K range of its columns, so the activation matrix is
// staged by every workgroup (L2 -> LDS traffic 4x that of the split kernel); the weights are still
// streamed exactly once.
//
// Work decomposition: one workgroup per 4 hidden units = 16 gate columns (H/4 workgroups: 256 for
// H = 1024, one per CU).  The weights are packed gate-interleaved (gt_pack_rec_weight mode 2: packed
// row 4*unit + gate), and the MFMA runs with the operands swapped, C[n'][m] = W[n'][k] X[m][k], so
// that a lane's four accumulator registers are the i,f,g,o pre-activations of ONE (unit, batch row).
// The 8 waves walk K together: chunk c = k-blocks [8c, 8c+8), wave w takes k-block 8c+w; its weight
// fragment (1 KiB, lane-linear) streams into a 4-deep register ring (32 KiB per CU in flight), the
// activation chunks are double-buffered in LDS behind a 4-deep register ring.  Epilogue: the 8
// K-partials are folded through LDS and MT waves (one per 16-row tile) run the cell.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int FC_CK = 8;                  // k-blocks (of 16) per chunk = waves per workgroup
constexpr int FC_LD = FC_CK * 16;         // LDS row of 32 float4, XOR-swizzled by the row (conflict-free b128, exactly 64 KiB)
constexpr int FC_THREADS = 512;
constexpr int FC_XR = 4;                  // activation chunks in flight (register ring)
constexpr int FC_WR = 4;                  // weight prefetch depth in chunks

struct FusedCell {
    const float* X; long ldx;             // (Bp, ldx) activations, zero padded to 16*KB columns
    const float* Wp;                      // gate-interleaved packed weights [H/4][KB][64][4]
    int H, KB;
    const float* pre; long ld_pre;        // added to the gate pre-activations (row stride 0 = a bias vector)
    const float* c_prev;                  // (Bp, H)
    const uint8_t* keep; float scale;     // dropout keep mask (Bp, H) or NULL
    float* gates; float* c_out; float* h_raw;
    float* h0; long ld0; float* h1; long ld1; float* h2; long ld2;
    int B;                                // valid rows (rows >= B are not written)
};
static_assert(sizeof(FusedCell) <= 1024, "FusedCell is a by-value kernel argument: keep it well below the 4 KB kernel-argument segment");

// DBG (diagnostics, GANTRON_DEBUG=fused=<bits>): bit 0 = no weight loads, bit 1 = no MFMAs, bit 2 = no activation staging,
// bit 3 = no chunk rotation
template <int MT, int DBG = 0>
__global__ __launch_bounds__(FC_THREADS) void rec_gemm_cell_kernel(FusedCell a) {
    __shared__ __attribute__((aligned(16))) float Xs[2][MT * 16 * FC_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = blockIdx.x;
    const int row0 = blockIdx.z * 64;
    const float* X = a.X + (long)row0 * a.ldx;
    const int KB = a.KB;
    const int nchunks = KB / FC_CK;           // KB is a multiple of FC_CK * FC_XR (host check): uniform, branch-free loop
    const float4* wp = reinterpret_cast<const float4*>(a.Wp) + ((long)nb * KB + wave) * 64 + lane;

    // Workgroups walk the chunks in rotated order: in lockstep they would all request the same 32 KiB of
    // activations at the same moment and hammer one or two L2 channels (measured: 2.3x slower).  Workgroup
    // nb runs on XCD nb % 8, so the rotation is taken from nb / 8.
    const int rot = (DBG & 8) ? 0 : (nb >> 3) % nchunks;
    // activation staging: MT*16 rows x 32 float4 per chunk, register ring of FC_XR chunks in flight.
    // Every load below is unconditional (KB is a multiple of 8*FC_XR, buffers are zero padded; prefetches
    // past the end re-read the last chunk): with predicated loads the compiler can no longer count the
    // outstanding ones and falls back to s_waitcnt vmcnt(0), which serialises the whole pipeline.
    constexpr int NV = MT * 16 * FC_CK * 4 / FC_THREADS;          // float4 per thread (MT=4: 4, MT=1: 1)
    static_assert(NV >= 1, "staging");
    // element i of a chunk: row tid / 32 + 16 i, float4 column tid % 32.  The register ring is sixteen NAMED
    // float4 variables driven by macros: arrays / structs passed to lambdas ended up in scratch memory.
    const int srow = tid / (FC_CK * 4), sc4 = tid % (FC_CK * 4);
    const float* xbase = X + (long)srow * a.ldx + 4 * sc4;
    const long xrow16 = 16L * a.ldx;
    const int xslot = srow * FC_LD + 4 * (sc4 ^ (srow & 15));      // (row + 16 i) & 15 == row & 15
#define FC_XLOAD(R, C)                                                                  \
    if (!(DBG & 4)) {                                                                   \
        const float* p_ = xbase + (long)(((C) + rot) % nchunks) * FC_CK * 16;           \
        R##0 = *reinterpret_cast<const float4*>(p_);                                    \
        if (NV > 1) R##1 = *reinterpret_cast<const float4*>(p_ + xrow16);               \
        if (NV > 2) R##2 = *reinterpret_cast<const float4*>(p_ + 2 * xrow16);           \
        if (NV > 3) R##3 = *reinterpret_cast<const float4*>(p_ + 3 * xrow16);           \
    }
#define FC_XSTORE(R, BUF)                                                               \
    if (!(DBG & 4)) {                                                                   \
        float* q_ = &Xs[BUF][xslot];                                                    \
        *reinterpret_cast<float4*>(q_) = R##0;                                          \
        if (NV > 1) *reinterpret_cast<float4*>(q_ + 16 * FC_LD) = R##1;                 \
        if (NV > 2) *reinterpret_cast<float4*>(q_ + 32 * FC_LD) = R##2;                 \
        if (NV > 3) *reinterpret_cast<float4*>(q_ + 48 * FC_LD) = R##3;                 \
    }
    auto wload = [&](int c) __attribute__((always_inline)) -> float4 {
        if (DBG & 1) return make_float4(1.f, 1.f, 1.f, 1.f);
        return wp[(long)((c + rot) % nchunks) * FC_CK * 64];
    };

    // the cell's own inputs are requested now and arrive while the product runs.
    // C/D map with swapped operands: column (batch row) = lane & 15, rows 4*(lane>>4) + reg -> unit lane>>4, gate reg
    const int H = a.H;
    const int m = row0 + wave * 16 + (lane & 15);
    const int u = nb * 4 + (lane >> 4);
    const bool cell_lane = wave < MT && m < a.B;
    const long idx = (long)m * H + u;
    float pre0 = 0.f, pre1 = 0.f, pre2 = 0.f, pre3 = 0.f;
    float cp = 0.f;
    bool kept = true;
    if (cell_lane) {
        if (a.pre != nullptr) {
            const float* p = a.pre + (long)m * a.ld_pre + u;
            pre0 = p[0]; pre1 = p[H]; pre2 = p[2L * H]; pre3 = p[3L * H];
        }
        cp = a.c_prev[idx];
        if (a.keep != nullptr) kept = a.keep[idx] != 0;
    }

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // software pipeline: chunk c's activations sit in LDS buffer c&1; chunks c+1..c+FC_XR-1 are in flight
    // in the register ring x[], the weights of chunks c..c+FC_WR-1 in w[] (global/L2 latency is several
    // chunks long: one chunk is only 16 MFMAs per wave)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 xa0 = z4, xa1 = z4, xa2 = z4, xa3 = z4, xb0 = z4, xb1 = z4, xb2 = z4, xb3 = z4;
    float4 xc0 = z4, xc1 = z4, xc2 = z4, xc3 = z4, xd0 = z4, xd1 = z4, xd2 = z4, xd3 = z4;
    FC_XLOAD(xa, 0) FC_XLOAD(xb, 1) FC_XLOAD(xc, 2) FC_XLOAD(xd, 3)
    float4 w0 = wload(0), w1 = wload(1), w2 = wload(2), w3 = wload(3);
    FC_XSTORE(xa, 0)
    __syncthreads();
    const int xoff = (lane & 15) * FC_LD + 4 * ((wave * 4 + (lane >> 4)) ^ (lane & 15));    // swizzled float4 slot

    auto mfmas = [&](const float4 b, int c) __attribute__((always_inline)) {
        const float* xs = &Xs[c & 1][xoff];
        float4 xr[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xr[mt] = *reinterpret_cast<const float4*>(xs + mt * 16 * FC_LD);
        if (DBG & 2) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { acc[mt][0] += b.x * xr[mt].x; acc[mt][1] += b.y * xr[mt].y; acc[mt][2] += b.z * xr[mt].z; acc[mt][3] += b.w * xr[mt].w; }
            return;
        }
        // swapped operands: rows of C = packed weight rows (4*unit + gate), columns = batch rows
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, xr[mt].x, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, xr[mt].y, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, xr[mt].z, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, xr[mt].w, acc[mt], 0, 0, 0);
    };
    // chunk C: refill the free register set with chunk C+4, run chunk C from LDS, prefetch the weights of
    // chunk C+4, publish chunk C+1 (after the last chunk this stores a re-read chunk nobody uses)
#define FC_CHUNK(W, XFREE, XNEXT, C)              \
    {                                             \
        FC_XLOAD(XFREE, (C) + FC_XR)              \
        const float4 b_ = W;                      \
        W = wload((C) + FC_WR);                   \
        mfmas(b_, (C));                           \
        FC_XSTORE(XNEXT, ((C) + 1) & 1)           \
        __syncthreads();                          \
    }
    static_assert(FC_XR == 4 && FC_WR == 4, "the unrolled schedule below is written for these depths");
    for (int c = 0; c < nchunks; c += 4) {       // nchunks % 4 == 0 (host check)
        FC_CHUNK(w0, xa, xb, c)
        FC_CHUNK(w1, xb, xc, c + 1)
        FC_CHUNK(w2, xc, xd, c + 2)
        FC_CHUNK(w3, xd, xa, c + 3)
    }
#undef FC_CHUNK
#undef FC_XLOAD
#undef FC_XSTORE

    // fold the 8 K-partials: red[(wave*MT + mt)*64 + lane] = float4 of the lane's four gate registers
    float4* red = reinterpret_cast<float4*>(&Xs[0][0]);
    static_assert(FC_CK * MT * 64 * 4 <= 2 * MT * 16 * FC_LD, "fold buffer fits the staging buffers");
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        red[(wave * MT + mt) * 64 + lane] = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
    __syncthreads();
    if (!cell_lane) return;
    float g[4] = {pre0, pre1, pre2, pre3};
#pragma unroll
    for (int sw = 0; sw < FC_CK; ++sw) {
        const float4 t = red[(sw * MT + wave) * 64 + lane];
        g[0] += t.x; g[1] += t.y; g[2] += t.z; g[3] += t.w;
    }
    const float gi = sigmoidf_(g[0]), gf = sigmoidf_(g[1]), gg = tanhf_(g[2]), go = sigmoidf_(g[3]);
    const float c = gf * cp + gi * gg;
    float h = go * tanhf_(c);
    if (a.keep != nullptr) h *= kept ? a.scale : 0.f;
    if (a.gates != nullptr) {
        float* gp = a.gates + (long)m * 4 * H + u;
        gp[0] = gi; gp[H] = gf; gp[2L * H] = gg; gp[3L * H] = go;
    }
    a.c_out[idx] = c;
    if (a.h_raw != nullptr) a.h_raw[idx] = h;
    if (a.h0 != nullptr) a.h0[(long)m * a.ld0 + u] = h;
    if (a.h1 != nullptr) a.h1[(long)m * a.ld1 + u] = h;
    if (a.h2 != nullptr) a.h2[(long)m * a.ld2 + u] = h;
}

}  // namespace

// K rounded up to whole pipeline rounds (FC_XR chunks of FC_CK 16-wide k-blocks = 512): the packed weights and
// the activation rows are zero padded to this many columns so that the kernel's loop is uniform
extern "C" int64_t gt_fused_k_padded(int32_t K) {
    const int64_t q = 16LL * FC_CK * FC_XR;
    return ((int64_t)K + q - 1) / q * q;
}

extern "C" int gt_lstm_step_fused(const float* X, int64_t ldx, const float* Wg, int32_t K, const float* pre,
                                  int64_t ld_pre, const float* c_prev, const uint8_t* keep, float p_drop, float* gates,
                                  float* c_out, float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2,
                                  int64_t ld2, int32_t B, int32_t Bp, int32_t H, void* stream) {
    GT_REQUIRE(X && Wg && c_prev && c_out, GT_E_ARG, "gt_lstm_step_fused: NULL pointer");
    GT_REQUIRE(B > 0 && Bp >= B && ((Bp <= 64 && Bp % 16 == 0) || Bp % 64 == 0) && H > 0 && H % 4 == 0 && K > 0, GT_E_ARG,
               "gt_lstm_step_fused: bad sizes B=%d Bp=%d H=%d K=%d", B, Bp, H, K);
    GT_REQUIRE(p_drop >= 0.f && p_drop < 1.f, GT_E_ARG, "gt_lstm_step_fused: p_drop");
    const int KB = (int)(gt_fused_k_padded(K) / 16);
    GT_REQUIRE(ldx >= (int64_t)KB * 16 && ldx % 4 == 0 && aligned16(X) && aligned16(Wg), GT_E_ALIGN,
               "gt_lstm_step_fused: X needs ldx >= gt_fused_k_padded(K) (=%d) zero-padded columns, ldx%%4==0 and 16-byte "
               "alignment", KB * 16);
    FusedCell a;
    a.X = X; a.ldx = ldx; a.Wp = Wg; a.H = H; a.KB = KB; a.pre = pre; a.ld_pre = ld_pre; a.c_prev = c_prev; a.keep = keep;
    a.scale = 1.0f / (1.0f - p_drop); a.gates = gates; a.c_out = c_out; a.h_raw = h_raw;
    a.h0 = h0; a.ld0 = ld0; a.h1 = h1; a.ld1 = ld1; a.h2 = h2; a.ld2 = ld2; a.B = B;
    hipStream_t st = (hipStream_t)stream;
    const int mtiles = Bp <= 64 ? Bp / 16 : 4;
    dim3 grid(H / 4, 1, Bp <= 64 ? 1 : Bp / 64);
    const double rows = (double)Bp, N = 4.0 * H;
    // algorithmic bytes: weights once + activations once + cell state / outputs
    GtProfScope prof(GT_FAM_REC_GEMM, 2.0 * rows * N * K, 4.0 * (N * K + rows * K + rows * H * 12.0), st);
    switch (mtiles) {
        case 1: hipLaunchKernelGGL((rec_gemm_cell_kernel<1>), grid, dim3(FC_THREADS), 0, st, a); break;
        case 2: hipLaunchKernelGGL((rec_gemm_cell_kernel<2>), grid, dim3(FC_THREADS), 0, st, a); break;
        case 3: hipLaunchKernelGGL((rec_gemm_cell_kernel<3>), grid, dim3(FC_THREADS), 0, st, a); break;
        default: {
            static const int dbg = gt_env_item("GANTRON_DEBUG", "fused", 0);
            switch (dbg) {      // diagnostic variants give wrong results by construction
                case 1: hipLaunchKernelGGL((rec_gemm_cell_kernel<4, 1>), grid, dim3(FC_THREADS), 0, st, a); break;
                case 2: hipLaunchKernelGGL((rec_gemm_cell_kernel<4, 2>), grid, dim3(FC_THREADS), 0, st, a); break;
                case 4: hipLaunchKernelGGL((rec_gemm_cell_kernel<4, 4>), grid, dim3(FC_THREADS), 0, st, a); break;
                case 5: hipLaunchKernelGGL((rec_gemm_cell_kernel<4, 5>), grid, dim3(FC_THREADS), 0, st, a); break;
                case 6: hipLaunchKernelGGL((rec_gemm_cell_kernel<4, 6>), grid, dim3(FC_THREADS), 0, st, a); break;
                case 7: hipLaunchKernelGGL((rec_gemm_cell_kernel<4, 7>), grid, dim3(FC_THREADS), 0, st, a); break;
                case 8: hipLaunchKernelGGL((rec_gemm_cell_kernel<4, 8>), grid, dim3(FC_THREADS), 0, st, a); break;
                default: hipLaunchKernelGGL((rec_gemm_cell_kernel<4>), grid, dim3(FC_THREADS), 0, st, a); break;
            }
        }
    }
    GT_LAUNCH_CHECK();
    return 0;
}
