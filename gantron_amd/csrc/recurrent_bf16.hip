// bf16-operand variant of the recurrent product (hparams.fp16_run; BASELINE config 3): the weights are packed as
// bf16 fragments of v_mfma_f32_16x16x32_bf16, the fp32 activations are rounded to bf16 while they are staged through
// LDS, accumulation and the output slabs stay fp32.
//
// At M <= 64 rows the fp32 product is bound by MFMA issue (10 us for N=4096, K=3072); in bf16 the same MFMAs take
// 1/16 of that, so the kernel is shaped for LATENCY instead: a workgroup's K-slice is at most 4 chunks of 256 and
// the global loads of the slice -- the wave's weight fragments (16 x 16 B per lane) and the activation rows of the
// first three chunks (24 float4 per thread; the fourth chunk reuses the first one's registers as soon as that one is
// in LDS) -- are issued before the first MFMA, then the chunks are converted, published and multiplied in turn.
// Same decomposition as gt_rec_gemm otherwise: 64 gate columns per workgroup, wave w owns columns [16*(w&3), +16)
// and the k-blocks of half (w>>2) of every chunk, K split over blockIdx.y into slabs.
//
// NP = 3 ("x3", gt_rec_gemm_x3 / gt_rec_cell_fwd_x3): the SAME kernels as an exact fp32 product.  The weights are packed as
// three bf16 planes w = w0 + w1 + w2 (w0 = bf16(w), w1 = bf16(w - w0), w2 = bf16(w - w0 - w1): all 24 significand bits),
// the activations are split the same way while they are staged through LDS, and a . b = a0 b0 + (a0 b1 + a1 b0) +
// (a0 b2 + a1 b1 + a2 b0) runs as six v_mfma_f32_16x16x32_bf16 with fp32 accumulation, smallest terms first -- every
// product exact, dropped terms <= 2^-24 |a||b|, the arithmetic of xgemm_kernel (gemm.hip).  At M = 64 rows the fp32 pipe
// needs 6.8 us for a workgroup's 64 columns x 512 k (v_mfma_f32_16x16x4_f32: 256 flop per clock and CU); the six bf16
// products take 2.6 us.  The three weight planes are held two chunks at a time and refilled two chunks ahead.
//
// NP = 2 ("h2", round 4: gt_rec_gemm_h2 / gt_rec_cell_fwd_h2): the fp32 product from two SCALED fp16 planes per operand, the
// arithmetic of gemm.hip's mode 4 -- x s = h0 + h1 (+ r), |r| <= 2^-22 |x s|, a . b = (a0 b0 + a0 b1 + a1 b0) / (sa sb) as three
// v_mfma_f32_16x16x32_f16 -- at 4 bytes per weight (what fp32 streams) and 3/16 of the fp32 pipe's matrix time.  The weights'
// power-of-two scale comes from max |W|, found when they are packed and kept in a 16-byte trailer of the image; the activations'
// from a device word the caller names (gt_absmax format) and a floor (a bound known a priori: h = o tanh c / (1 - p)).
#include <stdint.h>
#include "common.h"
#include "rec_cell.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));

// power-of-two scale that puts a tensor's largest magnitude (fp32 bit pattern) into [2^14, 2^15): gemm.hip's h2_scale_exp
__device__ __forceinline__ int rb_h2_exp(unsigned maxbits) {
    const int e = (int)((maxbits >> 23) & 0xffu);
    return min(127, max(-126, 141 - e));
}
__device__ __forceinline__ float rb_pow2(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }
// (x, y) scaled by s -> the pair's two fp16 planes, first element in the low half
__device__ __forceinline__ void rb_split_h2(float x, float y, float s, unsigned& p0, unsigned& p1) {
    float ra, rb;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(p0) : "s"(s), "v"(x));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(p0) : "s"(s), "v"(y));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ra) : "s"(s), "v"(x), "v"(p0));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "s"(s), "v"(y), "v"(p0));
    const f16x2 q = __builtin_convertvector((f32x2_){ra, rb}, f16x2);
    p1 = __builtin_bit_cast(unsigned, q);
}

constexpr int RB_THREADS = 512;
constexpr int RB_CB = 8;               // 32-wide k-blocks per chunk (256 k)
constexpr int RB_HB = RB_CB / 2;       // k-blocks per wave per chunk
constexpr int RB_NCH = 4;              // chunks per K-slice at most (the kernel is instantiated for 1..4)
constexpr int RB_XR = 3;               // activation chunks held in registers at a time
constexpr int RB_LD = RB_CB * 32 + 8;  // bf16 per LDS row: 528 B, rows 16 B apart mod 256 B (conflict-free b128 reads)

__device__ __forceinline__ unsigned pack2(float a, float b) {
    const bf16x2 p = {(__bf16)a, (__bf16)b};       // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return __builtin_bit_cast(unsigned, p);
}

// W (N x K, row stride ldw; or K x N when transposed) -> Wp[N/16][KB32][64 lanes][8 bf16]: lane l of block (nb, kb)
// holds B[k = 32 kb + 8 (l >> 4) + j][column 16 nb + (l & 15)], j = 0..7, zero padded.
// cellH > 0 (gt_pack_rec_cell_weight_bf16): W is a (4 cellH x K) LSTM weight and packed column 64 cg + 16 g + u is its row
// g * cellH + 16 cg + u, so that a 64-column group holds the four gates of 16 hidden units (gt_rec_cell_fwd_bf16).
// planes = 3: block (nb, kb) is followed by its second and third plane (residuals), ((nb KB + kb) planes + p) 64 + lane.
// planes = 2: two fp16 planes of W scaled by the power of two of *wmax (gt_absmax word of W).
__global__ __launch_bounds__(256) void pack_rec_weight_bf16_kernel(const float* W, long ldw, int transposed, int N,
                                                                   int n_valid, int K, int KB, int cellH, uint4* Wp, int planes,
                                                                   const unsigned long long* wmax) {
    const long total = (long)(N / 16) * KB * 64;
    float hs = 1.f;
    if (planes == 2) hs = rb_pow2(rb_h2_exp(__builtin_amdgcn_readfirstlane((unsigned)*wmax)));
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long blk = idx >> 6;
        const int kb = (int)(blk % KB), nb = (int)(blk / KB);
        int n = nb * 16 + (lane & 15);
        if (cellH > 0) n = ((n >> 4) & 3) * cellH + 16 * (n >> 6) + (n & 15);
        const int k = kb * 32 + 8 * (lane >> 4);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = k + j;
            v[j] = (kk < K && n < n_valid) ? (transposed ? W[(long)kk * ldw + n] : W[(long)n * ldw + kk]) : 0.f;
        }
        if (planes == 2) {
            unsigned h0[4], h1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) rb_split_h2(v[2 * j], v[2 * j + 1], hs, h0[j], h1[j]);
            Wp[(blk * 2 + 0) * 64 + lane] = make_uint4(h0[0], h0[1], h0[2], h0[3]);
            Wp[(blk * 2 + 1) * 64 + lane] = make_uint4(h1[0], h1[1], h1[2], h1[3]);
            continue;
        }
        for (int p = 0; p < planes; ++p) {
            const uint4 o = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
            Wp[(blk * planes + p) * 64 + lane] = o;
            const unsigned u[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {          // residuals (exact in fp32)
                v[2 * j] -= __uint_as_float(u[j] << 16);
                v[2 * j + 1] -= __uint_as_float(u[j] & 0xffff0000u);
            }
        }
    }
}

// float4 -> its NP bf16 planes, (x, y) packed in .x and (z, w) in .y of plane p
template <int NP>
__device__ __forceinline__ void split_planes(float4 x, uint2 (&out)[NP], float hs) {
    if constexpr (NP == 2) {
        unsigned a0, a1, b0, b1;
        rb_split_h2(x.x, x.y, hs, a0, a1);
        rb_split_h2(x.z, x.w, hs, b0, b1);
        out[0] = make_uint2(a0, b0);
        out[1] = make_uint2(a1, b1);
        return;
    }
    float v[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        unsigned u0 = pack2(v[0], v[1]), u1 = pack2(v[2], v[3]);
        asm("" : "+v"(u0));                      // keep the pair packed (otherwise every element is converted twice)
        asm("" : "+v"(u1));
        out[p] = make_uint2(u0, u1);
        if (p + 1 < NP) {
            v[0] -= __uint_as_float(u0 << 16); v[1] -= __uint_as_float(u0 & 0xffff0000u);
            v[2] -= __uint_as_float(u1 << 16); v[3] -= __uint_as_float(u1 & 0xffff0000u);
        }
    }
}

// NP = 2: where the operand scales come from (kernel argument; zeros for the other formats)
struct RecH2 { const unsigned long long* wmax; const unsigned long long* xmax; float xfloor; };
__device__ __forceinline__ void rb_h2_scales(const RecH2& h, float& hs, int& hk) {
    // (wave-uniform; agent-scope loads: the activations' word may have been written by atomics of the launch in front of this one)
    const unsigned wb = (unsigned)__hip_atomic_load(h.wmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned xb = h.xmax != nullptr ? (unsigned)__hip_atomic_load(h.xmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    xb = max(xb, __float_as_uint(h.xfloor));              // (non-negative floats order like their bit patterns)
    const int kw = rb_h2_exp(__builtin_amdgcn_readfirstlane(wb)), kx = rb_h2_exp(__builtin_amdgcn_readfirstlane(xb));
    hs = rb_pow2(kx);
    hk = kw + kx;
}

__device__ __forceinline__ void rb_setprio(int prio) {
    // issue priority of the waves: 3 = the loops' critical (attention) chain, 2 = the other chain, 0 = everything else
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
}

// One workgroup's [MT*16 rows x 64 columns] tile of X . W^T over the k-blocks [kb0, kb1) of packed weights Wz: returns it
// folded over the two K-halves in LDS (row stride REC_EP_LD floats, at the start of the staging array Xs), after a barrier.
template <int MT, int NCH, int NP>
__device__ __forceinline__ float* rec_bf16_tile(unsigned short* Xs, const float* __restrict__ Xz, long ldx, int kvalid,
                                                const uint4* __restrict__ Wz, int N, int KB, int kb0, int kb1, int cg,
                                                long long* stamps = nullptr, float hs = 1.f, int hk = 0) {
    // (NP = 2: hs = the activations' scale 2^kx, hk = kx + kw: the folded tile leaves as acc 2^-hk)
    constexpr int PL = MT * 16 * RB_LD;      // bf16 elements per activation plane in LDS
    constexpr int XV = MT * 2;         // float4 per thread per chunk (MT*16 rows x 64 float4 / 512 threads)
    constexpr int XR = NP >= 2 ? (NCH < 2 ? NCH : 2) : (NCH < RB_XR ? NCH : RB_XR);
    // weight chunks held in registers: all of the slice (NP = 1), or two at a time, refilled two chunks ahead (NP = 3:
    // three planes of a chunk are 48 registers; NP = 2: 32)
    constexpr int WR = NP >= 2 ? (NCH < 2 ? NCH : 2) : NCH;
    constexpr int EP_LD = REC_EP_LD;
    static_assert(MT * 16 * RB_LD * 2 >= (MT * 16 * (EP_LD + 20) + 4) * 4, "epilogue tile + scratch fit the staging buffer");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = wave >> 2;
    int nb = cg * 4 + (wave & 3);
    const bool active = nb * 16 < N;   // inactive waves (N % 64 != 0) still stage and join the barriers
    if (!active) nb = 0;
    const uint4* wp = Wz + ((long)nb * KB) * 64 * NP + lane;

    // ---- every load of the slice, issued together.  Out-of-range blocks are clamped to a valid address and
    // zeroed by a select (a predicated load would put s_waitcnt vmcnt(0) in front of everything behind it).
    uint4 w[WR][RB_HB][NP];
    float4 x[XR][XV];
    auto wload = [&](uint4 (&dst)[RB_HB][NP], int c) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < RB_HB; ++u) {
            const int kb = kb0 + c * RB_CB + half * RB_HB + u;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const uint4 t = wp[((long)min(kb, kb1 - 1) * NP + p) * 64];
                dst[u][p] = kb < kb1 ? t : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    auto xload = [&](float4 (&dst)[XV], int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int f = tid + RB_THREADS * i, row = f >> 6, c4 = f & 63;
            const int k = (kb0 + c * RB_CB) * 32 + 4 * c4;
            dst[i] = *reinterpret_cast<const float4*>(Xz + (long)row * ldx + min(k, kvalid - 4));
        }
    };
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c < WR) wload(w[c], c);
        if (c < XR) xload(x[c], c);
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned short* xs = Xs + (lane & 15) * RB_LD + 8 * (lane >> 4) + half * RB_HB * 32;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (kb0 + c * RB_CB < kb1) {   // uniform over the workgroup
#pragma unroll
            for (int i = 0; i < XV; ++i) {
                const int f = tid + RB_THREADS * i, row = f >> 6, c4 = f & 63;
                // activations beyond the valid columns meet zero weights; the clamped load keeps them finite
                uint2 pl[NP];
                split_planes<NP>(x[c % XR][i], pl, hs);
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(&Xs[p * PL + row * RB_LD + 4 * c4]) = pl[p];
            }
            if (c + XR < NCH) xload(x[c % XR], c + XR);      // registers of this chunk are free again
            __syncthreads();
            if (stamps != nullptr && tid == 0) { if (c == 0) { stamps[1] = wall_clock64(); stamps[7] = clock64(); } }
#pragma unroll
            for (int u = 0; u < RB_HB; ++u) {
                bf16x8 a[NP][MT];
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        a[p][mt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xs + p * PL + mt * 16 * RB_LD + u * 32));
                // plane products of order < NP, smallest terms first; the MT row tiles lie between two MFMAs on one accumulator
#pragma unroll
                for (int ord = NP - 1; ord >= 0; --ord)
#pragma unroll
                    for (int pa = 0; pa <= ord; ++pa) {
                        const bf16x8 b = __builtin_bit_cast(bf16x8, w[c % WR][u][ord - pa]);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if constexpr (NP == 2)
                                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[pa][mt]),
                                                                                 __builtin_bit_cast(f16x8, b), acc[mt], 0, 0, 0);
                            else
                                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pa][mt], b, acc[mt], 0, 0, 0);
                        }
                    }
            }
            if (c + WR < NCH) wload(w[c % WR], c + WR);      // registers of this chunk's weights are free again
            __syncthreads();
        }
    }
    if (stamps != nullptr && tid == 0) { stamps[2] = wall_clock64(); stamps[7] = clock64() - stamps[7]; }
    // ---- epilogue through LDS: fold the two K-halves, write the slab with full-line float4 stores
    float* red = reinterpret_cast<float*>(Xs);
    const int ecol = (wave & 3) * 16 + (lane & 15), erow = 4 * (lane >> 4);
    if constexpr (NP == 2) {           // the operand scales leave, exactly
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][r] = ldexpf(acc[mt][r], -hk);
    }
    if (half == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(mt * 16 + erow + r) * EP_LD + ecol] = acc[mt][r];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(mt * 16 + erow + r) * EP_LD + ecol] += acc[mt][r];
    }
    __syncthreads();
    return red;
}

template <int MT, int NCH, int NP>
__global__ __launch_bounds__(RB_THREADS) void rec_gemm_bf16_kernel(const float* __restrict__ X, long ldx, int kvalid,
                                                                   const uint4* __restrict__ Wp, float* __restrict__ part,
                                                                   int N, int KB, int KBS, long xz, long wz, long pz,
                                                                   long ps, int prio, RecH2 h2) {
    rb_setprio(prio);
    __shared__ __attribute__((aligned(16))) unsigned short Xs[NP * MT * 16 * RB_LD];
    const int s = blockIdx.y, z = blockIdx.z, cg = blockIdx.x;
    const int kb0 = s * KBS;
    float hs = 1.f;
    int hk = 0;
    if constexpr (NP == 2) rb_h2_scales(h2, hs, hk);
    const float* red = rec_bf16_tile<MT, NCH, NP>(Xs, X + (long)z * xz, ldx, kvalid, Wp + (long)z * wz, N, KB, kb0, min(KB, kb0 + KBS), cg,
                                                  nullptr, hs, hk);
    constexpr int EP_LD = REC_EP_LD;
    const int tid = threadIdx.x;
    float* out = part + (long)z * pz + (long)s * ps + (long)cg * 64;
    const int ncols = min(64, N - cg * 64);
#pragma unroll
    for (int i = 0; i < (MT * 16 * 16 + RB_THREADS - 1) / RB_THREADS; ++i) {
        const int f = tid + RB_THREADS * i;
        const int row = f >> 4, c4 = f & 15;
        if (row < MT * 16 && 4 * c4 < ncols)
            *reinterpret_cast<float4*>(out + (long)row * N + 4 * c4) = *reinterpret_cast<const float4*>(&red[row * EP_LD + 4 * c4]);
    }
}

// gt_rec_cell_fwd_bf16: the product above with the fused cell epilogue of rec_cell.h (gate-grouped packing, last-arriver
// split-K).  Same launch geometry as the fp32 rec_cell_kernel: blockIdx = (column group, K-slice, direction * nzb + row block).
template <int MT, int NCH, int NP>
__global__ __launch_bounds__(RB_THREADS) void rec_cell_bf16_kernel(const float* __restrict__ X, long ldx, int kvalid,
                                                                   const uint4* __restrict__ Wp, float* part, int N, int KB,
                                                                   int KBS, long xz, long pz, long ps, int prio, RecCellEpi e,
                                                                   RecH2 h2) {
    rb_setprio(prio);
    __shared__ __attribute__((aligned(16))) unsigned short Xs[NP * MT * 16 * RB_LD];
    const int cg = blockIdx.x, s = blockIdx.y, z = blockIdx.z;
    float hs = 1.f;
    int hk = 0;
    if constexpr (NP == 2) rb_h2_scales(h2, hs, hk);
    const int dir = z / e.nzb, rb = z - dir * e.nzb;
    const int kb0 = s * KBS;
    if (dir != 0) {
        X += e.dX; Wp += e.dW / 8; part += e.dPart;          // dW counts bf16 elements
        rec_cell_dir1(e);
    }
    REC_STAMP(e, 0);
    float* red = rec_bf16_tile<MT, NCH, NP>(Xs, X + (long)rb * xz, ldx, kvalid, Wp, N, KB, kb0, min(KB, kb0 + KBS), cg,
                                            e.stamp_on ? g_rec_stamps_dev + ((long)blockIdx.x + (long)gridDim.x * blockIdx.y) * 8 : nullptr,
                                            hs, hk);
    rec_cell_epilogue<MT>(red, e, part + (long)rb * pz, ps, N, cg, s, rb, e.counters + (long)z * gridDim.x + cg);
}

}  // namespace

extern "C" int64_t gt_rec_weight_bf16_elems(int32_t N, int32_t K) {
    return (int64_t)(N / 16) * cdiv(K, 32) * 64 * 8;
}
extern "C" int64_t gt_rec_weight_x3_elems(int32_t N, int32_t K) { return 3 * gt_rec_weight_bf16_elems(N, K); }
// two fp16 planes + a 16-byte trailer: the gt_absmax word of W (its low half = the bits of max |W|) the scale was taken from
extern "C" int64_t gt_rec_weight_h2_elems(int32_t N, int32_t K) { return 2 * gt_rec_weight_bf16_elems(N, K) + 8; }
static const unsigned long long* h2_trailer(const void* Wp, int N, int K) {
    return reinterpret_cast<const unsigned long long*>(static_cast<const unsigned short*>(Wp) + 2 * gt_rec_weight_bf16_elems(N, K));
}
// the activations' side of a planes == 2 launch: named by the caller right before it (this thread's next launch)
thread_local GtRecH2X g_rec_h2x = {nullptr, 0.f};

extern "C" int32_t gt_rec_bf16_min_splits(int32_t K) { return cdiv(cdiv(K, 32), RB_NCH * RB_CB); }
constexpr int RB_NCH_X3 = 4;           // chunks per K-slice (the three weight planes are held two chunks at a time)
extern "C" int32_t gt_rec_x3_min_splits(int32_t K) { return cdiv(cdiv(K, 32), RB_NCH_X3 * RB_CB); }

static int pack_rec_planes(const char* what, const float* W, int64_t ldw, int32_t transposed, int32_t N, int32_t n_valid,
                           int32_t K, int32_t cellH, void* Wp, int planes, void* stream) {
    GT_REQUIRE(W && Wp && N > 0 && K > 0 && N % 16 == 0 && n_valid > 0 && n_valid <= N && (transposed == 0 || transposed == 1),
               GT_E_ARG, "%s: N=%d must be a multiple of 16 and 0 < n_valid=%d <= N", what, N, n_valid);
    GT_REQUIRE(aligned16(Wp), GT_E_ALIGN, "%s: Wp must be 16-byte aligned", what);
    const int KB = cdiv(K, 32);
    const long total = (long)(N / 16) * KB * 64;
    const int blocks = (int)std::min<long>(cdiv64(total, 256), 4096);
    const unsigned long long* wmax = nullptr;
    if (planes == 2) {              // max |W| over the valid part, into the image's trailer (same stream: in front of the pack)
        wmax = h2_trailer(Wp, N, K);
        int rc = transposed ? gt_absmax(W, K, n_valid, ldw, (uint64_t*)wmax, stream) : gt_absmax(W, n_valid, K, ldw, (uint64_t*)wmax, stream);
        if (rc != 0) return rc;
    }
    hipLaunchKernelGGL(pack_rec_weight_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, (long)ldw,
                       transposed, N, n_valid, K, KB, cellH, reinterpret_cast<uint4*>(Wp), planes, wmax);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_pack_rec_weight_bf16(const float* W, int64_t ldw, int32_t transposed, int32_t N, int32_t n_valid,
                                       int32_t K, void* Wp, void* stream) {
    return pack_rec_planes("gt_pack_rec_weight_bf16", W, ldw, transposed, N, n_valid, K, 0, Wp, 1, stream);
}
extern "C" int gt_pack_rec_weight_x3(const float* W, int64_t ldw, int32_t transposed, int32_t N, int32_t n_valid,
                                     int32_t K, void* Wp, void* stream) {
    return pack_rec_planes("gt_pack_rec_weight_x3", W, ldw, transposed, N, n_valid, K, 0, Wp, 3, stream);
}

extern "C" int gt_pack_rec_weight_h2(const float* W, int64_t ldw, int32_t transposed, int32_t N, int32_t n_valid,
                                     int32_t K, void* Wp, void* stream) {
    return pack_rec_planes("gt_pack_rec_weight_h2", W, ldw, transposed, N, n_valid, K, 0, Wp, 2, stream);
}
extern "C" int gt_rec_gemm_h2(const float* X, int64_t ldx, const void* Wp, float* part, int32_t mtiles, int32_t N,
                              int32_t K, int32_t splits, int64_t part_sstride, int32_t nz, int64_t x_zstride,
                              int64_t part_zstride, const uint64_t* xmax, float xfloor, void* stream) {
    // (the product in recurrent.hip's geometry: rec_gemm_h2_kernel; this file's NP = 2 form stays behind GANTRON_DEBUG=rec_h2_form=2)
    static const int big = gt_env_item("GANTRON_DEBUG", "rec_h2_form", 0) & 2;
    if (!big) return gt_rec_gemm_h2_lean(X, ldx, Wp, part, mtiles, N, K, splits, part_sstride, nz, x_zstride, part_zstride, xmax, xfloor, 0, stream);
    g_rec_h2x = {xmax, xfloor};
    return gt_rec_gemm_bf16_prio(X, ldx, Wp, part, mtiles, N, K, splits, part_sstride, nz, x_zstride, 0, part_zstride, 0, stream, 2);
}

extern "C" int gt_rec_gemm_bf16(const float* X, int64_t ldx, const void* Wp, float* part, int32_t mtiles, int32_t N,
                                int32_t K, int32_t splits, int64_t part_sstride, int32_t nz, int64_t x_zstride,
                                int64_t w_zstride, int64_t part_zstride, void* stream) {
    return gt_rec_gemm_bf16_prio(X, ldx, Wp, part, mtiles, N, K, splits, part_sstride, nz, x_zstride, w_zstride, part_zstride, 0,
                                 stream, 1);
}
extern "C" int gt_rec_gemm_x3(const float* X, int64_t ldx, const void* Wp, float* part, int32_t mtiles, int32_t N,
                              int32_t K, int32_t splits, int64_t part_sstride, int32_t nz, int64_t x_zstride,
                              int64_t w_zstride, int64_t part_zstride, void* stream) {
    return gt_rec_gemm_bf16_prio(X, ldx, Wp, part, mtiles, N, K, splits, part_sstride, nz, x_zstride, w_zstride, part_zstride, 0,
                                 stream, 3);
}

int gt_rec_gemm_bf16_prio(const float* X, int64_t ldx, const void* Wp, float* part, int32_t mtiles, int32_t N, int32_t K,
                          int32_t splits, int64_t part_sstride, int32_t nz, int64_t x_zstride, int64_t w_zstride,
                          int64_t part_zstride, int prio, void* stream, int planes) {
    GT_REQUIRE(X && Wp && part, GT_E_ARG, "gt_rec_gemm_bf16: NULL pointer");
    GT_REQUIRE(mtiles >= 1 && mtiles <= 4 && N > 0 && N % 16 == 0 && K >= 4 && splits >= 1 && nz >= 1 && planes >= 1 && planes <= 3,
               GT_E_ARG, "gt_rec_gemm_bf16: bad sizes mtiles=%d N=%d K=%d splits=%d planes=%d", mtiles, N, K, splits, planes);
    RecH2 h2 = {nullptr, nullptr, 0.f};
    if (planes == 2) {
        GT_REQUIRE(w_zstride == 0, GT_E_ARG, "gt_rec_gemm_h2: one weight image (its trailer holds the scale)");
        GT_REQUIRE(g_rec_h2x.xmax != nullptr || g_rec_h2x.xfloor > 0.f, GT_E_ARG, "gt_rec_gemm_h2: no bound for the activations (xmax word or xfloor > 0)");
        h2.wmax = h2_trailer(Wp, N, K); h2.xmax = (const unsigned long long*)g_rec_h2x.xmax; h2.xfloor = g_rec_h2x.xfloor;
        g_rec_h2x = {nullptr, 0.f};
    }
    const int KB = cdiv(K, 32);
    const int kvalid = cdiv(K, 16) * 16;        // columns the caller guarantees (zero padded beyond K), as gt_rec_gemm
    GT_REQUIRE(ldx >= kvalid && ldx % 4 == 0 && aligned16(X) && aligned16(Wp) && x_zstride % 4 == 0 && w_zstride % 8 == 0,
               GT_E_ALIGN, "gt_rec_gemm_bf16: X needs ldx >= 16*ceil(K/16) (=%d), ldx%%4==0 and 16-byte alignment", kvalid);
    GT_REQUIRE(part_sstride >= (int64_t)mtiles * 16 * N && aligned16(part) && part_sstride % 4 == 0 && part_zstride % 4 == 0,
               GT_E_ARG, "gt_rec_gemm_bf16: part slabs too small or misaligned");
    const int KBS = cdiv(KB, splits);
    const int max_nch = planes >= 2 ? RB_NCH_X3 : RB_NCH;
    GT_REQUIRE(cdiv(KB, KBS) == splits, GT_E_ARG, "gt_rec_gemm_bf16: splits=%d leaves empty slabs for K=%d", splits, K);
    GT_REQUIRE(KBS <= max_nch * RB_CB, GT_E_ARG, "gt_rec_gemm_bf16: K-slice of %d exceeds %d (use >= %d splits)", KBS * 32,
               max_nch * RB_CB * 32, cdiv(KB, max_nch * RB_CB));
    dim3 grid(cdiv(N, 64), splits, nz);
    hipStream_t st = (hipStream_t)stream;
    GtProfScope prof(GT_FAM_REC_GEMM, 2.0 * 16 * mtiles * (double)N * K * nz,
                     (double)nz * (2.0 * planes * N * K + 4.0 * (16.0 * mtiles * K + 16.0 * mtiles * N * splits)), st, true);
    // (planes == 2: 4 bytes per weight, as fp32)
    const uint4* wp = reinterpret_cast<const uint4*>(Wp);
    const int nch = cdiv(KBS, RB_CB);           // chunks of a slice: one instantiation each (all loads issued up front)
#define GT_RECB(MT, NCH, NP)                                                                                          \
    GT_LAUNCH_TIMED((rec_gemm_bf16_kernel<MT, NCH, NP>), grid, dim3(RB_THREADS), 0, st, prof.e0, prof.e1, X, (long)ldx, \
                          kvalid, wp, part, N, KB, KBS, (long)x_zstride, (long)(w_zstride / 8), (long)part_zstride,        \
                          (long)part_sstride, prio, h2)
#define GT_RECB_M(MT)                                  \
    if (planes == 3) switch (nch) {                    \
        case 1: GT_RECB(MT, 1, 3); break;              \
        case 2: GT_RECB(MT, 2, 3); break;              \
        case 3: GT_RECB(MT, 3, 3); break;              \
        default: GT_RECB(MT, 4, 3); break;             \
    } else if (planes == 2) switch (nch) {             \
        case 1: GT_RECB(MT, 1, 2); break;              \
        case 2: GT_RECB(MT, 2, 2); break;              \
        case 3: GT_RECB(MT, 3, 2); break;              \
        default: GT_RECB(MT, 4, 2); break;             \
    } else switch (nch) {                              \
        case 1: GT_RECB(MT, 1, 1); break;              \
        case 2: GT_RECB(MT, 2, 1); break;              \
        case 3: GT_RECB(MT, 3, 1); break;              \
        default: GT_RECB(MT, 4, 1); break;             \
    }
    switch (mtiles) {
        case 1: GT_RECB_M(1); break;
        case 2: GT_RECB_M(2); break;
        case 3: GT_RECB_M(3); break;
        default: GT_RECB_M(4); break;
    }
#undef GT_RECB_M
#undef GT_RECB
    GT_LAUNCH_CHECK();
    return 0;
}

// packing for gt_rec_cell_fwd_bf16 / _x3: W is the (4H x K) LSTM weight in PyTorch gate order
extern "C" int gt_pack_rec_cell_weight_bf16(const float* W, int64_t ldw, int32_t H, int32_t K, void* Wp, void* stream) {
    GT_REQUIRE(H > 0 && H % 16 == 0, GT_E_ARG, "gt_pack_rec_cell_weight_bf16: H=%d must be a multiple of 16", H);
    return pack_rec_planes("gt_pack_rec_cell_weight_bf16", W, ldw, 0, 4 * H, 4 * H, K, H, Wp, 1, stream);
}
extern "C" int gt_pack_rec_cell_weight_x3(const float* W, int64_t ldw, int32_t H, int32_t K, void* Wp, void* stream) {
    GT_REQUIRE(H > 0 && H % 16 == 0, GT_E_ARG, "gt_pack_rec_cell_weight_x3: H=%d must be a multiple of 16", H);
    return pack_rec_planes("gt_pack_rec_cell_weight_x3", W, ldw, 0, 4 * H, 4 * H, K, H, Wp, 3, stream);
}

extern "C" int gt_pack_rec_cell_weight_h2(const float* W, int64_t ldw, int32_t H, int32_t K, void* Wp, void* stream) {
    GT_REQUIRE(H > 0 && H % 16 == 0, GT_E_ARG, "gt_pack_rec_cell_weight_h2: H=%d must be a multiple of 16", H);
    return pack_rec_planes("gt_pack_rec_cell_weight_h2", W, ldw, 0, 4 * H, 4 * H, K, H, Wp, 2, stream);
}

// launch of the bf16-operand (planes = 1), two-fp16-plane (2) or three-bf16-plane fp32 (3) fused product + cell; arguments already
// validated by gt_rec_cell_fwd_z (recurrent.hip)
int gt_launch_rec_cell_bf16(const float* X, int64_t ldx, const void* Wp, float* part, int32_t Bp, int32_t N, int32_t K,
                            int32_t splits, int32_t ndir, int prio, const RecCellEpi& e, hipEvent_t e0, hipEvent_t e1,
                            void* stream, int planes) {
    const int KB = cdiv(K, 32);
    const int kvalid = cdiv(K, 16) * 16;
    const int KBS = cdiv(KB, splits);
    const int max_nch = planes >= 2 ? RB_NCH_X3 : RB_NCH;
    GT_REQUIRE(planes >= 1 && planes <= 3, GT_E_ARG, "gt_rec_cell_fwd_bf16: planes");
    RecH2 h2 = {nullptr, nullptr, 0.f};
    if (planes == 2) {
        GT_REQUIRE(ndir == 1, GT_E_ARG, "gt_rec_cell_fwd_h2: one direction (one weight image: its trailer holds the scale)");
        GT_REQUIRE(g_rec_h2x.xmax != nullptr || g_rec_h2x.xfloor > 0.f, GT_E_ARG, "gt_rec_cell_fwd_h2: no bound for the activations (xmax word or xfloor > 0)");
        h2.wmax = h2_trailer(Wp, N, K); h2.xmax = (const unsigned long long*)g_rec_h2x.xmax; h2.xfloor = g_rec_h2x.xfloor;
        g_rec_h2x = {nullptr, 0.f};
    }
    GT_REQUIRE(cdiv(KB, KBS) == splits, GT_E_ARG, "gt_rec_cell_fwd_bf16: splits=%d leaves empty slabs for K=%d", splits, K);
    GT_REQUIRE(KBS <= max_nch * RB_CB, GT_E_ARG, "gt_rec_cell_fwd_bf16: K-slice of %d exceeds %d (use >= %d splits)", KBS * 32,
               max_nch * RB_CB * 32, cdiv(KB, max_nch * RB_CB));
    GT_REQUIRE(e.dW % 8 == 0, GT_E_ALIGN, "gt_rec_cell_fwd_bf16: direction offset of the weights must be a multiple of 8");
    const int mt = Bp <= 64 ? Bp / 16 : 4, nzb = Bp <= 64 ? 1 : Bp / 64;
    dim3 grid(N / 64, splits, nzb * ndir);
    hipStream_t st = (hipStream_t)stream;
    const uint4* wp = reinterpret_cast<const uint4*>(Wp);
    const long ps = (long)Bp * N;
    const int nch = cdiv(KBS, RB_CB);
#define GT_RCB(MT, NCH, NP)                                                                                              \
    GT_LAUNCH_TIMED((rec_cell_bf16_kernel<MT, NCH, NP>), grid, dim3(RB_THREADS), 0, st, e0, e1, X, (long)ldx, kvalid, wp, \
                          part, N, KB, KBS, (long)(64 * ldx), (long)64 * N, ps, prio, e, h2)
#define GT_RCB_M(MT)                                   \
    if (planes == 3) switch (nch) {                    \
        case 1: GT_RCB(MT, 1, 3); break;               \
        case 2: GT_RCB(MT, 2, 3); break;               \
        case 3: GT_RCB(MT, 3, 3); break;               \
        default: GT_RCB(MT, 4, 3); break;              \
    } else if (planes == 2) switch (nch) {             \
        case 1: GT_RCB(MT, 1, 2); break;               \
        case 2: GT_RCB(MT, 2, 2); break;               \
        case 3: GT_RCB(MT, 3, 2); break;               \
        default: GT_RCB(MT, 4, 2); break;              \
    } else switch (nch) {                              \
        case 1: GT_RCB(MT, 1, 1); break;               \
        case 2: GT_RCB(MT, 2, 1); break;               \
        case 3: GT_RCB(MT, 3, 1); break;               \
        default: GT_RCB(MT, 4, 1); break;              \
    }
    switch (mt) {
        case 1: GT_RCB_M(1); break;
        case 2: GT_RCB_M(2); break;
        case 3: GT_RCB_M(3); break;
        default: GT_RCB_M(4); break;
    }
#undef GT_RCB_M
#undef GT_RCB
    GT_LAUNCH_CHECK();
    return 0;
}
