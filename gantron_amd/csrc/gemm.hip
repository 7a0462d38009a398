// fp32 MFMA GEMM with conv-aware operand addressing (gfx950).
//
//   C[m,n] (=|+=) alpha * sum_k A(m,k) * B(n,k) (+ bias[n])
//
// One kernel family serves every dense contraction of the training step that is not inside a
// recurrence: Linear layers, Conv1d forward / input-grad / weight-grad as implicit GEMMs over
// channels-last activations (im2col addressing is done by the tile loader, nothing is
// materialised), the deferred weight gradients of the decoder LSTMs (K = To*B), the STFT.
//
// Tiling: BMxBNx16 block tile, 256 threads = 4 waves in a 2x2 arrangement; each wave owns a
// (BM/2)x(BN/2) sub-tile made of 32x32 MFMA tiles (v_mfma_f32_32x32x2_f32: exact fp32, runs at
// the chip's fp32 peak).  Both operand tiles are staged k-major in LDS ([16][rows+4]) so that an
// MFMA operand fetch is one conflict-free ds_read_b32 per lane; the next tile's global loads are
// issued into registers before the current tile's MFMAs (register prefetch).
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace {

struct OpDev {
    const float* p;
    int mode, T, C, dil, pad, vec;
    long ld;
};

struct GemmDev {
    OpDev A, B;
    const unsigned short* Ap; const unsigned short* Bp;     // pre-split bf16 planes of A / B (pgemm_kernel), else NULL
    long a_ps, b_ps;                                         // elements between planes
    float* C;
    const float* bias;
    long ldc, a_bs, b_bs, c_bs;
    int M, N, K, accumulate, split_k, k_per_split;
    int bf16;               // 0: fp32 MFMA, 1: operands rounded to bf16, 2 / 3: fp32 product from 2 / 3 bf16 planes (xgemm_kernel),
                            // 4: from two SCALED fp16 planes (xgemm_kernel, FMT = 1)
    int amax_slot;          // mode 4: g_amax_ring[amax_slot], [amax_slot + 1] hold (epoch << 32 | bits of max |a|), ... max |b|
    const unsigned long long* amax_a; const unsigned long long* amax_b;     // ... unless the caller brought the words (GtOperand.amax)
    int dbg;                // GANTRON_DEBUG=pgemm=<bits> (diagnostics, pgemm_kernel): 1 = no DMA after the first tile, 2 = every tile
                            // re-reads the first one's addresses (cache-hot sources), 4 = no MFMAs
    float alpha;
};
static_assert(sizeof(GemmDev) <= 1024, "GemmDev is a by-value kernel argument: keep it well below the 4 KB kernel-argument segment");

constexpr int BK = 16;

// float4 along k for row r (modes KC / IM2COL_K).  (b,t) = position of row r (im2col only).
template <int MODE>
__device__ __forceinline__ float4 load_kvec(const OpDev& o, const float* base, int r, int R, int k,
                                            int K, int b, int t, int tap, int c) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= R || k >= K) return v;
    const float* p;
    if (MODE == GT_OP_KC) {
        p = base + (long)r * o.ld + k;
    } else {
        const int tt = t + tap * o.dil - o.pad;
        if (tt < 0 || tt >= o.T) return v;
        p = base + ((long)b * o.T + tt) * o.C + c;
    }
    if (o.vec && k + 3 < K) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        v.x = p[0];
        if (k + 1 < K) v.y = p[1];
        if (k + 2 < K) v.z = p[2];
        if (k + 3 < K) v.w = p[3];
    }
    return v;
}

// float4 along r at reduction index k (modes RC / IM2COL_R).  (tap,c) = decomposition of row r.
template <int MODE>
__device__ __forceinline__ float4 load_rvec(const OpDev& o, const float* base, int r, int R, int k,
                                            int K, int tap, int c, int b, int t) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r >= R || k >= K) return v;
    const float* p;
    if (MODE == GT_OP_RC) {
        p = base + (long)k * o.ld + r;
    } else {
        const int tt = t + tap * o.dil - o.pad;
        if (tt < 0 || tt >= o.T) return v;
        p = base + ((long)b * o.T + tt) * o.C + c;
    }
    if (o.vec && r + 3 < R) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        v.x = p[0];
        if (r + 1 < R) v.y = p[1];
        if (r + 2 < R) v.z = p[2];
        if (r + 3 < R) v.w = p[3];
    }
    return v;
}

template <int MODE>
constexpr bool is_ktype() { return MODE == GT_OP_KC || MODE == GT_OP_IM2COL_K; }

// Per-thread loader state for one operand tile of BR rows.
template <int MODE, int BR>
struct TileLoader {
    static constexpr int NIT = BR / 64;
    static constexpr int LD = BR + 4;
    int p0[NIT], p1[NIT];   // (b,t) for IM2COL_K, (tap,c) for IM2COL_R: decomposition of the ROW index (fixed)
    int q0[NIT], q1[NIT];   // decomposition of the running REDUCTION index k of the next load: (tap,c) for
                            // IM2COL_K, (b,t) for IM2COL_R; advanced by BK per load instead of dividing every time
    float4 reg[NIT];

    // k-contiguous operands (KC / IM2COL_K): thread = (row tid >> 2 (+ 64 i), k-quad tid & 3), so that 4 consecutive
    // lanes read the 64 contiguous bytes (one 16-wide k-slab) of a row and a wave instruction touches 16 row segments
    // instead of 64 scattered 16-byte pieces (the earlier row = lane mapping: every lane in a different cache line).  The
    // k-major LDS store of that mapping has 2-way bank conflicts between k-quads 0/2 and 1/3, which ds_write_b32 hides.
    __device__ __forceinline__ void init(const OpDev& o, int r0, int kbeg, int tid) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            p0[i] = 0; p1[i] = 0; q0[i] = 0; q1[i] = 0;
            if (MODE == GT_OP_IM2COL_K) {
                const int k = kbeg + 4 * (tid & 3);
                q0[i] = k / o.C; q1[i] = k - q0[i] * o.C;
            } else if (MODE == GT_OP_IM2COL_R) {
                const int k = kbeg + (tid + 256 * i) / (BR / 4);
                q0[i] = k / o.T; q1[i] = k - q0[i] * o.T;
            }
            if (MODE == GT_OP_IM2COL_K) {
                const int r = r0 + (tid >> 2) + 64 * i;
                p0[i] = r / o.T; p1[i] = r - p0[i] * o.T;
            } else if (MODE == GT_OP_IM2COL_R) {
                const int f = tid + 256 * i;
                const int r = r0 + 4 * (f % (BR / 4));
                p0[i] = r / o.C; p1[i] = r - p0[i] * o.C;
            }
        }
    }
    __device__ __forceinline__ void load(const OpDev& o, const float* base, int r0, int R, int k0,
                                         int K, int tid) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (is_ktype<MODE>()) {
                const int r = r0 + (tid >> 2) + 64 * i;
                const int k = k0 + 4 * (tid & 3);
                reg[i] = load_kvec<MODE>(o, base, r, R, k, K, p0[i], p1[i], q0[i], q1[i]);
                if (MODE == GT_OP_IM2COL_K) {
                    q1[i] += BK;
                    while (q1[i] >= o.C) { q1[i] -= o.C; ++q0[i]; }
                }
            } else {
                const int f = tid + 256 * i;
                const int r = r0 + 4 * (f % (BR / 4));
                const int k = k0 + f / (BR / 4);
                reg[i] = load_rvec<MODE>(o, base, r, R, k, K, p0[i], p1[i], q0[i], q1[i]);
                if (MODE == GT_OP_IM2COL_R) {
                    q1[i] += BK;
                    while (q1[i] >= o.T) { q1[i] -= o.T; ++q0[i]; }
                }
            }
        }
    }
    __device__ __forceinline__ void store(float* S, int tid) const {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (is_ktype<MODE>()) {
                const int rl = (tid >> 2) + 64 * i;
                const int kq = tid & 3;
                S[(4 * kq + 0) * LD + rl] = reg[i].x;
                S[(4 * kq + 1) * LD + rl] = reg[i].y;
                S[(4 * kq + 2) * LD + rl] = reg[i].z;
                S[(4 * kq + 3) * LD + rl] = reg[i].w;
            } else {
                const int f = tid + 256 * i;
                const int rq = f % (BR / 4), kl = f / (BR / 4);
                *reinterpret_cast<float4*>(&S[kl * LD + 4 * rq]) = reg[i];
            }
        }
    }
};

template <int BM, int BN, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void sgemm_kernel(GemmDev g) {
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) float smem[BK * LDA + BK * LDB];
    float* As = smem;
    float* Bs = smem + BK * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int batch = blockIdx.z / g.split_k, kz = blockIdx.z - batch * g.split_k;
    const int kbeg = kz * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const float* Ab = g.A.p + (long)batch * g.a_bs;
    const float* Bb = g.B.p + (long)batch * g.b_bs;
    float* Cb = g.C + (long)batch * g.c_bs;

    TileLoader<AMODE, BM> la;
    TileLoader<BMODE, BN> lb;
    la.init(g.A, m0, kbeg, tid);
    lb.init(g.B, n0, kbeg, tid);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

    if (kbeg < kend) {
        la.load(g.A, Ab, m0, g.M, kbeg, kend, tid);
        lb.load(g.B, Bb, n0, g.N, kbeg, kend, tid);
    }
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        la.store(As, tid);
        lb.store(Bs, tid);
        __syncthreads();
        if (k0 + BK < kend) {
            la.load(g.A, Ab, m0, g.M, k0 + BK, kend, tid);
            lb.load(g.B, Bb, n0, g.N, k0 + BK, kend, tid);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int krow = 2 * kk + (lane >> 5);
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[krow * LDA + wm0 + 32 * i + (lane & 31)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[krow * LDB + wn0 + 32 * j + (lane & 31)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: C/D map of the 32x32 MFMA: n = lane&31, m = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + 32 * j + (lane & 31);
            if (n >= g.N) continue;
            const float bias = (g.bias != nullptr && kz == 0) ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float* c = Cb + (long)m * g.ldc + n;
                const float v = g.alpha * acc[i][j][r] + bias;
                if (g.split_k > 1) atomicAdd(c, v);
                else if (g.accumulate) *c += v;
                else *c = v;
            }
        }
}

// =====================================================================================================================
// bf16-operand variant (GtGemm.bf16, hparams.fp16_run): same contract and operand modes, but the fp32 operands are
// rounded to bf16 (nearest even) on their way into LDS and multiplied with v_mfma_f32_32x32x16_bf16; accumulation, C
// and the epilogue stay fp32.  The MFMA work of a tile drops to 1/16, so the kernel is organised around the operand
// traffic instead: 64-wide k-tiles (one barrier pair per 64 k instead of per 16), ROW-major LDS tiles [row][64 k + 8]
// bf16 (144-byte rows: the 16-byte fragments of 32 consecutive rows fall on distinct bank groups, a lane's fragment
// A[row = lane & 31][k = 16 s + 8 (lane >> 5) + 0..7] is one ds_read_b128), k-contiguous operands loaded as 256-byte
// row segments per 16 threads, reduction-major operands as 4-row x KPT-k register blocks that are transposed in
// registers and written as one 16-byte (8-byte for 64-row tiles) store per row.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
constexpr int BKH = 64;
constexpr int LDH = BKH + 8;
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const bf16x2_t p = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, p);
}

template <int MODE, int BR>
struct TileLoaderH {
    static constexpr int NK = BR / 16;            // k-type: float4 per thread (rows f/16, k-quad f%16)
    static constexpr int RQ = BR / 4;             // r-type: row quads per tile
    static constexpr int KPT = BKH / (256 / RQ);  // r-type: consecutive k per thread (8 for 128 rows, 4 for 64)
    static constexpr int NV = is_ktype<MODE>() ? NK : KPT;
    float4 reg[NV];
    int p0[is_ktype<MODE>() ? NK : 1], p1[is_ktype<MODE>() ? NK : 1];   // fixed decomposition of the row index
    int q0, q1;                                                        // running decomposition of the first k

    __device__ __forceinline__ void init(const OpDev& o, int r0, int kbeg, int tid) {
        q0 = q1 = 0;
#pragma unroll
        for (int i = 0; i < (is_ktype<MODE>() ? NK : 1); ++i) { p0[i] = 0; p1[i] = 0; }
        if (MODE == GT_OP_IM2COL_K) {
            const int k = kbeg + 4 * (tid & 15);
            q0 = k / o.C; q1 = k - q0 * o.C;                      // (tap, c) of this thread's k-quad
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const int r = r0 + (tid + 256 * i) / 16;
                p0[i] = r / o.T; p1[i] = r - p0[i] * o.T;         // (b, t) of the row
            }
        } else if (MODE == GT_OP_IM2COL_R) {
            const int r = r0 + 4 * (tid % RQ);
            p0[0] = r / o.C; p1[0] = r - p0[0] * o.C;             // (tap, c) of the row quad
            const int k = kbeg + KPT * (tid / RQ);
            q0 = k / o.T; q1 = k - q0 * o.T;                      // (b, t) of the first k
        }
    }
    __device__ __forceinline__ void load(const OpDev& o, const float* base, int r0, int R, int k0, int K, int tid) {
        if (is_ktype<MODE>()) {
            const int k = k0 + 4 * (tid & 15);
#pragma unroll
            for (int i = 0; i < NK; ++i)
                reg[i] = load_kvec<MODE>(o, base, r0 + (tid + 256 * i) / 16, R, k, K, p0[i], p1[i], q0, q1);
            if (MODE == GT_OP_IM2COL_K) {
                q1 += BKH;
                while (q1 >= o.C) { q1 -= o.C; ++q0; }
            }
        } else {
            const int r = r0 + 4 * (tid % RQ);
            const int k = k0 + KPT * (tid / RQ);
            int b = q0, t = q1;
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                reg[j] = load_rvec<MODE>(o, base, r, R, k + j, K, p0[0], p1[0], b, t);
                if (MODE == GT_OP_IM2COL_R) { if (++t >= o.T) { t = 0; ++b; } }
            }
            if (MODE == GT_OP_IM2COL_R) {
                q1 += BKH;
                while (q1 >= o.T) { q1 -= o.T; ++q0; }
            }
        }
    }
    __device__ __forceinline__ void store(unsigned short* S, int tid) const {
        if (is_ktype<MODE>()) {
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const int f = tid + 256 * i;
                *reinterpret_cast<uint2*>(&S[(f / 16) * LDH + 4 * (f & 15)]) =
                    make_uint2(pack_bf16(reg[i].x, reg[i].y), pack_bf16(reg[i].z, reg[i].w));
            }
        } else {
            const int rq = tid % RQ, kc = KPT * (tid / RQ);
            // reg[j] = rows 4rq..4rq+3 at k = kc + j: per row KPT consecutive k -> one 8 / 16 byte store
            unsigned short* d = &S[(4 * rq) * LDH + kc];
            if (KPT == 8) {
                *reinterpret_cast<uint4*>(d) = make_uint4(pack_bf16(reg[0].x, reg[1].x), pack_bf16(reg[2].x, reg[3].x),
                                                          pack_bf16(reg[4 % NV].x, reg[5 % NV].x), pack_bf16(reg[6 % NV].x, reg[7 % NV].x));
                *reinterpret_cast<uint4*>(d + LDH) = make_uint4(pack_bf16(reg[0].y, reg[1].y), pack_bf16(reg[2].y, reg[3].y),
                                                                pack_bf16(reg[4 % NV].y, reg[5 % NV].y), pack_bf16(reg[6 % NV].y, reg[7 % NV].y));
                *reinterpret_cast<uint4*>(d + 2 * LDH) = make_uint4(pack_bf16(reg[0].z, reg[1].z), pack_bf16(reg[2].z, reg[3].z),
                                                                    pack_bf16(reg[4 % NV].z, reg[5 % NV].z), pack_bf16(reg[6 % NV].z, reg[7 % NV].z));
                *reinterpret_cast<uint4*>(d + 3 * LDH) = make_uint4(pack_bf16(reg[0].w, reg[1].w), pack_bf16(reg[2].w, reg[3].w),
                                                                    pack_bf16(reg[4 % NV].w, reg[5 % NV].w), pack_bf16(reg[6 % NV].w, reg[7 % NV].w));
            } else {
                *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf16(reg[0].x, reg[1].x), pack_bf16(reg[2].x, reg[3].x));
                *reinterpret_cast<uint2*>(d + LDH) = make_uint2(pack_bf16(reg[0].y, reg[1].y), pack_bf16(reg[2].y, reg[3].y));
                *reinterpret_cast<uint2*>(d + 2 * LDH) = make_uint2(pack_bf16(reg[0].z, reg[1].z), pack_bf16(reg[2].z, reg[3].z));
                *reinterpret_cast<uint2*>(d + 3 * LDH) = make_uint2(pack_bf16(reg[0].w, reg[1].w), pack_bf16(reg[2].w, reg[3].w));
            }
        }
    }
};

template <int BM, int BN, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void hgemm_kernel(GemmDev g) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) unsigned short smem[(BM + BN) * LDH];
    unsigned short* Ah = smem;
    unsigned short* Bh = smem + BM * LDH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int batch = blockIdx.z / g.split_k, kz = blockIdx.z - batch * g.split_k;
    const int kbeg = kz * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const float* Ab = g.A.p + (long)batch * g.a_bs;
    const float* Bb = g.B.p + (long)batch * g.b_bs;
    float* Cb = g.C + (long)batch * g.c_bs;

    TileLoaderH<AMODE, BM> la;
    TileLoaderH<BMODE, BN> lb;
    la.init(g.A, m0, kbeg, tid);
    lb.init(g.B, n0, kbeg, tid);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    if (kbeg < kend) {
        la.load(g.A, Ab, m0, g.M, kbeg, kend, tid);
        lb.load(g.B, Bb, n0, g.N, kbeg, kend, tid);
    }
    const unsigned short* ar = Ah + (wm0 + (lane & 31)) * LDH + 8 * (lane >> 5);
    const unsigned short* br = Bh + (wn0 + (lane & 31)) * LDH + 8 * (lane >> 5);
    for (int k0 = kbeg; k0 < kend; k0 += BKH) {
        la.store(Ah, tid);
        lb.store(Bh, tid);
        __syncthreads();
        if (k0 + BKH < kend) {
            la.load(g.A, Ab, m0, g.M, k0 + BKH, kend, tid);
            lb.load(g.B, Bb, n0, g.N, k0 + BKH, kend, tid);
        }
#pragma unroll
        for (int s = 0; s < BKH / 16; ++s) {
            bf16x8_t a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ar + 32 * i * LDH + 16 * s));
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(br + 32 * j * LDH + 16 * s));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: as sgemm_kernel (the C/D map of the 32x32 MFMA does not depend on the operand type)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + 32 * j + (lane & 31);
            if (n >= g.N) continue;
            const float bias = (g.bias != nullptr && kz == 0) ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float* c = Cb + (long)m * g.ldc + n;
                const float v = g.alpha * acc[i][j][r] + bias;
                if (g.split_k > 1) atomicAdd(c, v);
                else if (g.accumulate) *c += v;
                else *c = v;
            }
        }
}


// =====================================================================================================================
// fp32 product on the bf16 matrix pipe ("split" mode, gt_sgemm_set_fp32_split).  gfx950 multiplies bf16 sixteen times
// faster than fp32 (v_mfma_f32_32x32x16_bf16: 32768 flop in 32 cycles, v_mfma_f32_32x32x2_f32: 4096 flop in 64), so an
// fp32 operand is staged as NP bf16 PLANES, x = x0 + x1 (+ x2) with x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)
// (each residual is exact in fp32; three planes hold all 24 significand bits), and a . b is accumulated in fp32 from the
// plane products of order < NP:  NP = 3: a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)  -- six MFMAs, every product
// exact, dropped terms <= 2^-24 |a||b| (the rounding unit of the fp32 MFMA itself);  NP = 2: a0 b0 + a0 b1 + a1 b0 --
// three MFMAs, dropped terms <= 2^-16 |a||b|.  Same contract, operand modes and epilogue as sgemm_kernel; 32-wide k-tiles;
// operands are split in registers on their way into LDS.  Inf / NaN inputs give NaN (inf - inf in the residual).
//
// LDS images (per plane): a k-contiguous operand (KC / IM2COL_K) is stored row-major [row][32 k + 8] (80-byte rows: the
// 16-byte fragments of 16 consecutive rows tile all 64 banks) and its MFMA fragment is one ds_read_b128; a
// reduction-major operand (RC / IM2COL_R) is stored AS IT ARRIVES, k-major [k][rows + 32] (row stride == 64 bytes mod
// 256), every lane writing the 4 consecutive rows of its float4 as one 8-byte store, and its fragment is gathered by two
// ds_read_b64_tr_b16 (the hardware transposes 4 k x 16 rows per 16 lanes): no register transpose, no bank conflicts on
// either side.
constexpr int BKX = 32;
constexpr int LDX = BKX + 8;
typedef short v4i16_t __attribute__((ext_vector_type(4)));

// NV float4 -> their NP bf16 planes, packed in pairs (x, y) -> lo, (z, w) -> hi with the first element in the low half.
// Per plane and pair: v_cvt_pk_bf16_f32, unpack both halves (shift, mask), two subtractions; the pairs advance plane by
// plane, side by side, so that no instruction waits for the one in front of it.  (The empty asm statements keep a pair
// packed -- otherwise every element is converted a second time for u << 16 -- and keep the subtractions apart:
// v_pk_add_f32 is slow to issue beside MFMAs.)
template <int NP, int NV>
__device__ __forceinline__ void split_all(const float4 (&reg)[NV], unsigned (&lo)[NV][NP], unsigned (&hi)[NV][NP]) {
    float v[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i][0] = reg[i].x; v[i][1] = reg[i].y; v[i][2] = reg[i].z; v[i][3] = reg[i].w; }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        unsigned u[NV][2];
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bf16x2_t b = {(__bf16)v[i][2 * h], (__bf16)v[i][2 * h + 1]};
                u[i][h] = __builtin_bit_cast(unsigned, b);
                asm("" : "+v"(u[i][h]));
            }
#pragma unroll
        for (int i = 0; i < NV; ++i) { lo[i][p] = u[i][0]; hi[i][p] = u[i][1]; }
        if (p + 1 < NP) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v[i][2 * h] -= __uint_as_float(u[i][h] << 16);
                    asm("" : "+v"(v[i][2 * h]));
                    v[i][2 * h + 1] -= __uint_as_float(u[i][h] & 0xffff0000u);
                    asm("" : "+v"(v[i][2 * h + 1]));
                }
        }
    }
}

// ---- two SCALED fp16 planes ("h2", gt_sgemm_set_fp32_split(4)) ------------------------------------------------------------
// x s = h0 + h1 + r with h0 = f16(x s), h1 = f16(x s - h0) (both nearest even, the residual x s - h0 is exact in fp32):
// |r| <= 2^-22 |x s|, i.e. 22 of the operand's 24 significand bits travel, and a . b = (a0 b0 + a0 b1 + a1 b0) / (sa sb) runs
// as THREE v_mfma_f32_32x32x16_f16 with fp32 accumulation (every product exact; a1 b1 <= 2^-22 |a||b| is dropped with the
// r terms).  fp16 has 5 exponent bits, so each operand is multiplied by a power of two s = 2^k that puts the operand
// TENSOR's largest magnitude into [2^14, 2^15) (gt_sgemm finds it with one pass over the operand in front of the product:
// absmax2_kernel): elements down to 2^-18 of that maximum keep all 22 bits (h1 >= 2^-14, fp16's smallest normal), smaller
// ones lose them gradually -- absolute error <= 2^-25 / s = 2^-39 of the tensor's maximum -- and nothing overflows.  The
// scaling is exact both ways (powers of two; the epilogue's v_ldexp_f32 takes sa sb out again).  Per PAIR of values the split
// is five vector instructions, scale included: v_fma_mixlo_f16 / v_fma_mixhi_f16 (h0 = f16(s x), one rounding),
// 2 x v_fma_mix_f32 (s x - h0, reading the fp16 halves in place), v_cvt_pk_f16_f32 -- against eleven for three bf16 planes;
// two plane stores instead of three, three MFMAs instead of six.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int NV>
__device__ __forceinline__ void split_h2(const float4 (&reg)[NV], float s, unsigned (&lo)[NV][2], unsigned (&hi)[NV][2]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float v[4] = {reg[i].x, reg[i].y, reg[i].z, reg[i].w};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned p0;
            float ra, rb;
            // (VALU only, operands in registers: nothing for the compiler to count or pad, cdna_hip_programming.md 5.7)
            asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(p0) : "s"(s), "v"(v[2 * h]));
            asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(p0) : "s"(s), "v"(v[2 * h + 1]));
            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ra) : "s"(s), "v"(v[2 * h]), "v"(p0));
            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "s"(s), "v"(v[2 * h + 1]), "v"(p0));
            const f16x2_t p1 = __builtin_convertvector((f32x2_t){ra, rb}, f16x2_t);
            (h == 0 ? lo : hi)[i][0] = p0;
            (h == 0 ? lo : hi)[i][1] = __builtin_bit_cast(unsigned, p1);
        }
    }
}

// operand scale of mode 4 from the bits of max |x|: s = 2^k with max |x| s in [2^14, 2^15); k is clamped to fp32's exponent
// range (an all-zero or denormal tensor gets 2^127, an Inf / NaN one any power of two: the product is NaN either way)
__device__ __forceinline__ int h2_scale_exp(unsigned maxbits) {
    const int e = (int)((maxbits >> 23) & 0xffu);
    return min(127, max(-126, 141 - e));
}
// (one 128-byte line per call: words of different calls -- different streams -- never share a cache line)
constexpr int AMAX_SLOTS = 8192, AMAX_STRIDE = 16;
__device__ __attribute__((aligned(128))) unsigned long long g_amax_ring[AMAX_SLOTS * AMAX_STRIDE];

// FAST loads: the tile lies inside the matrix, the k-tile inside the reduction range and the operand is 16-byte
// addressable, so nothing is checked per element; conv taps outside [0, T) read a clamped address and select zero
// (the zeroing is a bitwise AND: with a select the compiler moves the load under a branch, and a branch around a load
// hides the number of loads in flight from its s_waitcnt insertion, which then waits for all of them)
__device__ __forceinline__ float4 mask4(const float4& v, bool ok) {
    const unsigned m = ok ? 0xffffffffu : 0u;
    return make_float4(__uint_as_float(__float_as_uint(v.x) & m), __uint_as_float(__float_as_uint(v.y) & m),
                       __uint_as_float(__float_as_uint(v.z) & m), __uint_as_float(__float_as_uint(v.w) & m));
}
template <int MODE>
__device__ __forceinline__ float4 fast_kvec(const OpDev& o, const float* base, int r, int k, int b, int t, int tap, int c) {
    if (MODE == GT_OP_KC) return *reinterpret_cast<const float4*>(base + (long)r * o.ld + k);
    const int tt = t + tap * o.dil - o.pad;
    const bool ok = tt >= 0 && tt < o.T;
    return mask4(*reinterpret_cast<const float4*>(base + ((long)b * o.T + (ok ? tt : t)) * o.C + c), ok);
}
template <int MODE>
__device__ __forceinline__ float4 fast_rvec(const OpDev& o, const float* base, int r, int k, int tap, int c, int b, int t) {
    if (MODE == GT_OP_RC) return *reinterpret_cast<const float4*>(base + (long)k * o.ld + r);
    const int tt = t + tap * o.dil - o.pad;
    const bool ok = tt >= 0 && tt < o.T;
    return mask4(*reinterpret_cast<const float4*>(base + ((long)b * o.T + (ok ? tt : t)) * o.C + c), ok);
}

template <int MODE, int BR, int NP, int FMT = 0>
struct TileLoaderX {
    static constexpr bool KT = is_ktype<MODE>();
    static constexpr int NK = BR / 32;            // k-type: float4 per thread (row f / 8, k-quad f % 8)
    static constexpr int RQ = BR / 4;             // r-type: row quads per tile
    static constexpr int KPT = BKX / (256 / RQ);  // r-type: consecutive k per thread (4 for 128 rows, 2 for 64)
    static constexpr int NV = KT ? NK : KPT;
    static constexpr int SR = BR + 32;            // r-type: bf16 elements per k-row of the LDS image
    static constexpr int PLANE = KT ? BR * LDX : BKX * SR;        // bf16 elements per plane
    int p0[KT ? NK : 1], p1[KT ? NK : 1];         // fixed decomposition of the row index
    int q0, q1;                                   // running decomposition of the first k
    long foff;                                    // KC / RC: element offset of this thread's first load of the next k-tile

    __device__ __forceinline__ void init(const OpDev& o, int r0, int kbeg, int tid) {
        q0 = q1 = 0;
        foff = MODE == GT_OP_KC ? (long)(r0 + tid / 8) * o.ld + kbeg + 4 * (tid & 7)
                                : (long)(kbeg + KPT * (tid / RQ)) * o.ld + r0 + 4 * (tid % RQ);
#pragma unroll
        for (int i = 0; i < (KT ? NK : 1); ++i) { p0[i] = 0; p1[i] = 0; }
        if (MODE == GT_OP_IM2COL_K) {
            const int k = kbeg + 4 * (tid & 7);
            q0 = k / o.C; q1 = k - q0 * o.C;                      // (tap, c) of this thread's k-quad
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const int r = r0 + (tid + 256 * i) / 8;
                p0[i] = r / o.T; p1[i] = r - p0[i] * o.T;         // (b, t) of the row
            }
        } else if (MODE == GT_OP_IM2COL_R) {
            const int r = r0 + 4 * (tid % RQ);
            p0[0] = r / o.C; p1[0] = r - p0[0] * o.C;             // (tap, c) of the row quad
            const int k = kbeg + KPT * (tid / RQ);
            q0 = k / o.T; q1 = k - q0 * o.T;                      // (b, t) of the first k
        }
    }
    // (the k-tile's values go to / come from a register set of the CALLER: xgemm_kernel keeps PF of them in flight)
    template <bool FAST>
    __device__ __forceinline__ void load(float4 (&reg)[NV], const OpDev& o, const float* base, int r0, int R, int k0, int K, int tid) {
        if (KT) {
            const int k = k0 + 4 * (tid & 7);
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                const int r = r0 + (tid + 256 * i) / 8;
                if (FAST && MODE == GT_OP_KC) reg[i] = *reinterpret_cast<const float4*>(base + foff + 32 * i * o.ld);
                else reg[i] = FAST ? fast_kvec<MODE>(o, base, r, k, p0[i], p1[i], q0, q1)
                                   : load_kvec<MODE>(o, base, r, R, k, K, p0[i], p1[i], q0, q1);
            }
            foff += BKX;
            if (MODE == GT_OP_IM2COL_K) {
                q1 += BKX;
                while (q1 >= o.C) { q1 -= o.C; ++q0; }
            }
        } else {
            const int r = r0 + 4 * (tid % RQ);
            const int k = k0 + KPT * (tid / RQ);
            int b = q0, t = q1;
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                if (FAST && MODE == GT_OP_RC) reg[j] = *reinterpret_cast<const float4*>(base + foff + j * o.ld);
                else reg[j] = FAST ? fast_rvec<MODE>(o, base, r, k + j, p0[0], p1[0], b, t)
                                   : load_rvec<MODE>(o, base, r, R, k + j, K, p0[0], p1[0], b, t);
                if (MODE == GT_OP_IM2COL_R) { ++t; const int w = t >= o.T; t = w ? 0 : t; b += w; }
            }
            foff += BKX * o.ld;
            if (MODE == GT_OP_IM2COL_R) {
                q1 += BKX;
                while (q1 >= o.T) { q1 -= o.T; ++q0; }
            }
        }
    }
    __device__ __forceinline__ void store(const float4 (&reg)[NV], unsigned short* S, int tid, float scale = 1.f) const {
        unsigned lo[NV][NP], hi[NV][NP];
        if constexpr (FMT == 1) split_h2<NV>(reg, scale, lo, hi);
        else split_all<NP, NV>(reg, lo, hi);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            unsigned short* d;
            if (KT) {
                const int f = tid + 256 * i;
                d = &S[(f / 8) * LDX + 4 * (f & 7)];                       // [row][k .. k+3]
            } else {
                d = &S[(KPT * (tid / RQ) + i) * SR + 4 * (tid % RQ)];      // [k][row .. row+3]
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(d + p * PLANE) = make_uint2(lo[i][p], hi[i][p]);
        }
    }
    // MFMA operand of the 32x32x16 product for rows [w0 + 32 i, +32) of the tile, k-step s, plane p: lane l holds
    // [row l & 31][k = 16 s + 8 (l >> 5) .. + 7]
    static __device__ __forceinline__ int frag_base(int w0, int lane) {
        if (KT) return (w0 + (lane & 31)) * LDX + 8 * (lane >> 5);
        // transposed read: lane 4q + r of a 16-lane group addresses k-row q, columns 4r .. 4r+3 of the group's 16 rows
        return (8 * (lane >> 5) + ((lane & 15) >> 2)) * SR + w0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    }
    static __device__ __forceinline__ bf16x8_t frag(const unsigned short* S, int base, int p, int i, int s) {
        if (KT) return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(S + base + p * PLANE + 32 * i * LDX + 16 * s));
        typedef __attribute__((address_space(3))) v4i16_t* lds_v4;
        const unsigned short* q = S + base + p * PLANE + 32 * i + 16 * s * SR;
        const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q));
        const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q + 4 * SR));
        return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
};

// PF = k-tiles of operand loads in flight per workgroup (register sets of the kernel; 1 everywhere: see the note below the kernel)
template <int BM, int BN, int AMODE, int BMODE, int NP, int FMT = 0, int PF = 1>
__global__ __launch_bounds__(256, (BM * BN > 128 * 128 ? 2 : 1)) void xgemm_kernel(GemmDev g) {
    static_assert(FMT == 0 || NP == 2, "the fp16 form has two planes");
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    typedef TileLoaderX<AMODE, BM, NP, FMT> LA;
    typedef TileLoaderX<BMODE, BN, NP, FMT> LB;
    __shared__ __attribute__((aligned(16))) unsigned short smem[NP * (LA::PLANE + LB::PLANE)];
    unsigned short* Ax = smem;
    unsigned short* Bx = smem + NP * LA::PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int batch = blockIdx.z / g.split_k, kz = blockIdx.z - batch * g.split_k;
    const int kbeg = kz * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const float* Ab = g.A.p + (long)batch * g.a_bs;
    const float* Bb = g.B.p + (long)batch * g.b_bs;
    float* Cb = g.C + (long)batch * g.c_bs;
    // every lane of the workgroup takes the same branch: unchecked vector loads while the tile is interior
    const bool inside = m0 + BM <= g.M && n0 + BN <= g.N && g.A.vec && g.B.vec;

    LA la;
    LB lb;
    la.init(g.A, m0, kbeg, tid);
    lb.init(g.B, n0, kbeg, tid);
    // mode 4: the operands' scales (wave-uniform: scalar loads of what absmax2_kernel left in front of this launch)
    float sa = 1.f, sb = 1.f;
    int kab = 0;
    if constexpr (FMT == 1) {
        // (agent-scope loads: served by L2, past this CU's vector / scalar caches)
        const unsigned long long wa = __hip_atomic_load(g.amax_a != nullptr ? g.amax_a : &g_amax_ring[g.amax_slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long wb = __hip_atomic_load(g.amax_b != nullptr ? g.amax_b : &g_amax_ring[g.amax_slot + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ka = h2_scale_exp(__builtin_amdgcn_readfirstlane((unsigned)wa));
        const int kb = h2_scale_exp(__builtin_amdgcn_readfirstlane((unsigned)wb));
        sa = __uint_as_float((unsigned)(127 + ka) << 23);
        sb = __uint_as_float((unsigned)(127 + kb) << 23);
        kab = ka + kb;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    float4 ra[PF][LA::NV], rb[PF][LB::NV];
#define XG_LOAD(U, KT_)                                                                     \
    do {                                                                                    \
        if (inside && (KT_) + BKX <= kend) {                                                \
            la.template load<true>(ra[U], g.A, Ab, m0, g.M, (KT_), kend, tid);              \
            lb.template load<true>(rb[U], g.B, Bb, n0, g.N, (KT_), kend, tid);              \
        } else {                                                                            \
            la.template load<false>(ra[U], g.A, Ab, m0, g.M, (KT_), kend, tid);             \
            lb.template load<false>(rb[U], g.B, Bb, n0, g.N, (KT_), kend, tid);             \
        }                                                                                   \
    } while (0)
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (kbeg + u * BKX < kend) XG_LOAD(u, kbeg + u * BKX);
    const int abase = LA::frag_base(wm0, lane), bbase = LB::frag_base(wn0, lane);
    for (int k0 = kbeg; k0 < kend; k0 += PF * BKX) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int kt = k0 + u * BKX;
            if (kt >= kend) break;               // (uniform)
            la.store(ra[u], Ax, tid, sa);
            lb.store(rb[u], Bx, tid, sb);
            __syncthreads();
            if (kt + PF * BKX < kend) XG_LOAD(u, kt + PF * BKX);
#pragma unroll
            for (int s = 0; s < BKX / 16; ++s) {
                bf16x8_t a[NP][TM], b[NP][TN];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[p][i] = LA::frag(Ax, abase, p, i, s);
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[p][j] = LB::frag(Bx, bbase, p, j, s);
                }
                // smallest terms first; between two MFMAs on the same accumulator lie the other TM * TN - 1 tiles
#pragma unroll
                for (int ord = NP - 1; ord >= 0; --ord)
#pragma unroll
                    for (int pa = 0; pa <= ord; ++pa)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) {
                                if constexpr (FMT == 1)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a[pa][i]),
                                                                                        __builtin_bit_cast(f16x8_t, b[ord - pa][j]), acc[i][j], 0, 0, 0);
                                else
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][i], b[ord - pa][j], acc[i][j], 0, 0, 0);
                            }
            }
            __syncthreads();
        }
    }
#undef XG_LOAD
    // epilogue: as sgemm_kernel (mode 4: the operand scales leave first, exactly)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + 32 * j + (lane & 31);
            if (n >= g.N) continue;
            const float bias = (g.bias != nullptr && kz == 0) ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float* c = Cb + (long)m * g.ldc + n;
                const float sum = FMT == 1 ? ldexpf(acc[i][j][r], -kab) : acc[i][j][r];
                const float v = g.alpha * sum + bias;
                if (g.split_k > 1) atomicAdd(c, v);
                else if (g.accumulate) *c += v;
                else *c = v;
            }
        }
}

// (Measured and not adopted, round 4 -- numbers in DESIGN.md section 3: (a) PF = 2 / 3 k-tiles of operand loads in flight per
// workgroup: 245 / 155 against 274 TFLOP/s on the decoder weight gradients, the extra register sets cost occupancy 3 -> 2 -> 1;
// (b) h2gemm_kernel: two LDS images per workgroup, the split of tile t + 1 and the MFMAs of tile t in ONE instruction stream with
// no barrier between them, laid out by sched_group_barrier as {1 MFMA, 5 vector, 2 LDS, 1 load} x 24, loads three tiles ahead,
// one barrier per k-tile, 80 KB of LDS and 254 registers = two workgroups per CU: +21 % stand-alone where both operands are
// k-contiguous (PostNet convolutions 204 -> 247, 212 -> 256 TFLOP/s), -8 % where an operand is read through
// ds_read_b64_tr_b16 (decoder weight gradients 276 -> 255), and in the step 116.6-116.8 against 115.2-115.5 ms per pair -- what
// it gains alone it loses beside the loops' kernels; (c) s_setprio 1 around the MFMA block: +0.5-1.5 %.  The counters say why
// the simple form is hard to beat: per wave and k-tile 768 cycles of MFMA, 600 of vector instructions (116 of them), 315 of
// LDS instructions, and the SUM is what a k-tile takes (3086 cycles for the two waves of a SIMD, matrix pipe 39 % busy,
// SQ_WAIT_INST_ANY 46 %): the two waves of a SIMD do not overlap one's vector work with the other's MFMAs.)
// ---- mode 4's pass in front of the product: max |x| over the FINITE elements of both operands, as fp32 bit patterns (monotonic
// in |x|; an Inf / NaN element does not take part: it makes its own row's products NaN, as in the other modes, instead of
// ruining the scale of every row).  grid (blocks, 2): blockIdx.y = operand; an operand with rows == 0 is skipped.  The maxima meet in two 64-bit
// words of g_amax_ring whose HIGH half is the call's epoch: a later call's values are always larger than an earlier call's,
// so a slot is never reset (no memset in front of the launch, and a slot that comes round again after 8192 calls needs no
// care beyond the earlier product having finished) -- until the 32-bit tags run out, after 2^32 calls: amax_ring_slot then
// drains the device and clears the ring once.
struct AbsmaxOp { const float* p; long rows, cols, ld, batch_stride; int batch, vec; };
__device__ __forceinline__ unsigned finite_abs_bits(float x) {
    const unsigned u = __float_as_uint(x) & 0x7fffffffu;
    return u < 0x7f800000u ? u : 0u;
}

__global__ __launch_bounds__(256) void absmax2_kernel(AbsmaxOp a, AbsmaxOp b, unsigned long long* out, unsigned epoch) {
    const AbsmaxOp& o = blockIdx.y == 0 ? a : b;
    if (o.rows == 0) return;
    unsigned m = 0;
    // lines of `cols` contiguous floats (a contiguous operand is ONE line: absmax_extent folds it); workgroups take lines in
    // turn when there are several, and share the one long line otherwise -- no per-element index arithmetic either way
    const long nlines = o.rows * o.batch;
    const bool one = nlines == 1;
    for (long L = one ? 0 : blockIdx.x; L < nlines; L += gridDim.x) {
        const float* p = o.p + (L / o.rows) * o.batch_stride + (L % o.rows) * o.ld;
        const long stride = one ? (long)gridDim.x * 256 : 256, first = one ? (long)blockIdx.x * 256 + threadIdx.x : threadIdx.x;
        if (o.vec) {
            const float4* p4 = reinterpret_cast<const float4*>(p);
            const long n4 = o.cols >> 2;
            long i = first;
            for (; i + 3 * stride < n4; i += 4 * stride) {
                const float4 v0 = p4[i], v1 = p4[i + stride], v2 = p4[i + 2 * stride], v3 = p4[i + 3 * stride];
                m = max(max(max(m, finite_abs_bits(v0.x)), max(finite_abs_bits(v0.y), finite_abs_bits(v0.z))), finite_abs_bits(v0.w));
                m = max(max(max(m, finite_abs_bits(v1.x)), max(finite_abs_bits(v1.y), finite_abs_bits(v1.z))), finite_abs_bits(v1.w));
                m = max(max(max(m, finite_abs_bits(v2.x)), max(finite_abs_bits(v2.y), finite_abs_bits(v2.z))), finite_abs_bits(v2.w));
                m = max(max(max(m, finite_abs_bits(v3.x)), max(finite_abs_bits(v3.y), finite_abs_bits(v3.z))), finite_abs_bits(v3.w));
            }
            for (; i < n4; i += stride) {
                const float4 v0 = p4[i];
                m = max(max(max(m, finite_abs_bits(v0.x)), max(finite_abs_bits(v0.y), finite_abs_bits(v0.z))), finite_abs_bits(v0.w));
            }
        } else {
            for (long i = first; i < o.cols; i += stride) m = max(m, finite_abs_bits(p[i]));
        }
        if (one) break;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    __shared__ unsigned wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        atomicMax(&out[blockIdx.y], ((unsigned long long)epoch << 32) | m);
    }
}

// =====================================================================================================================
// The same three-plane product on operands that ARRIVE as bf16 planes (GtOperand.planes: plane i of element e at
// planes + i * plane_stride + e, e indexed exactly like the fp32 operand).  The split -- ~180 vector instructions per thread
// and k-tile in xgemm_kernel, repeated by every workgroup that shares an operand row -- is done once, where the operand
// is produced (gt_split3, or the producing kernel's epilogue), and this kernel has no vector work left: each wave moves its
// share of the tile global -> LDS with `global_load_lds_dwordx4` (1 KiB per instruction, no registers, no ds_write) and
// issues MFMAs.  LDS-DMA writes lane l's 16 bytes at base + 16 l, so the IMAGE is lane-linear and the bank-conflict
// swizzle sits on the per-lane SOURCE address plus the matching permutation of the fragment read:
//   k-contiguous operand (KC / IM2COL_K): image [row][4 chunks of 8 k] (64-byte rows), chunk c of row r stored in slot
//     c ^ ((r >> 2) & 3): the 16 rows of a ds_read_b128 lane group then cover all 16 bank quads;
//   reduction-major operand (RC / IM2COL_R): image [k][BR / 8 chunks of 8 rows], chunk m of k-row k stored in slot
//     (m + rot(k)) mod chunks with rot = 4 (k & 3) for 128-row tiles (k-rows 64 bytes apart mod 256) and 4 ((k >> 1) & 1)
//     for 64-row tiles: the four k-rows of a ds_read_b64_tr_b16 lane group fall on disjoint banks.
// One LDS buffer, every fragment of the k-tile in registers: read fragments -> barrier -> DMA of the next tile -> 48 MFMAs
// -> barrier (the DMA has landed), so the transfer of tile k + 1 runs under the MFMAs of tile k without a second buffer
// (48 KiB per workgroup: two workgroups per CU).  Conv taps outside [0, T) read 16 bytes of zeros (g_zero16).
constexpr int BKP = 32;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
__device__ __attribute__((aligned(16))) unsigned short g_zero16[8];

template <int MODE, int BR, int NW = 4>
struct TileDma {
    static constexpr bool KT = is_ktype<MODE>();
    static constexpr int PLANE = BR * BKP;                 // bf16 elements of one plane image (no padding)
    static constexpr int NI = PLANE * 2 / 1024;            // wave instructions per plane: 16 / 8 / 4 for 256 / 128 / 64 rows
    static constexpr int NJ = NI / NW;                     // ... per wave (NW waves share the tile)
    static constexpr int CPR = BR / 8;                     // reduction-major: 16-byte chunks per k-row
    static constexpr int KPI = 64 / CPR;                   // reduction-major: k-rows per instruction
    long off[NJ];                                          // KC / RC: element offset of this lane's chunk in the next k-tile
    int p0[NJ], p1[NJ], q0[NJ], q1[NJ];                    // im2col: fixed (row) and running (k) decompositions, as TileLoaderX

    static __device__ __forceinline__ int rot(int kk) { return BR >= 128 ? 4 * (kk & 3) : 4 * ((kk >> 1) & 1); }

    __device__ __forceinline__ void init(const OpDev& o, int r0, int R, int kbeg, int lane, int wave) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const int j = wave + NW * jj;
            p0[jj] = p1[jj] = q0[jj] = q1[jj] = 0;
            off[jj] = 0;
            if (KT) {
                const int rl = 16 * j + (lane >> 2);
                const int c = (lane & 3) ^ ((rl >> 2) & 3);
                const int row = min(r0 + rl, R - 1);                 // rows beyond the matrix re-read its last row (never stored)
                const int k = kbeg + 8 * c;
                if (MODE == GT_OP_KC) off[jj] = (long)row * o.ld + k;
                else { p0[jj] = row / o.T; p1[jj] = row - p0[jj] * o.T; q0[jj] = k / o.C; q1[jj] = k - q0[jj] * o.C; }
            } else {
                const int kk = KPI * j + lane / CPR;
                const int m = ((lane % CPR) - rot(kk)) & (CPR - 1);
                const int row = min(r0 + 8 * m, R - 8);
                const int k = kbeg + kk;
                if (MODE == GT_OP_RC) off[jj] = (long)k * o.ld + row;
                else { p0[jj] = row / o.C; p1[jj] = row - p0[jj] * o.C; q0[jj] = k / o.T; q1[jj] = k - q0[jj] * o.T; }
            }
        }
    }
    // source element offset of instruction slot jj for the NEXT k-tile (im2col: with its validity), then advance
    __device__ __forceinline__ long next_source(const OpDev& o, int jj, bool& ok) {
        long e = off[jj];
        ok = true;
        if (MODE == GT_OP_IM2COL_K) {
            const int tt = p1[jj] + q0[jj] * o.dil - o.pad;
            ok = tt >= 0 && tt < o.T;
            e = ((long)p0[jj] * o.T + (ok ? tt : p1[jj])) * o.C + q1[jj];
            q1[jj] += BKP;
            while (q1[jj] >= o.C) { q1[jj] -= o.C; ++q0[jj]; }
        } else if (MODE == GT_OP_IM2COL_R) {
            const int tt = q1[jj] + p0[jj] * o.dil - o.pad;
            ok = tt >= 0 && tt < o.T;
            e = ((long)q0[jj] * o.T + (ok ? tt : q1[jj])) * o.C + p1[jj];
            q1[jj] += BKP;
            while (q1[jj] >= o.T) { q1[jj] -= o.T; ++q0[jj]; }
        } else {
            off[jj] += MODE == GT_OP_KC ? (long)BKP : (long)BKP * o.ld;
        }
        return e;
    }
    // one LDS-DMA instruction: plane p of instruction slot jj (1 KiB of the image)
    static __device__ __forceinline__ void piece(const unsigned short* base, long ps, long e, bool ok, unsigned short* S, int wave,
                                                 int jj, int p) {
        const unsigned short* src = ok ? base + p * ps + e : g_zero16;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(S + p * PLANE + (wave + NW * jj) * 512), 16, 0, 0);
    }
    // this wave's whole share of the next k-tile at once (prologue)
    __device__ __forceinline__ void issue(const OpDev& o, const unsigned short* base, long ps, unsigned short* S, int wave) {
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            bool ok;
            const long e = next_source(o, jj, ok);
#pragma unroll
            for (int p = 0; p < 3; ++p) piece(base, ps, e, ok, S, wave, jj, p);
        }
    }
    // fragment of the 32x32x16 product for rows [w0 + 32 i, +32), k-step s, plane p (layouts: xgemm_kernel)
    static __device__ __forceinline__ bf16x8_t frag(const unsigned short* S, int w0, int lane, int p, int i, int s) {
        if (KT) {
            const int r = w0 + 32 * i + (lane & 31), c = 2 * s + (lane >> 5);
            return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(S + p * PLANE + r * BKP + 8 * (c ^ ((r >> 2) & 3))));
        }
        typedef __attribute__((address_space(3))) v4i16_t* lds_v4;
        const int m0 = w0 + 32 * i + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        const int kk = 16 * s + 8 * (lane >> 5) + ((lane & 15) >> 2);
        const unsigned short* q = S + p * PLANE + kk * BR + 8 * (((m0 >> 3) + rot(kk)) & (CPR - 1)) + (m0 & 7);
        const unsigned short* q4 = S + p * PLANE + (kk + 4) * BR + 8 * (((m0 >> 3) + rot(kk + 4)) & (CPR - 1)) + (m0 & 7);
        const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q));
        const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q4));
        return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
};

template <int BM, int BN, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void pgemm_kernel(GemmDev g) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    typedef TileDma<AMODE, BM> LA;
    typedef TileDma<BMODE, BN> LB;
    __shared__ __attribute__((aligned(1024))) unsigned short smem[3 * (LA::PLANE + LB::PLANE)];
    unsigned short* Ax = smem;
    unsigned short* Bx = smem + 3 * LA::PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so the ones that share an XCD (equal linear
    // id mod 8) get a CONTIGUOUS range of tiles, walked in 8-wide column strips -- the 64 tiles an XCD holds at a time are
    // then ~8 x 8 and its L2 serves 7 of 8 operand-panel reads (6 bytes per element make this kernel the first of the
    // GEMMs here that is bound by what leaves L2; placement is a speed matter only)
    int bx = blockIdx.x, by = blockIdx.y;
    if (!(g.dbg & 8)) {
        const int nx = gridDim.x, ny = gridDim.y, nwg = nx * ny;
        const int lin = by * nx + bx;
        const int xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);       // bijective remap
        const int strip = 8 * ny;                       // tiles per full 8-column strip
        const int sidx = t / strip, rem = t - sidx * strip;
        const int w = min(8, nx - 8 * sidx);            // width of this strip
        by = rem / w;
        bx = 8 * sidx + rem - by * w;
    }
    const int m0 = by * BM, n0 = bx * BN;
    const int batch = blockIdx.z / g.split_k, kz = blockIdx.z - batch * g.split_k;
    const int kbeg = kz * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const unsigned short* Ab = g.Ap + (long)batch * g.a_bs;
    const unsigned short* Bb = g.Bp + (long)batch * g.b_bs;
    float* Cb = g.C + (long)batch * g.c_bs;

    LA la;
    LB lb;
    la.init(g.A, m0, g.M, kbeg, lane, wave);
    lb.init(g.B, n0, g.N, kbeg, lane, wave);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    if (kbeg < kend) {
        la.issue(g.A, Ab, g.a_ps, Ax, wave);
        lb.issue(g.B, Bb, g.b_ps, Bx, wave);
    }
    __syncthreads();                      // (waits for the LDS-DMA in flight: vmcnt(0) + barrier)
    for (int k0 = kbeg; k0 < kend; k0 += BKP) {
        bf16x8_t a[2][3][TM], b[2][3][TN];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[s][p][i] = LA::frag(Ax, wm0, lane, p, i, s);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[s][p][j] = LB::frag(Bx, wn0, lane, p, j, s);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                  // every wave holds its fragments: the image may be overwritten
        // (Measured and not adopted: the next tile's LDS-DMA pieces issued BETWEEN the MFMAs, one per four, pinned with
        // sched_barrier: 145 instead of 190 TFLOP/s on the decoder weight gradients -- the pins cost the MFMA stream more
        // than the burst costs the pipe, the other workgroup of the CU computes while this one issues.)
        if (k0 + BKP < kend && !(g.dbg & 1)) {
            if (g.dbg & 2) { la.init(g.A, m0, g.M, kbeg, lane, wave); lb.init(g.B, n0, g.N, kbeg, lane, wave); }
            la.issue(g.A, Ab, g.a_ps, Ax, wave);
            lb.issue(g.B, Bb, g.b_ps, Bx, wave);
        }
        if (!(g.dbg & 4))
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ord = 2; ord >= 0; --ord)              // smallest terms first
#pragma unroll
                for (int pa = 0; pa <= ord; ++pa)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][pa][i], b[s][ord - pa][j], acc[i][j], 0, 0, 0);
        __syncthreads();                  // the next tile has landed (vmcnt(0) + barrier)
    }
    // epilogue: as sgemm_kernel
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn0 + 32 * j + (lane & 31);
            if (n >= g.N) continue;
            const float bias = (g.bias != nullptr && kz == 0) ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float* c = Cb + (long)m * g.ldc + n;
                const float v = g.alpha * acc[i][j][r] + bias;
                if (g.split_k > 1) atomicAdd(c, v);
                else if (g.accumulate) *c += v;
                else *c = v;
            }
        }
}

// ---- the same product on 256 x 128 tiles: 8 waves (4 x 2, wave tile 64 x 64), TWO LDS stages of 72 KiB, one workgroup per
// CU.  The DMA of tile k + 1 is issued before the fragment reads of tile k and lands under its 48 MFMAs per wave; ONE barrier
// per k-tile (it closes the reads of stage k & 1 and, behind vmcnt(0), opens stage (k + 1) & 1).  Against pgemm_kernel: a
// quarter less operand traffic per flop (384 instead of 512 rows staged per 128 x 256 outputs), 9 instead of 12 DMA
// instructions per wave and tile, half the barriers, and the second wave of each SIMD computes while the first one issues.
template <int AMODE, int BMODE>
__global__ __launch_bounds__(512) void pgemm2_kernel(GemmDev g) {
    constexpr int BM = 256, BN = 128, NW = 8;
    typedef TileDma<AMODE, BM, NW> LA;
    typedef TileDma<BMODE, BN, NW> LB;
    constexpr int STAGE = 3 * (LA::PLANE + LB::PLANE);             // bf16 elements per stage (72 KiB)
    extern __shared__ __attribute__((aligned(1024))) unsigned short smem2[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = blockIdx.x, by = blockIdx.y;
    {   // XCD-aware tile order (as pgemm_kernel)
        const int nx = gridDim.x, ny = gridDim.y, nwg = nx * ny;
        const int lin = by * nx + bx;
        const int xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
        const int strip = 8 * ny;
        const int sidx = t / strip, rem = t - sidx * strip;
        const int w = min(8, nx - 8 * sidx);
        by = rem / w;
        bx = 8 * sidx + rem - by * w;
    }
    const int m0 = by * BM, n0 = bx * BN;
    const int batch = blockIdx.z / g.split_k, kz = blockIdx.z - batch * g.split_k;
    const int kbeg = kz * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const unsigned short* Ab = g.Ap + (long)batch * g.a_bs;
    const unsigned short* Bb = g.Bp + (long)batch * g.b_bs;
    float* Cb = g.C + (long)batch * g.c_bs;

    LA la;
    LB lb;
    la.init(g.A, m0, g.M, kbeg, lane, wave);
    lb.init(g.B, n0, g.N, kbeg, lane, wave);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
    if (kbeg < kend) {
        la.issue(g.A, Ab, g.a_ps, smem2, wave);
        lb.issue(g.B, Bb, g.b_ps, smem2 + 3 * LA::PLANE, wave);
    }
    __syncthreads();                      // vmcnt(0) + barrier: stage 0 is there
    // (Measured and not adopted: a software-pipelined form -- fragments of the next k-step read under the MFMAs of the
    // current one, the barrier between the two k-steps, the DMA of tile k + 2 issued by waves 0..3 and 4..7 at different
    // times: 254 registers, 192 instead of 205 TFLOP/s on the decoder weight gradients, 78 instead of 181 on the PostNet
    // convolutions, whose im2col address arithmetic then sits inside the MFMA stream.)
    int st = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BKP, st ^= 1) {
        const unsigned short* Ax = smem2 + st * STAGE;
        const unsigned short* Bx = Ax + 3 * LA::PLANE;
        if (k0 + BKP < kend && !(g.dbg & 1)) {     // stage st ^ 1 was last read one iteration ago, behind the barrier below
            if (g.dbg & 2) { la.init(g.A, m0, g.M, kbeg, lane, wave); lb.init(g.B, n0, g.N, kbeg, lane, wave); }
            unsigned short* An = smem2 + (st ^ 1) * STAGE;
            la.issue(g.A, Ab, g.a_ps, An, wave);
            lb.issue(g.B, Bb, g.b_ps, An + 3 * LA::PLANE, wave);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8_t a[3][2], b[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[p][i] = LA::frag(Ax, wm0, lane, p, i, s);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[p][j] = LB::frag(Bx, wn0, lane, p, j, s);
            }
            if (g.dbg & 4) {              // knock-out: keep the fragment reads alive, no MFMAs
#pragma unroll
                for (int p = 0; p < 3; ++p) { asm volatile("" :: "v"(a[p][0]), "v"(a[p][1]), "v"(b[p][0]), "v"(b[p][1])); }
                continue;
            }
#pragma unroll
            for (int ord = 2; ord >= 0; --ord)
#pragma unroll
                for (int pa = 0; pa <= ord; ++pa)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][i], b[ord - pa][j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                  // reads of stage st done everywhere; (vmcnt(0)) stage st ^ 1 has landed
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn0 + 32 * j + (lane & 31);
            if (n >= g.N) continue;
            const float bias = (g.bias != nullptr && kz == 0) ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= g.M) continue;
                float* c = Cb + (long)m * g.ldc + n;
                const float v = g.alpha * acc[i][j][r] + bias;
                if (g.split_k > 1) atomicAdd(c, v);
                else if (g.accumulate) *c += v;
                else *c = v;
            }
        }
}

template <int AM, int BMD>
int launch_pgemm2_modes(const GemmDev& d, dim3 grid, hipStream_t st) {
    constexpr int LDS = 2 * 3 * (256 + 128) * BKP * 2;            // 147 456 bytes
    static bool once = false;
    if (!once) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pgemm2_kernel<AM, BMD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        once = true;
    }
    hipLaunchKernelGGL((pgemm2_kernel<AM, BMD>), grid, dim3(512), LDS, st, d);
    return 0;
}
int launch_pgemm2(const GemmDev& d, dim3 grid, hipStream_t st) {
    const int am = d.A.mode, bm = d.B.mode;
#define GT_CASE2(AM, BMD) if (am == AM && bm == BMD) return launch_pgemm2_modes<AM, BMD>(d, grid, st);
    GT_CASE2(GT_OP_KC, GT_OP_KC)
    GT_CASE2(GT_OP_KC, GT_OP_RC)
    GT_CASE2(GT_OP_RC, GT_OP_RC)
    GT_CASE2(GT_OP_RC, GT_OP_KC)
    GT_CASE2(GT_OP_IM2COL_K, GT_OP_KC)
    GT_CASE2(GT_OP_RC, GT_OP_IM2COL_R)
#undef GT_CASE2
    return GT_E_ARG;
}

// x -> its three bf16 planes (x = p0 + p1 + p2 exactly, as split_all): 8 elements per thread, flat
__global__ __launch_bounds__(256) void split3_kernel(const float* x, long n8, unsigned short* out, long ps) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        float4 reg[2] = {*reinterpret_cast<const float4*>(x + 8 * i), *reinterpret_cast<const float4*>(x + 8 * i + 4)};
        unsigned lo[2][3], hi[2][3];
        split_all<3, 2>(reg, lo, hi);
#pragma unroll
        for (int p = 0; p < 3; ++p)
            *reinterpret_cast<uint4*>(out + p * ps + 8 * i) = make_uint4(lo[0][p], hi[0][p], lo[1][p], hi[1][p]);
    }
}

// (Measured and not adopted: a producer / consumer form with 512 threads -- waves 4..7 load, split and store the next
// k-tile into a second LDS buffer while waves 0..3 only issue MFMAs, two k-tiles of loads in flight.  182-186 TFLOP/s
// against 191-193 for xgemm_kernel on the decoder weight gradients: SQ_VALU_MFMA_COEXEC_CYCLES stays at 1 %, the
// vector work of one wave does not run beside the MFMAs of its SIMD partner, so the two kinds of wave take turns.
// Also not adopted: a software-pipelined form, one 4-wave workgroup per CU with the whole register file, two LDS buffers,
// loads two k-tiles ahead with counted vmcnt waits, and -- pinned by sched_barrier -- one split stage (<= 5 vector
// instructions), one fragment read and one plane store in every MFMA gap of the same wave: 186 TFLOP/s.  Its parts:
// MFMAs + fragment reads alone 300, + plane stores 258; the split does not hide in the gaps (dependent cvt / shift / sub
// chains).  And 256 x 128 tiles with wave tiles of 128 x 64 (the split amortised over twice the MFMAs, one workgroup per
// CU): 178 against 196 on the same GEMMs.  XCD-contiguous tile order (each XCD's L2 sees 4 m-rows x 16 n-tiles instead of
// every m-row): 201.5 -> 205.7 on one GEMM, nothing on the others, pair 133.5 vs 133.2 ms.  xgemm_kernel with two workgroups
// per CU and the plain tile order stays.)
template <int BM, int BN>
int launch_tile(const GemmDev& d, dim3 grid, hipStream_t st, int lds_pad) {
    const int am = d.A.mode, bm = d.B.mode;
#define GT_CASE(AM, BMD)                                                                                    \
    if (am == AM && bm == BMD) {                                                                            \
        if (d.Ap != nullptr) hipLaunchKernelGGL((pgemm_kernel<BM, BN, AM, BMD>), grid, dim3(256), lds_pad, st, d);       \
        else if (d.bf16 == 1) hipLaunchKernelGGL((hgemm_kernel<BM, BN, AM, BMD>), grid, dim3(256), lds_pad, st, d);      \
        else if (d.bf16 == 2) hipLaunchKernelGGL((xgemm_kernel<BM, BN, AM, BMD, 2>), grid, dim3(256), lds_pad, st, d);   \
        else if (d.bf16 == 3) hipLaunchKernelGGL((xgemm_kernel<BM, BN, AM, BMD, 3>), grid, dim3(256), lds_pad, st, d);   \
        else if (d.bf16 == 4) hipLaunchKernelGGL((xgemm_kernel<BM, BN, AM, BMD, 2, 1>), grid, dim3(256), lds_pad, st, d); \
        else hipLaunchKernelGGL((sgemm_kernel<BM, BN, AM, BMD>), grid, dim3(256), lds_pad, st, d);                      \
        return 0;                                                                                           \
    }
    GT_CASE(GT_OP_KC, GT_OP_KC)
    GT_CASE(GT_OP_KC, GT_OP_RC)
    GT_CASE(GT_OP_RC, GT_OP_RC)
    GT_CASE(GT_OP_RC, GT_OP_KC)
    GT_CASE(GT_OP_IM2COL_K, GT_OP_KC)
    GT_CASE(GT_OP_RC, GT_OP_IM2COL_R)
#undef GT_CASE
    return GT_E_ARG;
}

// fp16-plane form only, other tile shapes: 128 x 256 (wave tiles 64 x 128) for large products of k-contiguous operands (gt_sgemm);
// GANTRON_DEBUG=h2_wide=1 / 2 (diagnostics): 128 x 256 / 256 x 128 tiles for every large product, 4: the default rule plus
// the weight gradients (reduction-major operands) with M >= 1024, N >= 512, 0: never
template <int BM, int BN>
int launch_tile_h2(const GemmDev& d, dim3 grid, hipStream_t st) {
    const int am = d.A.mode, bm = d.B.mode;
#define GT_CASEH(AM, BMD) if (am == AM && bm == BMD) { hipLaunchKernelGGL((xgemm_kernel<BM, BN, AM, BMD, 2, 1>), grid, dim3(256), 0, st, d); return 0; }
    GT_CASEH(GT_OP_KC, GT_OP_KC)
    GT_CASEH(GT_OP_KC, GT_OP_RC)
    GT_CASEH(GT_OP_RC, GT_OP_RC)
    GT_CASEH(GT_OP_RC, GT_OP_KC)
    GT_CASEH(GT_OP_IM2COL_K, GT_OP_KC)
    GT_CASEH(GT_OP_RC, GT_OP_IM2COL_R)
#undef GT_CASEH
    return GT_E_ARG;
}

int prep_operand(const GtOperand& s, OpDev& o, int rows, int K, const char* name) {
    GT_REQUIRE(s.p != nullptr || s.planes != nullptr, GT_E_ARG, "gt_sgemm: operand %s is NULL", name);
    o.p = s.p; o.mode = s.mode; o.T = s.T; o.C = s.C; o.dil = s.dil; o.pad = s.pad; o.ld = s.ld;
    switch (s.mode) {
        case GT_OP_KC:
            GT_REQUIRE(s.ld >= 1, GT_E_ARG, "gt_sgemm: %s ld", name);
            o.vec = aligned16(s.p) && (s.ld % 4 == 0) && (s.batch_stride % 4 == 0);
            break;
        case GT_OP_RC:
            GT_REQUIRE(s.ld >= rows, GT_E_ARG, "gt_sgemm: %s ld %ld < rows %d", name, (long)s.ld, rows);
            o.vec = aligned16(s.p) && (s.ld % 4 == 0) && (s.batch_stride % 4 == 0);
            break;
        case GT_OP_IM2COL_K:
            GT_REQUIRE(s.T > 0 && s.C > 0 && s.C % 4 == 0 && K % s.C == 0 && rows % s.T == 0, GT_E_ARG,
                       "gt_sgemm: %s im2col_k needs C%%4==0, K%%C==0, rows%%T==0 (T=%d C=%d K=%d rows=%d)",
                       name, s.T, s.C, K, rows);
            o.vec = aligned16(s.p);
            break;
        case GT_OP_IM2COL_R:
            GT_REQUIRE(s.T > 0 && s.C > 0 && s.C % 4 == 0 && rows % s.C == 0 && K % s.T == 0, GT_E_ARG,
                       "gt_sgemm: %s im2col_r needs C%%4==0, rows%%C==0, K%%T==0 (T=%d C=%d K=%d rows=%d)",
                       name, s.T, s.C, K, rows);
            o.vec = aligned16(s.p);
            break;
        default:
            GT_REQUIRE(false, GT_E_ARG, "gt_sgemm: %s bad mode %d", name, s.mode);
    }
    return 0;
}

// the memory an operand of `rows` x K addresses, as (rows of memory) x (contiguous columns) per batch entry
AbsmaxOp absmax_extent(const GtOperand& s, const OpDev& o, int rows, int K, int batch) {
    AbsmaxOp a;
    a.p = s.p; a.batch = batch; a.batch_stride = s.batch_stride;
    switch (s.mode) {
        case GT_OP_KC: a.rows = rows; a.cols = K; a.ld = s.ld; break;
        case GT_OP_RC: a.rows = K; a.cols = rows; a.ld = s.ld; break;
        case GT_OP_IM2COL_K: a.rows = rows; a.cols = s.C; a.ld = s.C; break;           // rows = B * T positions of C channels
        default: a.rows = K; a.cols = s.C; a.ld = s.C; break;                          // IM2COL_R: K = B * T positions
    }
    a.vec = o.vec && a.cols % 4 == 0;
    if (a.ld == a.cols) { a.cols *= a.rows; a.ld = a.cols; a.rows = 1; }             // contiguous: one long row
    return a;
}

#define GT_TRY_RC(call) do { int rc__ = (call); if (rc__ != 0) return rc__; } while (0)
// next pair of words of the device's g_amax_ring + a fresh epoch (see absmax2_kernel)
int amax_ring_slot(unsigned long long** ring, int* slot, unsigned* epoch) {
    struct Ring { unsigned long long* base; unsigned next_slot, next_epoch; };
    static std::mutex mu;
    static std::map<int, Ring> rings;                        // per device
    // GANTRON_DEBUG=amax_wrap=<n> (tests): the tags of a device start n calls in front of their wrap
    static const int wrap_in = gt_env_item("GANTRON_DEBUG", "amax_wrap", 0);
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    GT_REQUIRE(e == hipSuccess, (int)e, "gt_sgemm: hipGetDevice failed");
    std::lock_guard<std::mutex> lk(mu);
    auto it = rings.find(dev);
    if (it == rings.end()) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_amax_ring));
        GT_REQUIRE(e == hipSuccess && p != nullptr, (int)e, "gt_sgemm: hipGetSymbolAddress(g_amax_ring) failed");
        Ring r = {(unsigned long long*)p, 0u, wrap_in > 0 ? 0xffffffffu - (unsigned)wrap_in : 0u};
        it = rings.emplace(dev, r).first;
    }
    Ring& r = it->second;
    if (r.next_epoch == 0xffffffffu) {
        // the 32-bit tags are used up (2^32 calls: days of training).  A word only ever grows, so the tags cannot simply start
        // over: wait until every product that may still read a word has finished, clear the ring, begin again at 1
        e = hipDeviceSynchronize();
        GT_REQUIRE(e == hipSuccess, (int)e, "gt_sgemm: hipDeviceSynchronize failed (tag wrap of the maxima ring)");
        e = hipMemset(r.base, 0, sizeof(unsigned long long) * AMAX_SLOTS * AMAX_STRIDE);
        GT_REQUIRE(e == hipSuccess, (int)e, "gt_sgemm: hipMemset failed (tag wrap of the maxima ring)");
        e = hipDeviceSynchronize();
        GT_REQUIRE(e == hipSuccess, (int)e, "gt_sgemm: hipDeviceSynchronize failed (tag wrap of the maxima ring)");
        r.next_epoch = 0u;
    }
    *ring = r.base;
    *slot = AMAX_STRIDE * (int)(r.next_slot++ % (unsigned)AMAX_SLOTS);
    *epoch = ++r.next_epoch;
    return 0;
}

int g_fp32_planes = 4;       // gt_sgemm_set_fp32_split: how GtGemm.bf16 == 0 products are computed
int g_last_kernel = -1;      // gt_sgemm_last_kernel (tests): 0 sgemm, 1 hgemm, 2 xgemm, 3 pgemm, 4 pgemm2, 5 xgemm on fp16 planes

// can pgemm_kernel address this operand's planes? (16-byte chunks of 8 bf16 that never straddle a row / tap / matrix end)
bool planes_ok(const GtOperand& s, int rows) {
    if (!aligned16(s.planes) || s.plane_stride % 8 != 0 || s.batch_stride % 8 != 0) return false;
    switch (s.mode) {
        case GT_OP_KC: return s.ld % 8 == 0;
        case GT_OP_RC: return s.ld % 8 == 0 && rows % 8 == 0;
        case GT_OP_IM2COL_K: return s.C % BKP == 0;
        case GT_OP_IM2COL_R: return s.C % 8 == 0 && rows % 8 == 0;
    }
    return false;
}

}  // namespace

extern "C" int gt_sgemm_set_fp32_split(int32_t planes) {
    GT_REQUIRE(planes == 0 || planes == 2 || planes == 3 || planes == 4, GT_E_ARG,
               "gt_sgemm_set_fp32_split: 0 (fp32 pipe), 2 / 3 (bf16 planes) or 4 (two scaled fp16 planes)");
    g_fp32_planes = planes;
    return 0;
}

extern "C" int32_t gt_sgemm_get_fp32_split(void) { return g_fp32_planes; }
extern "C" int32_t gt_sgemm_last_kernel(void) { const int k = g_last_kernel; g_last_kernel = -1; return k; }

extern "C" int gt_split3(const float* x, int64_t n, void* planes, int64_t plane_stride, void* stream) {
    GT_REQUIRE(x != nullptr && planes != nullptr && n > 0 && n % 8 == 0 && plane_stride >= n && plane_stride % 8 == 0 &&
                   aligned16(x) && aligned16(planes), GT_E_ARG,
               "gt_split3: n (%ld) and plane_stride (%ld) must be multiples of 8, plane_stride >= n, 16-byte aligned buffers",
               (long)n, (long)plane_stride);
    const long n8 = n / 8;
    const unsigned blocks = (unsigned)std::min<long>(cdiv64(n8, 256), 4096);
    hipLaunchKernelGGL(split3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n8, (unsigned short*)planes, (long)plane_stride);
    GT_LAUNCH_CHECK();
    return 0;
}

// internal (loops.hip): max |x| of a matrix that SEVERAL products of one loop call share (the decoder LSTM's input weights:
// one tiled GEMM per block of steps), into the next ring word; *word stays valid for ~8000 later gt_sgemm calls
int gt_absmax_ring(const float* x, long rows, long cols, long ld, const uint64_t** word, hipStream_t st) {
    unsigned long long* ring = nullptr;
    int slot = 0;
    unsigned epoch = 0;
    GT_TRY_RC(amax_ring_slot(&ring, &slot, &epoch));
    AbsmaxOp a, none = {};
    a.p = x; a.rows = rows; a.cols = cols; a.ld = ld; a.batch = 1; a.batch_stride = 0;
    a.vec = aligned16(x) && cols % 4 == 0 && ld % 4 == 0;
    if (a.ld == a.cols) { a.cols *= a.rows; a.ld = a.cols; a.rows = 1; }
    const unsigned blocks = (unsigned)std::min<long>(256, std::max<long>(1, cdiv64(a.rows * a.cols, 256L * 16)));
    hipLaunchKernelGGL(absmax2_kernel, dim3(blocks, 1), dim3(256), 0, st, a, none, ring + slot, epoch);
    GT_LAUNCH_CHECK();
    *word = (const uint64_t*)(ring + slot);
    return 0;
}

extern "C" int gt_absmax(const float* x, int64_t rows, int64_t cols, int64_t ld, uint64_t* out, void* stream) {
    GT_REQUIRE(x != nullptr && out != nullptr && rows > 0 && cols > 0 && ld >= cols, GT_E_ARG, "gt_absmax: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(uint64_t), st);
    GT_REQUIRE(e == hipSuccess, (int)e, "gt_absmax: hipMemsetAsync failed");
    AbsmaxOp a, none = {};
    a.p = x; a.rows = rows; a.cols = cols; a.ld = ld; a.batch = 1; a.batch_stride = 0;
    a.vec = aligned16(x) && cols % 4 == 0 && ld % 4 == 0;
    if (a.ld == a.cols) { a.cols *= a.rows; a.ld = a.cols; a.rows = 1; }
    const unsigned blocks = (unsigned)std::min<long>(256, std::max<long>(1, cdiv64(a.rows * a.cols, 256L * 16)));
    hipLaunchKernelGGL(absmax2_kernel, dim3(blocks, 1), dim3(256), 0, st, a, none, (unsigned long long*)out, 1u);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_sgemm(const GtGemm* g, void* stream) {
    GT_REQUIRE(g != nullptr && g->C != nullptr, GT_E_ARG, "gt_sgemm: NULL argument");
    GT_REQUIRE(g->M > 0 && g->N > 0 && g->K > 0 && g->batch >= 1 && g->split_k >= 1, GT_E_ARG,
               "gt_sgemm: bad sizes M=%d N=%d K=%d batch=%d split=%d", g->M, g->N, g->K, g->batch, g->split_k);
    GT_REQUIRE(g->ldc >= g->N, GT_E_ARG, "gt_sgemm: ldc %ld < N %d", (long)g->ldc, g->N);
    GemmDev d;
    int rc;
    if ((rc = prep_operand(g->A, d.A, g->M, g->K, "A")) != 0) return rc;
    if ((rc = prep_operand(g->B, d.B, g->N, g->K, "B")) != 0) return rc;
    d.C = g->C; d.bias = g->bias; d.ldc = g->ldc;
    d.a_bs = g->A.batch_stride; d.b_bs = g->B.batch_stride; d.c_bs = g->c_batch_stride;
    d.M = g->M; d.N = g->N; d.K = g->K; d.accumulate = g->accumulate; d.alpha = g->alpha;
    d.split_k = g->split_k;
    GT_REQUIRE(g->bf16 >= 0 && g->bf16 <= 4, GT_E_ARG, "gt_sgemm: bf16 = %d (0 fp32, 1 bf16 operands, 2 / 3 bf16 planes, 4 fp16 planes)", g->bf16);
    d.bf16 = g->bf16 != 0 ? g->bf16 : g_fp32_planes;
    // (a captured launch would bake this call's slot and epoch into the graph: replays keep the three bf16 planes)
    if (d.bf16 == 4 && (g_gt_capturing || g->A.p == nullptr || g->B.p == nullptr)) d.bf16 = 3;
    static const int h2_skip = gt_env_item("GANTRON_DEBUG", "h2_skip", 0);      // diagnostics: classes of products kept on three bf16 planes
    if (d.bf16 == 4 && h2_skip) {
        if (((h2_skip & 1) && g->N < 128) || ((h2_skip & 2) && g->M <= 4096) || ((h2_skip & 4) && g->K < 512) ||
            ((h2_skip & 8) && (g->A.mode >= GT_OP_IM2COL_K || g->B.mode >= GT_OP_IM2COL_K)) || ((h2_skip & 16) && (g->A.amax || g->B.amax)) ||
            ((h2_skip & 32) && g->M > 4096 && g->N >= 128 && g->K >= 512 && g->A.mode < GT_OP_IM2COL_K && g->B.mode < GT_OP_IM2COL_K))
            d.bf16 = 3;
    }
    d.amax_slot = 0; d.amax_a = d.amax_b = nullptr;
    static const int pgemm_dbg = gt_env_item("GANTRON_DEBUG", "pgemm", 0);
    d.dbg = pgemm_dbg;
    // operands that arrive as bf16 planes: the pure three-plane kernel, when the shapes allow whole 16-byte chunks and
    // whole 32-wide k-tiles; otherwise the fp32 operands (which must then be there) go through xgemm_kernel's own split
    d.Ap = d.Bp = nullptr; d.a_ps = d.b_ps = 0;
    if (g->A.planes != nullptr && g->B.planes != nullptr && d.bf16 == 3 && g->K % BKP == 0 &&
        planes_ok(g->A, g->M) && planes_ok(g->B, g->N)) {
        d.Ap = (const unsigned short*)g->A.planes; d.Bp = (const unsigned short*)g->B.planes;
        d.a_ps = g->A.plane_stride; d.b_ps = g->B.plane_stride;
    } else {
        GT_REQUIRE(g->A.p != nullptr && g->B.p != nullptr, GT_E_ARG, "gt_sgemm: plane operands of this shape need their fp32 original too");
    }
    const int kstep = d.bf16 == 1 ? BKH : d.bf16 >= 2 ? BKX : BK;        // k-tile of the kernel: splits are whole k-tiles
    const int kblocks = cdiv(g->K, kstep);
    d.k_per_split = cdiv(kblocks, g->split_k) * kstep;
    d.split_k = cdiv(g->K, d.k_per_split);   // drop empty splits
    hipStream_t st = (hipStream_t)stream;

    GtProfScope prof(GT_FAM_SGEMM, 2.0 * g->M * (double)g->N * g->K * g->batch,
                     4.0 * g->batch * ((double)g->M * g->K + (double)g->N * g->K + (double)g->M * g->N), st);
    if (d.bf16 == 4) {
        // max |a|, max |b| over everything the product addresses: one pass per operand that did not bring its word
        // (GtOperand.amax), inside the family's timing bracket; blockIdx.y picks the operand
        d.amax_a = (const unsigned long long*)g->A.amax; d.amax_b = (const unsigned long long*)g->B.amax;
        if (h2_skip & 64) d.amax_a = d.amax_b = nullptr;        // diagnostics: ignore the words the caller brought
        if (d.amax_a == nullptr || d.amax_b == nullptr) {
            unsigned long long* ring = nullptr;
            unsigned epoch = 0;
            GT_TRY_RC(amax_ring_slot(&ring, &d.amax_slot, &epoch));
            AbsmaxOp ao = absmax_extent(g->A, d.A, g->M, g->K, g->batch), bo = absmax_extent(g->B, d.B, g->N, g->K, g->batch);
            if (d.amax_a != nullptr) ao.rows = 0;
            if (d.amax_b != nullptr) bo.rows = 0;
            const long work = std::max(ao.rows * ao.cols * ao.batch, bo.rows * bo.cols * bo.batch);
            const unsigned blocks = (unsigned)std::min<long>(512, std::max<long>(1, cdiv64(work, 256L * 32)));
            hipLaunchKernelGGL(absmax2_kernel, dim3(blocks, 2), dim3(256), 0, st, ao, bo, ring + d.amax_slot, epoch);
            static const int amax_dbg = gt_env_item("GANTRON_DEBUG", "amax", 0);
            if (amax_dbg) {         // diagnostics (synchronises): the two maxima this product is scaled by
                unsigned long long w[2] = {0, 0};
                (void)hipStreamSynchronize(st);
                (void)hipMemcpy(w, ring + d.amax_slot, sizeof(w), hipMemcpyDeviceToHost);
                float fa, fb; unsigned ua = (unsigned)w[0], ub = (unsigned)w[1];
                memcpy(&fa, &ua, 4); memcpy(&fb, &ub, 4);
                if (d.amax_a) { (void)hipMemcpy(w, d.amax_a, 8, hipMemcpyDeviceToHost); ua = (unsigned)w[0]; memcpy(&fa, &ua, 4); }
                if (d.amax_b) { (void)hipMemcpy(w, d.amax_b, 8, hipMemcpyDeviceToHost); ub = (unsigned)w[0]; memcpy(&fb, &ub, 4); }
                fprintf(stderr, "[amax] M=%d N=%d K=%d modes %d/%d batch %d split %d: max|a| %.4g (%s) max|b| %.4g (%s)\n", g->M, g->N, g->K,
                        g->A.mode, g->B.mode, g->batch, g->split_k, fa, d.amax_a ? "given" : "pass", fb, d.amax_b ? "given" : "pass");
            }
        }
    }
    const long tiles128 = (long)cdiv(g->M, 128) * cdiv(g->N, 128) * g->batch * d.split_k;
    // (Measured and not adopted, round 3: 64 x 64 tiles -- 31 KB of LDS instead of 61, so that the launch fits beside TWO workgroups
    // of the attention step on a CU -- for the forward loop's block GEMMs: attention 16.9 instead of 17.1 us, LSTM step 24.4 instead
    // of 25.4, pair 124.9-125.6 against 124.45 ms: the slower GEMM sits on the chain that is the longer one.)
    const bool big = tiles128 >= 192 && g->M > 64 && g->N > 64;
    const long gz = (long)g->batch * d.split_k;
    GT_REQUIRE(gz <= 65535, GT_E_LIMIT, "gt_sgemm: batch*split_k too large");
    // (Measured and not adopted: capping these launches at one or two workgroups per CU with a dynamic-LDS reservation, so
    // that the loops' latency-critical kernels always find free registers next to a GEMM running under them.  sgemm_kernel:
    // no effect on the pair, 160.5 vs 161.9-162.4 ms.  xgemm_kernel (61 KB LDS, 190 registers: two of its workgroups fill a
    // CU's register file) capped at one workgroup per CU while the loops run: 146.9 vs 134.0 ms -- the GEMMs slow down by
    // more than the loop kernels gain.)
    const int lds_pad = 0;
    static const int h2_wide = gt_env_item("GANTRON_DEBUG", "h2_wide", 3);
    static const int pgemm2_on = !gt_env_item("GANTRON_EXPERIMENTAL", "no_pgemm2", 0);
    const long tiles256 = (long)cdiv(g->M, 256) * cdiv(g->N, 128) * gz;
    if (d.Ap != nullptr && pgemm2_on && g->M >= 256 && g->N >= 128 && tiles256 >= 128) {
        dim3 grid(cdiv(g->N, 128), cdiv(g->M, 256), (unsigned)gz);
        GT_REQUIRE(grid.y <= 65535, GT_E_LIMIT, "gt_sgemm: M too large for grid.y");
        rc = launch_pgemm2(d, grid, st);
        g_last_kernel = 4;
    } else if (big && d.bf16 == 4 && g->N >= 256 && ((h2_wide == 1 && tiles128 >= 384) ||
               ((h2_wide == 3 || h2_wide == 4) && d.A.mode != GT_OP_RC && d.B.mode == GT_OP_KC && g->M >= 8192 && tiles128 >= 176) ||
               (h2_wide == 4 && d.A.mode == GT_OP_RC && d.B.mode != GT_OP_KC && g->M >= 1024 && g->N >= 512))) {
        // 128 x 256 tiles (wave tiles 64 x 128: three quarters of the vector and LDS work per MFMA, 234 registers = two workgroups
        // per CU) where BOTH operands are k-contiguous and the product is large: the convolutions' forward and input-gradient
        // passes and the prenet -> attention-LSTM projection, which run outside the decoder loops.  Stand-alone +20-25 % there
        // (PostNet convolution forward 206 -> 257, input gradient 216 -> 269 TFLOP/s), -10 % where ds_read_b64_tr_b16 feeds an
        // operand (weight gradients: they keep 128 x 128); pair 113.8-114.0 against 114.9-115.1 ms (the forward loop's block GEMMs
        // too, M = 1024: 114.8-114.9 against 114.4-114.5 -- they stay on 128 x 128; the discriminator's M = 2816 convolutions lose
        // 0.2-0.3 ms of the D step on the wide tile).  GANTRON_DEBUG=h2_wide=0: off.  The -10 % of the weight gradients is the
        // grid, not the transposed reads: M = 4096, N = 2048 is 256 wide tiles, one workgroup per CU; with split_k = 2 (512
        // workgroups) the same product runs at 326 instead of 283 TFLOP/s stand-alone (profiles/r04_gbench_wide.txt).  In the
        // step the decoder's weight gradients are M = 4096, N <= 1024 products per backward chunk that run UNDER the loop with
        // few workgroups on purpose (ops.SPLITK_OVERLAPPED); h2_wide=4 gives those the wide tile as well: 113.9-114.4 against
        // 113.7-114.0 ms per pair at split-K targets 64 / 128 / 256 -- the pair no longer follows the speed of these GEMMs.
        dim3 grid(cdiv(g->N, 256), cdiv(g->M, 128), (unsigned)gz);
        GT_REQUIRE(grid.y <= 65535, GT_E_LIMIT, "gt_sgemm: M too large for grid.y");
        rc = launch_tile_h2<128, 256>(d, grid, st);
    } else if (big && d.bf16 == 4 && h2_wide == 2 && g->M >= 256 && tiles128 >= 384) {
        dim3 grid(cdiv(g->N, 128), cdiv(g->M, 256), (unsigned)gz);
        GT_REQUIRE(grid.y <= 65535, GT_E_LIMIT, "gt_sgemm: M too large for grid.y");
        rc = launch_tile_h2<256, 128>(d, grid, st);
    } else if (big) {
        dim3 grid(cdiv(g->N, 128), cdiv(g->M, 128), (unsigned)gz);
        GT_REQUIRE(grid.y <= 65535, GT_E_LIMIT, "gt_sgemm: M too large for grid.y");
        rc = launch_tile<128, 128>(d, grid, st, lds_pad);
    } else {
        dim3 grid(cdiv(g->N, 64), cdiv(g->M, 64), (unsigned)gz);
        GT_REQUIRE(grid.y <= 65535, GT_E_LIMIT, "gt_sgemm: M too large for grid.y");
        rc = launch_tile<64, 64>(d, grid, st, lds_pad);
    }
    GT_REQUIRE(rc == 0, rc, "gt_sgemm: unsupported operand mode pair (%d,%d)", g->A.mode, g->B.mode);
    if (g_last_kernel != 4 || d.Ap == nullptr) g_last_kernel = d.Ap != nullptr ? 3 : d.bf16 == 4 ? 5 : d.bf16 >= 2 ? 2 : d.bf16;
    GT_LAUNCH_CHECK();
    return 0;
}
