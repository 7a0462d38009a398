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
#include <stdlib.h>
#include "common.h"
#include "rec_cell.h"

namespace {

// W (N x K, row stride ldw; or K x N when transposed == 1) -> Wp[N/16][KB][64][4], zero padded.
// transposed == 2: gate-interleaved rows for the fused step kernel (recurrent_fused.hip): packed row
// n = 4*unit + gate reads source row gate*(N/4) + unit of an (N x K) LSTM weight.
__global__ __launch_bounds__(256) void pack_rec_weight_kernel(const float* W, long ldw, int transposed,
                                                              int N, int n_valid, int K, int KB, float* Wp) {
    const long total = (long)(N / 16) * KB * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        const long blk = idx >> 6;
        const int kb = (int)(blk % KB);
        const int nb = (int)(blk / KB);
        int n = nb * 16 + (lane & 15);
        if (transposed == 2) n = (n & 3) * (N / 4) + (n >> 2);
        // mode 3 (gt_rec_cell_fwd): packed column 64*cg + 16*gate + u holds gate `gate` of hidden unit 16*cg + u, so
        // that a workgroup's 64 columns are the four gates of 16 units (wave & 3 = gate)
        if (transposed == 3) n = ((n >> 4) & 3) * (N / 4) + (n >> 6) * 16 + (n & 15);
        const int k = kb * 16 + 4 * (lane >> 4);
        float4 v;
        float* vv = reinterpret_cast<float*>(&v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = k + j;
            vv[j] = (kk < K && n < n_valid) ? (transposed == 1 ? W[(long)kk * ldw + n] : W[(long)n * ldw + kk]) : 0.f;
        }
        reinterpret_cast<float4*>(Wp)[idx] = v;
    }
}

// Activation chunks (MT*16 rows x 64 k) are staged through LDS once per workgroup with fully
// coalesced 256-byte row segments and shared by all waves; the next chunk's global loads are issued
// into registers before the current chunk's MFMAs.  A workgroup has 8 waves = 2 per SIMD: wave w owns
// gate columns [16*(w&3), +16) of the workgroup's 64 and the k-blocks of half (w>>2) of every chunk, so
// two waves interleave on each matrix pipe (one's LDS/VMEM waits hide under the other's MFMAs; a single
// wave per SIMD reached only ~60 % of the MFMA rate).
// The two K-halves are folded through LDS at the end.  The wave's weight fragments (1 KiB per 16-wide
// k-block, lane-linear) stream straight into VGPRs, one chunk ahead (two chunks ahead through three rotating buffers, round 3:
// 126.9-127.2 ms per pair against 125.3-126.1; the ACTIVATION chunks two ahead instead: the fused step 27.2 us against 24.9 --
// more requests in flight have made this kernel slower every time.  The chunks of a slice visited in an order rotated by the
// column group, so that the 64 workgroups sharing a K-slice do not ask for the same activation lines at the same moment: 25.9-26.3
// us against 24.9, the backward product 21.6-22.0 against 20 -- running in lockstep is what makes those lines cheap).
constexpr int REC_CK = 4;                 // k-blocks (of 16) per LDS chunk.  Round 3, pair time / fused forward step / backward
                                          // product in the loop: 16 blocks 140 ms / 29.3 us / 28 us; 8 (rounds 1-3) 127.4-128.0 /
                                          // 26.2 / 20.5-21; 4: 125.3-126.1 / 25.3 / 19.7-20.4; 2: 136.4-137.0 / 34 / 23.5
constexpr int REC_HK = REC_CK / 2;        // k-blocks per wave per chunk
constexpr int REC_LD = REC_CK * 16 + 4;   // padded LDS row: conflict-free ds_read_b128 / ds_write_b128
constexpr int REC_THREADS = 512;

template <int MT>
struct RecChunk {                         // per-thread registers of one activation chunk
    static constexpr int NV = (MT * 16 * REC_CK * 16 / 4 + REC_THREADS - 1) / REC_THREADS;   // float4 per thread
    float4 v[NV];
    __device__ __forceinline__ void load(const float* X, long ldx, int kb, int kb1, int tid) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + REC_THREADS * i;
            const int row = f / (REC_CK * 4), c4 = f % (REC_CK * 4);
            const int kblk = kb + (c4 >> 2);
            v[i] = (kblk < kb1 && row < MT * 16) ? *reinterpret_cast<const float4*>(X + (long)row * ldx + (long)kb * 16 + 4 * c4)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __device__ __forceinline__ void store(float* S, int tid) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + REC_THREADS * i;
            const int row = f / (REC_CK * 4), c4 = f % (REC_CK * 4);
            if (row < MT * 16) *reinterpret_cast<float4*>(&S[row * REC_LD + 4 * c4]) = v[i];
        }
    }
};

struct RecW {                             // weight fragments of this wave's half of one chunk
    float4 w[REC_HK];
    __device__ __forceinline__ void load(const float4* wp, int kb, int kb1) {
#pragma unroll
        for (int u = 0; u < REC_HK; ++u) w[u] = (kb + u < kb1) ? wp[u * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
};

// DBG (diagnostics, GANTRON_DEBUG=rec=<bits>): bit 0 = no weight loads, bit 1 = no MFMAs, bit 2 = no activation staging
// One workgroup of the product: 64 columns starting at column group `cg`, K-slice [kb0, kb1) of 16-wide blocks;
// `out` is the slab this K-slice writes (row stride N).
constexpr int EP_LD = 68;                 // epilogue tile row stride: 4 rows apart = 16 banks apart, conflict-free scalar writes

// Product of one workgroup: 64 columns starting at column group `cg`, K-slice [kb0, kb1).  Returns (after a barrier)
// the folded [MT*16 rows][64 columns] tile in LDS, row stride EP_LD; the LDS array has MT*16*REC_LD floats, of which
// the tile uses the first MT*16*EP_LD.
// (Measured with the phase stamps of GANTRON_DEBUG=rec_stamps and not adopted, attention LSTM of the training step, 64 rows x
// 64 columns x K-slice 688 per workgroup, in the loop: this form -- first operands in LDS after 1.6 us, 2.4 us per chunk
// iteration (1.7 us of MFMAs), products done at 16.2 us; every weight fragment of the slice requested up front: first
// operands after 5.2 us, 2.1 us per chunk, done at 17.7; weights three chunks ahead, activations two: 3.6 / 2.2 / 16.9.)
// NBUF = 2 (the forward fused step): two staging buffers, chunk c + 1 is published into the other one while waves still read
// chunk c -- ONE barrier per chunk (products done 13.3 us after the first operands instead of 14.6).  The backward products
// keep one buffer: with 68 KB of LDS instead of 34 their workgroups no longer fit beside the weight-gradient GEMMs' (2 x 61 KB
// per CU) and run 25 us instead of 20 in the loop.
template <int MT, int DBG, int NBUF = 1>
__device__ __forceinline__ float* rec_gemm_tile(const float* __restrict__ Xz, long ldx, const float* __restrict__ Wpz,
                                                int N, int KB, int kb0, int kb1, int cg, long long* stamps = nullptr) {
    static_assert(MT == 4 || MT == 2 || MT == 1 || MT == 3, "row tiles");
    // the staging buffer(s); at least as large as what the callers' epilogues use of it afterwards
    constexpr int SLOT = MT * 16 * REC_LD;
    // (the plain product's epilogue is the folded tile alone, the fused ones add their scratch: with 17.4 KB the attention chain's
    // and the decoder-LSTM chain's backward products TOGETHER fit the 37 KB that two weight-gradient GEMM workgroups leave on a CU)
    constexpr int EPI = NBUF == 1 ? MT * 16 * REC_EP_LD : MT * 16 * (REC_EP_LD + 20) + 4;
    __shared__ __attribute__((aligned(16))) float Xs[NBUF * SLOT > EPI ? NBUF * SLOT : EPI];
    if (DBG & 8) return Xs;               // diagnostics: pure launch + wave start-up cost
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = wave >> 2;
    int nb = cg * 4 + (wave & 3);
    const bool active = nb * 16 < N;      // inactive waves still help staging and join the barriers
    if (!active) nb = 0;
    // this wave's k-blocks of chunk c: kb0 + 8c + 4*half + {0..3}
    const float4* wp = reinterpret_cast<const float4*>(Wpz) + ((long)nb * KB + kb0 + REC_HK * half) * 64 + lane;

    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    RecChunk<MT> xc;                      // MT float4 per thread: MT*16 rows x 32 float4 over 512 threads
    RecW wa, wb;
    auto xload = [&](int kb) __attribute__((always_inline)) { xc.load(Xz, ldx, kb, kb1, tid); };
    auto xstore = [&](int slot) __attribute__((always_inline)) { xc.store(Xs + (slot % NBUF) * SLOT, tid); };
    xload(kb0);
    wa.load(wp, kb0 + REC_HK * half, kb1);
    if (DBG & 1) wb = wa;
    // one chunk: publish the staged activations, prefetch the next chunk (activations + weights),
    // run the chunk's MFMAs.  `cur`/`nxt` are bound statically (two calls per loop trip) so that the
    // register arrays are never indexed at run time.
    auto chunk = [&](const RecW& cur, RecW& nxt, int kb, int slot) __attribute__((always_inline)) {
        if (!(DBG & 4) || kb == kb0) xstore(slot);
        __syncthreads();                  // also: every wave is done with the chunk before last, which lived in this slot
        const float* xs = Xs + (slot % NBUF) * SLOT + (lane & 15) * REC_LD + 4 * (lane >> 4) + REC_HK * half * 16;
        if (stamps != nullptr && tid == 0) {
            if (kb == kb0) { stamps[1] = wall_clock64(); stamps[7] = clock64(); }       // first activation chunk arrived
        }
        const int kn = kb + REC_CK;
        if (!(DBG & 4) && kn < kb1) xload(kn);
        if (!(DBG & 1)) nxt.load(wp + (long)REC_CK * 64, kn + REC_HK * half, kb1);
#pragma unroll
        for (int u = 0; u < REC_HK; ++u) {
            if (kb + REC_HK * half + u >= kb1) break;       // (wave-uniform) k-blocks beyond the slice: zero weights, nothing to add
            float4 a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = *reinterpret_cast<const float4*>(xs + mt * 16 * REC_LD + u * 16);
            const float4 b = cur.w[u];
            if (DBG & 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { acc[mt][0] += a[mt].x * b.x; acc[mt][1] += a[mt].y * b.y; acc[mt][2] += a[mt].z * b.z; acc[mt][3] += a[mt].w * b.w; }
                continue;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].x, b.x, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].y, b.y, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].z, b.z, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].w, b.w, acc[mt], 0, 0, 0);
        }
        if (NBUF == 1) __syncthreads();   // one buffer: nobody may publish the next chunk while a wave still reads this one
        wp += (long)REC_CK * 64;
    };
    for (int kb = kb0; kb < kb1; kb += 2 * REC_CK) {
        chunk(wa, wb, kb, 0);
        if (kb + REC_CK < kb1) chunk(wb, wa, kb + REC_CK, 1);
    }
    if (NBUF > 1) __syncthreads();        // the staging buffers are free: the epilogue reuses the first one
    if (stamps != nullptr && tid == 0) { stamps[2] = wall_clock64(); stamps[7] = clock64() - stamps[7]; }      // MFMAs done; [7] = shader clocks of the loop
    // Epilogue through LDS (Xs is free after the last barrier): the two K-halves are folded into a row-major
    // [row][64 columns] tile so that the slab is written with full-line float4 stores by all 8 waves (the MFMA
    // C layout would give 16 scalar stores of 64-byte pieces per lane, by half of the waves only).
    float* red = Xs;
    const int ecol = (wave & 3) * 16 + (lane & 15), erow = 4 * (lane >> 4);
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

template <int MT, int DBG>
__device__ __forceinline__ void rec_gemm_body(const float* __restrict__ Xz, long ldx, const float* __restrict__ Wpz,
                                              float* __restrict__ out_slab, int N, int KB, int kb0, int kb1, int cg) {
    const float* red = rec_gemm_tile<MT, DBG>(Xz, ldx, Wpz, N, KB, kb0, kb1, cg);
    if (DBG & 8) return;
    const int tid = threadIdx.x;
    float* out = out_slab + (long)cg * 64;
    const int ncols = min(64, N - cg * 64);      // N % 16 == 0: whole float4s
#pragma unroll
    for (int i = 0; i < (MT * 16 * 16 + REC_THREADS - 1) / REC_THREADS; ++i) {
        const int f = tid + REC_THREADS * i;
        const int row = f >> 4, c4 = f & 15;
        if (row < MT * 16 && 4 * c4 < ncols)
            *reinterpret_cast<float4*>(out + (long)row * N + 4 * c4) = *reinterpret_cast<const float4*>(&red[row * EP_LD + 4 * c4]);
    }
}

// (Measured and not adopted, round 3: capped at 96 registers (amdgpu_waves_per_eu(5, 5), 52 bytes of scratch) so that a workgroup
// fits beside TWO weight-gradient GEMM workgroups on a CU instead of one: 25 us per backward product instead of 20.  Restructured
// by hand to 95 registers without scratch -- one weight buffer, each fragment refilled with the next chunk's as soon as its
// value is copied out: 20.5-21.1 us against 21.0-21.1, pair 127.7 against 128.0-128.3.  Registers are not what it waits for.)
template <int MT, int DBG = 0>
__global__ __launch_bounds__(REC_THREADS) void rec_gemm_kernel(const float* __restrict__ X, long ldx,
                                                               const float* __restrict__ Wp,
                                                               float* __restrict__ part, int N, int KB, int KBS,
                                                               long xz, long wz, long pz, long ps, int prio) {
    // issue priority of the waves: 3 = the loops' critical (attention) chain, 2 = the other chain, 0 = everything else
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    const int s = blockIdx.y, z = blockIdx.z;
    const int kb0 = s * KBS;
    rec_gemm_body<MT, DBG>(X + (long)z * xz, ldx, Wp + (long)z * wz, part + (long)z * pz + (long)s * ps, N, KB, kb0,
                           min(KB, kb0 + KBS), blockIdx.x);
}

// ---- The same product from two SCALED fp16 planes per operand ("h2": recurrent_bf16.hip's NP = 2 arithmetic and weight image)
// in THIS kernel's geometry: 512 threads, 64-k chunks, one 18 KB staging buffer.  recurrent_bf16.hip's own form (256-k chunks,
// 68 KB of LDS, 156-210 registers) is 4.6 us faster than the fp32 product stand-alone and SLOWER in the backward loop, where
// its workgroups wait for room beside the weight-gradient GEMMs' (engine.REC_H2); this one keeps the footprint the fp32
// product has there.  Weights: Wp[N/16][KB32][2 planes][64 lanes] uint4 (lane l: B[k = 32 kb + 8 (l >> 4) + j][16 nb + (l & 15)]),
// the gt_absmax word of W in the image's trailer.  Wave w: columns [16 (w & 3), +16), the (w >> 2)-th 32-wide k-block of every
// chunk.  The activations are split on their way into LDS (two fp16 planes of [row][64 k + 8]).
constexpr int RH_LD = 64 + 8;             // halfs per staged row: 144 B, 16 rows' 16-byte fragments tile all 64 banks
constexpr int RH_NBUF_BWD = 1;            // staging buffers of the plain product (rec_gemm_h2_kernel): 2 = 37 KB, 9.7 against 10.2 us alone, nothing in the step
typedef _Float16 rh_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rh_f16x2 __attribute__((ext_vector_type(2)));
typedef float rh_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int rh_exp(unsigned maxbits) { return min(127, max(-126, 141 - (int)((maxbits >> 23) & 0xffu))); }
__device__ __forceinline__ void rh_split(float x, float y, float s, unsigned& p0, unsigned& p1) {
    float ra, rb;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(p0) : "s"(s), "v"(x));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(p0) : "s"(s), "v"(y));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ra) : "s"(s), "v"(x), "v"(p0));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "s"(s), "v"(y), "v"(p0));
    const rh_f16x2 q = __builtin_convertvector((rh_f32x2){ra, rb}, rh_f16x2);
    p1 = __builtin_bit_cast(unsigned, q);
}

// NBUF = 2 (the fused forward step, as rec_gemm_tile's): two staging buffers, ONE barrier per chunk, and room for the cell epilogue's
// scratch behind the folded tile.
template <int MT, int NBUF = 1>
__device__ __forceinline__ float* rec_h2_tile(const float* __restrict__ Xz, long ldx, int kvalid, const uint4* __restrict__ Wz,
                                              int N, int KB, int kb0, int kb1, int cg, float hs, int hk) {
    constexpr int PL = MT * 16 * RH_LD;                         // halfs per plane
    constexpr int NV = (MT * 16 * 16 + REC_THREADS - 1) / REC_THREADS;      // float4 per thread per chunk (MT*16 rows x 16 float4)
    constexpr int SLOT = 2 * PL;                                            // halfs per staging buffer (two planes)
    constexpr int STAGE = NBUF * SLOT * 2;                                  // bytes
    constexpr int EPI = (NBUF == 1 ? MT * 16 * REC_EP_LD : MT * 16 * (REC_EP_LD + 20) + 4) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char raw[STAGE > EPI ? STAGE : EPI];
    unsigned short* Xs0 = reinterpret_cast<unsigned short*>(raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = wave >> 2;
    int nb = cg * 4 + (wave & 3);
    const bool active = nb * 16 < N;
    if (!active) nb = 0;
    const uint4* wp = Wz + (((long)nb * KB + kb0 + half) * 2) * 64 + lane;     // this wave's block of chunk 0, plane 0
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 xv[NV];
    uint4 wa[2], wb[2];
    auto xload = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + REC_THREADS * i, row = f >> 4, c4 = f & 15;
            const int k = kb * 32 + 4 * c4;
            xv[i] = (row < MT * 16 && k + 3 < kvalid && (k >> 5) < kb1) ? *reinterpret_cast<const float4*>(Xz + (long)row * ldx + k)
                                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto xstore = [&](int slot) __attribute__((always_inline)) {
        unsigned short* Xs = Xs0 + (slot % NBUF) * SLOT;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + REC_THREADS * i, row = f >> 4, c4 = f & 15;
            if (row >= MT * 16) continue;
            unsigned a0, a1, b0, b1;
            rh_split(xv[i].x, xv[i].y, hs, a0, a1);
            rh_split(xv[i].z, xv[i].w, hs, b0, b1);
            *reinterpret_cast<uint2*>(&Xs[row * RH_LD + 4 * c4]) = make_uint2(a0, b0);
            *reinterpret_cast<uint2*>(&Xs[PL + row * RH_LD + 4 * c4]) = make_uint2(a1, b1);
        }
    };
    auto wload = [&](uint4 (&w)[2], const uint4* q, int kb) __attribute__((always_inline)) {
        // (blocks beyond the slice: a valid address, the value replaced by zeros -- a predicated load would serialise the stream)
        const bool ok = kb + half < kb1;
        const uint4* qq = ok ? q : Wz + lane;
        const uint4 t0 = qq[0], t1 = qq[64];
        w[0] = ok ? t0 : make_uint4(0u, 0u, 0u, 0u);
        w[1] = ok ? t1 : make_uint4(0u, 0u, 0u, 0u);
    };
    xload(kb0);
    wload(wa, wp, kb0);
    auto chunk = [&](const uint4 (&cur)[2], uint4 (&nxt)[2], int kb, int slot) __attribute__((always_inline)) {
        xstore(slot);
        __syncthreads();                  // (NBUF = 2: also, every wave is done with the chunk before last, which lived in this slot)
        const unsigned short* Xs = Xs0 + (slot % NBUF) * SLOT;
        const int kn = kb + 2;
        if (kn < kb1) xload(kn);
        wload(nxt, wp + 2 * 2 * 64, kn);
        if (kb + half < kb1) {                      // (wave-uniform)
            const unsigned short* xs = Xs + (lane & 15) * RH_LD + half * 32 + 8 * (lane >> 4);
            const rh_f16x8 b0 = __builtin_bit_cast(rh_f16x8, cur[0]), b1 = __builtin_bit_cast(rh_f16x8, cur[1]);
            rh_f16x8 a0[MT], a1[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                a0[mt] = __builtin_bit_cast(rh_f16x8, *reinterpret_cast<const uint4*>(xs + mt * 16 * RH_LD));
                a1[mt] = __builtin_bit_cast(rh_f16x8, *reinterpret_cast<const uint4*>(xs + PL + mt * 16 * RH_LD));
            }
            // smallest terms first; the MT row tiles lie between two MFMAs on one accumulator
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[mt], b1, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[mt], b0, acc[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[mt], b0, acc[mt], 0, 0, 0);
        }
        if (NBUF == 1) __syncthreads();   // one buffer: nobody may publish the next chunk while a wave still reads this one
        wp += 2 * 2 * 64;
    };
    for (int kb = kb0; kb < kb1; kb += 4) {
        chunk(wa, wb, kb, 0);
        if (kb + 2 < kb1) chunk(wb, wa, kb + 2, 1);
    }
    if (NBUF > 1) __syncthreads();        // the staging buffers are free: the epilogue reuses them
    // the operand scales leave (exactly), then the two K-halves are folded as in rec_gemm_tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][r] = ldexpf(acc[mt][r], -hk);
    float* red = reinterpret_cast<float*>(raw);
    const int ecol = (wave & 3) * 16 + (lane & 15), erow = 4 * (lane >> 4);
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

template <int MT>
__global__ __launch_bounds__(REC_THREADS) void rec_gemm_h2_kernel(const float* __restrict__ X, long ldx, int kvalid,
                                                                  const uint4* __restrict__ Wp, float* __restrict__ part, int N,
                                                                  int KB, int KBS, long xz, long pz, long ps, int prio,
                                                                  const unsigned long long* wmax, const unsigned long long* xmax,
                                                                  float xfloor) {
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    // (wave-uniform; agent-scope loads: the activations' word may come from atomics of the launch in front of this one)
    const unsigned wbits = (unsigned)__hip_atomic_load(wmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned xbits = xmax != nullptr ? (unsigned)__hip_atomic_load(xmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    xbits = max(xbits, __float_as_uint(xfloor));
    const int kw = rh_exp(__builtin_amdgcn_readfirstlane(wbits)), kx = rh_exp(__builtin_amdgcn_readfirstlane(xbits));
    const float hs = __uint_as_float((unsigned)(127 + kx) << 23);
    const int s = blockIdx.y, z = blockIdx.z, cg = blockIdx.x;
    const int kb0 = s * KBS;
    const float* red = rec_h2_tile<MT, RH_NBUF_BWD>(X + (long)z * xz, ldx, kvalid, Wp, N, KB, kb0, min(KB, kb0 + KBS), cg, hs, kw + kx);
    const int tid = threadIdx.x;
    float* out = part + (long)z * pz + (long)s * ps + (long)cg * 64;
    const int ncols = min(64, N - cg * 64);
#pragma unroll
    for (int i = 0; i < (MT * 16 * 16 + REC_THREADS - 1) / REC_THREADS; ++i) {
        const int f = tid + REC_THREADS * i;
        const int row = f >> 4, c4 = f & 15;
        if (row < MT * 16 && 4 * c4 < ncols)
            *reinterpret_cast<float4*>(out + (long)row * N + 4 * c4) = *reinterpret_cast<const float4*>(&red[row * EP_LD + 4 * c4]);
    }
}

// Two independent products with their own shapes in ONE launch (blockIdx.x is linear over both): the decoder loops
// issue the attention-LSTM and the decoder-LSTM products of a step together, so that they share the device as two
// co-resident workgroups per CU without any cross-stream synchronisation.
struct RecProblem {
    const float* X; long ldx;
    const float* Wp;
    float* part; long ps;
    int N, KB, KBS, groups, splits;
};
static_assert(sizeof(RecProblem) <= 1024, "RecProblem is a by-value kernel argument: keep it well below the 4 KB kernel-argument segment");

template <int MT>
__global__ __launch_bounds__(REC_THREADS) void rec_gemm2_kernel(RecProblem p0, RecProblem p1) {
    const int n0 = p0.groups * p0.splits;
    const bool second = (int)blockIdx.x >= n0;
    const RecProblem& p = second ? p1 : p0;
    const int id = second ? blockIdx.x - n0 : blockIdx.x;
    const int s = id / p.groups, cg = id - s * p.groups;
    const int kb0 = s * p.KBS;
    rec_gemm_body<MT, 0>(p.X, p.ldx, p.Wp, p.part + (long)s * p.ps, p.N, p.KB, kb0, min(p.KB, kb0 + p.KBS), cg);
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused recurrent product + LSTM cell (+ attention query projection): gt_rec_cell_fwd.
// The K-splits of one 64-column group (= the four gates of 16 hidden units, packing mode 3) meet INSIDE the launch:
// every split writes its partial tile as a slab with write-through (sc1) 16-byte stores, drains them, and draws a
// ticket from the group's counter; the workgroup that draws the last ticket adds the other slabs (sc1 loads, which
// bypass its CU's L1) to the tile it still holds in LDS and runs the cell for its 16 units x all rows: gates, cell
// state, dropout, the h scatter into the next operands -- and, for the attention LSTM, its share of the query
// projection h . Wq^T as one more slab.  This is the hand-off form "sc1 stores, every storing wave drained, one
// agent-scope atomic add per workgroup, the last adder loads with sc1 loads" of MI355X_MICROARCH.md (section
// visibility): no fence, no spinning, no assumption about placement or dispatch order.  It replaces the dependent
// launch of the cell kernel (launch + slab round trip, ~8 us) by ~3 us of epilogue in the last workgroup per group.
template <int MT>
__global__ __launch_bounds__(REC_THREADS) void rec_cell_kernel(const float* __restrict__ X, long ldx,
                                                               const float* __restrict__ Wp, float* part, int N, int KB,
                                                               int KBS, int KBlim, long xz, long pz, long ps, int prio, RecCellEpi e) {
    // KBlim <= KB: the k-blocks beyond it are not this launch's (e.Sx: an earlier launch has left their products as slabs)
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    const int cg = blockIdx.x, s = blockIdx.y, z = blockIdx.z;
    const int dir = z / e.nzb, rb = z - dir * e.nzb;
    const int kb0 = s * KBS;
    if (dir != 0) {
        X += e.dX; Wp += e.dW; part += e.dPart;
        rec_cell_dir1(e);
    }
    REC_STAMP(e, 0);
    float* red = rec_gemm_tile<MT, 0, 2>(X + (long)rb * xz, ldx, Wp, N, KB, kb0, min(KBlim, kb0 + KBS), cg,
                                          e.stamp_on ? g_rec_stamps_dev + ((long)blockIdx.x + (long)gridDim.x * blockIdx.y) * 8 : nullptr);
    rec_cell_epilogue<MT>(red, e, part + (long)rb * pz, ps, N, cg, s, rb, e.counters + (long)z * gridDim.x + cg);
}

// gt_rec_cell_fwd_h2 in this file's geometry: rec_h2_tile (two staging buffers) + the fused cell epilogue; one direction.
template <int MT>
__global__ __launch_bounds__(REC_THREADS) void rec_cell_h2_kernel(const float* __restrict__ X, long ldx, int kvalid,
                                                                  const uint4* __restrict__ Wp, float* part, int N, int KB,
                                                                  int KBS, long xz, long pz, long ps, int prio, RecCellEpi e,
                                                                  const unsigned long long* wmax, const unsigned long long* xmax,
                                                                  float xfloor) {
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    const unsigned wbits = (unsigned)__hip_atomic_load(wmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned xbits = xmax != nullptr ? (unsigned)__hip_atomic_load(xmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    xbits = max(xbits, __float_as_uint(xfloor));
    const int kw = rh_exp(__builtin_amdgcn_readfirstlane(wbits)), kx = rh_exp(__builtin_amdgcn_readfirstlane(xbits));
    const float hs = __uint_as_float((unsigned)(127 + kx) << 23);
    const int cg = blockIdx.x, s = blockIdx.y, rb = blockIdx.z;
    const int kb0 = s * KBS;
    float* red = rec_h2_tile<MT, 2>(X + (long)rb * xz, ldx, kvalid, Wp, N, KB, kb0, min(KB, kb0 + KBS), cg, hs, kw + kx);
    rec_cell_epilogue<MT>(red, e, part + (long)rb * pz, ps, N, cg, s, rb, e.counters + (long)rb * gridDim.x + cg);
}

// gt_rec_cellbwd: see rec_cell.h.  Same launch geometry as rec_cell_kernel: blockIdx = (64-unit column group, K-slice,
// 64-row block).
template <int MT>
__global__ __launch_bounds__(REC_THREADS) void rec_cellbwd_kernel(const float* __restrict__ X, long ldx,
                                                                  const float* __restrict__ Wp, float* part, int N, int KB,
                                                                  int KBS, long xz, long pz, long ps, int prio, RecCellBwdEpi e) {
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    const int cg = blockIdx.x, s = blockIdx.y, rb = blockIdx.z;
    const int kb0 = s * KBS;
    float* red = rec_gemm_tile<MT, 0, 2>(X + (long)rb * xz, ldx, Wp, N, KB, kb0, min(KB, kb0 + KBS), cg);
    rec_cellbwd_epilogue<MT>(red, e, part + (long)rb * pz, ps, N, cg, s, rb, e.counters + (long)rb * gridDim.x + cg);
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
    // blockIdx.y = 1 (second BiLSTM direction): element offsets added to the pointers above
    long z_part, z_pre, z_c, z_hprev, z_gates, z_hraw, z_h0;
};
static_assert(sizeof(CellFwd) <= 1024, "CellFwd is a by-value kernel argument: keep it well below the 4 KB kernel-argument segment");

// V = hidden units per thread (4: float4 accesses when every pointer / leading dimension allows it)
template <int V>
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(CellFwd a) {
    if (blockIdx.y != 0) {
        a.part.p += a.z_part;
        if (a.pre) a.pre += a.z_pre;
        a.c_prev += a.z_c; a.c_out += a.z_c;
        if (a.h_prev) a.h_prev += a.z_hprev;
        if (a.gates) a.gates += a.z_gates;
        if (a.h_raw) a.h_raw += a.z_hraw;
        if (a.h0) a.h0 += a.z_h0;
    }
    const long vid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int HV = a.H / V;
    if (vid >= (long)a.B * HV) return;
    const int b = (int)(vid / HV), j = (int)(vid - (long)b * HV) * V;
    const long idx = (long)b * a.H + j;
    const bool act = (a.active == nullptr) || (a.active[b] != 0);
    float gi[4][V];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long po = (long)b * a.part.ld + q * a.H + j;
#pragma unroll
        for (int e = 0; e < V; ++e) gi[q][e] = 0.f;
        if (a.pre != nullptr) {
            if (V == 4) {
                const float4 t = *reinterpret_cast<const float4*>(a.pre + (long)b * a.ld_pre + q * a.H + j);
                gi[q][0] = t.x; gi[q][1 % V] = t.y; gi[q][2 % V] = t.z; gi[q][3 % V] = t.w;
            } else {
                gi[q][0] = a.pre[(long)b * a.ld_pre + q * a.H + j];
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {                 // all slab loads of a gate are issued together
            if (s < a.part.n) {
                const float* p = a.part.p + (long)s * a.part.slab_stride + po;
                if (V == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(p);
                    gi[q][0] += t.x; gi[q][1 % V] += t.y; gi[q][2 % V] += t.z; gi[q][3 % V] += t.w;
                } else {
                    gi[q][0] += p[0];
                }
            }
        }
    }
    float cp[V], cn[V], hout[V], hcarry[V], gv[4][V];
    unsigned char kp[V];
    if (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(a.c_prev + idx);
        cp[0] = t.x; cp[1 % V] = t.y; cp[2 % V] = t.z; cp[3 % V] = t.w;
        if (a.keep != nullptr) {
            const uchar4 k = *reinterpret_cast<const uchar4*>(a.keep + idx);
            kp[0] = k.x; kp[1 % V] = k.y; kp[2 % V] = k.z; kp[3 % V] = k.w;
        }
    } else {
        cp[0] = a.c_prev[idx];
        if (a.keep != nullptr) kp[0] = a.keep[idx];
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const float i = sigmoidf_(gi[0][e]), f = sigmoidf_(gi[1][e]), g = tanhf_(gi[2][e]), o = sigmoidf_(gi[3][e]);
        gv[0][e] = i; gv[1][e] = f; gv[2][e] = g; gv[3][e] = o;
        float c = f * cp[e] + i * g;
        const float h = o * tanhf_(c);
        if (act) {
            hout[e] = (a.keep != nullptr) ? h * (kp[e] ? a.scale : 0.f) : h;
            hcarry[e] = hout[e];
        } else {
            // packed-sequence semantics: state passes through, the emitted output is zero
            c = cp[e];
            hout[e] = 0.f;
            hcarry[e] = (a.h_prev != nullptr) ? a.h_prev[(long)b * a.ld_hprev + j + e] : 0.f;
        }
        cn[e] = c;
    }
    auto put = [&](float* base, long off, const float (&vals)[V]) __attribute__((always_inline)) {
        if (V == 4) *reinterpret_cast<float4*>(base + off) = make_float4(vals[0], vals[1 % V], vals[2 % V], vals[3 % V]);
        else base[off] = vals[0];
    };
    if (a.gates != nullptr) {
#pragma unroll
        for (int q = 0; q < 4; ++q) put(a.gates, (long)b * 4 * a.H + q * a.H + j, gv[q]);
    }
    put(a.c_out, idx, cn);
    if (a.h_raw != nullptr) put(a.h_raw, idx, hout);          // emitted output (zero when inactive)
    if (a.h0 != nullptr) put(a.h0, (long)b * a.ld0 + j, hcarry);
    if (a.h1 != nullptr) put(a.h1, (long)b * a.ld1 + j, hcarry);
    if (a.h2 != nullptr) put(a.h2, (long)b * a.ld2 + j, hcarry);
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
    long z_da, z_direct, z_dc, z_gates, z_c, z_dgates;     // blockIdx.y = 1: second BiLSTM direction
};
static_assert(sizeof(CellBwd) <= 1024, "CellBwd is a by-value kernel argument: keep it well below the 4 KB kernel-argument segment");

template <int V>
__device__ __forceinline__ void vload(const float* p, float (&o)[V]) {
    if (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        o[0] = t.x; o[1 % V] = t.y; o[2 % V] = t.z; o[3 % V] = t.w;
    } else {
        o[0] = p[0];
    }
}
template <int V>
__device__ __forceinline__ void vstore(float* p, const float (&o)[V]) {
    if (V == 4) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1 % V], o[2 % V], o[3 % V]);
    else p[0] = o[0];
}
template <int V>
__device__ __forceinline__ void slab_acc(const GtSlabs& sl, long off, float (&o)[V]) {
    if (sl.p == nullptr) return;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        if (s < sl.n) {
            float t[V];
            vload<V>(sl.p + (long)s * sl.slab_stride + off, t);
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] += t[e];
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(CellBwd a) {
    if (blockIdx.y != 0) {
        if (a.da.p) a.da.p += a.z_da;
        if (a.dh_direct) a.dh_direct += a.z_direct;
        if (a.dc_next) a.dc_next += a.z_dc;
        a.dc_prev += a.z_dc;
        a.gates += a.z_gates; a.c_new += a.z_c; a.c_prev += a.z_c;
        a.dgates += a.z_dgates;
    }
    const long vid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int HV = a.H / V;
    if (vid >= (long)a.B * HV) return;
    const int b = (int)(vid / HV), j = (int)(vid - (long)b * HV) * V;
    const long idx = (long)b * a.H + j;
    // dh_carry: gradient w.r.t. the hidden state carried into later steps (slab sources),
    // dh_out  : gradient w.r.t. the emitted output of this step (direct source).
    float dh_carry[V], dh_out[V], dcn[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { dh_carry[e] = 0.f; dh_out[e] = 0.f; dcn[e] = 0.f; }
    slab_acc<V>(a.da, (long)b * a.da.ld + j, dh_carry);
    slab_acc<V>(a.db, (long)b * a.db.ld + j, dh_carry);
    slab_acc<V>(a.dc, (long)b * a.dc.ld + j, dh_carry);
    if (a.dh_direct != nullptr) vload<V>(a.dh_direct + (long)b * a.ld_direct + j, dh_out);
    if (a.dc_next != nullptr) vload<V>(a.dc_next + idx, dcn);
    const bool act = (a.active == nullptr) || (a.active[b] != 0);
    float* dg = a.dgates + (long)b * 4 * a.H + j;
    float dgi[V], dgf[V], dgg[V], dgo[V], dcp[V], dpass[V];
    if (!act) {
        // inactive packed-sequence row: state passed through untouched, output was zero
#pragma unroll
        for (int e = 0; e < V; ++e) { dgi[e] = dgf[e] = dgg[e] = dgo[e] = 0.f; dcp[e] = dcn[e]; dpass[e] = dh_carry[e]; }
    } else {
        float gi[V], gf[V], gg[V], go[V], cn[V], cp[V];
        const float* gt = a.gates + (long)b * 4 * a.H + j;
        vload<V>(gt, gi); vload<V>(gt + a.H, gf); vload<V>(gt + 2 * a.H, gg); vload<V>(gt + 3 * a.H, go);
        vload<V>(a.c_new + idx, cn);
        vload<V>(a.c_prev + idx, cp);
        unsigned char kp[V];
#pragma unroll
        for (int e = 0; e < V; ++e) kp[e] = 1;
        if (a.keep != nullptr) {
            if (V == 4) {
                const uchar4 k = *reinterpret_cast<const uchar4*>(a.keep + idx);
                kp[0] = k.x; kp[1 % V] = k.y; kp[2 % V] = k.z; kp[3 % V] = k.w;
            } else {
                kp[0] = a.keep[idx];
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float dh = dh_carry[e] + dh_out[e];
            if (a.keep != nullptr) dh *= (kp[e] ? a.scale : 0.f);
            const float tc = tanhf_(cn[e]);
            const float dcell = dcn[e] + dh * go[e] * (1.f - tc * tc);
            dgi[e] = dcell * gg[e] * gi[e] * (1.f - gi[e]);
            dgf[e] = dcell * cp[e] * gf[e] * (1.f - gf[e]);
            dgg[e] = dcell * gi[e] * (1.f - gg[e] * gg[e]);
            dgo[e] = dh * tc * go[e] * (1.f - go[e]);
            dcp[e] = dcell * gf[e];
            dpass[e] = 0.f;
        }
    }
    vstore<V>(dg, dgi); vstore<V>(dg + a.H, dgf); vstore<V>(dg + 2 * a.H, dgg); vstore<V>(dg + 3 * a.H, dgo);
    vstore<V>(a.dc_prev + idx, dcp);
    if (a.dh_pass != nullptr) vstore<V>(a.dh_pass + idx, dpass);
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

// gate-interleaved packing for gt_lstm_step_fused: KB covers gt_fused_k_padded(K) columns (zeros beyond K)
extern "C" int gt_pack_gates(const float* W, int64_t ldw, int32_t H, int32_t K, float* Wg, void* stream) {
    GT_REQUIRE(W && Wg && H > 0 && H % 4 == 0 && K > 0, GT_E_ARG, "gt_pack_gates: H=%d must be a multiple of 4", H);
    GT_REQUIRE(aligned16(Wg), GT_E_ALIGN, "gt_pack_gates: Wg must be 16-byte aligned");
    const int N = 4 * H;
    const int KB = (int)(gt_fused_k_padded(K) / 16);
    const long total = (long)(N / 16) * KB * 64;
    const int blocks = (int)std::min<long>(cdiv64(total, 256), 4096);
    hipLaunchKernelGGL(pack_rec_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, (long)ldw, 2, N, N, K, KB, Wg);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_rec_gemm(const float* X, int64_t ldx, const float* Wp, float* part, int32_t mtiles,
                           int32_t N, int32_t K, int32_t splits, int64_t part_sstride, int32_t nz,
                           int64_t x_zstride, int64_t w_zstride, int64_t part_zstride, void* stream) {
    return gt_rec_gemm_prio(X, ldx, Wp, part, mtiles, N, K, splits, part_sstride, nz, x_zstride, w_zstride, part_zstride, 0,
                            stream);
}

// prio != 0: the waves run at raised issue priority (s_setprio): used by the decoder loops for the products of the
// attention chain, which is the serial critical path, while the other chain's kernels share the CUs
int gt_rec_gemm_prio(const float* X, int64_t ldx, const float* Wp, float* part, int32_t mtiles, int32_t N, int32_t K,
                     int32_t splits, int64_t part_sstride, int32_t nz, int64_t x_zstride, int64_t w_zstride,
                     int64_t part_zstride, int prio, void* stream) {
    GT_REQUIRE(X && Wp && part, GT_E_ARG, "gt_rec_gemm: NULL pointer");
    GT_REQUIRE(mtiles >= 1 && mtiles <= 4 && N > 0 && N % 16 == 0 && K > 0 && splits >= 1 && nz >= 1, GT_E_ARG,
               "gt_rec_gemm: bad sizes mtiles=%d N=%d K=%d splits=%d", mtiles, N, K, splits);
    const int KB = cdiv(K, 16);
    GT_REQUIRE(ldx >= (int64_t)KB * 16 && ldx % 4 == 0 && aligned16(X) && aligned16(Wp) && x_zstride % 4 == 0 &&
                   w_zstride % 4 == 0,
               GT_E_ALIGN, "gt_rec_gemm: X needs ldx >= 16*ceil(K/16) (=%d), ldx%%4==0 and 16-byte alignment", KB * 16);
    GT_REQUIRE(part_sstride >= (int64_t)mtiles * 16 * N, GT_E_ARG, "gt_rec_gemm: part_sstride too small");
    GT_REQUIRE(aligned16(part) && part_sstride % 4 == 0 && part_zstride % 4 == 0, GT_E_ALIGN,
               "gt_rec_gemm: part must be 16-byte aligned with slab strides that are multiples of 4");
    const int KBS = cdiv(KB, splits);
    const int ns = cdiv(KB, KBS);
    GT_REQUIRE(ns == splits, GT_E_ARG, "gt_rec_gemm: splits=%d leaves empty slabs for K=%d (use <= %d)", splits, K, ns);
    dim3 grid(cdiv(N, 64), splits, nz);
    hipStream_t st = (hipStream_t)stream;
    GtProfScope prof(GT_FAM_REC_GEMM, 2.0 * 16 * mtiles * (double)N * K * nz,
                     4.0 * nz * ((double)N * K + 16.0 * mtiles * K + 16.0 * mtiles * N * splits), st, true);
    // reserve enough LDS that at most ONE workgroup fits a CU: the grid is sized to the CU count and
    // two co-resident workgroups would serialise on one CU's matrix pipes while another CU idles
    static const int lds_pad = gt_env_item("GANTRON_DEBUG", "rec_lds_pad", 0);
#define GT_REC(MT)                                                                                    \
    GT_LAUNCH_TIMED((rec_gemm_kernel<MT>), grid, dim3(REC_THREADS), lds_pad, st, prof.e0, prof.e1, X, (long)ldx, \
                          Wp, part, N, KB, KBS, (long)x_zstride, (long)w_zstride, (long)part_zstride, (long)part_sstride, prio)
    static const int dbg = gt_env_item("GANTRON_DEBUG", "rec", 0);
    if (dbg && mtiles == 4) {   // diagnostic variants (wrong results by construction)
#define GT_RECD(D)                                                                                       \
    GT_LAUNCH_TIMED((rec_gemm_kernel<4, D>), grid, dim3(REC_THREADS), lds_pad, st, prof.e0, prof.e1, X, (long)ldx, \
                          Wp, part, N, KB, KBS, (long)x_zstride, (long)w_zstride, (long)part_zstride, (long)part_sstride, prio)
        switch (dbg) {
            case 1: GT_RECD(1); break;
            case 2: GT_RECD(2); break;
            case 3: GT_RECD(3); break;
            case 4: GT_RECD(4); break;
            case 5: GT_RECD(5); break;
            case 6: GT_RECD(6); break;
            case 8: GT_RECD(8); break;
            default: GT_RECD(7); break;
        }
#undef GT_RECD
        GT_LAUNCH_CHECK();
        return 0;
    }
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

// gt_rec_gemm_h2 in this file's geometry (rec_gemm_h2_kernel): Wp is a gt_pack_rec_weight_h2 image (its trailer holds the word
// the weights' scale comes from), xmax / xfloor bound the activations
int gt_rec_gemm_h2_lean(const float* X, int64_t ldx, const void* Wp, float* part, int32_t mtiles, int32_t N, int32_t K,
                        int32_t splits, int64_t part_sstride, int32_t nz, int64_t x_zstride, int64_t part_zstride,
                        const uint64_t* xmax, float xfloor, int prio, void* stream) {
    GT_REQUIRE(X && Wp && part, GT_E_ARG, "gt_rec_gemm_h2: NULL pointer");
    GT_REQUIRE(mtiles >= 1 && mtiles <= 4 && N > 0 && N % 16 == 0 && K >= 4 && splits >= 1 && nz >= 1, GT_E_ARG,
               "gt_rec_gemm_h2: bad sizes mtiles=%d N=%d K=%d splits=%d", mtiles, N, K, splits);
    GT_REQUIRE(xmax != nullptr || xfloor > 0.f, GT_E_ARG, "gt_rec_gemm_h2: no bound for the activations (xmax word or xfloor > 0)");
    const int KB = cdiv(K, 32);
    const int kvalid = cdiv(K, 16) * 16;
    GT_REQUIRE(ldx >= kvalid && ldx % 4 == 0 && aligned16(X) && aligned16(Wp) && x_zstride % 4 == 0, GT_E_ALIGN,
               "gt_rec_gemm_h2: X needs ldx >= 16*ceil(K/16) (=%d), ldx%%4==0 and 16-byte alignment", kvalid);
    GT_REQUIRE(part_sstride >= (int64_t)mtiles * 16 * N && aligned16(part) && part_sstride % 4 == 0 && part_zstride % 4 == 0,
               GT_E_ARG, "gt_rec_gemm_h2: part slabs too small or misaligned");
    const int KBS = cdiv(KB, splits);
    GT_REQUIRE(cdiv(KB, KBS) == splits, GT_E_ARG, "gt_rec_gemm_h2: splits=%d leaves empty slabs for K=%d", splits, K);
    const unsigned long long* wmax = reinterpret_cast<const unsigned long long*>(
        static_cast<const unsigned short*>(Wp) + 2 * (int64_t)(N / 16) * KB * 64 * 8);
    dim3 grid(cdiv(N, 64), splits, nz);
    hipStream_t st = (hipStream_t)stream;
    GtProfScope prof(GT_FAM_REC_GEMM, 2.0 * 16 * mtiles * (double)N * K * nz,
                     4.0 * nz * ((double)N * K + 16.0 * mtiles * K + 16.0 * mtiles * N * splits), st, true);
#define GT_RECH(MT)                                                                                               \
    GT_LAUNCH_TIMED((rec_gemm_h2_kernel<MT>), grid, dim3(REC_THREADS), 0, st, prof.e0, prof.e1, X, (long)ldx, kvalid, \
                          reinterpret_cast<const uint4*>(Wp), part, N, KB, KBS, (long)x_zstride, (long)part_zstride,   \
                          (long)part_sstride, prio, wmax, (const unsigned long long*)xmax, xfloor)
    switch (mtiles) {
        case 1: GT_RECH(1); break;
        case 2: GT_RECH(2); break;
        case 3: GT_RECH(3); break;
        default: GT_RECH(4); break;
    }
#undef GT_RECH
    GT_LAUNCH_CHECK();
    return 0;
}

// packing for gt_rec_cell_fwd (mode 3): W is the (4H x K) LSTM weight in PyTorch gate order
extern "C" int gt_pack_rec_cell_weight(const float* W, int64_t ldw, int32_t H, int32_t K, float* Wp, void* stream) {
    GT_REQUIRE(W && Wp && H > 0 && H % 16 == 0 && K > 0, GT_E_ARG, "gt_pack_rec_cell_weight: H=%d must be a multiple of 16", H);
    GT_REQUIRE(aligned16(Wp), GT_E_ALIGN, "gt_pack_rec_cell_weight: Wp must be 16-byte aligned");
    const int N = 4 * H, KB = cdiv(K, 16);
    const long total = (long)(N / 16) * KB * 64;
    const int blocks = (int)std::min<long>(cdiv64(total, 256), 4096);
    hipLaunchKernelGGL(pack_rec_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, (long)ldw, 3, N, N, K, KB, Wp);
    GT_LAUNCH_CHECK();
    return 0;
}

struct RecCellEpi;
int gt_launch_rec_cell_bf16(const float* X, int64_t ldx, const void* Wp, float* part, int32_t Bp, int32_t N, int32_t K,
                            int32_t splits, int32_t ndir, int prio, const RecCellEpi& e, hipEvent_t e0, hipEvent_t e1,
                            void* stream, int planes);      // recurrent_bf16.hip

static long long* g_rec_stamps = nullptr;
static int g_rec_stamp_wgs = 0;
static bool g_rec_stamps_set = false;
// diagnostics: copies the stamps of the last stamped launch (8 per workgroup) to `out` (HOST, room for 8 * 4096) and returns
// the number of workgroups; synchronises the device
extern "C" int gt_debug_rec_stamps(int64_t* out) {
    if (g_rec_stamps == nullptr || out == nullptr) return 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(out, g_rec_stamps, sizeof(long long) * 8 * (size_t)g_rec_stamp_wgs, hipMemcpyDeviceToHost);
    return g_rec_stamp_wgs;
}

int gt_rec_cell_fwd_z(const float* X, int64_t ldx, const void* Wpv, int wbf16, float* part, uint32_t* counters, int32_t Bp, int32_t B,
                      int32_t H, int32_t K, int32_t splits, const float* pre, int64_t ld_pre, const float* c_prev,
                      const uint8_t* keep, float p_drop, float* gates, float* c_out, float* h_raw, float* h0, int64_t ld0,
                      float* h1, int64_t ld1, float* h2, int64_t ld2, const float* Wq, int64_t ldwq, float* qpart,
                      int64_t q_sstride, int32_t A, int32_t Apad, const int32_t* active, const float* h_prev, int64_t ld_hprev,
                      int32_t ndir, const int64_t* dz, int32_t prio, void* stream, int32_t k_own, int32_t extra_slabs) {
    const float* Wp = static_cast<const float*>(Wpv);
    // k_own > 0: only the first k_own (a multiple of 16) of the K reduction columns are this launch's; the products of the
    // rest lie in `extra_slabs` slabs behind the K-slices' own (part + (splits + j) * Bp * 4H), written by an earlier launch
    GT_REQUIRE(k_own == 0 || (k_own > 0 && k_own % 16 == 0 && k_own <= K && wbf16 == 0 && ndir == 1), GT_E_ARG,
               "gt_rec_cell_fwd: k_own=%d must be a multiple of 16 <= K (fp32, one direction)", k_own);
    GT_REQUIRE(extra_slabs >= 0 && extra_slabs + splits <= 16 && (extra_slabs == 0 || wbf16 == 0), GT_E_ARG,
               "gt_rec_cell_fwd: extra_slabs=%d (with %d splits: at most 16 slabs, fp32 only)", extra_slabs, splits);
    GT_REQUIRE(X && Wp && part && counters && c_prev && c_out, GT_E_ARG, "gt_rec_cell_fwd: NULL pointer");
    GT_REQUIRE(B > 0 && Bp >= B && ((Bp <= 64 && Bp % 16 == 0) || Bp % 64 == 0) && H > 0 && H % 16 == 0 && K > 0 && splits >= 1 &&
                   splits <= 16, GT_E_ARG, "gt_rec_cell_fwd: bad sizes B=%d Bp=%d H=%d K=%d splits=%d", B, Bp, H, K, splits);
    GT_REQUIRE(p_drop >= 0.f && p_drop < 1.f, GT_E_ARG, "gt_rec_cell_fwd: p_drop");
    GT_REQUIRE(ndir == 1 || (ndir == 2 && dz != nullptr && keep == nullptr && h1 == nullptr && h2 == nullptr && Wq == nullptr),
               GT_E_ARG, "gt_rec_cell_fwd: the paired launch takes offsets, no dropout, one h destination and no query");
    const int N = 4 * H, KB = cdiv(K, 16);
    GT_REQUIRE(ldx >= (int64_t)KB * 16 && ldx % 4 == 0 && aligned16(X) && aligned16(Wp) && aligned16(part), GT_E_ALIGN,
               "gt_rec_cell_fwd: X needs ldx >= 16*ceil(K/16), ldx %% 4 == 0 and 16-byte alignment");
    auto ok4 = [](const void* p, int64_t ld) { return p == nullptr || (aligned16(p) && ld % 4 == 0); };
    GT_REQUIRE(ok4(pre, ld_pre) && aligned16(c_prev) && (keep == nullptr || (((uintptr_t)keep) & 3) == 0) && ok4(gates, 0) &&
                   aligned16(c_out) && ok4(h_raw, 0) && ok4(h0, ld0) && ok4(h1, ld1) && ok4(h2, ld2) && ok4(Wq, ldwq) &&
                   ok4(h_prev, ld_hprev),
               GT_E_ALIGN, "gt_rec_cell_fwd: every pointer must be 16-byte aligned with leading dimensions %% 4 == 0");
    if (Wq != nullptr)
        GT_REQUIRE(qpart && A > 0 && Apad >= A && Apad % 16 == 0 && Apad <= 256 && q_sstride >= (int64_t)Bp * Apad, GT_E_LIMIT,
                   "gt_rec_cell_fwd: attention_dim %d (padded %d) must fit 16 column blocks", A, Apad);
    const int KBlim = k_own > 0 ? k_own / 16 : KB;
    const int KBS = cdiv(KBlim, splits);
    GT_REQUIRE(cdiv(KBlim, KBS) == splits, GT_E_ARG, "gt_rec_cell_fwd: splits=%d leaves empty slabs for K=%d", splits, 16 * KBlim);
    const int mt = Bp <= 64 ? Bp / 16 : 4, nzb = Bp <= 64 ? 1 : Bp / 64;
    RecCellEpi e;
    e.Sx = extra_slabs;
    e.pre = pre; e.ld_pre = ld_pre; e.c_prev = c_prev; e.c_out = c_out; e.keep = keep; e.scale = 1.0f / (1.0f - p_drop);
    e.gates = gates; e.h_raw = h_raw; e.h0 = h0; e.ld0 = ld0; e.h1 = h1; e.ld1 = ld1; e.h2 = h2; e.ld2 = ld2;
    // (query slabs laid out [row][column group][Apad] instead, so that a row's 64 slabs are one contiguous 32 KB for the attention
    // step that sums them: no difference, 126.5-127.6 against 127.0 ms per pair, round 3)
    e.Wq = Wq; e.ldwq = ldwq; e.qpart = qpart; e.q_sstride = q_sstride; e.counters = counters;
    e.B = B; e.H = H; e.A = A; e.Apad = Apad; e.S = splits;
    e.active = active; e.h_prev = h_prev; e.ld_hprev = ld_hprev; e.nzb = nzb;
    e.dX = e.dW = e.dPart = e.dPre = e.dC = e.dGates = e.dHraw = e.dH0 = e.dHprev = 0;
    e.stamp_on = 0;
    {   // diagnostics: GANTRON_DEBUG=rec_stamps keeps the phase stamps of the LAST attention-LSTM launch (gt_debug_rec_stamps)
        static const int want = gt_env_item("GANTRON_DEBUG", "rec_stamps", 0);
        if (want && Wq != nullptr && ndir == 1) {
            if (g_rec_stamps == nullptr) {
                (void)hipMalloc(&g_rec_stamps, sizeof(long long) * 8 * 4096);
                fprintf(stderr, "[rec_stamps] N=%d K=%d KB=%d splits=%d Bp=%d\n", N, K, KB, splits, Bp);
            }
            if (!g_rec_stamps_set) {
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rec_stamps_dev), &g_rec_stamps, sizeof(g_rec_stamps));
                g_rec_stamps_set = true;
            }
            e.stamp_on = wbf16 ? 0 : 1;         // (the fp32 kernel of this translation unit)
            g_rec_stamp_wgs = (N / 64) * splits;
            (void)hipMemsetAsync(g_rec_stamps, 0, sizeof(long long) * 8 * (size_t)g_rec_stamp_wgs, (hipStream_t)stream);
        }
    }
    if (ndir == 2) {
        e.dX = dz[0]; e.dW = dz[1]; e.dPart = dz[2]; e.dPre = dz[3]; e.dC = dz[4]; e.dGates = dz[5]; e.dHraw = dz[6];
        e.dH0 = dz[7]; e.dHprev = dz[8];
        for (int i = 0; i < 9; ++i) GT_REQUIRE(dz[i] % 4 == 0, GT_E_ALIGN, "gt_rec_cell_fwd: direction offsets must be multiples of 4");
    }
    dim3 grid(N / 64, splits, nzb * ndir);
    hipStream_t st = (hipStream_t)stream;
    const double Kown = 16.0 * KBlim < K ? 16.0 * KBlim : (double)K;        // the reduction columns this launch multiplies
    GtProfScope prof(GT_FAM_REC_CELL, ndir * (2.0 * Bp * (double)N * Kown + (Wq ? 2.0 * Bp * (double)H * A : 0.0)),
                     ndir * ((wbf16 == 3 ? 6.0 : wbf16 == 1 ? 2.0 : 4.0) * (double)N * Kown +
                             4.0 * ((double)Bp * Kown + (double)Bp * N * ((splits > 1 ? 2 * splits - 1 : 0) + extra_slabs) +
                                    (double)B * H * 12)), st, true);
    // (GANTRON_DEBUG=rec_h2_form=<bits>: bit 0 = the forward step in THIS file's geometry -- rec_cell_h2_kernel: 64-k chunks,
    // 37 KB of LDS -- instead of recurrent_bf16.hip's 256-k-chunk form, which is the faster one in the forward loop: 110.5
    // against 113.0 ms per pair where this one gives 110.3 against 111.6; bit 1 = the backward product in recurrent_bf16.hip's
    // form, which loses in the backward loop)
    static const int h2_form = gt_env_item("GANTRON_DEBUG", "rec_h2_form", 0);
    if (wbf16 == 2 && (h2_form & 1)) {  // two scaled fp16 planes in this file's geometry (rec_cell_h2_kernel)
        GT_REQUIRE(ndir == 1 && k_own == 0 && extra_slabs == 0, GT_E_ARG, "gt_rec_cell_fwd_h2: one direction, the whole reduction");
        const GtRecH2X hx = g_rec_h2x;
        g_rec_h2x = {nullptr, 0.f};
        GT_REQUIRE(hx.xmax != nullptr || hx.xfloor > 0.f, GT_E_ARG, "gt_rec_cell_fwd_h2: no bound for the activations (xmax word or xfloor > 0)");
        const int KB32 = cdiv(K, 32), KBS32 = cdiv(KB32, splits), kvalid = KB * 16;
        GT_REQUIRE(cdiv(KB32, KBS32) == splits, GT_E_ARG, "gt_rec_cell_fwd_h2: splits=%d leaves empty slabs for K=%d", splits, K);
        const unsigned long long* wmax = reinterpret_cast<const unsigned long long*>(
            static_cast<const unsigned short*>(Wpv) + 2 * (int64_t)(N / 16) * KB32 * 64 * 8);
        const long psl = (long)Bp * N;
        dim3 grid2(N / 64, splits, nzb);
#define GT_RCH(MT)                                                                                                     \
    GT_LAUNCH_TIMED((rec_cell_h2_kernel<MT>), grid2, dim3(REC_THREADS), 0, st, prof.e0, prof.e1, X, (long)ldx, kvalid,   \
                          reinterpret_cast<const uint4*>(Wpv), part, N, KB32, KBS32, (long)(64 * ldx), (long)64 * N, psl, prio, e, \
                          wmax, (const unsigned long long*)hx.xmax, hx.xfloor)
        switch (mt) {
            case 1: GT_RCH(1); break;
            case 2: GT_RCH(2); break;
            case 3: GT_RCH(3); break;
            default: GT_RCH(4); break;
        }
#undef GT_RCH
        GT_LAUNCH_CHECK();
        return 0;
    }
    if (wbf16) return gt_launch_rec_cell_bf16(X, ldx, Wpv, part, Bp, N, K, splits, ndir, prio, e, prof.e0, prof.e1, stream,
                                              wbf16 == 3 ? 3 : wbf16 == 2 ? 2 : 1);
    const long ps = (long)Bp * N;
#define GT_RC(MT)                                                                                                      \
    GT_LAUNCH_TIMED((rec_cell_kernel<MT>), grid, dim3(REC_THREADS), 0, st, prof.e0, prof.e1, X, (long)ldx, Wp, part,  \
                          N, KB, KBS, KBlim, (long)(64 * ldx), (long)64 * N, ps, prio, e)
    switch (mt) {
        case 1: GT_RC(1); break;
        case 2: GT_RC(2); break;
        case 3: GT_RC(3); break;
        default: GT_RC(4); break;
    }
#undef GT_RC
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_rec_cell_fwd(const float* X, int64_t ldx, const float* Wp, float* part, uint32_t* counters, int32_t Bp,
                               int32_t B, int32_t H, int32_t K, int32_t splits, const float* pre, int64_t ld_pre,
                               const float* c_prev, const uint8_t* keep, float p_drop, float* gates, float* c_out,
                               float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2,
                               const float* Wq, int64_t ldwq, float* qpart, int64_t q_sstride, int32_t A, int32_t Apad,
                               int32_t prio, void* stream) {
    return gt_rec_cell_fwd_z(X, ldx, Wp, 0, part, counters, Bp, B, H, K, splits, pre, ld_pre, c_prev, keep, p_drop, gates, c_out,
                             h_raw, h0, ld0, h1, ld1, h2, ld2, Wq, ldwq, qpart, q_sstride, A, Apad, nullptr, nullptr, 0, 1,
                             nullptr, prio, stream);
}

extern "C" int gt_rec_cell_fwd_part(const float* X, int64_t ldx, const float* Wp, float* part, uint32_t* counters, int32_t Bp,
                                    int32_t B, int32_t H, int32_t K, int32_t splits, const float* pre, int64_t ld_pre,
                                    const float* c_prev, const uint8_t* keep, float p_drop, float* gates, float* c_out,
                                    float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2,
                                    const float* Wq, int64_t ldwq, float* qpart, int64_t q_sstride, int32_t A, int32_t Apad,
                                    int32_t k_own, int32_t extra_slabs, int32_t prio, void* stream) {
    return gt_rec_cell_fwd_z(X, ldx, Wp, 0, part, counters, Bp, B, H, K, splits, pre, ld_pre, c_prev, keep, p_drop, gates, c_out,
                             h_raw, h0, ld0, h1, ld1, h2, ld2, Wq, ldwq, qpart, q_sstride, A, Apad, nullptr, nullptr, 0, 1,
                             nullptr, prio, stream, k_own, extra_slabs);
}

extern "C" int gt_rec_cell_fwd_bf16(const float* X, int64_t ldx, const void* Wp, float* part, uint32_t* counters, int32_t Bp,
                                    int32_t B, int32_t H, int32_t K, int32_t splits, const float* pre, int64_t ld_pre,
                                    const float* c_prev, const uint8_t* keep, float p_drop, float* gates, float* c_out,
                                    float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2,
                                    const float* Wq, int64_t ldwq, float* qpart, int64_t q_sstride, int32_t A, int32_t Apad,
                                    int32_t prio, void* stream) {
    return gt_rec_cell_fwd_z(X, ldx, Wp, 1, part, counters, Bp, B, H, K, splits, pre, ld_pre, c_prev, keep, p_drop, gates, c_out,
                             h_raw, h0, ld0, h1, ld1, h2, ld2, Wq, ldwq, qpart, q_sstride, A, Apad, nullptr, nullptr, 0, 1,
                             nullptr, prio, stream);
}

extern "C" int gt_rec_cell_fwd_h2(const float* X, int64_t ldx, const void* Wp, float* part, uint32_t* counters, int32_t Bp,
                                  int32_t B, int32_t H, int32_t K, int32_t splits, const float* pre, int64_t ld_pre,
                                  const float* c_prev, const uint8_t* keep, float p_drop, float* gates, float* c_out,
                                  float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2,
                                  const float* Wq, int64_t ldwq, float* qpart, int64_t q_sstride, int32_t A, int32_t Apad,
                                  const uint64_t* xmax, float xfloor, int32_t prio, void* stream) {
    g_rec_h2x = {xmax, xfloor};
    return gt_rec_cell_fwd_z(X, ldx, Wp, 2, part, counters, Bp, B, H, K, splits, pre, ld_pre, c_prev, keep, p_drop, gates, c_out,
                             h_raw, h0, ld0, h1, ld1, h2, ld2, Wq, ldwq, qpart, q_sstride, A, Apad, nullptr, nullptr, 0, 1,
                             nullptr, prio, stream);
}

extern "C" int gt_rec_cell_fwd_x3(const float* X, int64_t ldx, const void* Wp, float* part, uint32_t* counters, int32_t Bp,
                                  int32_t B, int32_t H, int32_t K, int32_t splits, const float* pre, int64_t ld_pre,
                                  const float* c_prev, const uint8_t* keep, float p_drop, float* gates, float* c_out,
                                  float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2,
                                  const float* Wq, int64_t ldwq, float* qpart, int64_t q_sstride, int32_t A, int32_t Apad,
                                  int32_t prio, void* stream) {
    return gt_rec_cell_fwd_z(X, ldx, Wp, 3, part, counters, Bp, B, H, K, splits, pre, ld_pre, c_prev, keep, p_drop, gates, c_out,
                             h_raw, h0, ld0, h1, ld1, h2, ld2, Wq, ldwq, qpart, q_sstride, A, Apad, nullptr, nullptr, 0, 1,
                             nullptr, prio, stream);
}

extern "C" int gt_rec_cellbwd(const float* X, int64_t ldx, const float* Wp, float* part, uint32_t* counters, int32_t Bp,
                              int32_t B, int32_t H, int32_t K, int32_t splits, const float* dh_direct, int64_t ld_direct,
                              float* dc, const float* gates, const float* c_new, const float* c_prev, const uint8_t* keep,
                              float p_drop, float* dgates, int32_t prio, void* stream) {
    GT_REQUIRE(X && Wp && part && counters && dc && gates && c_new && c_prev && dgates, GT_E_ARG, "gt_rec_cellbwd: NULL pointer");
    GT_REQUIRE(B > 0 && Bp >= B && ((Bp <= 64 && Bp % 16 == 0) || Bp % 64 == 0) && H > 0 && H % 64 == 0 && K > 0 && splits >= 1 &&
                   splits <= 16, GT_E_ARG, "gt_rec_cellbwd: bad sizes B=%d Bp=%d H=%d (%% 64) K=%d splits=%d", B, Bp, H, K, splits);
    GT_REQUIRE(p_drop >= 0.f && p_drop < 1.f, GT_E_ARG, "gt_rec_cellbwd: p_drop");
    const int N = H, KB = cdiv(K, 16);
    GT_REQUIRE(ldx >= (int64_t)KB * 16 && ldx % 4 == 0 && aligned16(X) && aligned16(Wp) && aligned16(part), GT_E_ALIGN,
               "gt_rec_cellbwd: X needs ldx >= 16*ceil(K/16), ldx %% 4 == 0 and 16-byte alignment");
    GT_REQUIRE(aligned16(dc) && aligned16(gates) && aligned16(c_new) && aligned16(c_prev) && aligned16(dgates) &&
                   (dh_direct == nullptr || (aligned16(dh_direct) && ld_direct % 4 == 0)) &&
                   (keep == nullptr || (((uintptr_t)keep) & 3) == 0),
               GT_E_ALIGN, "gt_rec_cellbwd: every pointer must be 16-byte aligned with leading dimensions %% 4 == 0");
    const int KBS = cdiv(KB, splits);
    GT_REQUIRE(cdiv(KB, KBS) == splits, GT_E_ARG, "gt_rec_cellbwd: splits=%d leaves empty slabs for K=%d", splits, K);
    const int mt = Bp <= 64 ? Bp / 16 : 4, nzb = Bp <= 64 ? 1 : Bp / 64;
    RecCellBwdEpi e;
    e.dh_direct = dh_direct; e.ld_direct = ld_direct; e.gates = gates; e.c_new = c_new; e.c_prev = c_prev; e.keep = keep;
    e.scale = 1.0f / (1.0f - p_drop); e.dc = dc; e.dgates = dgates; e.counters = counters; e.B = B; e.H = H; e.S = splits;
    dim3 grid(N / 64, splits, nzb);
    hipStream_t st = (hipStream_t)stream;
    GtProfScope prof(GT_FAM_REC_GEMM, 2.0 * Bp * (double)N * K,
                     4.0 * ((double)N * K + (double)Bp * K + (double)Bp * N * (splits > 1 ? 2 * splits - 1 : 0) + (double)B * H * 14), st,
                     true);
    const long ps = (long)Bp * N;
#define GT_RCB(MT)                                                                                                     \
    GT_LAUNCH_TIMED((rec_cellbwd_kernel<MT>), grid, dim3(REC_THREADS), 0, st, prof.e0, prof.e1, X, (long)ldx, Wp, part,  \
                          N, KB, KBS, (long)(64 * ldx), (long)64 * N, ps, prio, e)
    switch (mt) {
        case 1: GT_RCB(1); break;
        case 2: GT_RCB(2); break;
        case 3: GT_RCB(3); break;
        default: GT_RCB(4); break;
    }
#undef GT_RCB
    GT_LAUNCH_CHECK();
    return 0;
}

static int rec_problem(RecProblem& p, const float* X, int64_t ldx, const float* Wp, float* part, int N, int K, int splits,
                       int64_t sstride, int mtiles, const char* which) {
    GT_REQUIRE(X && Wp && part && N > 0 && N % 16 == 0 && K > 0 && splits >= 1, GT_E_ARG, "gt_rec_gemm2: bad problem %s", which);
    const int KB = cdiv(K, 16);
    GT_REQUIRE(ldx >= (int64_t)KB * 16 && ldx % 4 == 0 && aligned16(X) && aligned16(Wp) && aligned16(part) && sstride % 4 == 0,
               GT_E_ALIGN, "gt_rec_gemm2: alignment of problem %s", which);
    GT_REQUIRE(sstride >= (int64_t)mtiles * 16 * N, GT_E_ARG, "gt_rec_gemm2: slab stride of problem %s", which);
    const int KBS = cdiv(KB, splits);
    GT_REQUIRE(cdiv(KB, KBS) == splits, GT_E_ARG, "gt_rec_gemm2: splits=%d leaves empty slabs for K=%d (%s)", splits, K, which);
    p.X = X; p.ldx = ldx; p.Wp = Wp; p.part = part; p.ps = sstride; p.N = N; p.KB = KB; p.KBS = KBS; p.groups = cdiv(N, 64);
    p.splits = splits;
    return 0;
}

extern "C" int gt_rec_gemm2(const float* X0, int64_t ldx0, const float* Wp0, float* part0, int32_t N0, int32_t K0,
                            int32_t splits0, int64_t sstride0, const float* X1, int64_t ldx1, const float* Wp1,
                            float* part1, int32_t N1, int32_t K1, int32_t splits1, int64_t sstride1, int32_t mtiles,
                            void* stream) {
    GT_REQUIRE(mtiles >= 1 && mtiles <= 4, GT_E_ARG, "gt_rec_gemm2: mtiles");
    RecProblem p0, p1;
    int rc;
    if ((rc = rec_problem(p0, X0, ldx0, Wp0, part0, N0, K0, splits0, sstride0, mtiles, "0")) != 0) return rc;
    if ((rc = rec_problem(p1, X1, ldx1, Wp1, part1, N1, K1, splits1, sstride1, mtiles, "1")) != 0) return rc;
    hipStream_t st = (hipStream_t)stream;
    const double fl = 2.0 * 16 * mtiles * ((double)N0 * K0 + (double)N1 * K1);
    GtProfScope prof(GT_FAM_REC_GEMM, fl, 4.0 * ((double)N0 * K0 + (double)N1 * K1 + 16.0 * mtiles * (K0 + K1) +
                                                 16.0 * mtiles * ((double)N0 * splits0 + (double)N1 * splits1)), st);
    dim3 grid(p0.groups * p0.splits + p1.groups * p1.splits);
    switch (mtiles) {
        case 1: hipLaunchKernelGGL((rec_gemm2_kernel<1>), grid, dim3(REC_THREADS), 0, st, p0, p1); break;
        case 2: hipLaunchKernelGGL((rec_gemm2_kernel<2>), grid, dim3(REC_THREADS), 0, st, p0, p1); break;
        case 3: hipLaunchKernelGGL((rec_gemm2_kernel<3>), grid, dim3(REC_THREADS), 0, st, p0, p1); break;
        default: hipLaunchKernelGGL((rec_gemm2_kernel<4>), grid, dim3(REC_THREADS), 0, st, p0, p1); break;
    }
    GT_LAUNCH_CHECK();
    return 0;
}

int gt_cell_fwd_z(GtSlabs part, const float* pre, int64_t ld_pre, const float* c_prev, const float* h_prev,
                  int64_t ld_hprev, const uint8_t* keep, float p_drop, const int32_t* active, float* gates, float* c_out,
                  float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2, int32_t B, int32_t H,
                  int nz, const int64_t* z, void* stream) {
    GT_REQUIRE(part.p && c_prev && c_out && B > 0 && H > 0 && part.n >= 1, GT_E_ARG, "gt_lstm_cell_fwd: bad argument");
    GT_REQUIRE(p_drop >= 0.f && p_drop < 1.f, GT_E_ARG, "gt_lstm_cell_fwd: p_drop");
    GT_REQUIRE(part.n <= 16, GT_E_LIMIT, "gt_lstm_cell_fwd: at most 16 slabs");
    GT_REQUIRE(nz == 1 || (nz == 2 && z != nullptr && keep == nullptr && h1 == nullptr && h2 == nullptr), GT_E_ARG,
               "gt_lstm_cell_fwd: paired launch takes offsets, no dropout and one h destination");
    CellFwd a;
    a.part = part; a.pre = pre; a.ld_pre = ld_pre; a.c_prev = c_prev; a.h_prev = h_prev; a.ld_hprev = ld_hprev;
    a.keep = keep; a.scale = 1.0f / (1.0f - p_drop); a.active = active; a.gates = gates; a.c_out = c_out;
    a.h_raw = h_raw; a.h0 = h0; a.ld0 = ld0; a.h1 = h1; a.ld1 = ld1; a.h2 = h2; a.ld2 = ld2; a.B = B; a.H = H;
    a.z_part = z ? z[0] : 0; a.z_pre = z ? z[1] : 0; a.z_c = z ? z[2] : 0; a.z_hprev = z ? z[3] : 0;
    a.z_gates = z ? z[4] : 0; a.z_hraw = z ? z[5] : 0; a.z_h0 = z ? z[6] : 0;
    const long n = (long)B * H;
    GtProfScope prof(GT_FAM_CELL_FWD, 0.0, nz * 4.0 * n * (4.0 * part.n + 4 + 1 + 4 + 1 + 3), (hipStream_t)stream);
    auto ok4 = [](const void* p, int64_t ld) { return p == nullptr || (aligned16(p) && ld % 4 == 0); };
    const bool vec = H % 4 == 0 && ok4(part.p, part.ld) && part.slab_stride % 4 == 0 && ok4(pre, ld_pre) && aligned16(c_prev) &&
                     (keep == nullptr || (((uintptr_t)keep) & 3) == 0) && ok4(gates, 0) && aligned16(c_out) && ok4(h_raw, 0) &&
                     ok4(h0, ld0) && ok4(h1, ld1) && ok4(h2, ld2) &&
                     (a.z_part | a.z_pre | a.z_c | a.z_hprev | a.z_gates | a.z_hraw | a.z_h0) % 4 == 0;
    if (vec)
        hipLaunchKernelGGL((lstm_cell_fwd_kernel<4>), dim3((unsigned)cdiv64(n / 4, 64), nz), dim3(64), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((lstm_cell_fwd_kernel<1>), dim3((unsigned)cdiv64(n, 256), nz), dim3(256), 0, (hipStream_t)stream, a);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_lstm_cell_fwd(GtSlabs part, const float* pre, int64_t ld_pre, const float* c_prev,
                                const float* h_prev, int64_t ld_hprev, const uint8_t* keep, float p_drop,
                                const int32_t* active, float* gates, float* c_out, float* h_raw, float* h0,
                                int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2, int32_t B,
                                int32_t H, void* stream) {
    return gt_cell_fwd_z(part, pre, ld_pre, c_prev, h_prev, ld_hprev, keep, p_drop, active, gates, c_out, h_raw, h0, ld0,
                         h1, ld1, h2, ld2, B, H, 1, nullptr, stream);
}

int gt_cell_bwd_z(GtSlabs dh_a, GtSlabs dh_b, GtSlabs dh_c, const float* dh_direct, int64_t ld_direct,
                  const float* dc_next, const float* gates, const float* c_new, const float* c_prev, const uint8_t* keep,
                  float p_drop, const int32_t* active, float* dgates, float* dc_prev, float* dh_pass, int32_t B, int32_t H,
                  int nz, const int64_t* z, void* stream) {
    GT_REQUIRE(gates && c_new && c_prev && dgates && dc_prev && B > 0 && H > 0, GT_E_ARG, "gt_lstm_cell_bwd: bad argument");
    GT_REQUIRE(dh_a.n <= 16 && dh_b.n <= 16 && dh_c.n <= 16, GT_E_LIMIT, "gt_lstm_cell_bwd: at most 16 slabs per source");
    GT_REQUIRE(nz == 1 || (nz == 2 && z != nullptr && keep == nullptr && dh_b.p == nullptr && dh_c.p == nullptr &&
                           dh_pass == nullptr), GT_E_ARG, "gt_lstm_cell_bwd: paired launch takes offsets and one slab source");
    CellBwd a;
    a.da = dh_a; a.db = dh_b; a.dc = dh_c; a.dh_direct = dh_direct; a.ld_direct = ld_direct; a.dc_next = dc_next;
    a.gates = gates; a.c_new = c_new; a.c_prev = c_prev; a.keep = keep; a.scale = 1.0f / (1.0f - p_drop);
    a.active = active; a.dgates = dgates; a.dc_prev = dc_prev; a.dh_pass = dh_pass; a.B = B; a.H = H;
    a.z_da = z ? z[0] : 0; a.z_direct = z ? z[1] : 0; a.z_dc = z ? z[2] : 0; a.z_gates = z ? z[3] : 0;
    a.z_c = z ? z[4] : 0; a.z_dgates = z ? z[5] : 0;
    const long n = (long)B * H;
    GtProfScope prof(GT_FAM_CELL_BWD, 0.0, nz * 4.0 * n * (dh_a.n + dh_b.n + dh_c.n + 1 + 1 + 4 + 2 + 4 + 1), (hipStream_t)stream);
    auto ok4 = [](const void* p, int64_t ld) { return p == nullptr || (aligned16(p) && ld % 4 == 0); };
    auto sl4 = [&](const GtSlabs& sl) { return sl.p == nullptr || (aligned16(sl.p) && sl.ld % 4 == 0 && sl.slab_stride % 4 == 0); };
    const bool vec = H % 4 == 0 && sl4(dh_a) && sl4(dh_b) && sl4(dh_c) && ok4(dh_direct, ld_direct) && ok4(dc_next, 0) &&
                     aligned16(gates) && aligned16(c_new) && aligned16(c_prev) && (keep == nullptr || (((uintptr_t)keep) & 3) == 0) &&
                     aligned16(dgates) && aligned16(dc_prev) && ok4(dh_pass, 0) &&
                     (a.z_da | a.z_direct | a.z_dc | a.z_gates | a.z_c | a.z_dgates) % 4 == 0;
    if (vec)
        hipLaunchKernelGGL((lstm_cell_bwd_kernel<4>), dim3((unsigned)cdiv64(n / 4, 64), nz), dim3(64), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((lstm_cell_bwd_kernel<1>), dim3((unsigned)cdiv64(n, 256), nz), dim3(256), 0, (hipStream_t)stream, a);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_lstm_cell_bwd(GtSlabs dh_a, GtSlabs dh_b, GtSlabs dh_c, const float* dh_direct,
                                int64_t ld_direct, const float* dc_next, const float* gates, const float* c_new,
                                const float* c_prev, const uint8_t* keep, float p_drop, const int32_t* active,
                                float* dgates, float* dc_prev, float* dh_pass, int32_t B, int32_t H,
                                void* stream) {
    return gt_cell_bwd_z(dh_a, dh_b, dh_c, dh_direct, ld_direct, dc_next, gates, c_new, c_prev, keep, p_drop, active,
                         dgates, dc_prev, dh_pass, B, H, 1, nullptr, stream);
}
