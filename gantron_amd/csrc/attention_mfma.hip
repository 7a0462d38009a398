// Location-sensitive attention energies, forward and backward, for the reference's default
// dimensions (attention_dim 128, 32 location filters, kernel 31) on the matrix cores.
//
// Per workgroup (batch row b, 32 encoder positions) every contraction of the step is a small GEMM
// on v_mfma_f32_16x16x4_f32 (exact fp32, same rate as the fp32 VALU peak but ~1000x fewer LDS
// operand reads than a scalar MAC loop, which is what bounded the VALU version):
//   G1 conv   : locf[32 x 32]   = im2col(window)[32 x 62] . Wc^T[62 x 32]
//   G2 dense  : s[32 x 128]     = locf[32 x 32] . Wd^T[32 x 128]          (+ q + pm, tanh, .v)
//   G3        : dlocf[32 x 32]  = ds[32 x 128] . Wd[128 x 32]
//   G4        : dWd[128 x 32]   = ds^T[128 x 32] . locf[32 x 32]
//   G5        : dWc[32 x 62]    = dlocf^T[32 x 32] . im2col(window)[32 x 62]
//   G6        : g[32 x 62]      = dlocf[32 x 32] . Wc[32 x 62]   (then d window = diagonal sums of g)
// Operand fragments are read from LDS as float4 along k where the operand is k-contiguous (the four
// MFMAs of a 16-wide k-step then take elements j=0..3, i.e. k = 16*s + 4*(lane>>4) + j for BOTH
// operands) and as four ds_read_b32 otherwise; row strides are chosen so that every such read is
// bank-conflict free (see the LD* constants).
#include <stddef.h>
#include "common.h"
#include "attn_ctx_bwd.h"
#include "rec_tail.h"

namespace {

constexpr int CT = 32, A = 128, F = 32, KC = 31, KK = 2 * KC;   // KK = 62 (c,k) pairs
constexpr int LDC = 68;    // Wc rows [f][kk], zero padded to 64 (+4: 4*68 % 32 == 16)
constexpr int LDD = 36;    // Wd rows [d][f]
constexpr int LDL = 36;    // locf / dlocf rows [tl][f]
constexpr int LDS_ = 132;  // ds rows [tl][d]
constexpr int LDG = 66;    // g rows [tl][kk]
constexpr int WL = 64;     // window row length (32 + 30 used)
constexpr int PER_MFMA = A + 256 * 16 + 256 * 8;   // accumulator slab of one workgroup in register order

struct SharedM {
    alignas(16) float wc[F * LDC];
    alignas(16) float wd[A * LDD];
    alignas(16) float locf[CT * LDL];
    alignas(16) float dlocf[CT * LDL];
    alignas(16) float ds[CT * LDS_];
    alignas(16) float g[CT * LDG];
    float win[2][WL];
    float q[A];
    float v[A];
    float de[CT];
    float epart[4][CT];
    float red[16];
};

__device__ __forceinline__ f32x4 mfma4(const float4& a, const float4& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// four strided scalars as one fragment
__device__ __forceinline__ float4 ld4s(const float* p, int stride) {
    return make_float4(p[0], p[stride], p[2 * stride], p[3 * stride]);
}

// wpack (optional): the LDS image of [wc | wd] prepared once per loop by gt_attn_pack_weights, so that staging is
// 7 float4 copies per thread instead of 26 scalar loads with index arithmetic
constexpr int WPACK_F4 = (F * LDC + A * LDD) / 4;
static_assert((F * LDC) % 4 == 0 && (A * LDD) % 4 == 0 && offsetof(SharedM, wd) == sizeof(float) * F * LDC, "[wc | wd] is one block");

__device__ __forceinline__ void stage_common(SharedM& S, const GtAttnDims& d, const float* Wc, const float* Wd,
                                             const float* v, const float* w_prev, const float* w_cum, int b, int t0,
                                             const float* wpack, bool with_windows = true) {
    if (wpack != nullptr) {
        const float4* src = reinterpret_cast<const float4*>(wpack);
        float4* dst = reinterpret_cast<float4*>(&S.wc[0]);
#pragma unroll
        for (int i = 0; i < (WPACK_F4 + 255) / 256; ++i) {
            const int j = min((int)threadIdx.x + 256 * i, WPACK_F4 - 1);      // clamped: a few threads copy the last one twice
            dst[j] = src[j];
        }
    } else {
    for (int i = threadIdx.x; i < F * LDC; i += 256) {
        const int f = i / LDC, kk = i - f * LDC;
        const float wv = Wc[f * KK + min(kk, KK - 1)];        // unconditional load, selected below: predicated
        S.wc[i] = kk < KK ? wv : 0.f;                         // loads would force s_waitcnt vmcnt(0) everywhere
    }
    for (int i = threadIdx.x; i < A * F; i += 256) S.wd[(i >> 5) * LDD + (i & 31)] = Wd[i];
    }
    if (threadIdx.x < A) S.v[threadIdx.x] = v[threadIdx.x];
    if (with_windows && threadIdx.x < 2 * WL) {
        const int c = threadIdx.x / WL, p = threadIdx.x - c * WL;
        const int t = t0 + p - (KC - 1) / 2;
        const float* src = c == 0 ? w_prev : w_cum;
        const float wv = src[(long)b * d.Ti + min(max(t, 0), d.Ti - 1)];
        S.win[c][p] = (p < CT + KC - 1 && t >= 0 && t < d.Ti) ? wv : 0.f;
    }
}

__global__ void attn_pack_weights_kernel(const float* Wc, const float* Wd, float* out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < F * LDC + A * LDD; i += gridDim.x * blockDim.x) {
        float val = 0.f;
        if (i < F * LDC) {
            const int f = i / LDC, kk = i - f * LDC;
            if (kk < KK) val = Wc[f * KK + kk];
        } else {
            const int j = i - F * LDC, dd = j / LDD, f = j - dd * LDD;
            if (f < F) val = Wd[dd * F + f];
        }
        out[i] = val;
    }
}

// im2col element (tl, kk) of the window; kk >= 62 is multiplied by zero weights
__device__ __forceinline__ float im2col(const SharedM& S, int tl, int kk) {
    kk = min(kk, KK - 1);
    const int c = kk >= KC ? 1 : 0;
    return S.win[c][tl + kk - c * KC];
}

// G1: wave w computes the 16x16 tile (mt = w>>1, nt = w&1) of locf and stores it to LDS.
__device__ __forceinline__ void conv_tile(SharedM& S, int wave, int lane) {
    const int mt = wave >> 1, nt = wave & 1;
    const int r = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int k0 = 16 * s4 + 4 * kq;
        const float4 a = make_float4(im2col(S, 16 * mt + r, k0), im2col(S, 16 * mt + r, k0 + 1),
                                     im2col(S, 16 * mt + r, k0 + 2), im2col(S, 16 * mt + r, k0 + 3));
        const float4 bb = ld4(&S.wc[(16 * nt + r) * LDC + k0]);
        acc = mfma4(a, bb, acc);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) S.locf[(16 * mt + 4 * kq + e) * LDL + 16 * nt + r] = acc[e];
}

// G2: wave w owns attention channels [32w, 32w+32): tiles (mt, nl) -> acc[mt][nl]
__device__ __forceinline__ void dense_tiles(const SharedM& S, int wave, int lane, f32x4 (&acc)[2][2]) {
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nl = 0; nl < 2; ++nl) acc[mt][nl] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s4 = 0; s4 < 2; ++s4) {
        const int k0 = 16 * s4 + 4 * kq;
        float4 a[2], bb[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a[mt] = ld4(&S.locf[(16 * mt + r) * LDL + k0]);
#pragma unroll
        for (int nl = 0; nl < 2; ++nl) bb[nl] = ld4(&S.wd[(32 * wave + 16 * nl + r) * LDD + k0]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nl = 0; nl < 2; ++nl) acc[mt][nl] = mfma4(a[mt], bb[nl], acc[mt][nl]);
    }
}

__global__ __launch_bounds__(256) void attn_energies_fwd_mfma(GtAttnDims d, GtSlabs qpart, const float* w_prev,
                                                              const float* w_cum, const float* pm, const float* Wc,
                                                              const float* Wd, const float* v, const int* lengths,
                                                              float* q_out, float* energies, const float* wpack,
                                                              float* u_out) {
    __builtin_amdgcn_s_setprio(3);      // the attention chain is the serial critical path of the decoder loops: its waves
                                        // issue ahead of the co-resident waves of the other chain and of the leaf GEMMs
    __shared__ SharedM S;
    // grid (B, chunks): the chunks of a batch row share an XCD (workgroup ids congruent mod 8), so the row's query
    // slabs are fetched into one L2
    const int b = blockIdx.x, chunk = blockIdx.y, t0 = chunk * CT;
    if (lengths != nullptr && t0 >= lengths[b]) {      // chunk entirely in the padding: energies are -inf
        if (threadIdx.x < CT && t0 + threadIdx.x < d.Ti) energies[(long)b * d.Ti + t0 + threadIdx.x] = -INFINITY;
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    // processed-memory tile [32 positions x 128 channels]: coalesced float4 loads issued before anything else,
    // parked in LDS (S.ds is free in the forward kernel) and picked up in accumulator layout after the barrier
    float4 pm4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, tl = f >> 5, c4 = f & 31;
        pm4[i] = *reinterpret_cast<const float4*>(pm + ((long)b * d.Ti + min(t0 + tl, d.Ti - 1)) * A + 4 * c4);
    }
    // query = sum of the slabs of the query product (up to H/16 = 64 of them).  Vector form: thread (sg, c4) takes slabs
    // sg, sg + 8, ... at channels 4 c4 .. 4 c4 + 3, so that ALL slab loads of the workgroup are in flight together (one
    // memory round trip); the 8 partial rows are folded through LDS (S.g is free in the forward kernel).
    const bool qvec = ((qpart.ld | qpart.slab_stride) & 3) == 0 && (reinterpret_cast<uintptr_t>(qpart.p) & 15) == 0;
    float qv = 0.f;
    if (qvec) {
        const int sg = threadIdx.x >> 5, c4 = threadIdx.x & 31;
        const float* qp = qpart.p + (long)b * qpart.ld + 4 * c4;
        float4 qa = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = 0; base < qpart.n; base += 64) {
            float4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const float4*>(qp + (long)min(base + sg + 8 * j, qpart.n - 1) * qpart.slab_stride);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w = base + sg + 8 * j < qpart.n ? 1.f : 0.f;
                qa.x += w * t[j].x; qa.y += w * t[j].y; qa.z += w * t[j].z; qa.w += w * t[j].w;
            }
        }
        *reinterpret_cast<float4*>(&S.g[sg * A + 4 * c4]) = qa;
    } else {
        const float* qp = qpart.p + (long)b * qpart.ld + (threadIdx.x & (A - 1));
#pragma unroll 8
        for (int s2 = 0; s2 < qpart.n; ++s2) qv += qp[(long)s2 * qpart.slab_stride];
    }
    stage_common(S, d, Wc, Wd, v, w_prev, w_cum, b, t0, wpack);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, tl = f >> 5, c4 = f & 31;
        *reinterpret_cast<float4*>(&S.ds[tl * LDS_ + 4 * c4]) = pm4[i];
    }
    if (qvec) {
        __syncthreads();
        if (threadIdx.x < A) {
            qv = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) qv += S.g[j * A + threadIdx.x];
        }
    }
    if (threadIdx.x < A) {
        S.q[threadIdx.x] = qv;
        if (chunk == 0 && q_out != nullptr) q_out[(long)b * A + threadIdx.x] = qv;
    }
    __syncthreads();
    conv_tile(S, wave, lane);
    __syncthreads();
    f32x4 acc[2][2];
    dense_tiles(S, wave, lane, acc);
    // epilogue: e[tl] += sum_d v[d] * tanh(s + q[d] + pm)
    float ep[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sum = 0.f;
#pragma unroll
            for (int nl = 0; nl < 2; ++nl) {
                const int dd = 32 * wave + 16 * nl + r;
                const float u = tanhf_(acc[mt][nl][e] + S.q[dd] + S.ds[(16 * mt + 4 * kq + e) * LDS_ + dd]);
                sum += S.v[dd] * u;
                // hoisted backward (gt_attn_param_grads): keep tanh(.) of every step instead of recomputing it
                if (u_out != nullptr && t0 + 16 * mt + 4 * kq + e < d.Ti)
                    u_out[((long)b * d.Ti + t0 + 16 * mt + 4 * kq + e) * A + dd] = u;
            }
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 4, 64);
            sum += __shfl_xor(sum, 8, 64);
            ep[mt][e] = sum;
        }
    if (r == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) S.epart[wave][16 * mt + 4 * kq + e] = ep[mt][e];
    }
    __syncthreads();
    if (threadIdx.x < CT) {
        const int t = t0 + threadIdx.x;
        if (t < d.Ti) {
            float e = (S.epart[0][threadIdx.x] + S.epart[1][threadIdx.x]) + (S.epart[2][threadIdx.x] + S.epart[3][threadIdx.x]);
            if (lengths != nullptr && t >= lengths[b]) e = -INFINITY;
            energies[(long)b * d.Ti + t] = e;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One launch for a whole forward attention step (model.py:70-91): energies of this workgroup's 32 positions, then --
// INSIDE the launch -- the workgroups of a batch row exchange their energies, every one of them normalises the row
// (softmax) and accumulates the context for a slice of the E columns over all valid positions.
//
// Hand-off (MI355X_MICROARCH.md, visibility, "sc1 stores + drained + one agent-scope atomic per workgroup; the consumer
// polls the counter with sc1 loads and reads the payload with sc1 loads"): the 32 energies are written through, the
// storing wave drains, ONE lane adds to the row's arrival counter; ONE lane polls that counter (relaxed agent loads,
// s_sleep between polls) until the row's NCv workgroups of this step have arrived, the workgroup's barrier releases the
// others, every energy is then read with an sc1 load.  Counters are monotonic over the loop (target = NCv * step): zeroed
// once per loop call, never reset.  Progress: only the <= NCv workgroups of one row wait for each other; workgroup ids are
// laid out so that the rows of a group of 8 are contiguous (at most one group is ever partially dispatched) and the
// chunks of a row share an XCD (id % 8 == row % 8); nothing outside this launch ever waits for it, so slots always free
// up.  The poll is nevertheless bounded: on time-out the workgroup raises `timeout_flag` and carries on with what it has
// (the step is then wrong, the launch still ends).
struct AttnStepFwd {
    GtAttnDims d; GtSlabs qpart;
    const float* w_prev; const float* w_cum; const float* pm; const float* Wc; const float* Wd; const float* v;
    const int* lengths; float* q_out; const float* wpack; float* u_out;
    float* erow; unsigned* row_cnt; unsigned step1; int* timeout_flag;
    const float* memory; long ld_mem; float* w_out; const float* cum_in; float* cum_out;
    float* c0; long ld0; float* c1; long ld1; float* c2; long ld2;
    int NC;
};
static_assert(sizeof(AttnStepFwd) <= 1024, "AttnStepFwd is a by-value kernel argument: keep it well below the 4 KB kernel-argument segment");

__global__ __launch_bounds__(256) void attn_step_fwd_mfma(AttnStepFwd a) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ SharedM S;
    const GtAttnDims& d = a.d;
    // workgroup id -> (row, chunk): groups of 8 rows; inside a group id = chunk * 8 + (row % 8)
    const int L = blockIdx.x;
    const int grp = L / (8 * a.NC), rem = L - grp * 8 * a.NC;
    const int chunk = rem >> 3, b = grp * 8 + (rem & 7);
    if (b >= d.B) return;
    const int t0 = chunk * CT;
    const int Tv = a.lengths != nullptr ? min(a.lengths[b], d.Ti) : d.Ti;      // valid positions of this row
    if (t0 >= Tv) return;                                                    // chunk entirely in the padding: not a member
    const int NCv = (Tv + CT - 1) / CT;                                      // members of this row's exchange
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    // ---- phase 1: energies of positions [t0, t0 + 32)  (as attn_energies_fwd_mfma)
    float4 pm4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, tl = f >> 5, c4 = f & 31;
        pm4[i] = *reinterpret_cast<const float4*>(a.pm + ((long)b * d.Ti + min(t0 + tl, d.Ti - 1)) * A + 4 * c4);
    }
    const bool qvec = ((a.qpart.ld | a.qpart.slab_stride) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.qpart.p) & 15) == 0;
    float qv = 0.f;
    if (qvec) {
        const int sg = threadIdx.x >> 5, c4 = threadIdx.x & 31;
        const float* qp = a.qpart.p + (long)b * a.qpart.ld + 4 * c4;
        float4 qa = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = 0; base < a.qpart.n; base += 64) {
            float4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const float4*>(qp + (long)min(base + sg + 8 * j, a.qpart.n - 1) * a.qpart.slab_stride);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w = base + sg + 8 * j < a.qpart.n ? 1.f : 0.f;
                qa.x += w * t[j].x; qa.y += w * t[j].y; qa.z += w * t[j].z; qa.w += w * t[j].w;
            }
        }
        *reinterpret_cast<float4*>(&S.g[sg * A + 4 * c4]) = qa;
    } else {
        const float* qp = a.qpart.p + (long)b * a.qpart.ld + (threadIdx.x & (A - 1));
#pragma unroll 8
        for (int s2 = 0; s2 < a.qpart.n; ++s2) qv += qp[(long)s2 * a.qpart.slab_stride];
    }
    stage_common(S, d, a.Wc, a.Wd, a.v, a.w_prev, a.w_cum, b, t0, a.wpack);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, tl = f >> 5, c4 = f & 31;
        *reinterpret_cast<float4*>(&S.ds[tl * LDS_ + 4 * c4]) = pm4[i];
    }
    if (qvec) {
        __syncthreads();
        if (threadIdx.x < A) {
            qv = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) qv += S.g[j * A + threadIdx.x];
        }
    }
    if (threadIdx.x < A) {
        S.q[threadIdx.x] = qv;
        if (chunk == 0 && a.q_out != nullptr) a.q_out[(long)b * A + threadIdx.x] = qv;
    }
    __syncthreads();
    conv_tile(S, wave, lane);
    __syncthreads();
    f32x4 acc[2][2];
    dense_tiles(S, wave, lane, acc);
    float ep[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sum = 0.f;
#pragma unroll
            for (int nl = 0; nl < 2; ++nl) {
                const int dd = 32 * wave + 16 * nl + r;
                const float u = tanhf_(acc[mt][nl][e] + S.q[dd] + S.ds[(16 * mt + 4 * kq + e) * LDS_ + dd]);
                sum += S.v[dd] * u;
                if (a.u_out != nullptr && t0 + 16 * mt + 4 * kq + e < d.Ti)
                    a.u_out[((long)b * d.Ti + t0 + 16 * mt + 4 * kq + e) * A + dd] = u;
            }
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 4, 64);
            sum += __shfl_xor(sum, 8, 64);
            ep[mt][e] = sum;
        }
    if (r == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) S.epart[wave][16 * mt + 4 * kq + e] = ep[mt][e];
    }
    __syncthreads();
    // ---- publish this chunk's energies (wave 0), then wait for the row
    float* erow = a.erow + (long)b * d.Ti;
    if (wave == 0) {
        if (lane < CT && t0 + lane < d.Ti) {
            float e = (S.epart[0][lane] + S.epart[1][lane]) + (S.epart[2][lane] + S.epart[3][lane]);
            if (t0 + lane >= Tv) e = -INFINITY;
            __hip_atomic_store(erow + t0 + lane, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through (sc1)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the storing wave drains (its u_out stores included)
        if (lane == 0) {
            unsigned* cnt = a.row_cnt + b;
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)NCv * a.step1;
            unsigned spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 21)) {                    // seconds: something is badly wrong, do not hang the GPU
                    if (a.timeout_flag != nullptr) *a.timeout_flag = 1;
                    break;
                }
            }
        }
    }
    __syncthreads();
    // ---- phase 2: softmax over the row (every member computes it; energies read with sc1 loads), S.g holds w
    float* w = S.g;                                  // Ti <= 256 (host check)
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < Tv; t += 256) {
        const float e = __hip_atomic_load(erow + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w[t] = e;
        mx = fmaxf(mx, e);
    }
    mx = block_max(mx, S.red);
    float sum = 0.f;
    for (int t = threadIdx.x; t < Tv; t += 256) {
        const float ex = __expf(w[t] - mx);
        w[t] = ex;
        sum += ex;
    }
    sum = block_sum(sum, S.red);
    const float inv = 1.0f / sum;
    for (int t = threadIdx.x; t < Tv; t += 256) w[t] *= inv;
    __syncthreads();
    if (chunk == 0) {
        for (int t = threadIdx.x; t < d.Ti; t += 256) {
            const float wt = t < Tv ? w[t] : 0.f;
            a.w_out[(long)b * d.Ti + t] = wt;
            if (a.cum_out != nullptr) a.cum_out[(long)b * d.Ti + t] = a.cum_in[(long)b * d.Ti + t] + wt;
        }
    }
    // ---- phase 3: context for this member's slice of the E columns, over all valid positions
    const int W = ((d.E + NCv - 1) / NCv + 3) & ~3;          // columns per member (multiple of 4)
    const int col0 = chunk * W;
    const int W4 = max(0, min(W, d.E - col0)) >> 2;           // float4 columns of this slice
    if (W4 <= 0) return;
    const int W4e = min(W4, 256), PG = 256 / W4e;             // position groups folded through LDS
    const int cq = threadIdx.x % W4e, pg = threadIdx.x / W4e;
    float4* fold = reinterpret_cast<float4*>(S.ds);           // [PG][W4e] float4 (S.ds is free after phase 1)
    for (int cbase = 0; cbase < W4; cbase += W4e) {           // one trip unless a slice is wider than 1024 columns
        const int c4 = cbase + cq;
        const bool live = pg < PG && c4 < W4;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        if (live) {
            const float* mp = a.memory + (long)b * d.Ti * a.ld_mem + col0 + 4 * c4;
            int p = pg;
            for (; p + 3 * PG < Tv; p += 4 * PG) {            // four rows in flight per thread
                const float4 m0 = *reinterpret_cast<const float4*>(mp + (long)p * a.ld_mem);
                const float4 m1 = *reinterpret_cast<const float4*>(mp + (long)(p + PG) * a.ld_mem);
                const float4 m2 = *reinterpret_cast<const float4*>(mp + (long)(p + 2 * PG) * a.ld_mem);
                const float4 m3 = *reinterpret_cast<const float4*>(mp + (long)(p + 3 * PG) * a.ld_mem);
                const float w0 = w[p], w1 = w[p + PG], w2 = w[p + 2 * PG], w3 = w[p + 3 * PG];
                a0.x += w0 * m0.x + w2 * m2.x; a0.y += w0 * m0.y + w2 * m2.y; a0.z += w0 * m0.z + w2 * m2.z; a0.w += w0 * m0.w + w2 * m2.w;
                a1.x += w1 * m1.x + w3 * m3.x; a1.y += w1 * m1.y + w3 * m3.y; a1.z += w1 * m1.z + w3 * m3.z; a1.w += w1 * m1.w + w3 * m3.w;
            }
            for (; p < Tv; p += PG) {
                const float4 m0 = *reinterpret_cast<const float4*>(mp + (long)p * a.ld_mem);
                const float w0 = w[p];
                a0.x += w0 * m0.x; a0.y += w0 * m0.y; a0.z += w0 * m0.z; a0.w += w0 * m0.w;
            }
            fold[pg * W4e + cq] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
        }
        __syncthreads();
        if (pg == 0 && c4 < W4) {
            float4 c = fold[cq];
            for (int j = 1; j < PG; ++j) {
                const float4 t = fold[j * W4e + cq];
                c.x += t.x; c.y += t.y; c.z += t.z; c.w += t.w;
            }
            const int col = col0 + 4 * c4;
            if (a.c0 != nullptr) *reinterpret_cast<float4*>(a.c0 + (long)b * a.ld0 + col) = c;
            if (a.c1 != nullptr) *reinterpret_cast<float4*>(a.c1 + (long)b * a.ld1 + col) = c;
            if (a.c2 != nullptr) *reinterpret_cast<float4*>(a.c2 + (long)b * a.ld2 + col) = c;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward attention step as ONE launch WITHOUT any in-launch wait (gt_attn_step2_fwd): every (row, 32-position chunk)
// workgroup
//   0. requests its 32 x E tile of the attention memory straight away -- it depends on nothing of this step --
//   1. computes the energies of its positions (as attn_energies_fwd_mfma),
//   2. turns them into softmax NUMERATORS relative to the chunk's own maximum, p_i = exp(e_i - m_c), s_c = sum p_i, and
//      forms the chunk's PARTIAL context sum_i p_i memory_i from the tile that arrived meanwhile,
//   3. publishes (partial context, p, m_c, s_c) with write-through stores, drains, and draws a ticket on its row's counter;
//   4. the LAST arriver of the row combines: M = max m_c, ctx = sum_c e^{m_c - M} part_c / sum_c e^{m_c - M} s_c, the
//      normalised weights w_i = p_i e^{m_c - M} / S and the cumulative weights, and writes the step's outputs.
// No workgroup ever waits for another (the hand-off form of rec_cell.h: sc1 stores, every storing wave drained, one
// agent-scope add per workgroup, the last adder reads with sc1 loads), the memory stream runs under the energies phase
// instead of after a kernel boundary, and the softmax is the usual online form: results equal the two-launch path up to
// rounding.  Counters are monotonic over a loop call (ticket NCv * step - 1 is the last arriver's), zeroed once by the caller.
struct AttnStep2Fwd {
    GtAttnDims d; GtSlabs qpart;
    const float* w_prev; const float* w_cum; const float* pm; const float* Wc; const float* Wd; const float* v;
    const int* lengths; float* q_out; const float* wpack; float* u_out;
    float* xbuf; long xs; unsigned* row_cnt; unsigned step1;
    const float* memory; long ld_mem; float* w_out; const float* cum_in; float* cum_out;
    float* c0; long ld0; float* c1; long ld1; float* c2; long ld2;
    int NC;
    int dbg;       // GANTRON_DEBUG=attn2=<bits> (timing knock-outs, results wrong): 1 no memory tile, 2 stop after the energies,
                   // 4 stop after the ticket (no combine)
    long long* stamps;     // diagnostics (GANTRON_DEBUG=attn_stamps): 8 wall-clock stamps per workgroup, or NULL
    long long* tstamps;    // ... and 4 per workgroup for the tail (gt_debug_attn_tail_stamps)
    GtRecTail tail;        // recurrent-product items for workgroups with nothing (more) to do (rec_tail.h); work == NULL: none
};
// (by-value kernel argument: far below the 4 KB the runtime copies per launch)
static_assert(sizeof(AttnStep2Fwd) <= 512, "AttnStep2Fwd grew: check the kernel-argument segment");
// (a separate instantiation, stamps kept in registers until the end)
#define AT_STAMP(k) do { if constexpr (STAMPS) { tstamp[k] = (long long)__builtin_amdgcn_s_memrealtime(); if ((k) >= 6 && threadIdx.x == 0) { for (int q_ = 0; q_ <= (k); ++q_) a.stamps[(long)blockIdx.x * 8 + q_] = tstamp[q_]; } } } while (0)

typedef unsigned au32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t at_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

// TAIL = false is the default launch: exactly the step.  Round 4: with the tail's code in the SAME kernel (tail.work == NULL at
// run time) the step's own results went wrong now and then beside other kernels -- one register of a quarter wave (lanes
// 48..63) of the partial context, 1e-2 off, from the second block of a loop on, never alone on the chip and never with the
// round-3 build of the same source (tools/nondet_probe.py; profiles/r04_attn_step2_regression.txt).  The tail (experimental,
// measured zero-sum) therefore lives in an instantiation of its own and the default kernel contains none of its code.
template <bool STAMPS>
__global__ __launch_bounds__(256) void attn_step2_fwd_mfma(AttnStep2Fwd a) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ SharedM S;
#define AT2_DONE return
#include "attn_step2_body.h"
#undef AT2_DONE
}

// the same step followed by recurrent-product items for workgroups that are done (GtRecTail; experimental)
template <bool STAMPS>
__global__ __launch_bounds__(256) void attn_step2_tail_fwd_mfma(AttnStep2Fwd a) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ SharedM S;
    static_assert(sizeof(SharedM) >= sizeof(float) * RT_LDS_FLOATS, "the tail's staging buffers live in the step's LDS");
    if constexpr (STAMPS) {
        if (threadIdx.x == 0 && a.tstamps != nullptr) a.tstamps[(long)blockIdx.x * 4] = (long long)__builtin_amdgcn_s_memrealtime();
    }
    {
#define AT2_DONE goto at2_done
#include "attn_step2_body.h"
#undef AT2_DONE
    }
at2_done:
    // this workgroup's share of the step is done (or it never had one: its chunk lies in the row's padding): the next LSTM
    // step's early half, at the default issue priority -- the step's own waves on this CU come first
    if (a.tail.work != nullptr) {
        if (!(a.dbg & 8)) __builtin_amdgcn_s_setprio(0);       // (GANTRON_DEBUG=attn2=8: the tail keeps the step's priority)
        rec_tail_run(a.tail, reinterpret_cast<float*>(&S), STAMPS && a.tstamps != nullptr ? a.tstamps + (long)blockIdx.x * 4 : nullptr);
    }
}

__global__ __launch_bounds__(256) void attn_energies_bwd_mfma(GtAttnDims d, const float* dw, const float* w,
                                                              const float* q, const float* w_prev, const float* w_cum,
                                                              const float* pm, const float* Wc, const float* Wd,
                                                              const float* v, const int* lengths, float* dq, long ld_dq,
                                                              float* dpm, float* dwcat_out, float* acc_slabs, const float* wpack) {
    __shared__ SharedM S;
    const int b = blockIdx.y, t0 = blockIdx.x * CT;
    if (lengths != nullptr && t0 >= lengths[b]) return;      // all-padding chunk: every gradient is zero
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    // this workgroup's accumulator slab in the kernel's own register order (gt_attn_acc_reduce un-permutes it):
    // [dv: 128][dWd: thread*16 + (ml*2+nt)*4 + e][dWc: thread*8 + nl*4 + e] -> float4 loads / stores
    float* acc_wg = acc_slabs + ((long)b * gridDim.x + blockIdx.x) * PER_MFMA;
    // every global input is requested here, unconditionally (clamped indices): pm and the dpm accumulator in
    // accumulator layout, this workgroup's dv / dWd / dWc accumulator slab, and the softmax row.  The kernel is
    // then one memory round trip deep instead of five.
    float4 pm4[4], dpm4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, tl = f >> 5, c4 = f & 31;
        const long pi = ((long)b * d.Ti + min(t0 + tl, d.Ti - 1)) * A + 4 * c4;
        pm4[i] = *reinterpret_cast<const float4*>(pm + pi);
        dpm4[i] = *reinterpret_cast<const float4*>(dpm + pi);
    }
    float old_dv[2], old_wd[2][2][4], old_wc[2][4];
#pragma unroll
    for (int nl = 0; nl < 2; ++nl) old_dv[nl] = acc_wg[32 * wave + 16 * nl + r];
    {
        const float4* pw = reinterpret_cast<const float4*>(acc_wg + A + threadIdx.x * 16);
#pragma unroll
        for (int ml = 0; ml < 2; ++ml)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float4 t = pw[ml * 2 + nt];
                old_wd[ml][nt][0] = t.x; old_wd[ml][nt][1] = t.y; old_wd[ml][nt][2] = t.z; old_wd[ml][nt][3] = t.w;
            }
        const float4* pc = reinterpret_cast<const float4*>(acc_wg + A + A * F + threadIdx.x * 8);
#pragma unroll
        for (int nl = 0; nl < 2; ++nl) {
            const float4 t = pc[nl];
            old_wc[nl][0] = t.x; old_wc[nl][1] = t.y; old_wc[nl][2] = t.z; old_wc[nl][3] = t.w;
        }
    }
    // softmax backward: de = w * (dw - sum_t w*dw); the first 256 positions are prefetched, one per thread
    const long ti = (long)b * d.Ti + min((int)threadIdx.x, d.Ti - 1);
    const float w_t = w[ti], dw_t = dw[ti];
    const long tci = (long)b * d.Ti + min(t0 + (int)(threadIdx.x & (CT - 1)), d.Ti - 1);
    const float w_c = w[tci], dw_c = dw[tci];
    const float q_t = q[(long)b * A + (threadIdx.x & (A - 1))];
    stage_common(S, d, Wc, Wd, v, w_prev, w_cum, b, t0, wpack);
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // pm tile parked in S.ds; every thread later overwrites exactly the elements it read
        const int f = threadIdx.x + 256 * i, tl = f >> 5, c4 = f & 31;
        *reinterpret_cast<float4*>(&S.ds[tl * LDS_ + 4 * c4]) = pm4[i];
    }
    if (threadIdx.x < A) S.q[threadIdx.x] = q_t;
    float part = (int)threadIdx.x < d.Ti ? w_t * dw_t : 0.f;
    for (int t = threadIdx.x + 256; t < d.Ti; t += 256) part += w[(long)b * d.Ti + t] * dw[(long)b * d.Ti + t];
    const float dot = block_sum(part, S.red);
    if (threadIdx.x < CT) {
        const int t = t0 + threadIdx.x;
        S.de[threadIdx.x] = (t < d.Ti) ? w_c * (dw_c - dot) : 0.f;
    }
    __syncthreads();
    conv_tile(S, wave, lane);
    __syncthreads();
    f32x4 acc[2][2];
    dense_tiles(S, wave, lane, acc);
    // ds = de * v * (1 - u^2); dq, dv column sums; dpm += ds; ds -> LDS
    float dqs[2] = {0.f, 0.f}, dvs[2] = {0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nl = 0; nl < 2; ++nl) {
            const int dd = 32 * wave + 16 * nl + r;
            const float qd = S.q[dd], vd = S.v[dd];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tl = 16 * mt + 4 * kq + e;
                float dsv = 0.f;
                if (t0 + tl < d.Ti) {
                    const float u = tanhf_(acc[mt][nl][e] + qd + S.ds[tl * LDS_ + dd]);      // pm parked here
                    const float de = S.de[tl];
                    dsv = de * vd * (1.f - u * u);
                    dvs[nl] += de * u;
                    dqs[nl] += dsv;
                }
                S.ds[tl * LDS_ + dd] = dsv;
            }
        }
#pragma unroll
    for (int nl = 0; nl < 2; ++nl) {
        dqs[nl] += __shfl_xor(dqs[nl], 16, 64);
        dqs[nl] += __shfl_xor(dqs[nl], 32, 64);
        dvs[nl] += __shfl_xor(dvs[nl], 16, 64);
        dvs[nl] += __shfl_xor(dvs[nl], 32, 64);
        if (kq == 0) {
            const int dd = 32 * wave + 16 * nl + r;
            atomicAdd(&dq[(long)b * ld_dq + dd], dqs[nl]);
            acc_wg[dd] = old_dv[nl] + dvs[nl];
        }
    }
    __syncthreads();                                   // ds complete
    // dpm += ds as coalesced float4 read-modify-write (the dpm tile was requested at the top)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, tl = f >> 5, c4 = f & 31;
        if (t0 + tl < d.Ti) {
            const float4 dsv = *reinterpret_cast<const float4*>(&S.ds[tl * LDS_ + 4 * c4]);
            *reinterpret_cast<float4*>(dpm + ((long)b * d.Ti + t0 + tl) * A + 4 * c4) =
                make_float4(dpm4[i].x + dsv.x, dpm4[i].y + dsv.y, dpm4[i].z + dsv.z, dpm4[i].w + dsv.w);
        }
    }

    // G3: dlocf tile (mt = wave>>1, nt = wave&1) = ds[32x128] . Wd[128x32]
    {
        const int mt = wave >> 1, nt = wave & 1;
        f32x4 c3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const int k0 = 16 * s4 + 4 * kq;
            const float4 a = ld4(&S.ds[(16 * mt + r) * LDS_ + k0]);
            const float4 bb = ld4s(&S.wd[k0 * LDD + 16 * nt + r], LDD);
            c3 = mfma4(a, bb, c3);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) S.dlocf[(16 * mt + 4 * kq + e) * LDL + 16 * nt + r] = c3[e];
    }
    // G4: dWd rows [32w, 32w+32) x 32 filters = ds^T . locf ; read-modify-write of this workgroup's slab
    {
        float* acc_wd = acc_wg + A;
        f32x4 c4[2][2];
#pragma unroll
        for (int ml = 0; ml < 2; ++ml)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) c4[ml][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 2; ++s4) {
            const int k0 = 16 * s4 + 4 * kq;     // tl
            float4 a[2], bb[2];
#pragma unroll
            for (int ml = 0; ml < 2; ++ml) a[ml] = ld4s(&S.ds[k0 * LDS_ + 32 * wave + 16 * ml + r], LDS_);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bb[nt] = ld4s(&S.locf[k0 * LDL + 16 * nt + r], LDL);
#pragma unroll
            for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) c4[ml][nt] = mfma4(a[ml], bb[nt], c4[ml][nt]);
        }
        float4* pw = reinterpret_cast<float4*>(acc_wd + threadIdx.x * 16);
#pragma unroll
        for (int ml = 0; ml < 2; ++ml)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                pw[ml * 2 + nt] = make_float4(old_wd[ml][nt][0] + c4[ml][nt][0], old_wd[ml][nt][1] + c4[ml][nt][1],
                                              old_wd[ml][nt][2] + c4[ml][nt][2], old_wd[ml][nt][3] + c4[ml][nt][3]);
    }
    __syncthreads();                                   // dlocf complete
    // G5: dWc[f][kk] += dlocf^T . im2col ; wave w: filter tile mt = w>>1, kk tiles nt in {2*(w&1), 2*(w&1)+1}
    // G6: g[tl][kk]  = dlocf . Wc       ; same tile assignment (rows = tl)
    {
        float* acc_wc = acc_wg + A + A * F;
        const int mt = wave >> 1;
        f32x4 c5[2], c6[2];
#pragma unroll
        for (int nl = 0; nl < 2; ++nl) {
            c5[nl] = (f32x4){0.f, 0.f, 0.f, 0.f};
            c6[nl] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s4 = 0; s4 < 2; ++s4) {
            const int k0 = 16 * s4 + 4 * kq;
            const float4 a5 = ld4s(&S.dlocf[k0 * LDL + 16 * mt + r], LDL);        // A[row=f][k=tl]
            const float4 a6 = ld4(&S.dlocf[(16 * mt + r) * LDL + k0]);            // A[row=tl][k=f]
#pragma unroll
            for (int nl = 0; nl < 2; ++nl) {
                const int kk = 16 * (2 * (wave & 1) + nl) + r;
                const float4 b5 = make_float4(im2col(S, k0, kk), im2col(S, k0 + 1, kk), im2col(S, k0 + 2, kk),
                                              im2col(S, k0 + 3, kk));            // B[k=tl][col=kk]
                const float4 b6 = ld4s(&S.wc[k0 * LDC + kk], LDC);               // B[k=f][col=kk]
                c5[nl] = mfma4(a5, b5, c5[nl]);
                c6[nl] = mfma4(a6, b6, c6[nl]);
            }
        }
#pragma unroll
        for (int nl = 0; nl < 2; ++nl) {
            const int kk = 16 * (2 * (wave & 1) + nl) + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) S.g[(16 * mt + 4 * kq + e) * LDG + kk] = c6[nl][e];
            reinterpret_cast<float4*>(acc_wc + threadIdx.x * 8)[nl] =
                make_float4(old_wc[nl][0] + c5[nl][0], old_wc[nl][1] + c5[nl][1], old_wc[nl][2] + c5[nl][2], old_wc[nl][3] + c5[nl][3]);
        }
    }
    __syncthreads();                                   // g complete
    // d[w_prev; w_cum][p] = sum_k g[p-k][(c,k)]  (p = window position, t = t0 + p - 15)
    if (threadIdx.x < 2 * (CT + KC - 1)) {
        const int c = threadIdx.x / (CT + KC - 1), p = threadIdx.x - c * (CT + KC - 1);
        const int t = t0 + p - (KC - 1) / 2;
        if (t >= 0 && t < d.Ti) {
            float a = 0.f;
            const int klo = max(0, p - (CT - 1)), khi = min(KC - 1, p);
            for (int k = klo; k <= khi; ++k) a += S.g[(p - k) * LDG + c * KC + k];
            atomicAdd(&dwcat_out[(long)c * d.B * d.Ti + (long)b * d.Ti + t], a);
        }
    }
}

// ---- hoisted backward ------------------------------------------------------------------------------------------------
// With the tanh outputs u of every forward step kept (4.9 MB per step at B=64, Ti=150: sized for 288 GB of HBM), the
// backward loop only needs the RECURRENT terms of the energies backward -- dq and d[w_prev; w_cum] -- and the
// softmax-backward vector de; the parameter gradients (dv, dWd, dWc) and d(processed memory) are sums over all steps
// with no consumer inside the loop and are produced by attn_param_grads_mfma in one batched pass per loop chunk.
// Recurrent terms of the energies backward for one (b, 32-position) tile from the softmax-backward vector: S.de holds de
// of the tile's positions, S.ds the tanh outputs u (both in LDS, weights staged): dq[b] += sum_t de v (1 - u^2) and
// d[w_prev; w_cum] += conv^T(dense^T(.)) by atomics.  Every thread of the workgroup calls it (barriers inside).
__device__ __forceinline__ void energies_rec_tail(SharedM& S, const GtAttnDims& d, int b, int t0, float* dq, long ld_dq,
                                                  float* dwcat_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    float dqs[2] = {0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nl = 0; nl < 2; ++nl) {
            const int dd = 32 * wave + 16 * nl + r;
            const float vd = S.v[dd];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tl = 16 * mt + 4 * kq + e;
                float dsv = 0.f;
                if (t0 + tl < d.Ti) {
                    const float uu = S.ds[tl * LDS_ + dd];
                    dsv = S.de[tl] * vd * (1.f - uu * uu);
                    dqs[nl] += dsv;
                }
                S.ds[tl * LDS_ + dd] = dsv;
            }
        }
#pragma unroll
    for (int nl = 0; nl < 2; ++nl) {
        dqs[nl] += __shfl_xor(dqs[nl], 16, 64);
        dqs[nl] += __shfl_xor(dqs[nl], 32, 64);
        if (kq == 0) atomicAdd(&dq[(long)b * ld_dq + 32 * wave + 16 * nl + r], dqs[nl]);
    }
    __syncthreads();
    {   // G3: dlocf = ds . Wd
        const int mt = wave >> 1, nt = wave & 1;
        f32x4 c3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const int k0 = 16 * s4 + 4 * kq;
            c3 = mfma4(ld4(&S.ds[(16 * mt + r) * LDS_ + k0]), ld4s(&S.wd[k0 * LDD + 16 * nt + r], LDD), c3);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) S.dlocf[(16 * mt + 4 * kq + e) * LDL + 16 * nt + r] = c3[e];
    }
    __syncthreads();
    {   // G6: g = dlocf . Wc
        const int mt = wave >> 1;
        f32x4 c6[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s4 = 0; s4 < 2; ++s4) {
            const int k0 = 16 * s4 + 4 * kq;
            const float4 a6 = ld4(&S.dlocf[(16 * mt + r) * LDL + k0]);
#pragma unroll
            for (int nl = 0; nl < 2; ++nl) {
                const int kk = 16 * (2 * (wave & 1) + nl) + r;
                c6[nl] = mfma4(a6, ld4s(&S.wc[k0 * LDC + kk], LDC), c6[nl]);
            }
        }
#pragma unroll
        for (int nl = 0; nl < 2; ++nl) {
            const int kk = 16 * (2 * (wave & 1) + nl) + r;
#pragma unroll
            for (int e = 0; e < 4; ++e) S.g[(16 * mt + 4 * kq + e) * LDG + kk] = c6[nl][e];
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * (CT + KC - 1)) {
        const int c = threadIdx.x / (CT + KC - 1), p = threadIdx.x - c * (CT + KC - 1);
        const int t = t0 + p - (KC - 1) / 2;
        if (t >= 0 && t < d.Ti) {
            float a = 0.f;
            const int klo = max(0, p - (CT - 1)), khi = min(KC - 1, p);
            for (int k = klo; k <= khi; ++k) a += S.g[(p - k) * LDG + c * KC + k];
            atomicAdd(&dwcat_out[(long)c * d.B * d.Ti + (long)b * d.Ti + t], a);
        }
    }
}

__global__ __launch_bounds__(256) void attn_energies_bwd_rec_mfma(GtAttnDims d, const float* dw, const float* w,
                                                                  const float* u, const float* Wc, const float* Wd,
                                                                  const float* v, const int* lengths, float* dq, long ld_dq,
                                                                  float* dwcat_out, float* de_out, const float* wpack) {
    __builtin_amdgcn_s_setprio(3);      // the attention chain is the serial critical path of the decoder loops: its waves
                                        // issue ahead of the co-resident waves of the other chain and of the leaf GEMMs
    __shared__ SharedM S;
    const int b = blockIdx.x, t0 = blockIdx.y * CT;       // grid (B, chunks), as the forward kernel
    if (lengths != nullptr && t0 >= lengths[b]) {
        if (threadIdx.x < CT && t0 + threadIdx.x < d.Ti) de_out[(long)b * d.Ti + t0 + threadIdx.x] = 0.f;
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    float4 u4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, tl = f >> 5, c4 = f & 31;
        u4[i] = *reinterpret_cast<const float4*>(u + ((long)b * d.Ti + min(t0 + tl, d.Ti - 1)) * A + 4 * c4);
    }
    const long ti = (long)b * d.Ti + min((int)threadIdx.x, d.Ti - 1);
    const float w_t = w[ti], dw_t = dw[ti];
    const long tci = (long)b * d.Ti + min(t0 + (int)(threadIdx.x & (CT - 1)), d.Ti - 1);
    const float w_c = w[tci], dw_c = dw[tci];
    stage_common(S, d, Wc, Wd, v, nullptr, nullptr, b, t0, wpack, false);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, tl = f >> 5, c4 = f & 31;
        *reinterpret_cast<float4*>(&S.ds[tl * LDS_ + 4 * c4]) = u4[i];
    }
    float part = (int)threadIdx.x < d.Ti ? w_t * dw_t : 0.f;
    for (int t = threadIdx.x + 256; t < d.Ti; t += 256) part += w[(long)b * d.Ti + t] * dw[(long)b * d.Ti + t];
    const float dot = block_sum(part, S.red);
    if (threadIdx.x < CT) {
        const int t = t0 + threadIdx.x;
        const float de = (t < d.Ti) ? w_c * (dw_c - dot) : 0.f;
        S.de[threadIdx.x] = de;
        if (t < d.Ti) de_out[(long)b * d.Ti + t] = de;
    }
    __syncthreads();
    energies_rec_tail(S, d, b, t0, dq, ld_dq, dwcat_out);
}

// ---- fused backward step ---------------------------------------------------------------------------------------------
// The backward attention chain was context_bwd (dw for all positions of a row) -> energies_bwd_rec (softmax backward needs
// the ROW sum s = sum_t w dw, then everything is per position) -- two dependent launches because of one scalar per row.
// Everything downstream of the softmax backward is LINEAR in de = w (dw - s) = w dw - s w.  With
//     M[t][d] = v[d] (1 - u[t][d]^2)   and   P = (M . Wd) . Wc   ([t][(c,k)], 62 columns per position),
// both functions of forward quantities only,
//     dq[d] = sum_t de[t] M[t][d],      d[w_prev; w_cum][c][t0 + p - 15] = sum_k de[p-k] P[p-k][(c,k)].
// So
//  * gt_attn_bwd_wterms (ONE batched launch over all steps, before the loop, off the critical path) computes P for every
//    step (the two small GEMMs of the energies backward, on M instead of on de M) and the s-terms dq2 = Lq(w_t),
//    dwcat2 = Lw(w_t);
//  * attn_bwd_step_kernel (one launch per step) computes dw for its 32 positions, e1 = w dw, the tile's share of s, and
//    applies e1 to M and P with a few dozen multiply-adds per thread -- no GEMM, no weights, no cross-workgroup dependency;
//  * the consumers subtract s times the precomputed term: the next step's carried alignment gradient (in this kernel),
//    the attention cell backward's dq (gt_cell_bwd_query_c), and -- off the critical path -- the stored dq / de histories
//    (gt_attn_bwd_fixup before the weight-gradient passes that read them).
constexpr int PW = 64;      // columns of a P row in memory (62 used)

// dq[b][d] += sum_t e[t] M[t][d],  dwcat[c][b][t0 + p - 15] += sum_k e[p-k] P[p-k][c KC + k]  (e, M, P in LDS: e[CT],
// M rows LDS_ apart, P rows LDG apart)
__device__ __forceinline__ void apply_e(const float* e, const float* M, const float* P, const GtAttnDims& d, int b, int t0,
                                        float* dq, long ld_dq, float* dwcat_out) {
    {
        const int dd = threadIdx.x & (A - 1), half = threadIdx.x >> 7;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < CT / 2; ++i) acc += e[16 * half + i] * M[(16 * half + i) * LDS_ + dd];
        atomicAdd(&dq[(long)b * ld_dq + dd], acc);
    }
    if (threadIdx.x < 2 * (CT + KC - 1)) {
        const int c = threadIdx.x / (CT + KC - 1), p = threadIdx.x - c * (CT + KC - 1);
        const int t = t0 + p - (KC - 1) / 2;
        if (t >= 0 && t < d.Ti) {
            float acc = 0.f;
            const int klo = max(0, p - (CT - 1)), khi = min(KC - 1, p);
            for (int k = klo; k <= khi; ++k) acc += e[p - k] * P[(p - k) * LDG + c * KC + k];
            atomicAdd(&dwcat_out[(long)c * d.B * d.Ti + (long)b * d.Ti + t], acc);
        }
    }
}

// (round 4: a workgroup walks WT_STEPS steps of its (row, chunk) instead of one -- the packed weights, 24 KB, are staged once
// per workgroup, not once per 16 KB tile of tanh outputs: 8.3 -> see DESIGN.md section 5 -- and the next step's tile is
// requested under the current step's two small GEMMs)
constexpr int WT_STEPS = 8;
__global__ __launch_bounds__(256) void attn_bwd_wterms_mfma(GtAttnDims d, int nsteps, const float* w, long w_step, const float* u,
                                                            const float* Wc, const float* Wd, const float* v,
                                                            const int* lengths, float* dq2, long dq2_step, long ld_dq2,
                                                            float* dwcat2, float* p_hist, const float* wpack) {
    __shared__ SharedM S;
    const int b = blockIdx.x, t0 = blockIdx.y * CT, step0 = blockIdx.z * WT_STEPS;       // grid (B, chunks, ceil(steps / WT_STEPS))
    if (lengths != nullptr && t0 >= lengths[b]) return;                      // dq2 / dwcat2 are zero-initialised; P unread
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const long BTi = (long)d.B * d.Ti;
    const int step1 = min(nsteps, step0 + WT_STEPS);
    stage_common(S, d, Wc, Wd, v, nullptr, nullptr, b, t0, wpack, false);
    float4 u4[4];
    float w_c;
#define WT_LOAD(ST_)                                                                                                         \
    do {                                                                                                                     \
        const float* u_ = u + (long)(ST_) * BTi * A;                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                      \
            const int f = threadIdx.x + 256 * i, tl = f >> 5, c4 = f & 31;                                                   \
            u4[i] = *reinterpret_cast<const float4*>(u_ + ((long)b * d.Ti + min(t0 + tl, d.Ti - 1)) * A + 4 * c4);          \
        }                                                                                                                    \
        w_c = (w + (long)(ST_) * w_step)[(long)b * d.Ti + min(t0 + (int)(threadIdx.x & (CT - 1)), d.Ti - 1)];                \
    } while (0)
    WT_LOAD(step0);
    __syncthreads();                                   // S.v, S.wc, S.wd
    for (int step = step0; step < step1; ++step) {
        float* dq2_s = dq2 + (long)step * dq2_step;
        float* dwcat2_s = dwcat2 + (long)step * 2 * BTi;
        float* p_s = p_hist + (long)step * BTi * PW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {                      // M = v (1 - u^2), zero beyond Ti
            const int f = threadIdx.x + 256 * i, tl = f >> 5, c4 = f & 31;
            const float4 vv = *reinterpret_cast<const float4*>(&S.v[4 * c4]);
            const float m = (t0 + tl < d.Ti) ? 1.f : 0.f;
            *reinterpret_cast<float4*>(&S.ds[tl * LDS_ + 4 * c4]) =
                make_float4(m * vv.x * (1.f - u4[i].x * u4[i].x), m * vv.y * (1.f - u4[i].y * u4[i].y),
                            m * vv.z * (1.f - u4[i].z * u4[i].z), m * vv.w * (1.f - u4[i].w * u4[i].w));
        }
        if (threadIdx.x < CT) S.de[threadIdx.x] = (t0 + (int)threadIdx.x < d.Ti) ? w_c : 0.f;
        if (step + 1 < step1) WT_LOAD(step + 1);           // the next step's tile travels under this step's products
        __syncthreads();
        {   // M . Wd
            const int mt = wave >> 1, nt = wave & 1;
            f32x4 c3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                const int k0 = 16 * s4 + 4 * kq;
                c3 = mfma4(ld4(&S.ds[(16 * mt + r) * LDS_ + k0]), ld4s(&S.wd[k0 * LDD + 16 * nt + r], LDD), c3);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) S.dlocf[(16 * mt + 4 * kq + e) * LDL + 16 * nt + r] = c3[e];
        }
        __syncthreads();
        {   // P = (M . Wd) . Wc
            const int mt = wave >> 1;
            f32x4 c6[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s4 = 0; s4 < 2; ++s4) {
                const int k0 = 16 * s4 + 4 * kq;
                const float4 a6 = ld4(&S.dlocf[(16 * mt + r) * LDL + k0]);
#pragma unroll
                for (int nl = 0; nl < 2; ++nl) {
                    const int kk = 16 * (2 * (wave & 1) + nl) + r;
                    c6[nl] = mfma4(a6, ld4s(&S.wc[k0 * LDC + kk], LDC), c6[nl]);
                }
            }
#pragma unroll
            for (int nl = 0; nl < 2; ++nl) {
                const int kk = 16 * (2 * (wave & 1) + nl) + r;
#pragma unroll
                for (int e = 0; e < 4; ++e) S.g[(16 * mt + 4 * kq + e) * LDG + kk] = c6[nl][e];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < CT * PW; i += 256) {               // P rows of the tile's valid positions
            const int tl = i / PW, kk = i - tl * PW;
            if (t0 + tl < d.Ti) p_s[((long)b * d.Ti + t0 + tl) * PW + kk] = S.g[tl * LDG + kk];
        }
        apply_e(S.de, S.ds, S.g, d, b, t0, dq2_s, ld_dq2, dwcat2_s);
        __syncthreads();                                   // S.ds / S.de / S.g are rewritten by the next step
    }
#undef WT_LOAD
}

struct AttnBwdStep {
    GtAttnDims d;
    GtSlabs da, db;
    const float* dctx_direct; long ld_direct;
    const float* memory; long ld_mem;
    const int* lengths;
    float* dwcat_in; float* dcum_acc; const float* dw_extern;
    float* dctx_out; long ld_dctx_out;
    const float* w; const float* u; const float* v; const float* p;       // p: this step's (B, Ti, PW) rows of P
    const float* sum_prev; const float* dwcat2_prev;      // NULL on the last step (nothing carried yet)
    float* sum_out;                                       // (B, chunks) this step's partial row sums
    float* dq; long ld_dq; float* dwcat_out; float* e1_out;
    // TAIL instantiation (gt_attn_bwd_step_cell): the attention LSTM's cell backward of the step, done by each row's LAST
    // workgroup -- what gt_cell_bwd_query_c computes in a launch of its own
    struct Tail {
        GtSlabs dh_a, dh_c;
        const float* dq2; const float* Wq; long ldwq;
        const float* dc_next; const float* gates; const float* c_new; const float* c_prev;
        const uint8_t* keep; float scale;
        float* dgates; float* dc_prev;
        int H;
        unsigned* row_cnt; unsigned launch;
        unsigned long long* dg_amax;            // optional: max |dgates| of the step (gt_absmax format), for the product behind it
    } tail;
};
static_assert(sizeof(AttnBwdStep) <= 1024, "AttnBwdStep is a by-value kernel argument: keep it well below the 4 KB kernel-argument segment");

struct SharedStep {
    alignas(16) float M[CT * LDS_];
    alignas(16) float P[CT * LDG];
    float e[CT];
    float dots[CT];
};

// (amdgpu_waves_per_eu(1, 2): the grid never puts more than two of these workgroups on a CU, so the compiler may spend registers
// -- without it the generic-width instantiation, 79 registers, keeps the tail's 128 loads of Wq to a handful in flight so as not to
// lose an occupancy it cannot use: 50.6 us per launch at E = 1096 against 21.7 at E = 1024, tools/attn_bwd_cell_bench.py)
template <bool VEC, int NE4, bool TAIL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void attn_bwd_step_kernel(AttnBwdStep a) {
    __builtin_amdgcn_s_setprio(3);      // the attention chain is the serial critical path of the decoder loops
    extern __shared__ __attribute__((aligned(16))) float dctx[];   // E floats
    __shared__ SharedStep S;
    const GtAttnDims& d = a.d;
    const int b = blockIdx.x, chunk = blockIdx.y, t0 = chunk * CT, nch = gridDim.y;      // grid (B, chunks)
    const int Tv = a.lengths != nullptr ? min(a.lengths[b], d.Ti) : d.Ti;
    if (t0 >= Tv) {
        if (threadIdx.x < CT && t0 + threadIdx.x < d.Ti) {
            const long i = (long)b * d.Ti + t0 + threadIdx.x;
            if (a.dwcat_in != nullptr) { a.dwcat_in[i] = 0.f; a.dwcat_in[(long)d.B * d.Ti + i] = 0.f; }
            a.e1_out[i] = 0.f;
        }
        if (threadIdx.x == 0) a.sum_out[(long)b * nch + chunk] = 0.f;
        return;
    }
    // requests that do not depend on the context gradient go out first: the tile's tanh outputs (-> M), its P rows, the
    // previous step's row sum
    float4 u4[4], p4[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, tl = f >> 5, c4 = f & 31;
        u4[i] = *reinterpret_cast<const float4*>(a.u + ((long)b * d.Ti + min(t0 + tl, d.Ti - 1)) * A + 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = threadIdx.x + 256 * i, tl = f >> 4, c4 = f & 15;
        p4[i] = *reinterpret_cast<const float4*>(a.p + ((long)b * d.Ti + min(t0 + tl, Tv - 1)) * PW + 4 * c4);
    }
    const float w_c = a.w[(long)b * d.Ti + min(t0 + (int)(threadIdx.x & (CT - 1)), d.Ti - 1)];
    float s_prev = 0.f;
    if (a.sum_prev != nullptr) {
        float sp[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) sp[c] = a.sum_prev[(long)b * nch + min(c, nch - 1)];
#pragma unroll
        for (int c = 0; c < 16; ++c) s_prev += c < nch ? sp[c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 256 * i, tl = f >> 5, c4 = f & 31;
        const float4 vv = *reinterpret_cast<const float4*>(a.v + 4 * c4);
        const float m = (t0 + tl < d.Ti) ? 1.f : 0.f;
        *reinterpret_cast<float4*>(&S.M[tl * LDS_ + 4 * c4]) =
            make_float4(m * vv.x * (1.f - u4[i].x * u4[i].x), m * vv.y * (1.f - u4[i].y * u4[i].y),
                        m * vv.z * (1.f - u4[i].z * u4[i].z), m * vv.w * (1.f - u4[i].w * u4[i].w));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = threadIdx.x + 256 * i, tl = f >> 4, c4 = f & 15;
        float* dst = &S.P[tl * LDG + 4 * c4];         // LDG = 66: rows are 8-byte aligned only
        dst[0] = p4[i].x; dst[1] = p4[i].y; dst[2] = p4[i].z; dst[3] = p4[i].w;
    }
    const float dw = attn_ctx_bwd_dw<VEC, NE4>(d, a.da, a.db, a.dctx_direct, a.ld_direct, a.memory, a.ld_mem, Tv, a.dwcat_in,
                                               a.dcum_acc, a.dw_extern, a.dctx_out, a.ld_dctx_out, a.dwcat2_prev, s_prev,
                                               dctx, S.dots, b, chunk);
    if (threadIdx.x < 64) {         // wave 0: e1 of the tile's positions and their sum
        const bool pos = threadIdx.x < CT && t0 + (int)threadIdx.x < d.Ti;
        const float e1 = pos ? w_c * dw : 0.f;
        if (threadIdx.x < CT) S.e[threadIdx.x] = e1;
        if (pos) a.e1_out[(long)b * d.Ti + t0 + threadIdx.x] = e1;
        const float tot = wave_sum(threadIdx.x < CT ? e1 : 0.f);
        if (threadIdx.x == 0) {
            // (TAIL: the row's last workgroup reads it inside this launch -> written through)
            if constexpr (TAIL) __hip_atomic_store(&a.sum_out[(long)b * nch + chunk], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else a.sum_out[(long)b * nch + chunk] = tot;
        }
    }
    __syncthreads();
    apply_e(S.e, S.M, S.P, d, b, t0, a.dq, a.ld_dq, a.dwcat_out);
    if constexpr (TAIL) {
        // ---- the row's last arriver runs the attention LSTM's cell backward for the row (gt_cell_bwd_query_c's arithmetic:
        // dh = slabs + (dq - s dq2) . Wq, dropout, cell).  Hand-off as in the forward step: every wave's atomics and stores
        // have left (vmcnt), barrier, one agent-scope ticket per workgroup on the row's monotonic counter; what the last
        // arriver reads of this launch's results (dq, the tile sums) it reads with agent-scope loads.
        const AttnBwdStep::Tail& c = a.tail;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nv_row = (Tv + CT - 1) / CT;
        if (threadIdx.x == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(c.row_cnt + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            S.dots[0] = ticket == (unsigned)nv_row * c.launch - 1u ? 1.f : 0.f;
        }
        __syncthreads();
        if (S.dots[0] == 0.f) return;
        float* dqe = S.M;                                   // (apply_e is done with M, P, e in every wave: barriers above)
        if (threadIdx.x < A) {
            float sp[16];
#pragma unroll
            for (int cc = 0; cc < 16; ++cc)
                sp[cc] = __hip_atomic_load(&a.sum_out[(long)b * nch + min(cc, nv_row - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float sm = 0.f;
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) sm += cc < nv_row ? sp[cc] : 0.f;    // (the tiles beyond the row's length write zeros --
                                                                                 // perhaps later than this: not read)
            const float q = __hip_atomic_load(&a.dq[(long)b * a.ld_dq + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float q2 = c.dq2[(long)b * a.ld_dq + threadIdx.x];
            dqe[threadIdx.x] = q - sm * q2;
        }
        const int H = c.H;
        unsigned dg_mx = 0u;
        for (int uq0 = 0; uq0 < (H >> 2); uq0 += 256) {     // (uniform trip count: the barrier below is legal)
            const bool on = uq0 + (int)threadIdx.x < (H >> 2);
            const int u = 4 * (on ? uq0 + (int)threadIdx.x : 0);
            const long idx = (long)b * H + u;
            // the cell's own inputs first: they do not wait for dqe
            float4 dh = make_float4(0.f, 0.f, 0.f, 0.f), dcn = dh;
            auto add_slabs = [&](const GtSlabs& sl) __attribute__((always_inline)) {
                if (sl.p == nullptr) return;
                const float* p = sl.p + (long)b * sl.ld + u;
#pragma unroll 4
                for (int s_ = 0; s_ < sl.n; ++s_) {
                    const float4 t = *reinterpret_cast<const float4*>(p + (long)s_ * sl.slab_stride);
                    dh.x += t.x; dh.y += t.y; dh.z += t.z; dh.w += t.w;
                }
            };
            add_slabs(c.dh_a);
            add_slabs(c.dh_c);
            if (c.dc_next != nullptr) dcn = *reinterpret_cast<const float4*>(c.dc_next + idx);
            const float* gt = c.gates + (long)b * 4 * H + u;
            const float4 gi = *reinterpret_cast<const float4*>(gt), gf = *reinterpret_cast<const float4*>(gt + H);
            const float4 gg = *reinterpret_cast<const float4*>(gt + 2 * H), go = *reinterpret_cast<const float4*>(gt + 3 * H);
            const float4 cn = *reinterpret_cast<const float4*>(c.c_new + idx);
            const float4 cp = *reinterpret_cast<const float4*>(c.c_prev + idx);
            uchar4 kp = make_uchar4(1, 1, 1, 1);
            if (c.keep != nullptr) kp = *reinterpret_cast<const uchar4*>(c.keep + idx);
            if (uq0 == 0) __syncthreads();                  // dqe is there
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* wq = c.Wq + u;
            // (512 KB of Wq per row through one CU's vector cache: ~3.5 us, the larger part of what this tail adds -- 8.9 us on
            // average against gt_cell_bwd_query_c's 9.8 in a launch of its own.  Requesting the first 16 rows in front of the
            // barrier changes nothing.)
#pragma unroll 16
            for (int k = 0; k < A; ++k) {
                const float4 w = *reinterpret_cast<const float4*>(wq + (long)k * c.ldwq);
                const float q = dqe[k];
                acc.x += q * w.x; acc.y += q * w.y; acc.z += q * w.z; acc.w += q * w.w;
            }
            const float dhv[4] = {dh.x + acc.x, dh.y + acc.y, dh.z + acc.z, dh.w + acc.w};
            const float dcv[4] = {dcn.x, dcn.y, dcn.z, dcn.w};
            const float giv[4] = {gi.x, gi.y, gi.z, gi.w}, gfv[4] = {gf.x, gf.y, gf.z, gf.w};
            const float ggv[4] = {gg.x, gg.y, gg.z, gg.w}, gov[4] = {go.x, go.y, go.z, go.w};
            const float cnv[4] = {cn.x, cn.y, cn.z, cn.w}, cpv[4] = {cp.x, cp.y, cp.z, cp.w};
            const unsigned char kpv[4] = {kp.x, kp.y, kp.z, kp.w};
            float dgi[4], dgf[4], dgg[4], dgo[4], dcp[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float dd = dhv[e];
                if (c.keep != nullptr) dd *= (kpv[e] ? c.scale : 0.f);
                const float tc = tanhf_(cnv[e]);
                const float dcell = dcv[e] + dd * gov[e] * (1.f - tc * tc);
                dgi[e] = dcell * ggv[e] * giv[e] * (1.f - giv[e]);
                dgf[e] = dcell * cpv[e] * gfv[e] * (1.f - gfv[e]);
                dgg[e] = dcell * giv[e] * (1.f - ggv[e] * ggv[e]);
                dgo[e] = dd * tc * gov[e] * (1.f - gov[e]);
                dcp[e] = dcell * gfv[e];
            }
            if (c.dg_amax != nullptr) {
                // the largest finite |gate gradient| of the row (gathered below, after the last trip)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g4[4] = {dgi[e], dgf[e], dgg[e], dgo[e]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned ub = __float_as_uint(g4[j]) & 0x7fffffffu;
                        dg_mx = max(dg_mx, (on && ub < 0x7f800000u) ? ub : 0u);
                    }
                }
            }
            if (!on) continue;
            float* dg = c.dgates + (long)b * 4 * H + u;
            *reinterpret_cast<float4*>(dg) = make_float4(dgi[0], dgi[1], dgi[2], dgi[3]);
            *reinterpret_cast<float4*>(dg + H) = make_float4(dgf[0], dgf[1], dgf[2], dgf[3]);
            *reinterpret_cast<float4*>(dg + 2 * H) = make_float4(dgg[0], dgg[1], dgg[2], dgg[3]);
            *reinterpret_cast<float4*>(dg + 3 * H) = make_float4(dgo[0], dgo[1], dgo[2], dgo[3]);
            *reinterpret_cast<float4*>(c.dc_prev + idx) = make_float4(dcp[0], dcp[1], dcp[2], dcp[3]);
        }
        if (c.dg_amax != nullptr) {
            // ... for the fp16-plane product that multiplies them next (a launch boundary away): ONE 64-bit atomicMax per row,
            // tag 1 in the high half as gt_absmax leaves it (one per wave -- 256 atomics on one word at the end of every launch --
            // cost 1.7 us of kernel time)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) dg_mx = max(dg_mx, (unsigned)__shfl_xor((int)dg_mx, off));
            __syncthreads();                                // (dqe is no longer read)
            unsigned* wmx = reinterpret_cast<unsigned*>(S.M);
            if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = dg_mx;
            __syncthreads();
            if (threadIdx.x == 0) atomicMax(c.dg_amax, (1ull << 32) | max(max(wmx[0], wmx[1]), max(wmx[2], wmx[3])));
        }
    }
}

// de[t] = e1[t] - s_t w_t and dq[t] -= s_t dq2[t] for `nsteps` steps (the histories the weight-gradient passes read):
// one thread per element of max(B*Ti, B*A) per step, grid (blocks, steps)
__global__ __launch_bounds__(256) void attn_bwd_fixup_kernel(GtAttnDims d, int nch, const float* sum, const float* w, long w_step,
                                                             float* de, float* dq, long dq_step, long ld_dq, const float* dq2) {
    const int step = blockIdx.y;
    const long BTi = (long)d.B * d.Ti;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const float* sm = sum + (long)step * d.B * nch;
    if (i < BTi) {
        const int b = (int)(i / d.Ti);
        float s = 0.f;
        for (int c = 0; c < nch; ++c) s += sm[(long)b * nch + c];
        de[(long)step * BTi + i] -= s * w[(long)step * w_step + i];
    }
    if (i < (long)d.B * A) {
        const int b = (int)(i / A), c0 = (int)(i - (long)b * A);
        float s = 0.f;
        for (int c = 0; c < nch; ++c) s += sm[(long)b * nch + c];
        dq[(long)step * dq_step + (long)b * ld_dq + c0] -= s * dq2[(long)step * dq_step + (long)b * ld_dq + c0];
    }
}

// Parameter gradients of `nsteps` consecutive steps for one (b, 32-position) tile: per step ds = de v (1 - u^2),
// locf = conv(windows), dlocf = ds . Wd;  dv += de u, dpm += ds, dWd += ds^T . locf, dWc += dlocf^T . im2col, all
// accumulated in REGISTERS over the steps and added to dpm / the workgroup's accumulator slab once at the end
// (same slab layout as attn_energies_bwd_mfma, reduced by gt_attn_acc_reduce).
__global__ __launch_bounds__(256) void attn_param_grads_mfma(GtAttnDims d, int nsteps, const float* u, const float* de,
                                                             const float* w_prev, const float* w_cum, long step_stride,
                                                             const float* Wc, const float* Wd, const float* v,
                                                             const int* lengths, float* dpm, float* acc_slabs,
                                                             const float* wpack) {
    __shared__ SharedM S;
    const int b = blockIdx.y, t0 = blockIdx.x * CT;
    if (lengths != nullptr && t0 >= lengths[b]) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const long BTi = (long)d.B * d.Ti;
    float* acc_wg = acc_slabs + ((long)b * gridDim.x + blockIdx.x) * PER_MFMA;
    stage_common(S, d, Wc, Wd, v, w_prev, w_cum, b, t0, wpack, false);      // weights once; windows per step below
    float dv_acc[2] = {0.f, 0.f}, dpm_acc[2][2][4];
    f32x4 c4[2][2], c5[2];
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_) {
        c5[a_] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_) {
            c4[a_][b_] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) dpm_acc[a_][b_][e] = 0.f;
        }
    }
    // software pipeline: the loads of step s + 1 are issued before the arithmetic of step s
    float4 u4[4];
    float de_r = 0.f, win_r = 0.f;
#define GT_FETCH(STEP)                                                                                              \
    {                                                                                                               \
        const float* us = u + (long)(STEP) * BTi * A;                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                             \
            const int f = threadIdx.x + 256 * i, tl = f >> 5, c4i = f & 31;                                         \
            u4[i] = *reinterpret_cast<const float4*>(us + ((long)b * d.Ti + min(t0 + tl, d.Ti - 1)) * A + 4 * c4i);  \
        }                                                                                                           \
        de_r = de[(long)(STEP) * BTi + (long)b * d.Ti + min(t0 + (int)(threadIdx.x & (CT - 1)), d.Ti - 1)];         \
        const int c = (threadIdx.x & 127) / WL, p = (threadIdx.x & 127) - c * WL;                                   \
        const int t = t0 + p - (KC - 1) / 2;                                                                        \
        const float* src = (c == 0 ? w_prev : w_cum) + (long)(STEP) * step_stride;                                  \
        const float wv = src[(long)b * d.Ti + min(max(t, 0), d.Ti - 1)];                                            \
        win_r = (p < CT + KC - 1 && t >= 0 && t < d.Ti) ? wv : 0.f;                                                 \
    }
    GT_FETCH(0)
    for (int s = 0; s < nsteps; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = threadIdx.x + 256 * i, tl = f >> 5, c4i = f & 31;
            *reinterpret_cast<float4*>(&S.ds[tl * LDS_ + 4 * c4i]) = u4[i];
        }
        if (threadIdx.x < CT) S.de[threadIdx.x] = (t0 + (int)threadIdx.x < d.Ti) ? de_r : 0.f;
        if (threadIdx.x < 2 * WL) S.win[threadIdx.x / WL][threadIdx.x % WL] = win_r;
        __syncthreads();
        if (s + 1 < nsteps) GT_FETCH(s + 1)
        conv_tile(S, wave, lane);                         // locf (reads win, wc)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nl = 0; nl < 2; ++nl) {
                const int dd = 32 * wave + 16 * nl + r;
                const float vd = S.v[dd];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int tl = 16 * mt + 4 * kq + e;
                    float dsv = 0.f;
                    if (t0 + tl < d.Ti) {
                        const float uu = S.ds[tl * LDS_ + dd], dee = S.de[tl];
                        dsv = dee * vd * (1.f - uu * uu);
                        dv_acc[nl] += dee * uu;
                        dpm_acc[mt][nl][e] += dsv;
                    }
                    S.ds[tl * LDS_ + dd] = dsv;
                }
            }
        __syncthreads();                                   // ds, locf complete
        {   // G3: dlocf = ds . Wd
            const int mt = wave >> 1, nt = wave & 1;
            f32x4 c3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 8; ++s4) {
                const int k0 = 16 * s4 + 4 * kq;
                c3 = mfma4(ld4(&S.ds[(16 * mt + r) * LDS_ + k0]), ld4s(&S.wd[k0 * LDD + 16 * nt + r], LDD), c3);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) S.dlocf[(16 * mt + 4 * kq + e) * LDL + 16 * nt + r] = c3[e];
        }
        // G4: dWd rows [32w, 32w+32) += ds^T . locf
#pragma unroll
        for (int s4 = 0; s4 < 2; ++s4) {
            const int k0 = 16 * s4 + 4 * kq;
            float4 a[2], bb[2];
#pragma unroll
            for (int ml = 0; ml < 2; ++ml) a[ml] = ld4s(&S.ds[k0 * LDS_ + 32 * wave + 16 * ml + r], LDS_);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bb[nt] = ld4s(&S.locf[k0 * LDL + 16 * nt + r], LDL);
#pragma unroll
            for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) c4[ml][nt] = mfma4(a[ml], bb[nt], c4[ml][nt]);
        }
        __syncthreads();                                   // dlocf complete
        {   // G5: dWc += dlocf^T . im2col
            const int mt = wave >> 1;
#pragma unroll
            for (int s4 = 0; s4 < 2; ++s4) {
                const int k0 = 16 * s4 + 4 * kq;
                const float4 a5 = ld4s(&S.dlocf[k0 * LDL + 16 * mt + r], LDL);
#pragma unroll
                for (int nl = 0; nl < 2; ++nl) {
                    const int kk = 16 * (2 * (wave & 1) + nl) + r;
                    const float4 b5 = make_float4(im2col(S, k0, kk), im2col(S, k0 + 1, kk), im2col(S, k0 + 2, kk),
                                                  im2col(S, k0 + 3, kk));
                    c5[nl] = mfma4(a5, b5, c5[nl]);
                }
            }
        }
        __syncthreads();                                   // before the next step overwrites ds / win / dlocf
    }
#undef GT_FETCH
    // ---- write-out: dpm tile (+=), accumulator slab (+=) in the register order of attn_energies_bwd_mfma
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nl = 0; nl < 2; ++nl)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int tl = 16 * mt + 4 * kq + e;
                if (t0 + tl < d.Ti) dpm[((long)b * d.Ti + t0 + tl) * A + 32 * wave + 16 * nl + r] += dpm_acc[mt][nl][e];
            }
#pragma unroll
    for (int nl = 0; nl < 2; ++nl) {
        dv_acc[nl] += __shfl_xor(dv_acc[nl], 16, 64);
        dv_acc[nl] += __shfl_xor(dv_acc[nl], 32, 64);
        if (kq == 0) acc_wg[32 * wave + 16 * nl + r] += dv_acc[nl];
    }
    {
        float4* pw = reinterpret_cast<float4*>(acc_wg + A + threadIdx.x * 16);
#pragma unroll
        for (int ml = 0; ml < 2; ++ml)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float4 o = pw[ml * 2 + nt];
                o.x += c4[ml][nt][0]; o.y += c4[ml][nt][1]; o.z += c4[ml][nt][2]; o.w += c4[ml][nt][3];
                pw[ml * 2 + nt] = o;
            }
        float4* pc = reinterpret_cast<float4*>(acc_wg + A + A * F + threadIdx.x * 8);
#pragma unroll
        for (int nl = 0; nl < 2; ++nl) {
            float4 o = pc[nl];
            o.x += c5[nl][0]; o.y += c5[nl][1]; o.z += c5[nl][2]; o.w += c5[nl][3];
            pc[nl] = o;
        }
    }
}

}  // namespace

// entry points used by attention.hip when the dimensions match (128, 32, 31)
void gt_launch_attn_energies_fwd_mfma(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum,
                                      const float* pm, const float* Wc, const float* Wd, const float* v,
                                      const int* lengths, float* q_out, float* energies, const float* wpack, hipStream_t st,
                                      float* u_out) {
    dim3 grid(d.B, cdiv(d.Ti, CT));
    hipLaunchKernelGGL(attn_energies_fwd_mfma, grid, dim3(256), 0, st, d, qpart, w_prev, w_cum, pm, Wc, Wd, v, lengths,
                       q_out, energies, wpack, u_out);
}

void gt_launch_attn_step_fwd_mfma(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum, const float* pm,
                                  const float* Wc, const float* Wd, const float* v, const int* lengths, float* q_out,
                                  const float* wpack, float* u_out, float* erow, unsigned* row_cnt, unsigned step1,
                                  int* timeout_flag, const float* memory, long ld_mem, float* w_out, const float* cum_in,
                                  float* cum_out, float* c0, long ld0, float* c1, long ld1, float* c2, long ld2,
                                  hipStream_t st) {
    AttnStepFwd a;
    a.d = d; a.qpart = qpart; a.w_prev = w_prev; a.w_cum = w_cum; a.pm = pm; a.Wc = Wc; a.Wd = Wd; a.v = v; a.lengths = lengths;
    a.q_out = q_out; a.wpack = wpack; a.u_out = u_out; a.erow = erow; a.row_cnt = row_cnt; a.step1 = step1;
    a.timeout_flag = timeout_flag; a.memory = memory; a.ld_mem = ld_mem; a.w_out = w_out; a.cum_in = cum_in; a.cum_out = cum_out;
    a.c0 = c0; a.ld0 = ld0; a.c1 = c1; a.ld1 = ld1; a.c2 = c2; a.ld2 = ld2;
    a.NC = cdiv(d.Ti, CT);
    const int groups = cdiv(d.B, 8);
    hipLaunchKernelGGL(attn_step_fwd_mfma, dim3(groups * 8 * a.NC), dim3(256), 0, st, a);
}

static long long* g_attn_stamps = nullptr;
static int g_attn_stamp_wgs = 0;
// diagnostics: the stamps of the last stamped gt_attn_step2_fwd launch (8 per workgroup) into a HOST array of 8 * 4096;
// returns the number of workgroups; synchronises the device
extern "C" int gt_debug_attn_stamps(int64_t* out) {
    if (g_attn_stamps == nullptr || out == nullptr) return 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(out, g_attn_stamps, sizeof(long long) * 8 * (size_t)g_attn_stamp_wgs, hipMemcpyDeviceToHost);
    return g_attn_stamp_wgs;
}

extern "C" int gt_debug_attn_tail_stamps(int64_t* out) {
    if (g_attn_stamps == nullptr || out == nullptr) return 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(out, g_attn_stamps + 8 * 4096, sizeof(long long) * 4 * (size_t)g_attn_stamp_wgs, hipMemcpyDeviceToHost);
    return g_attn_stamp_wgs;
}

void gt_launch_attn_step2_fwd_mfma(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum, const float* pm,
                                   const float* Wc, const float* Wd, const float* v, const int* lengths, float* q_out,
                                   const float* wpack, float* u_out, float* xbuf, long xs, unsigned* row_cnt, unsigned step1,
                                   const float* memory, long ld_mem, float* w_out, const float* cum_in, float* cum_out,
                                   float* c0, long ld0, float* c1, long ld1, float* c2, long ld2, const GtRecTail* tail,
                                   hipStream_t st) {
    AttnStep2Fwd a;
    memset(&a.tail, 0, sizeof(a.tail));
    if (tail != nullptr && tail->work != nullptr) a.tail = *tail;
    a.d = d; a.qpart = qpart; a.w_prev = w_prev; a.w_cum = w_cum; a.pm = pm; a.Wc = Wc; a.Wd = Wd; a.v = v; a.lengths = lengths;
    a.q_out = q_out; a.wpack = wpack; a.u_out = u_out; a.xbuf = xbuf; a.xs = xs; a.row_cnt = row_cnt; a.step1 = step1;
    a.memory = memory; a.ld_mem = ld_mem; a.w_out = w_out; a.cum_in = cum_in; a.cum_out = cum_out;
    a.c0 = c0; a.ld0 = ld0; a.c1 = c1; a.ld1 = ld1; a.c2 = c2; a.ld2 = ld2;
    a.NC = cdiv(d.Ti, CT);
    static const int dbg = gt_env_item("GANTRON_DEBUG", "attn2", 0);
    a.dbg = dbg;
    const int groups = cdiv(d.B, 8);
    a.stamps = nullptr; a.tstamps = nullptr;
    static const int want_stamps = gt_env_item("GANTRON_DEBUG", "attn_stamps", 0);
    if (want_stamps && groups * 8 * a.NC <= 4096) {
        if (g_attn_stamps == nullptr) (void)hipMalloc(&g_attn_stamps, sizeof(long long) * 12 * 4096);
        a.stamps = g_attn_stamps;
        g_attn_stamp_wgs = groups * 8 * a.NC;
        (void)hipMemsetAsync(g_attn_stamps, 0, sizeof(long long) * 8 * (size_t)g_attn_stamp_wgs, st);
        if (a.tail.work != nullptr) {       // (the last step of a loop carries no tail: the stamps of the one before it stay)
            a.tstamps = g_attn_stamps + 8 * 4096;
            (void)hipMemsetAsync(a.tstamps, 0, sizeof(long long) * 4 * (size_t)g_attn_stamp_wgs, st);
        }
    }
    const dim3 grid(groups * 8 * a.NC);
    if (a.tail.work != nullptr) {
        if (a.stamps != nullptr) hipLaunchKernelGGL(attn_step2_tail_fwd_mfma<true>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(attn_step2_tail_fwd_mfma<false>, grid, dim3(256), 0, st, a);
    } else {
        if (a.stamps != nullptr) hipLaunchKernelGGL(attn_step2_fwd_mfma<true>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(attn_step2_fwd_mfma<false>, grid, dim3(256), 0, st, a);
    }
}

void gt_launch_attn_energies_bwd_rec_mfma(GtAttnDims d, const float* dw, const float* w, const float* u, const float* Wc,
                                          const float* Wd, const float* v, const int* lengths, float* dq, long ld_dq,
                                          float* dwcat_out, float* de_out, const float* wpack, hipStream_t st) {
    dim3 grid(d.B, cdiv(d.Ti, CT));
    hipLaunchKernelGGL(attn_energies_bwd_rec_mfma, grid, dim3(256), 0, st, d, dw, w, u, Wc, Wd, v, lengths, dq, ld_dq,
                       dwcat_out, de_out, wpack);
}

void gt_launch_attn_bwd_wterms_mfma(GtAttnDims d, int nsteps, const float* w, long w_step, const float* u, const float* Wc,
                                     const float* Wd, const float* v, const int* lengths, float* dq2, long dq2_step,
                                     long ld_dq2, float* dwcat2, float* p_hist, const float* wpack, hipStream_t st) {
    dim3 grid(d.B, cdiv(d.Ti, CT), cdiv(nsteps, WT_STEPS));
    hipLaunchKernelGGL(attn_bwd_wterms_mfma, grid, dim3(256), 0, st, d, nsteps, w, w_step, u, Wc, Wd, v, lengths, dq2, dq2_step,
                       ld_dq2, dwcat2, p_hist, wpack);
}

void gt_launch_attn_bwd_step_mfma(GtAttnDims d, GtSlabs da, GtSlabs db, const float* dctx_direct, long ld_direct,
                                  const float* memory, long ld_mem, const int* lengths, float* dwcat_in, float* dcum_acc,
                                  const float* dw_extern, float* dctx_out, long ld_dctx_out, const float* w, const float* u,
                                  const float* v, const float* p, const float* sum_prev, const float* dwcat2_prev,
                                  float* sum_out, float* dq, long ld_dq, float* dwcat_out, float* e1_out, bool vec,
                                  bool vec_slabs, hipStream_t st, const GtCellBwdTail* cell) {
    AttnBwdStep a;
    a.d = d; a.da = da; a.db = db; a.dctx_direct = dctx_direct; a.ld_direct = ld_direct; a.memory = memory; a.ld_mem = ld_mem;
    a.lengths = lengths; a.dwcat_in = dwcat_in; a.dcum_acc = dcum_acc; a.dw_extern = dw_extern; a.dctx_out = dctx_out;
    a.ld_dctx_out = ld_dctx_out; a.w = w; a.u = u; a.v = v; a.p = p; a.sum_prev = sum_prev;
    a.dwcat2_prev = dwcat2_prev; a.sum_out = sum_out; a.dq = dq; a.ld_dq = ld_dq; a.dwcat_out = dwcat_out; a.e1_out = e1_out;
    a.tail = {};
    dim3 grid(d.B, cdiv(d.Ti, CT));
    const size_t dyn = (size_t)d.E * 4;
    if (cell != nullptr) {
        AttnBwdStep::Tail& c = a.tail;
        c.dh_a = cell->dh_a; c.dh_c = cell->dh_c; c.dq2 = cell->dq2; c.Wq = cell->Wq; c.ldwq = (long)cell->ldwq;
        c.dc_next = cell->dc_next; c.gates = cell->gates; c.c_new = cell->c_new; c.c_prev = cell->c_prev; c.keep = cell->keep;
        c.scale = 1.0f / (1.0f - cell->p_drop); c.dgates = cell->dgates; c.dc_prev = cell->dc_prev; c.H = cell->H;
        c.row_cnt = cell->row_cnt; c.launch = cell->launch; c.dg_amax = (unsigned long long*)cell->dg_amax;
        if (vec && d.E == 1024 && vec_slabs) hipLaunchKernelGGL((attn_bwd_step_kernel<true, 4, true>), grid, dim3(256), dyn, st, a);
        else if (vec) hipLaunchKernelGGL((attn_bwd_step_kernel<true, 0, true>), grid, dim3(256), dyn, st, a);
        else hipLaunchKernelGGL((attn_bwd_step_kernel<false, 0, true>), grid, dim3(256), dyn, st, a);
        return;
    }
    if (vec && d.E == 1024 && vec_slabs) hipLaunchKernelGGL((attn_bwd_step_kernel<true, 4>), grid, dim3(256), dyn, st, a);
    else if (vec) hipLaunchKernelGGL((attn_bwd_step_kernel<true, 0>), grid, dim3(256), dyn, st, a);
    else hipLaunchKernelGGL((attn_bwd_step_kernel<false, 0>), grid, dim3(256), dyn, st, a);
}

void gt_launch_attn_bwd_fixup(GtAttnDims d, int nsteps, int nch, const float* sum, const float* w, long w_step, float* de,
                              float* dq, long dq_step, long ld_dq, const float* dq2, hipStream_t st) {
    const long n = (long)d.B * (d.Ti > A ? d.Ti : A);
    dim3 grid((unsigned)((n + 255) / 256), nsteps);
    hipLaunchKernelGGL(attn_bwd_fixup_kernel, grid, dim3(256), 0, st, d, nch, sum, w, w_step, de, dq, dq_step, ld_dq, dq2);
}

void gt_launch_attn_param_grads_mfma(GtAttnDims d, int nsteps, const float* u, const float* de, const float* w_prev,
                                     const float* w_cum, long step_stride, const float* Wc, const float* Wd,
                                     const float* v, const int* lengths, float* dpm, float* acc, const float* wpack,
                                     hipStream_t st) {
    dim3 grid(cdiv(d.Ti, CT), d.B);
    hipLaunchKernelGGL(attn_param_grads_mfma, grid, dim3(256), 0, st, d, nsteps, u, de, w_prev, w_cum, step_stride, Wc, Wd,
                       v, lengths, dpm, acc, wpack);
}

void gt_launch_attn_energies_bwd_mfma(GtAttnDims d, const float* dw, const float* w, const float* q,
                                      const float* w_prev, const float* w_cum, const float* pm, const float* Wc,
                                      const float* Wd, const float* v, const int* lengths, float* dq, long ld_dq,
                                      float* dpm, float* dwcat_out, float* acc, const float* wpack, hipStream_t st) {
    dim3 grid(cdiv(d.Ti, CT), d.B);
    hipLaunchKernelGGL(attn_energies_bwd_mfma, grid, dim3(256), 0, st, d, dw, w, q, w_prev, w_cum, pm, Wc, Wd, v, lengths,
                       dq, ld_dq, dpm, dwcat_out, acc, wpack);
}

__global__ void attn_acc_reduce_mfma_kernel(const float* acc, int nwg, float* dv, float* dWd, float* dWc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PER_MFMA) return;
    float s = 0.f;
    for (int w = 0; w < nwg; ++w) s += acc[(long)w * PER_MFMA + i];
    if (i < A) { dv[i] += s; return; }
    int j = i - A;
    if (j < 256 * 16) {            // dWd: thread*16 + (ml*2 + nt)*4 + e  ->  row 32w + 16ml + 4kq + e, column 16nt + r
        const int tid = j >> 4, q = j & 15, ml = q >> 3, nt = (q >> 2) & 1, e = q & 3;
        const int wave = tid >> 6, lane = tid & 63, r = lane & 15, kq = lane >> 4;
        dWd[(32 * wave + 16 * ml + 4 * kq + e) * F + 16 * nt + r] += s;
        return;
    }
    j -= 256 * 16;                 // dWc: thread*8 + nl*4 + e  ->  filter 16*(wave>>1) + 4kq + e, tap 16*(2*(wave&1)+nl) + r
    const int tid = j >> 3, q = j & 7, nl = q >> 2, e = q & 3;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 15, kq = lane >> 4;
    const int kk = 16 * (2 * (wave & 1) + nl) + r;
    if (kk < KK) dWc[(16 * (wave >> 1) + 4 * kq + e) * KK + kk] += s;
}

int gt_attn_acc_per_mfma() { return PER_MFMA; }
void gt_launch_attn_acc_reduce_mfma(const float* acc, int nwg, float* dv, float* dWd, float* dWc, hipStream_t st) {
    hipLaunchKernelGGL(attn_acc_reduce_mfma_kernel, dim3((PER_MFMA + 255) / 256), dim3(256), 0, st, acc, nwg, dv, dWd, dWc);
}
int gt_attn_wpack_elems_mfma() { return F * LDC + A * LDD; }
void gt_launch_attn_pack_weights(const float* Wc, const float* Wd, float* out, hipStream_t st) {
    hipLaunchKernelGGL(attn_pack_weights_kernel, dim3(8), dim3(256), 0, st, Wc, Wd, out);
}
