// Location-sensitive attention, one decoder step (reference model.py:27-31, 48-91), forward and
// backward, split so that every phase fills the chip:
//   energies  : grid (chunks of 32 encoder positions, batch)  -- VALU-bound (conv31 + 32->128 dense + tanh)
//   context   : grid (256-column chunks of E, batch)           -- streams `memory` (B*Ti*E floats / step)
// Both kernels keep their small weights (location conv/dense, v), the window of previous/cumulative
// alignments and the per-position intermediates in LDS; softmax reductions use wave64 shuffles.
// The backward kernels recompute the tanh arguments (the reference's autograd saves a
// (B,Ti,128) tensor per step: 3.9 GB over 800 steps) and accumulate the location-layer parameter
// gradients in per-workgroup slabs (no atomics) that gt_attn_acc_reduce folds after the loop.
#include <stdlib.h>
#include "common.h"
#include "rec_tail.h"
#include "attn_ctx_bwd.h"

// matrix-core versions for the reference's default dimensions (attention_mfma.hip)
void gt_launch_attn_energies_bwd_rec_mfma(GtAttnDims d, const float* dw, const float* w, const float* u, const float* Wc,
                                          const float* Wd, const float* v, const int* lengths, float* dq, long ld_dq,
                                          float* dwcat_out, float* de_out, const float* wpack, hipStream_t st);
void gt_launch_attn_param_grads_mfma(GtAttnDims d, int nsteps, const float* u, const float* de, const float* w_prev,
                                     const float* w_cum, long step_stride, const float* Wc, const float* Wd,
                                     const float* v, const int* lengths, float* dpm, float* acc, const float* wpack,
                                     hipStream_t st);
void gt_launch_attn_energies_fwd_mfma(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum,
                                      const float* pm, const float* Wc, const float* Wd, const float* v,
                                      const int* lengths, float* q_out, float* energies, const float* wpack, hipStream_t st,
                                      float* u_out);
void gt_launch_attn_energies_bwd_mfma(GtAttnDims d, const float* dw, const float* w, const float* q,
                                      const float* w_prev, const float* w_cum, const float* pm, const float* Wc,
                                      const float* Wd, const float* v, const int* lengths, float* dq, long ld_dq,
                                      float* dpm, float* dwcat_out, float* acc, const float* wpack, hipStream_t st);
void gt_launch_attn_step_fwd_mfma(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum, const float* pm,
                                  const float* Wc, const float* Wd, const float* v, const int* lengths, float* q_out,
                                  const float* wpack, float* u_out, float* erow, unsigned* row_cnt, unsigned step1,
                                  int* timeout_flag, const float* memory, long ld_mem, float* w_out, const float* cum_in,
                                  float* cum_out, float* c0, long ld0, float* c1, long ld1, float* c2, long ld2,
                                  hipStream_t st);
void gt_launch_attn_step2_fwd_mfma(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum, const float* pm,
                                   const float* Wc, const float* Wd, const float* v, const int* lengths, float* q_out,
                                   const float* wpack, float* u_out, float* xbuf, long xs, unsigned* row_cnt, unsigned step1,
                                   const float* memory, long ld_mem, float* w_out, const float* cum_in, float* cum_out,
                                   float* c0, long ld0, float* c1, long ld1, float* c2, long ld2, const GtRecTail* tail,
                                   hipStream_t st);
void gt_launch_attn_bwd_wterms_mfma(GtAttnDims d, int nsteps, const float* w, long w_step, const float* u, const float* Wc,
                                     const float* Wd, const float* v, const int* lengths, float* dq2, long dq2_step,
                                     long ld_dq2, float* dwcat2, float* p_hist, const float* wpack, hipStream_t st);
void gt_launch_attn_bwd_step_mfma(GtAttnDims d, GtSlabs da, GtSlabs db, const float* dctx_direct, long ld_direct,
                                  const float* memory, long ld_mem, const int* lengths, float* dwcat_in, float* dcum_acc,
                                  const float* dw_extern, float* dctx_out, long ld_dctx_out, const float* w, const float* u,
                                  const float* v, const float* p, const float* sum_prev, const float* dwcat2_prev,
                                  float* sum_out, float* dq, long ld_dq, float* dwcat_out, float* e1_out, bool vec,
                                  bool vec_slabs, hipStream_t st, const GtCellBwdTail* cell);
void gt_launch_attn_bwd_fixup(GtAttnDims d, int nsteps, int nch, const float* sum, const float* w, long w_step, float* de,
                              float* dq, long dq_step, long ld_dq, const float* dq2, hipStream_t st);
int gt_attn_wpack_elems_mfma();
int gt_attn_acc_per_mfma();
void gt_launch_attn_acc_reduce_mfma(const float* acc, int nwg, float* dv, float* dWd, float* dWc, hipStream_t st);
void gt_launch_attn_pack_weights(const float* Wc, const float* Wd, float* out, hipStream_t st);

namespace {

constexpr int CT = 32;       // encoder positions per workgroup
constexpr int MAXA = 128;    // attention_dim
constexpr int MAXF = 32;     // attention_location_n_filters
constexpr int MAXK = 31;     // attention_location_kernel_size
constexpr int WINW = CT + MAXK - 1;
constexpr int LDW = MAXF + 1;      // Wd row stride in LDS
constexpr int LDL = MAXF + 4;      // locf row stride in LDS (16-byte aligned rows)
constexpr int LDS_A = MAXA + 1;    // [CT][A] scratch row stride
constexpr int MAX_TI = 2048;

struct Shared {
    alignas(16) float locf[CT * LDL];     // 16-byte aligned rows: read with ds_read_b128
    alignas(16) float dlocf[CT * LDL];
    alignas(16) float buf[CT * LDS_A];    // fwd: v*tanh ; bwd: ds
    float q[MAXA];
    float v[MAXA];
    float wc[MAXF * 2 * MAXK];
    float wd[MAXA * LDW];
    float win[2][WINW + 2];
    float de[CT];
    float dvs[MAXA];
    float red[16];
};

// Compile-time copies of the attention dimensions (0 = take the runtime value): with constants the
// inner loops unroll and their LDS reads are issued in batches instead of one exposed read per MAC.
template <int CA, int CF, int CK>
struct Dims {
    int A, F, Kc, Ti, B;
    __device__ __forceinline__ explicit Dims(const GtAttnDims& d)
        : A(CA ? CA : d.A), F(CF ? CF : d.F), Kc(CK ? CK : d.Kc), Ti(d.Ti), B(d.B) {}
};

template <class D>
__device__ __forceinline__ void load_weights(Shared& S, const D& d, const float* Wc, const float* Wd,
                                             const float* v) {
    for (int i = threadIdx.x; i < d.F * 2 * d.Kc; i += blockDim.x) S.wc[i] = Wc[i];
    for (int i = threadIdx.x; i < d.A * d.F; i += blockDim.x) {
        const int dd = i / d.F, f = i - dd * d.F;
        S.wd[dd * LDW + f] = Wd[i];
    }
    for (int i = threadIdx.x; i < d.A; i += blockDim.x) S.v[i] = v[i];
}

template <class D>
__device__ __forceinline__ void load_window(Shared& S, const D& d, const float* w_prev, const float* w_cum,
                                            int b, int t0) {
    const int pad = (d.Kc - 1) / 2;
    const int ww = CT + d.Kc - 1;
    for (int i = threadIdx.x; i < 2 * ww; i += blockDim.x) {
        const int c = i / ww, p = i - c * ww;
        const int t = t0 + p - pad;
        const float* src = c == 0 ? w_prev : w_cum;
        S.win[c][p] = (t >= 0 && t < d.Ti) ? src[(long)b * d.Ti + t] : 0.f;
    }
}

// locf[tl][f] = sum_{c,k} Wc[f][c][k] * win[c][tl+k]
template <class D>
__device__ __forceinline__ void location_conv(Shared& S, const D& d) {
    for (int o = threadIdx.x; o < CT * d.F; o += blockDim.x) {
        const int tl = o / d.F, f = o - tl * d.F;
        float acc0 = 0.f, acc1 = 0.f;
        const float* w0 = &S.wc[(f * 2 + 0) * d.Kc];
        const float* w1 = &S.wc[(f * 2 + 1) * d.Kc];
        const float* x0 = &S.win[0][tl];
        const float* x1 = &S.win[1][tl];
#pragma unroll
        for (int k = 0; k < d.Kc; ++k) {
            acc0 += w0[k] * x0[k];
            acc1 += w1[k] * x1[k];
        }
        S.locf[tl * LDL + f] = acc0 + acc1;
    }
}

// Wd row of this thread's attention channel in registers (zero beyond F)
template <class D>
__device__ __forceinline__ void load_wrow(const Shared& S, const D& d, int dd, float (&wrow)[MAXF]) {
#pragma unroll
    for (int f = 0; f < MAXF; ++f) wrow[f] = (f < d.F) ? S.wd[dd * LDW + f] : 0.f;
}

// sum_f wrow[f] * locf[tl][f] with 16-byte LDS reads (locf rows are zero padded to MAXF)
__device__ __forceinline__ float dense_dot(const Shared& S, int tl, const float (&wrow)[MAXF]) {
    const float4* l = reinterpret_cast<const float4*>(&S.locf[tl * LDL]);
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int f4 = 0; f4 < MAXF / 4; ++f4) {
        const float4 v = l[f4];
        s0 += wrow[4 * f4 + 0] * v.x;
        s1 += wrow[4 * f4 + 1] * v.y;
        s0 += wrow[4 * f4 + 2] * v.z;
        s1 += wrow[4 * f4 + 3] * v.w;
    }
    return s0 + s1;
}

__device__ __forceinline__ void zero_locf(Shared& S) {
    for (int i = threadIdx.x; i < CT * LDL; i += blockDim.x) { S.locf[i] = 0.f; S.dlocf[i] = 0.f; }
}

// NIT = CT / (256 / A): positions handled by each thread (attention_dim 128 -> 16)
template <int NIT, int CA, int CF, int CK>
__global__ __launch_bounds__(256) void attn_energies_fwd_kernel(GtAttnDims dims, GtSlabs qpart, const float* w_prev,
                                                                const float* w_cum, const float* pm, const float* Wc,
                                                                const float* Wd, const float* v, const int* lengths,
                                                                float* q_out, float* energies) {
    __shared__ Shared S;
    const Dims<CA, CF, CK> d(dims);
    const int b = blockIdx.y, t0 = blockIdx.x * CT;
    if (lengths != nullptr && t0 >= lengths[b]) {      // chunk entirely in the padding: energies are -inf
        if (threadIdx.x < CT && t0 + threadIdx.x < d.Ti) energies[(long)b * d.Ti + t0 + threadIdx.x] = -INFINITY;
        return;
    }
    const int G = CT / NIT;
    const int dd = threadIdx.x % d.A, g = threadIdx.x / d.A;
    // first long-latency global load issued before the LDS set-up; later ones one iteration ahead
    const float* pm_b = pm + (long)b * d.Ti * d.A + dd;
    float pm_next = (t0 + g < d.Ti) ? pm_b[(long)(t0 + g) * d.A] : 0.f;
    zero_locf(S);
    load_weights(S, d, Wc, Wd, v);
    for (int i = threadIdx.x; i < d.A; i += blockDim.x) {
        const float qv = slab_sum(qpart.p, qpart.n, qpart.slab_stride, (long)b * qpart.ld + i);
        S.q[i] = qv;
        if (blockIdx.x == 0 && q_out != nullptr) q_out[(long)b * d.A + i] = qv;
    }
    load_window(S, d, w_prev, w_cum, b, t0);
    __syncthreads();
    location_conv(S, d);
    __syncthreads();
    float wrow[MAXF];
    load_wrow(S, d, dd, wrow);
    const float qd = S.q[dd], vd = S.v[dd];
#pragma unroll 2
    for (int i = 0; i < NIT; ++i) {
        const int tl = g + G * i;
        const float pm_cur = pm_next;
        const int tn = t0 + tl + G;
        pm_next = (i + 1 < NIT && tn < d.Ti) ? pm_b[(long)tn * d.A] : 0.f;
        const float sarg = qd + pm_cur + dense_dot(S, tl, wrow);
        S.buf[tl * LDS_A + dd] = (t0 + tl < d.Ti) ? vd * tanhf_(sarg) : 0.f;
    }
    __syncthreads();
    // reduce over the attention channels: 8 lanes per position
    {
        const int tl = threadIdx.x >> 3, part = threadIdx.x & 7;
        float e = 0.f;
        for (int i = part; i < d.A; i += 8) e += S.buf[tl * LDS_A + i];
        e += __shfl_xor(e, 4, 64);
        e += __shfl_xor(e, 2, 64);
        e += __shfl_xor(e, 1, 64);
        const int t = t0 + tl;
        if (part == 0 && t < d.Ti) {
            if (lengths != nullptr && t >= lengths[b]) e = -INFINITY;
            energies[(long)b * d.Ti + t] = e;
        }
    }
}

// softmax over Ti (block-wide) into LDS; returns nothing, w[] holds the normalised weights
__device__ __forceinline__ void softmax_row(const GtAttnDims& d, const float* energies, int b, float* w, float* red) {
    float mx = -INFINITY;
    for (int t = threadIdx.x; t < d.Ti; t += blockDim.x) {
        const float e = energies[(long)b * d.Ti + t];
        w[t] = e;
        mx = fmaxf(mx, e);
    }
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int t = threadIdx.x; t < d.Ti; t += blockDim.x) {
        const float ex = __expf(w[t] - mx);
        w[t] = ex;
        sum += ex;
    }
    sum = block_sum(sum, red);
    const float inv = 1.0f / sum;
    for (int t = threadIdx.x; t < d.Ti; t += blockDim.x) w[t] *= inv;
    __syncthreads();
}

// grid (256-column chunks of E, B).  VEC: E, ld_mem multiples of 4 and 16-byte aligned memory: every
// wave streams float4 columns of a quarter of the rows (8-16 rows in flight), partial sums fold through LDS.
template <bool VEC>
__global__ __launch_bounds__(256) void attn_context_fwd_kernel(GtAttnDims d, const float* energies, const float* memory,
                                                               long ld_mem, const int* lengths, float* w_out,
                                                               const float* cum_in, float* cum_out, float* c0, long ld0,
                                                               float* c1, long ld1, float* c2, long ld2) {
    __builtin_amdgcn_s_setprio(3);      // the attention chain is the serial critical path of the decoder loops: its waves
                                        // issue ahead of the co-resident waves of the other chain and of the leaf GEMMs
    __shared__ float w[MAX_TI];
    __shared__ float red[16];
    __shared__ __attribute__((aligned(16))) float psum[4][256];
    const int b = blockIdx.y;
    // rows beyond the memory length have weight exactly 0 (energy -inf): never read them
    const int Tv = lengths != nullptr ? min(lengths[b], d.Ti) : d.Ti;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col4 = min(blockIdx.x * 256 + 4 * lane, (d.E - 4) & ~3);        // clamped: loads stay unconditional
    // VEC: the first 8 memory rows of every wave are requested BEFORE the softmax (they do not depend on it);
    // afterwards 8 rows are in flight per wave while the previous 8 are consumed.  Row indices are clamped and
    // invalid rows get weight 0, so no load is predicated (predicated loads make the compiler wait with vmcnt(0)).
    const float* mrow = memory + (long)b * d.Ti * ld_mem + col4;
    float4 ra[8], rb[8];
    if (VEC) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            ra[j] = *reinterpret_cast<const float4*>(mrow + (long)min(wave + 4 * j, Tv - 1) * ld_mem);
    }
    softmax_row(d, energies, b, w, red);
    if (blockIdx.x == 0) {
        for (int t = threadIdx.x; t < d.Ti; t += blockDim.x) {
            w_out[(long)b * d.Ti + t] = w[t];
            if (cum_out != nullptr) cum_out[(long)b * d.Ti + t] = cum_in[(long)b * d.Ti + t] + w[t];
        }
    }
    float c = 0.f;
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (VEC) {
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        auto fetch = [&](float4 (&r)[8], int base) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                r[j] = *reinterpret_cast<const float4*>(mrow + (long)min(base + wave + 4 * j, Tv - 1) * ld_mem);
        };
        auto consume = [&](const float4 (&r)[8], int base) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const int t0 = base + wave + 4 * j, t1 = t0 + 4;
                const float w0 = t0 < Tv ? w[t0] : 0.f, w1 = t1 < Tv ? w[t1] : 0.f;
                a0.x += w0 * r[j].x; a0.y += w0 * r[j].y; a0.z += w0 * r[j].z; a0.w += w0 * r[j].w;
                a1.x += w1 * r[j + 1].x; a1.y += w1 * r[j + 1].y; a1.z += w1 * r[j + 1].z; a1.w += w1 * r[j + 1].w;
            }
        };
        for (int base = 0; base < Tv; base += 64) {       // two batches of 32 rows (8 per wave) per trip
            fetch(rb, base + 32);
            consume(ra, base);
            fetch(ra, base + 64);
            consume(rb, base + 32);
        }
        *reinterpret_cast<float4*>(&psum[wave][4 * lane]) = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
        __syncthreads();
        c = (psum[0][threadIdx.x] + psum[1][threadIdx.x]) + (psum[2][threadIdx.x] + psum[3][threadIdx.x]);
    } else if (col < d.E) {
        const float* m = memory + (long)b * d.Ti * ld_mem + col;
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
        int t = 0;
        for (; t + 3 < Tv; t += 4) {
            acc0 += w[t] * m[(long)t * ld_mem];
            acc1 += w[t + 1] * m[(long)(t + 1) * ld_mem];
            acc2 += w[t + 2] * m[(long)(t + 2) * ld_mem];
            acc3 += w[t + 3] * m[(long)(t + 3) * ld_mem];
        }
        for (; t < Tv; ++t) acc0 += w[t] * m[(long)t * ld_mem];
        c = (acc0 + acc1) + (acc2 + acc3);
    }
    if (col < d.E) {
        if (c0 != nullptr) c0[(long)b * ld0 + col] = c;
        if (c1 != nullptr) c1[(long)b * ld1 + col] = c;
        if (c2 != nullptr) c2[(long)b * ld2 + col] = c;
    }
}

// grid (B, chunks of CT positions).  dw[t'] = memory[b][t'] . dctx + carried alignment gradients (attn_ctx_bwd.h).
template <bool VEC, int NE4 = 0>
__global__ __launch_bounds__(256) void attn_context_bwd_kernel(GtAttnDims d, GtSlabs da, GtSlabs db,
                                                               const float* dctx_direct, long ld_direct,
                                                               const float* memory, long ld_mem, const int* lengths,
                                                               float* dwcat_in, float* dcum_acc, const float* dw_extern,
                                                               float* dctx_out, long ld_dctx_out, float* dq_zero, long ld_dq,
                                                               float* dw_out) {
    __builtin_amdgcn_s_setprio(3);      // the attention chain is the serial critical path of the decoder loops: its waves
                                        // issue ahead of the co-resident waves of the other chain and of the leaf GEMMs
    extern __shared__ __attribute__((aligned(16))) float dctx[];   // E floats
    __shared__ float dots[CT];
    // grid (B, chunks): b is the fast index of the linear workgroup id, so the chunks of one batch row have ids
    // congruent mod 8 and land on the same XCD -- the 13 K-split slabs of that row's context gradient are then
    // fetched into one L2 instead of into every XCD that runs one of its chunks
    const int b = blockIdx.x, chunk = blockIdx.y, t0 = chunk * CT;
    // padded positions carry weight 0: their dw never reaches an energy gradient, so it is set to 0
    // and neither the upstream context gradient nor their memory rows are read
    const int Tv = lengths != nullptr ? min(lengths[b], d.Ti) : d.Ti;
    if (t0 >= Tv) {
        if (threadIdx.x < CT && t0 + threadIdx.x < d.Ti) {
            const long i = (long)b * d.Ti + t0 + threadIdx.x;
            if (dwcat_in != nullptr) { dwcat_in[i] = 0.f; dwcat_in[(long)d.B * d.Ti + i] = 0.f; }
            dw_out[i] = 0.f;
        }
        return;
    }
    if (chunk == 0 && dq_zero != nullptr)
        for (int i = threadIdx.x; i < d.A; i += blockDim.x) dq_zero[(long)b * ld_dq + i] = 0.f;
    const float dw = attn_ctx_bwd_dw<VEC, NE4>(d, da, db, dctx_direct, ld_direct, memory, ld_mem, Tv, dwcat_in, dcum_acc,
                                               dw_extern, dctx_out, ld_dctx_out, nullptr, 0.f, dctx, dots, b, chunk);
    if (threadIdx.x < CT && t0 + threadIdx.x < d.Ti) dw_out[(long)b * d.Ti + t0 + threadIdx.x] = dw;
}

// grid (chunks, B).  Energies backward for CT positions; see file header.
template <int NIT, int CA, int CF, int CK>
__global__ __launch_bounds__(256) void attn_energies_bwd_kernel(GtAttnDims dims, const float* dw, const float* w,
                                                                const float* q, const float* w_prev, const float* w_cum,
                                                                const float* pm, const float* Wc, const float* Wd,
                                                                const float* v, const int* lengths, float* dq, long ld_dq,
                                                                float* dpm, float* dwcat_out, float* acc) {
    __shared__ Shared S;
    const Dims<CA, CF, CK> d(dims);
    const int b = blockIdx.y, t0 = blockIdx.x * CT;
    if (lengths != nullptr && t0 >= lengths[b]) return;      // all-padding chunk: every gradient is zero
    const int G = CT / NIT;
    const int dd = threadIdx.x % d.A, g = threadIdx.x / d.A;
    // first long-latency global loads issued before the LDS set-up; later ones one iteration ahead
    const long row0 = (long)b * d.Ti * d.A + dd;
    float pm_next = 0.f, dpm_next = 0.f;
    if (t0 + g < d.Ti) { pm_next = pm[row0 + (long)(t0 + g) * d.A]; dpm_next = dpm[row0 + (long)(t0 + g) * d.A]; }
    zero_locf(S);
    load_weights(S, d, Wc, Wd, v);
    for (int i = threadIdx.x; i < d.A; i += blockDim.x) S.q[i] = q[(long)b * d.A + i];
    load_window(S, d, w_prev, w_cum, b, t0);
    // softmax backward: de = w * (dw - sum_t w*dw)
    float part = 0.f;
    for (int t = threadIdx.x; t < d.Ti; t += blockDim.x) part += w[(long)b * d.Ti + t] * dw[(long)b * d.Ti + t];
    const float dot = block_sum(part, S.red);
    if (threadIdx.x < CT) {
        const int t = t0 + threadIdx.x;
        S.de[threadIdx.x] = (t < d.Ti) ? w[(long)b * d.Ti + t] * (dw[(long)b * d.Ti + t] - dot) : 0.f;
    }
    __syncthreads();
    location_conv(S, d);
    __syncthreads();

    const int per_wg = d.A + d.A * d.F + d.F * 2 * d.Kc;
    float* acc_wg = acc + ((long)b * gridDim.x + blockIdx.x) * per_wg;
    float wrow[MAXF], dwd[MAXF];
    load_wrow(S, d, dd, wrow);
#pragma unroll
    for (int f = 0; f < MAXF; ++f) dwd[f] = 0.f;
    const float qd = S.q[dd], vd = S.v[dd];
    float dq_acc = 0.f, dv_acc = 0.f;
#pragma unroll 2
    for (int i = 0; i < NIT; ++i) {
        const int tl = g + G * i;
        const float pm_cur = pm_next, dpm_cur = dpm_next;
        const int tn = t0 + tl + G;
        if (i + 1 < NIT && tn < d.Ti) { pm_next = pm[row0 + (long)tn * d.A]; dpm_next = dpm[row0 + (long)tn * d.A]; }
        const float4* l = reinterpret_cast<const float4*>(&S.locf[tl * LDL]);
        float4 lv[MAXF / 4];
#pragma unroll
        for (int f4 = 0; f4 < MAXF / 4; ++f4) lv[f4] = l[f4];
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int f4 = 0; f4 < MAXF / 4; ++f4) {
            s0 += wrow[4 * f4 + 0] * lv[f4].x; s1 += wrow[4 * f4 + 1] * lv[f4].y;
            s0 += wrow[4 * f4 + 2] * lv[f4].z; s1 += wrow[4 * f4 + 3] * lv[f4].w;
        }
        float ds = 0.f;
        if (t0 + tl < d.Ti) {
            const float u = tanhf_(qd + pm_cur + s0 + s1);
            const float de = S.de[tl];
            ds = de * vd * (1.f - u * u);
            dv_acc += de * u;
            dq_acc += ds;
            dpm[row0 + (long)(t0 + tl) * d.A] = dpm_cur + ds;
        }
        // dWd[dd][f] += ds * locf[tl][f]
#pragma unroll
        for (int f4 = 0; f4 < MAXF / 4; ++f4) {
            dwd[4 * f4 + 0] += ds * lv[f4].x; dwd[4 * f4 + 1] += ds * lv[f4].y;
            dwd[4 * f4 + 2] += ds * lv[f4].z; dwd[4 * f4 + 3] += ds * lv[f4].w;
        }
        S.buf[tl * LDS_A + dd] = ds;
    }
    atomicAdd(&dq[(long)b * ld_dq + dd], dq_acc);
    if (threadIdx.x < d.A) S.dvs[threadIdx.x] = 0.f;
    __syncthreads();                                   // ds complete in S.buf, dvs zeroed
    atomicAdd(&S.dvs[dd], dv_acc);

    // dlocf[tl][f] = sum_d ds[tl][d] * Wd[d][f]: thread -> (f, 4 positions tl = tq + 8*j)
    {
        const int f = threadIdx.x % MAXF, tq = threadIdx.x / MAXF;      // tq in [0, 8)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (f < d.F) {
#pragma unroll 8
            for (int i = 0; i < d.A; ++i) {
                const float wv = S.wd[i * LDW + f];
                a0 += S.buf[(tq + 0) * LDS_A + i] * wv;
                a1 += S.buf[(tq + 8) * LDS_A + i] * wv;
                a2 += S.buf[(tq + 16) * LDS_A + i] * wv;
                a3 += S.buf[(tq + 24) * LDS_A + i] * wv;
            }
            S.dlocf[(tq + 0) * LDL + f] = a0;
            S.dlocf[(tq + 8) * LDL + f] = a1;
            S.dlocf[(tq + 16) * LDL + f] = a2;
            S.dlocf[(tq + 24) * LDL + f] = a3;
        }
    }
    __syncthreads();                                   // dlocf + dvs complete; S.buf free for reuse
    // fold the G partial dWd rows through LDS (S.buf as [G][A][F] needs G*A*F <= CT*LDS_A: 256*32 vs 32*129)
    float* acc_wd = acc_wg + d.A;
    for (int gg = 0; gg < G; ++gg) {
        if (g == gg) {
#pragma unroll
            for (int f = 0; f < MAXF; ++f) {
                float* slot = &S.wd[dd * LDW + f];      // Wd no longer needed: reuse as the dWd accumulator
                *slot = (gg == 0 ? 0.f : *slot) + dwd[f];
            }
        }
        __syncthreads();
    }
    // batched read-modify-write of this workgroup's slab: loads first, then adds and stores
    {
        const int nwd = d.A * d.F;
        float old[(MAXA * MAXF) / 256];
#pragma unroll
        for (int i = 0; i < (MAXA * MAXF) / 256; ++i) {
            const int o = threadIdx.x + 256 * i;
            old[i] = (o < nwd) ? acc_wd[o] : 0.f;
        }
        const float dv_old = (threadIdx.x < d.A) ? acc_wg[threadIdx.x] : 0.f;
#pragma unroll
        for (int i = 0; i < (MAXA * MAXF) / 256; ++i) {
            const int o = threadIdx.x + 256 * i;
            if (o < nwd) acc_wd[o] = old[i] + S.wd[(o / d.F) * LDW + (o % d.F)];
        }
        if (threadIdx.x < d.A) acc_wg[threadIdx.x] = dv_old + S.dvs[threadIdx.x];
    }
    // dWc[f][c][k] += sum_tl dlocf[tl][f] * win[c][tl+k]
    float* acc_wc = acc_wg + d.A + d.A * d.F;
    {
        constexpr int NWC = (MAXF * 2 * MAXK + 255) / 256;
        const int nwc = d.F * 2 * d.Kc;
        float old[NWC], val[NWC];
#pragma unroll
        for (int i = 0; i < NWC; ++i) {
            const int o = threadIdx.x + 256 * i;
            old[i] = (o < nwc) ? acc_wc[o] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NWC; ++i) {
            const int o = threadIdx.x + 256 * i;
            float a = 0.f;
            if (o < nwc) {
                const int f = o / (2 * d.Kc), r = o - f * 2 * d.Kc, c = r / d.Kc, k = r - c * d.Kc;
#pragma unroll
                for (int tl = 0; tl < CT; ++tl) a += S.dlocf[tl * LDL + f] * S.win[c][tl + k];
            }
            val[i] = a;
        }
#pragma unroll
        for (int i = 0; i < NWC; ++i) {
            const int o = threadIdx.x + 256 * i;
            if (o < nwc) acc_wc[o] = old[i] + val[i];
        }
    }
    // d[w_prev; w_cum][p] = sum_{f,k} dlocf[p-k][f] * Wc[f][c][k]   (transposed conv over the window)
    const int pad = (d.Kc - 1) / 2, ww = CT + d.Kc - 1;
    {
        // 2 lanes per output (split the filter range), 2*ww <= 124 outputs -> 248 threads busy
        const int o = threadIdx.x >> 1, half = threadIdx.x & 1;
        float a = 0.f;
        int t = -1, c = 0;
        if (o < 2 * ww) {
            c = o / ww;
            const int p = o - c * ww;
            t = t0 + p - pad;
            if (t >= 0 && t < d.Ti) {
                const int fh = (d.F + 1) / 2;
                const int f0 = half * fh, f1 = min(d.F, f0 + fh);
                for (int k = 0; k < d.Kc; ++k) {
                    const int tl = p - k;
                    if (tl < 0 || tl >= CT) continue;
                    for (int f = f0; f < f1; ++f) a += S.dlocf[tl * LDL + f] * S.wc[(f * 2 + c) * d.Kc + k];
                }
            }
        }
        a += __shfl_xor(a, 1, 64);
        if (half == 0 && o < 2 * ww && t >= 0 && t < d.Ti)
            atomicAdd(&dwcat_out[(long)c * d.B * d.Ti + (long)b * d.Ti + t], a);
    }
}

__global__ void attn_acc_reduce_kernel(const float* acc, int nwg, int per, int A, int AF, float* dv, float* dWd,
                                       float* dWc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per) return;
    float s = 0.f;
    for (int w = 0; w < nwg; ++w) s += acc[(long)w * per + i];
    if (i < A) dv[i] += s;
    else if (i < A + AF) dWd[i - A] += s;
    else dWc[i - A - AF] += s;
}

int check_dims(const GtAttnDims& d, const char* who) {
    GT_REQUIRE(d.B > 0 && d.Ti > 0 && d.Ti <= MAX_TI && d.E > 0, GT_E_ARG, "%s: bad B/Ti/E (%d,%d,%d)", who, d.B, d.Ti, d.E);
    GT_REQUIRE(d.A >= 8 && d.A <= MAXA && 256 % d.A == 0, GT_E_LIMIT, "%s: attention_dim %d must be a power of two in [8, %d]", who, d.A, MAXA);
    GT_REQUIRE(d.F > 0 && d.F <= MAXF && d.Kc > 0 && d.Kc <= MAXK && (d.Kc & 1), GT_E_LIMIT,
               "%s: location filters %d (<=%d) / kernel %d (odd, <=%d)", who, d.F, MAXF, d.Kc, MAXK);
    return 0;
}

}  // namespace

extern "C" int32_t gt_attn_chunks(int32_t Ti) { return cdiv(Ti, CT); }

static bool use_mfma(const GtAttnDims& d) { return d.A == 128 && d.F == 32 && d.Kc == 31 && !gt_env_item("GANTRON_DEBUG", "attn_valu", 0); }

extern "C" int64_t gt_attn_acc_elems(GtAttnDims d) {
    // per workgroup: the logical [dv | dWd | dWc] block, or the matrix-core kernel's register-order block
    const int per = use_mfma(d) ? gt_attn_acc_per_mfma() : d.A + d.A * d.F + d.F * 2 * d.Kc;
    return (int64_t)d.B * cdiv(d.Ti, CT) * per;
}

extern "C" int64_t gt_attn_wpack_elems(GtAttnDims d) {
    return (d.A == 128 && d.F == 32 && d.Kc == 31) ? gt_attn_wpack_elems_mfma() : 0;
}

extern "C" int gt_attn_pack_weights(GtAttnDims d, const float* Wc, const float* Wd, float* wpack, void* stream) {
    GT_REQUIRE(Wc && Wd && wpack && aligned16(wpack), GT_E_ARG, "gt_attn_pack_weights: bad argument");
    GT_REQUIRE(gt_attn_wpack_elems(d) > 0, GT_E_ARG, "gt_attn_pack_weights: only the (128, 32, 31) matrix-core path stages packed weights");
    gt_launch_attn_pack_weights(Wc, Wd, wpack, (hipStream_t)stream);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_attn_energies_fwd(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum,
                                    const float* pm, const float* Wc, const float* Wd, const float* v,
                                    const int32_t* lengths, float* q_out, float* energies, void* stream) {
    return gt_attn_energies_fwd_w(d, qpart, w_prev, w_cum, pm, Wc, Wd, v, lengths, q_out, energies, nullptr, stream);
}

extern "C" int gt_attn_energies_fwd_w(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum,
                                      const float* pm, const float* Wc, const float* Wd, const float* v,
                                      const int32_t* lengths, float* q_out, float* energies, const float* wpack,
                                      void* stream) {
    return gt_attn_energies_fwd_u(d, qpart, w_prev, w_cum, pm, Wc, Wd, v, lengths, q_out, energies, wpack, nullptr, stream);
}

extern "C" int32_t gt_attn_hoist_supported(GtAttnDims d) { return use_mfma(d) ? 1 : 0; }

extern "C" int gt_attn_energies_fwd_u(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum,
                                      const float* pm, const float* Wc, const float* Wd, const float* v,
                                      const int32_t* lengths, float* q_out, float* energies, const float* wpack,
                                      float* u_out, void* stream) {
    int rc = check_dims(d, "gt_attn_energies_fwd");
    GT_REQUIRE(u_out == nullptr || use_mfma(d), GT_E_ARG, "gt_attn_energies_fwd_u: u_out needs the (128, 32, 31) dimensions");
    if (rc) return rc;
    GT_REQUIRE(qpart.p && w_prev && w_cum && pm && Wc && Wd && v && energies, GT_E_ARG, "gt_attn_energies_fwd: NULL pointer");
    dim3 grid(cdiv(d.Ti, CT), d.B);
    GtProfScope prof(GT_FAM_ATTN_E_FWD, 2.0 * d.B * d.Ti * (double)(d.F * 2 * d.Kc + d.A * d.F + 2 * d.A),
                     4.0 * d.B * d.Ti * (d.A + 3.0), (hipStream_t)stream);
#define GT_EF(NIT, CA, CF, CK)                                                                                   \
    hipLaunchKernelGGL((attn_energies_fwd_kernel<NIT, CA, CF, CK>), grid, dim3(256), 0, (hipStream_t)stream, d,  \
                       qpart, w_prev, w_cum, pm, Wc, Wd, v, lengths, q_out, energies)
    if (d.A == 128 && d.F == 32 && d.Kc == 31 && !gt_env_item("GANTRON_DEBUG", "attn_valu", 0)) {
        gt_launch_attn_energies_fwd_mfma(d, qpart, w_prev, w_cum, pm, Wc, Wd, v, lengths, q_out, energies, wpack,
                                         (hipStream_t)stream, u_out);      // the reference's default hparams
    } else if (d.A == 128 && d.F == 32 && d.Kc == 31) {
        GT_EF(16, 128, 32, 31);
    } else {
        switch (CT / (256 / d.A)) {
            case 1: GT_EF(1, 0, 0, 0); break;
            case 2: GT_EF(2, 0, 0, 0); break;
            case 4: GT_EF(4, 0, 0, 0); break;
            case 8: GT_EF(8, 0, 0, 0); break;
            default: GT_EF(16, 0, 0, 0); break;
        }
    }
#undef GT_EF
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int32_t gt_attn_step_supported(GtAttnDims d) {
    return (use_mfma(d) && d.Ti <= 256 && d.E % 4 == 0) ? 1 : 0;
}

extern "C" int gt_attn_step_fwd(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum, const float* pm,
                                const float* Wc, const float* Wd, const float* v, const int32_t* lengths, float* q_out,
                                const float* wpack, float* u_out, float* erow, uint32_t* row_cnt, uint32_t step1,
                                int32_t* timeout_flag, const float* memory, int64_t ld_mem, float* w_out, const float* cum_in,
                                float* cum_out, float* c0, int64_t ld0, float* c1, int64_t ld1, float* c2, int64_t ld2,
                                void* stream) {
    int rc = check_dims(d, "gt_attn_step_fwd");
    if (rc) return rc;
    GT_REQUIRE(gt_attn_step_supported(d), GT_E_LIMIT, "gt_attn_step_fwd: needs attention dims (128, 32, 31), Ti <= 256 and E %% 4 == 0");
    GT_REQUIRE(qpart.p && w_prev && w_cum && pm && Wc && Wd && v && erow && row_cnt && memory && w_out && step1 >= 1, GT_E_ARG,
               "gt_attn_step_fwd: NULL pointer");
    GT_REQUIRE(cum_out == nullptr || cum_in != nullptr, GT_E_ARG, "gt_attn_step_fwd: cum_in missing");
    auto ok4 = [](const void* p, int64_t ld) { return p == nullptr || (aligned16(p) && ld % 4 == 0); };
    GT_REQUIRE(ok4(memory, ld_mem) && ld_mem >= d.E && ok4(c0, ld0) && ok4(c1, ld1) && ok4(c2, ld2), GT_E_ALIGN,
               "gt_attn_step_fwd: memory and the context destinations must be 16-byte aligned with leading dimensions %% 4 == 0");
    const double ctx_rows = g_gt_prof_ctx_rows > 0.0 ? g_gt_prof_ctx_rows : (double)d.B * d.Ti;
    GtProfScope prof(GT_FAM_ATTN_E_FWD, 2.0 * d.B * d.Ti * (double)(d.F * 2 * d.Kc + d.A * d.F + 2 * d.A) + 2.0 * ctx_rows * d.E,
                     4.0 * (d.B * d.Ti * (d.A + 3.0) + ctx_rows * d.E + d.B * (3.0 * d.E + 4.0 * d.Ti)), (hipStream_t)stream);
    gt_launch_attn_step_fwd_mfma(d, qpart, w_prev, w_cum, pm, Wc, Wd, v, lengths, q_out, wpack, u_out, erow, row_cnt, step1,
                                 timeout_flag, memory, (long)ld_mem, w_out, cum_in, cum_out, c0, (long)ld0, c1, (long)ld1, c2,
                                 (long)ld2, (hipStream_t)stream);
    GT_LAUNCH_CHECK();
    return 0;
}

// gt_attn_step2_fwd: see attention_mfma.hip (attn_step2_fwd_mfma).  xbuf: B * ceil(Ti / 32) rows of gt_attn_step2_xstride(d)
// floats; row_cnt: B words zeroed once per loop call; step1 = 1, 2, ... within that call.
extern "C" int32_t gt_attn_step2_supported(GtAttnDims d) {
    return (use_mfma(d) && d.Ti <= 256 && d.E % 4 == 0 && d.E <= 1024) ? 1 : 0;
}
extern "C" int64_t gt_attn_step2_xstride(GtAttnDims d) { return d.E + 64; }

extern "C" int gt_attn_step2_fwd(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum, const float* pm,
                                 const float* Wc, const float* Wd, const float* v, const int32_t* lengths, float* q_out,
                                 const float* wpack, float* u_out, float* xbuf, uint32_t* row_cnt, uint32_t step1,
                                 const float* memory, int64_t ld_mem, float* w_out, const float* cum_in, float* cum_out,
                                 float* c0, int64_t ld0, float* c1, int64_t ld1, float* c2, int64_t ld2, void* stream) {
    return gt_attn_step2_fwd_tail(d, qpart, w_prev, w_cum, pm, Wc, Wd, v, lengths, q_out, wpack, u_out, xbuf, row_cnt, step1, memory,
                                  ld_mem, w_out, cum_in, cum_out, c0, ld0, c1, ld1, c2, ld2, nullptr, stream);
}

extern "C" int gt_attn_step2_fwd_tail(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum, const float* pm,
                                      const float* Wc, const float* Wd, const float* v, const int32_t* lengths, float* q_out,
                                      const float* wpack, float* u_out, float* xbuf, uint32_t* row_cnt, uint32_t step1,
                                      const float* memory, int64_t ld_mem, float* w_out, const float* cum_in, float* cum_out,
                                      float* c0, int64_t ld0, float* c1, int64_t ld1, float* c2, int64_t ld2, const GtRecTail* tail,
                                      void* stream) {
    int rc = check_dims(d, "gt_attn_step2_fwd");
    if (rc) return rc;
    if ((rc = gt_rec_tail_check(tail, "gt_attn_step2_fwd_tail")) != 0) return rc;
    GT_REQUIRE(gt_attn_step2_supported(d), GT_E_LIMIT, "gt_attn_step2_fwd: needs attention dims (128, 32, 31), Ti <= 256, E %% 4 == 0 and E <= 1024");
    GT_REQUIRE(qpart.p && w_prev && w_cum && pm && Wc && Wd && v && xbuf && row_cnt && memory && w_out && wpack && step1 >= 1, GT_E_ARG,
               "gt_attn_step2_fwd: NULL pointer (wpack, the gt_attn_pack_weights image, is required)");
    GT_REQUIRE(cum_out == nullptr || cum_in != nullptr, GT_E_ARG, "gt_attn_step2_fwd: cum_in missing");
    auto ok4 = [](const void* p, int64_t ld) { return p == nullptr || (aligned16(p) && ld % 4 == 0); };
    GT_REQUIRE(ok4(memory, ld_mem) && ld_mem >= d.E && ok4(c0, ld0) && ok4(c1, ld1) && ok4(c2, ld2) && aligned16(xbuf), GT_E_ALIGN,
               "gt_attn_step2_fwd: memory, the context destinations and xbuf must be 16-byte aligned with leading dimensions %% 4 == 0");
    const double ctx_rows = g_gt_prof_ctx_rows > 0.0 ? g_gt_prof_ctx_rows : (double)d.B * d.Ti;
    // the tail's product is charged to this launch: flops, packed weights + activations read once per column group + slabs
    const bool has_tail = tail != nullptr && tail->work != nullptr;
    const double tk = has_tail ? 16.0 * (tail->kb1 - tail->kb0) : 0.0, tn = has_tail ? 64.0 * tail->groups : 0.0;
    const double trows = has_tail ? 16.0 * tail->mtiles : 0.0, tsl = has_tail ? (double)cdiv(tail->kb1 - tail->kb0, tail->kbs) : 0.0;
    GtProfScope prof(GT_FAM_ATTN_E_FWD, 2.0 * d.B * d.Ti * (double)(d.F * 2 * d.Kc + d.A * d.F + 2 * d.A) + 2.0 * ctx_rows * d.E +
                                            2.0 * trows * tn * tk,
                     4.0 * (d.B * d.Ti * (d.A + 3.0) + ctx_rows * d.E + d.B * (3.0 * d.E + 4.0 * d.Ti) + tn * tk + trows * tk +
                            trows * tn * tsl), (hipStream_t)stream);
    gt_launch_attn_step2_fwd_mfma(d, qpart, w_prev, w_cum, pm, Wc, Wd, v, lengths, q_out, wpack, u_out, xbuf,
                                  (long)gt_attn_step2_xstride(d), row_cnt, step1, memory, (long)ld_mem, w_out, cum_in, cum_out,
                                  c0, (long)ld0, c1, (long)ld1, c2, (long)ld2, tail, (hipStream_t)stream);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_attn_context_fwd(GtAttnDims d, const float* energies, const float* memory, int64_t ld_mem,
                                   const int32_t* lengths, float* w_out, const float* cum_in, float* cum_out, float* c0, int64_t ld0,
                                   float* c1, int64_t ld1, float* c2, int64_t ld2, void* stream) {
    int rc = check_dims(d, "gt_attn_context_fwd");
    if (rc) return rc;
    GT_REQUIRE(energies && memory && w_out && ld_mem >= d.E, GT_E_ARG, "gt_attn_context_fwd: bad argument");
    GT_REQUIRE(cum_out == nullptr || cum_in != nullptr, GT_E_ARG, "gt_attn_context_fwd: cum_in missing");
    dim3 grid(cdiv(d.E, 256), d.B);
    // the kernels skip padded memory rows: charge the rows they really read when the caller has told us how many
    const double ctx_rows = g_gt_prof_ctx_rows > 0.0 ? g_gt_prof_ctx_rows : (double)d.B * d.Ti;
    GtProfScope prof(GT_FAM_ATTN_C_FWD, 2.0 * ctx_rows * (double)d.E, 4.0 * (ctx_rows * d.E + d.B * (3.0 * d.E + 4.0 * d.Ti)),
                     (hipStream_t)stream);
    const bool vec = (d.E % 4 == 0) && (ld_mem % 4 == 0) && aligned16(memory);
    if (vec)
        hipLaunchKernelGGL((attn_context_fwd_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, d, energies, memory,
                           (long)ld_mem, lengths, w_out, cum_in, cum_out, c0, (long)ld0, c1, (long)ld1, c2, (long)ld2);
    else
        hipLaunchKernelGGL((attn_context_fwd_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, d, energies, memory,
                           (long)ld_mem, lengths, w_out, cum_in, cum_out, c0, (long)ld0, c1, (long)ld1, c2, (long)ld2);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_attn_context_bwd(GtAttnDims d, GtSlabs dctx_a, GtSlabs dctx_b, const float* dctx_direct,
                                     int64_t ld_direct, const float* memory, int64_t ld_mem, const int32_t* lengths,
                                     float* dwcat_in, float* dcum_acc, const float* dw_extern, float* dctx_out, int64_t ld_dctx_out,
                                     float* dq_zero, int64_t ld_dq, float* dw_out, void* stream) {
    int rc = check_dims(d, "gt_attn_context_bwd");
    if (rc) return rc;
    GT_REQUIRE(memory && dw_out && ld_mem >= d.E, GT_E_ARG, "gt_attn_context_bwd: bad argument");
    GT_REQUIRE((size_t)d.E * 4 <= 64 * 1024, GT_E_LIMIT, "gt_attn_context_bwd: E too large for LDS");
    dim3 grid(d.B, cdiv(d.Ti, CT));
    const double ctx_rows = g_gt_prof_ctx_rows > 0.0 ? g_gt_prof_ctx_rows : (double)d.B * d.Ti;
    GtProfScope prof(GT_FAM_ATTN_C_BWD, 2.0 * ctx_rows * (double)d.E,
                     4.0 * (ctx_rows * d.E + d.B * ((1.0 + dctx_a.n + dctx_b.n) * d.E + 6.0 * d.Ti)), (hipStream_t)stream);
    const bool vec = (d.E % 4 == 0) && (ld_mem % 4 == 0) && aligned16(memory);
    auto sl4 = [](const GtSlabs& sl) { return sl.p == nullptr || (aligned16(sl.p) && sl.ld % 4 == 0 && sl.slab_stride % 4 == 0); };
    const bool vec_slabs = sl4(dctx_a) && sl4(dctx_b) && (dctx_direct == nullptr || (aligned16(dctx_direct) && ld_direct % 4 == 0)) &&
                           (dctx_out == nullptr || (aligned16(dctx_out) && ld_dctx_out % 4 == 0));
    if (vec && d.E == 1024 && vec_slabs)
        hipLaunchKernelGGL((attn_context_bwd_kernel<true, 4>), grid, dim3(256), (size_t)d.E * 4, (hipStream_t)stream, d,
                           dctx_a, dctx_b, dctx_direct, (long)ld_direct, memory, (long)ld_mem, lengths, dwcat_in, dcum_acc,
                           dw_extern, dctx_out, (long)ld_dctx_out, dq_zero, (long)ld_dq, dw_out);
    else if (vec)
        hipLaunchKernelGGL((attn_context_bwd_kernel<true>), grid, dim3(256), (size_t)d.E * 4, (hipStream_t)stream, d,
                           dctx_a, dctx_b, dctx_direct, (long)ld_direct, memory, (long)ld_mem, lengths, dwcat_in, dcum_acc,
                           dw_extern, dctx_out, (long)ld_dctx_out, dq_zero, (long)ld_dq, dw_out);
    else
        hipLaunchKernelGGL((attn_context_bwd_kernel<false>), grid, dim3(256), (size_t)d.E * 4, (hipStream_t)stream, d,
                           dctx_a, dctx_b, dctx_direct, (long)ld_direct, memory, (long)ld_mem, lengths, dwcat_in, dcum_acc,
                           dw_extern, dctx_out, (long)ld_dctx_out, dq_zero, (long)ld_dq, dw_out);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_attn_energies_bwd(GtAttnDims d, const float* dw, const float* w, const float* q,
                                      const float* w_prev, const float* w_cum, const float* pm, const float* Wc,
                                      const float* Wd, const float* v, const int32_t* lengths, float* dq, int64_t ld_dq,
                                      float* dpm, float* dwcat_out, float* acc, void* stream) {
    return gt_attn_energies_bwd_w(d, dw, w, q, w_prev, w_cum, pm, Wc, Wd, v, lengths, dq, ld_dq, dpm, dwcat_out, acc, nullptr, stream);
}

extern "C" int gt_attn_energies_bwd_w(GtAttnDims d, const float* dw, const float* w, const float* q,
                                      const float* w_prev, const float* w_cum, const float* pm, const float* Wc,
                                      const float* Wd, const float* v, const int32_t* lengths, float* dq, int64_t ld_dq,
                                      float* dpm, float* dwcat_out, float* acc, const float* wpack, void* stream) {
    int rc = check_dims(d, "gt_attn_energies_bwd");
    if (rc) return rc;
    GT_REQUIRE(dw && w && q && w_prev && w_cum && pm && Wc && Wd && v && dq && dpm && dwcat_out && acc, GT_E_ARG,
               "gt_attn_energies_bwd: NULL pointer");
    dim3 grid(cdiv(d.Ti, CT), d.B);
    GtProfScope prof(GT_FAM_ATTN_E_BWD, 2.0 * d.B * d.Ti * (double)(3 * d.F * 2 * d.Kc + 3 * d.A * d.F + 4 * d.A),
                     4.0 * d.B * d.Ti * (3.0 * d.A + 6.0), (hipStream_t)stream);
#define GT_EB(NIT, CA, CF, CK)                                                                                   \
    hipLaunchKernelGGL((attn_energies_bwd_kernel<NIT, CA, CF, CK>), grid, dim3(256), 0, (hipStream_t)stream, d,  \
                       dw, w, q, w_prev, w_cum, pm, Wc, Wd, v, lengths, dq, (long)ld_dq, dpm, dwcat_out, acc)
    if (d.A == 128 && d.F == 32 && d.Kc == 31 && !gt_env_item("GANTRON_DEBUG", "attn_valu", 0)) {
        gt_launch_attn_energies_bwd_mfma(d, dw, w, q, w_prev, w_cum, pm, Wc, Wd, v, lengths, dq, (long)ld_dq, dpm, dwcat_out, acc,
                                         wpack, (hipStream_t)stream);
    } else if (d.A == 128 && d.F == 32 && d.Kc == 31) {
        GT_EB(16, 128, 32, 31);
    } else {
        switch (CT / (256 / d.A)) {
            case 1: GT_EB(1, 0, 0, 0); break;
            case 2: GT_EB(2, 0, 0, 0); break;
            case 4: GT_EB(4, 0, 0, 0); break;
            case 8: GT_EB(8, 0, 0, 0); break;
            default: GT_EB(16, 0, 0, 0); break;
        }
    }
#undef GT_EB
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_attn_energies_bwd_rec(GtAttnDims d, const float* dw, const float* w, const float* u, const float* Wc,
                                        const float* Wd, const float* v, const int32_t* lengths, float* dq, int64_t ld_dq,
                                        float* dwcat_out, float* de_out, const float* wpack, void* stream) {
    int rc = check_dims(d, "gt_attn_energies_bwd_rec");
    if (rc) return rc;
    GT_REQUIRE(use_mfma(d), GT_E_ARG, "gt_attn_energies_bwd_rec: needs the (128, 32, 31) dimensions");
    GT_REQUIRE(dw && w && u && Wc && Wd && v && dq && dwcat_out && de_out, GT_E_ARG, "gt_attn_energies_bwd_rec: NULL pointer");
    GtProfScope prof(GT_FAM_ATTN_E_BWD, 2.0 * d.B * d.Ti * (double)(d.F * 2 * d.Kc + d.A * d.F + 4 * d.A),
                     4.0 * d.B * d.Ti * (d.A + 6.0), (hipStream_t)stream);
    gt_launch_attn_energies_bwd_rec_mfma(d, dw, w, u, Wc, Wd, v, lengths, dq, (long)ld_dq, dwcat_out, de_out, wpack,
                                         (hipStream_t)stream);
    GT_LAUNCH_CHECK();
    return 0;
}

// the one-launch backward attention step (gt_attn_bwd_wterms + gt_attn_bwd_step + gt_attn_bwd_fixup) exists for these shapes
extern "C" int32_t gt_attn_bwd_step_supported(GtAttnDims d) {
    return (use_mfma(d) && d.Ti <= 512 && (size_t)d.E * 4 <= 32 * 1024 && d.E % 4 == 0) ? 1 : 0;
}

extern "C" int gt_attn_bwd_wterms(GtAttnDims d, int32_t nsteps, const float* w, int64_t w_step_stride, const float* u,
                                  const float* Wc, const float* Wd, const float* v, const int32_t* lengths, float* dq2,
                                  int64_t dq2_step_stride, int64_t ld_dq2, float* dwcat2, float* p_hist, const float* wpack,
                                  void* stream) {
    int rc = check_dims(d, "gt_attn_bwd_wterms");
    if (rc) return rc;
    GT_REQUIRE(use_mfma(d), GT_E_ARG, "gt_attn_bwd_wterms: needs the (128, 32, 31) dimensions");
    GT_REQUIRE(nsteps > 0 && nsteps <= 65535 && w && u && Wc && Wd && v && dq2 && dwcat2 && p_hist && ld_dq2 >= d.A && aligned16(p_hist),
               GT_E_ARG, "gt_attn_bwd_wterms: bad argument");
    gt_launch_attn_bwd_wterms_mfma(d, nsteps, w, (long)w_step_stride, u, Wc, Wd, v, lengths, dq2, (long)dq2_step_stride,
                                   (long)ld_dq2, dwcat2, p_hist, wpack, (hipStream_t)stream);
    GT_LAUNCH_CHECK();
    return 0;
}

static int attn_bwd_step_impl(GtAttnDims d, GtSlabs dctx_a, GtSlabs dctx_b, const float* dctx_direct, int64_t ld_direct,
                              const float* memory, int64_t ld_mem, const int32_t* lengths, float* dwcat_in, float* dcum_acc,
                              const float* dw_extern, float* dctx_out, int64_t ld_dctx_out, const float* w, const float* u,
                              const float* v, const float* p, const float* sum_prev, const float* dwcat2_prev,
                              float* sum_out, float* dq, int64_t ld_dq, float* dwcat_out, float* e1_out,
                              const GtCellBwdTail* cell, void* stream) {
    int rc = check_dims(d, "gt_attn_bwd_step");
    if (rc) return rc;
    GT_REQUIRE(use_mfma(d), GT_E_ARG, "gt_attn_bwd_step: needs the (128, 32, 31) dimensions");
    GT_REQUIRE(memory && ld_mem >= d.E && w && u && v && p && aligned16(p) && aligned16(u) && aligned16(v) && sum_out && dq &&
                   dwcat_out && e1_out && ld_dq >= d.A && cdiv(d.Ti, CT) <= 16,
               GT_E_ARG, "gt_attn_bwd_step: bad argument (Ti <= 512)");
    GT_REQUIRE((sum_prev == nullptr) == (dwcat2_prev == nullptr), GT_E_ARG, "gt_attn_bwd_step: sum_prev and dwcat2_prev go together");
    GT_REQUIRE((size_t)d.E * 4 <= 32 * 1024, GT_E_LIMIT, "gt_attn_bwd_step: E too large for LDS");
    const double ctx_rows = g_gt_prof_ctx_rows > 0.0 ? g_gt_prof_ctx_rows : (double)d.B * d.Ti;
    // (with the cell inside: + dq . Wq of every row and what gt_cell_bwd_query_c reads and writes; algorithmic bytes: Wq ONCE --
    // the 512 KB per row that the tails actually pull come out of L2)
    const double cell_flop = cell != nullptr ? 2.0 * d.B * (double)cell->H * d.A : 0.0;
    const double cell_bytes = cell != nullptr ? 4.0 * ((double)d.A * cell->H + d.B * (double)cell->H * (cell->dh_a.n + cell->dh_c.n + 1 + 4 + 2 + 4 + 1)) : 0.0;
    GtProfScope prof(GT_FAM_ATTN_C_BWD, 2.0 * ctx_rows * (double)d.E + 2.0 * d.B * d.Ti * (double)(d.A + 2 * d.Kc) + cell_flop,
                     4.0 * (ctx_rows * d.E + d.B * ((1.0 + dctx_a.n + dctx_b.n) * d.E + 6.0 * d.Ti) + d.B * d.Ti * (d.A + 64 + 6.0)) + cell_bytes,
                     (hipStream_t)stream);
    const bool vec = (d.E % 4 == 0) && (ld_mem % 4 == 0) && aligned16(memory);
    auto sl4 = [](const GtSlabs& sl) { return sl.p == nullptr || (aligned16(sl.p) && sl.ld % 4 == 0 && sl.slab_stride % 4 == 0); };
    const bool vec_slabs = sl4(dctx_a) && sl4(dctx_b) && (dctx_direct == nullptr || (aligned16(dctx_direct) && ld_direct % 4 == 0)) &&
                           (dctx_out == nullptr || (aligned16(dctx_out) && ld_dctx_out % 4 == 0));
    if (cell != nullptr) {
        auto sl16 = [](const GtSlabs& sl) { return sl.p == nullptr || (sl.n >= 1 && aligned16(sl.p) && sl.ld % 4 == 0 && sl.slab_stride % 4 == 0); };
        GT_REQUIRE(cell->dq2 && cell->Wq && cell->gates && cell->c_new && cell->c_prev && cell->dgates && cell->dc_prev &&
                       cell->row_cnt && cell->launch >= 1 && cell->H > 0 && cell->H % 4 == 0 && cell->ldwq >= cell->H &&
                       cell->p_drop >= 0.f && cell->p_drop < 1.f,
                   GT_E_ARG, "gt_attn_bwd_step_cell: bad cell argument (H %% 4 == 0, launch >= 1, row_cnt ...)");
        GT_REQUIRE(sl16(cell->dh_a) && sl16(cell->dh_c) && aligned16(cell->Wq) && cell->ldwq % 4 == 0 && aligned16(cell->gates) &&
                       aligned16(cell->c_new) && aligned16(cell->c_prev) && aligned16(cell->dgates) && aligned16(cell->dc_prev) &&
                       (cell->dc_next == nullptr || aligned16(cell->dc_next)) && (cell->keep == nullptr || (((uintptr_t)cell->keep) & 3) == 0),
                   GT_E_ALIGN, "gt_attn_bwd_step_cell: cell pointers must be 16-byte aligned with leading dimensions %% 4 == 0");
        GT_REQUIRE((uint64_t)cell->launch * 16u < 0xffffffffull, GT_E_LIMIT,
                   "gt_attn_bwd_step_cell: launch number too large for the row counters");
    }
    gt_launch_attn_bwd_step_mfma(d, dctx_a, dctx_b, dctx_direct, (long)ld_direct, memory, (long)ld_mem, lengths, dwcat_in,
                                 dcum_acc, dw_extern, dctx_out, (long)ld_dctx_out, w, u, v, p, sum_prev, dwcat2_prev,
                                 sum_out, dq, (long)ld_dq, dwcat_out, e1_out, vec, vec_slabs, (hipStream_t)stream, cell);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_attn_bwd_step(GtAttnDims d, GtSlabs dctx_a, GtSlabs dctx_b, const float* dctx_direct, int64_t ld_direct,
                                const float* memory, int64_t ld_mem, const int32_t* lengths, float* dwcat_in, float* dcum_acc,
                                const float* dw_extern, float* dctx_out, int64_t ld_dctx_out, const float* w, const float* u,
                                const float* v, const float* p, const float* sum_prev, const float* dwcat2_prev,
                                float* sum_out, float* dq, int64_t ld_dq, float* dwcat_out, float* e1_out, void* stream) {
    return attn_bwd_step_impl(d, dctx_a, dctx_b, dctx_direct, ld_direct, memory, ld_mem, lengths, dwcat_in, dcum_acc, dw_extern, dctx_out,
                              ld_dctx_out, w, u, v, p, sum_prev, dwcat2_prev, sum_out, dq, ld_dq, dwcat_out, e1_out, nullptr, stream);
}

extern "C" int gt_attn_bwd_step_cell(GtAttnDims d, GtSlabs dctx_a, GtSlabs dctx_b, const float* dctx_direct, int64_t ld_direct,
                                     const float* memory, int64_t ld_mem, const int32_t* lengths, float* dwcat_in, float* dcum_acc,
                                     const float* dw_extern, float* dctx_out, int64_t ld_dctx_out, const float* w, const float* u,
                                     const float* v, const float* p, const float* sum_prev, const float* dwcat2_prev,
                                     float* sum_out, float* dq, int64_t ld_dq, float* dwcat_out, float* e1_out,
                                     const GtCellBwdTail* cell, void* stream) {
    GT_REQUIRE(cell != nullptr, GT_E_ARG, "gt_attn_bwd_step_cell: NULL cell");
    return attn_bwd_step_impl(d, dctx_a, dctx_b, dctx_direct, ld_direct, memory, ld_mem, lengths, dwcat_in, dcum_acc, dw_extern, dctx_out,
                              ld_dctx_out, w, u, v, p, sum_prev, dwcat2_prev, sum_out, dq, ld_dq, dwcat_out, e1_out, cell, stream);
}

extern "C" int gt_attn_bwd_fixup(GtAttnDims d, int32_t nsteps, const float* sum, const float* w, int64_t w_step_stride,
                                 float* de, float* dq, int64_t dq_step_stride, int64_t ld_dq, const float* dq2, void* stream) {
    int rc = check_dims(d, "gt_attn_bwd_fixup");
    if (rc) return rc;
    GT_REQUIRE(nsteps > 0 && nsteps <= 65535 && sum && w && de && dq && dq2 && d.A == 128, GT_E_ARG, "gt_attn_bwd_fixup: bad argument");
    gt_launch_attn_bwd_fixup(d, nsteps, cdiv(d.Ti, CT), sum, w, (long)w_step_stride, de, dq, (long)dq_step_stride, (long)ld_dq,
                             dq2, (hipStream_t)stream);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_attn_param_grads(GtAttnDims d, int32_t nsteps, const float* u, const float* de, const float* w_prev,
                                   const float* w_cum, int64_t step_stride, const float* Wc, const float* Wd,
                                   const float* v, const int32_t* lengths, float* dpm, float* acc, const float* wpack,
                                   void* stream) {
    int rc = check_dims(d, "gt_attn_param_grads");
    if (rc) return rc;
    GT_REQUIRE(use_mfma(d), GT_E_ARG, "gt_attn_param_grads: needs the (128, 32, 31) dimensions");
    GT_REQUIRE(nsteps > 0 && u && de && w_prev && w_cum && Wc && Wd && v && dpm && acc, GT_E_ARG, "gt_attn_param_grads: bad argument");
    gt_launch_attn_param_grads_mfma(d, nsteps, u, de, w_prev, w_cum, (long)step_stride, Wc, Wd, v, lengths, dpm, acc, wpack,
                                    (hipStream_t)stream);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_attn_acc_reduce(GtAttnDims d, const float* acc, float* dv, float* dWd, float* dWc, void* stream) {
    int rc = check_dims(d, "gt_attn_acc_reduce");
    if (rc) return rc;
    GT_REQUIRE(acc && dv && dWd && dWc, GT_E_ARG, "gt_attn_acc_reduce: NULL pointer");
    const int per = d.A + d.A * d.F + d.F * 2 * d.Kc;
    const int nwg = d.B * cdiv(d.Ti, CT);
    if (use_mfma(d)) {
        gt_launch_attn_acc_reduce_mfma(acc, nwg, dv, dWd, dWc, (hipStream_t)stream);
        GT_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(attn_acc_reduce_kernel, dim3(cdiv(per, 256)), dim3(256), 0, (hipStream_t)stream, acc, nwg, per,
                       d.A, d.A * d.F, dv, dWd, dWc);
    GT_LAUNCH_CHECK();
    return 0;
}
