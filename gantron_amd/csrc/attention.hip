// Location-sensitive attention, one decoder step (reference model.py:27-31, 48-91), forward and
// backward, split so that every phase fills the chip:
//   energies  : grid (chunks of 32 encoder positions, batch)  -- VALU-bound (conv31 + 32->128 dense + tanh)
//   context   : grid (256-column chunks of E, batch)           -- streams `memory` (B*Ti*E floats / step)
// Both kernels keep their small weights (location conv/dense, v), the window of previous/cumulative
// alignments and the per-position intermediates in LDS; softmax reductions use wave64 shuffles.
// The backward kernels recompute the tanh arguments (the reference's autograd saves a
// (B,Ti,128) tensor per step: 3.9 GB over 800 steps) and accumulate the location-layer parameter
// gradients in per-workgroup slabs (no atomics) that gt_attn_acc_reduce folds after the loop.
#include "common.h"

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
    float q[MAXA];
    float v[MAXA];
    float wc[MAXF * 2 * MAXK];
    float wd[MAXA * LDW];
    float win[2][WINW + 1];
    float locf[CT * LDL];
    float buf[CT * LDS_A];      // fwd: v*tanh ; bwd: ds
    float dlocf[CT * LDL];
    float de[CT];
    float dvs[MAXA];
    float red[16];
};

__device__ __forceinline__ void load_weights(Shared& S, const GtAttnDims& d, const float* Wc, const float* Wd,
                                             const float* v) {
    for (int i = threadIdx.x; i < d.F * 2 * d.Kc; i += blockDim.x) S.wc[i] = Wc[i];
    for (int i = threadIdx.x; i < d.A * d.F; i += blockDim.x) {
        const int dd = i / d.F, f = i - dd * d.F;
        S.wd[dd * LDW + f] = Wd[i];
    }
    for (int i = threadIdx.x; i < d.A; i += blockDim.x) S.v[i] = v[i];
}

__device__ __forceinline__ void load_window(Shared& S, const GtAttnDims& d, const float* w_prev, const float* w_cum,
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
__device__ __forceinline__ void location_conv(Shared& S, const GtAttnDims& d) {
    for (int o = threadIdx.x; o < CT * d.F; o += blockDim.x) {
        const int tl = o / d.F, f = o - tl * d.F;
        float acc = 0.f;
        for (int c = 0; c < 2; ++c) {
            const float* wrow = &S.wc[(f * 2 + c) * d.Kc];
            const float* x = &S.win[c][tl];
            for (int k = 0; k < d.Kc; ++k) acc += wrow[k] * x[k];
        }
        S.locf[tl * LDL + f] = acc;
    }
}

// tanh argument for (tl, dd): q + Wd . locf[tl] + pm
__device__ __forceinline__ float dense_arg(const Shared& S, const GtAttnDims& d, int tl, int dd, float pmv) {
    float s = S.q[dd] + pmv;
    const float* wrow = &S.wd[dd * LDW];
    const float* l = &S.locf[tl * LDL];
    for (int f = 0; f < d.F; ++f) s += wrow[f] * l[f];
    return s;
}

__global__ __launch_bounds__(256) void attn_energies_fwd_kernel(GtAttnDims d, GtSlabs qpart, const float* w_prev,
                                                                const float* w_cum, const float* pm, const float* Wc,
                                                                const float* Wd, const float* v, const int* lengths,
                                                                float* q_out, float* energies) {
    __shared__ Shared S;
    const int b = blockIdx.y, t0 = blockIdx.x * CT;
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
    const int G = blockDim.x / d.A;
    const int dd = threadIdx.x % d.A, g = threadIdx.x / d.A;
    if (g < G) {
        for (int tl = g; tl < CT; tl += G) {
            const int t = t0 + tl;
            float val = 0.f;
            if (t < d.Ti) {
                const float s = dense_arg(S, d, tl, dd, pm[((long)b * d.Ti + t) * d.A + dd]);
                val = S.v[dd] * tanhf(s);
            }
            S.buf[tl * LDS_A + dd] = val;
        }
    }
    __syncthreads();
    if (threadIdx.x < CT) {
        const int t = t0 + threadIdx.x;
        if (t < d.Ti) {
            float e = 0.f;
            for (int i = 0; i < d.A; ++i) e += S.buf[threadIdx.x * LDS_A + i];
            if (lengths != nullptr && t >= lengths[b]) e = -INFINITY;
            energies[(long)b * d.Ti + t] = e;
        }
    }
}

__global__ __launch_bounds__(256) void attn_context_fwd_kernel(GtAttnDims d, const float* energies, const float* memory,
                                                               long ld_mem, float* w_out, const float* cum_in,
                                                               float* cum_out, float* c0, long ld0, float* c1, long ld1,
                                                               float* c2, long ld2) {
    __shared__ float w[MAX_TI];
    __shared__ float red[16];
    const int b = blockIdx.y;
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
    for (int t = threadIdx.x; t < d.Ti; t += blockDim.x) {
        const float wv = w[t] * inv;
        w[t] = wv;
        if (blockIdx.x == 0) {
            w_out[(long)b * d.Ti + t] = wv;
            if (cum_out != nullptr) cum_out[(long)b * d.Ti + t] = cum_in[(long)b * d.Ti + t] + wv;
        }
    }
    __syncthreads();
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col < d.E) {
        const float* m = memory + (long)b * d.Ti * ld_mem + col;
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
        int t = 0;
        for (; t + 3 < d.Ti; t += 4) {
            acc0 += w[t] * m[(long)t * ld_mem];
            acc1 += w[t + 1] * m[(long)(t + 1) * ld_mem];
            acc2 += w[t + 2] * m[(long)(t + 2) * ld_mem];
            acc3 += w[t + 3] * m[(long)(t + 3) * ld_mem];
        }
        for (; t < d.Ti; ++t) acc0 += w[t] * m[(long)t * ld_mem];
        const float c = (acc0 + acc1) + (acc2 + acc3);
        if (c0 != nullptr) c0[(long)b * ld0 + col] = c;
        if (c1 != nullptr) c1[(long)b * ld1 + col] = c;
        if (c2 != nullptr) c2[(long)b * ld2 + col] = c;
    }
}

// grid (chunks of CT positions, B).  dw[t'] = memory[b][t'] . dctx + carried alignment gradients.
__global__ __launch_bounds__(256) void attn_context_bwd_kernel(GtAttnDims d, GtSlabs da, GtSlabs db,
                                                               const float* dctx_direct, long ld_direct,
                                                               const float* memory, long ld_mem, float* dwcat_in,
                                                               float* dcum_acc, const float* dw_extern, float* dctx_out,
                                                               long ld_dctx_out, float* dq_zero, long ld_dq, float* dw_out) {
    extern __shared__ float dctx[];   // E floats
    const int b = blockIdx.y, t0 = blockIdx.x * CT;
    for (int e = threadIdx.x; e < d.E; e += blockDim.x) {
        float g = 0.f;
        if (da.p != nullptr) g += slab_sum(da.p, da.n, da.slab_stride, (long)b * da.ld + e);
        if (db.p != nullptr) g += slab_sum(db.p, db.n, db.slab_stride, (long)b * db.ld + e);
        if (dctx_direct != nullptr) g += dctx_direct[(long)b * ld_direct + e];
        dctx[e] = g;
        if (blockIdx.x == 0 && dctx_out != nullptr) dctx_out[(long)b * ld_dctx_out + e] = g;
    }
    if (blockIdx.x == 0 && dq_zero != nullptr)
        for (int i = threadIdx.x; i < d.A; i += blockDim.x) dq_zero[(long)b * ld_dq + i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int tl = wave; tl < CT; tl += nw) {
        const int t = t0 + tl;
        if (t >= d.Ti) break;
        const float* m = memory + ((long)b * d.Ti + t) * ld_mem;
        float acc = 0.f;
        for (int e = lane; e < d.E; e += 64) acc += m[e] * dctx[e];
        acc = wave_sum(acc);
        if (lane == 0) {
            const long i = (long)b * d.Ti + t;
            const long plane = (long)d.B * d.Ti;
            float carried = 0.f;
            if (dwcat_in != nullptr) {
                const float cum = dcum_acc[i] + dwcat_in[plane + i];
                dcum_acc[i] = cum;
                carried = dwcat_in[i] + cum;
                dwcat_in[i] = 0.f;
                dwcat_in[plane + i] = 0.f;
            }
            if (dw_extern != nullptr) carried += dw_extern[i];
            dw_out[i] = acc + carried;
        }
    }
}

// grid (chunks, B).  Energies backward for CT positions; see file header.
__global__ __launch_bounds__(256) void attn_energies_bwd_kernel(GtAttnDims d, const float* dw, const float* w,
                                                                const float* q, const float* w_prev, const float* w_cum,
                                                                const float* pm, const float* Wc, const float* Wd,
                                                                const float* v, float* dq, long ld_dq, float* dpm, float* dwcat_out,
                                                                float* acc) {
    __shared__ Shared S;
    const int b = blockIdx.y, t0 = blockIdx.x * CT;
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

    float* acc_wg = acc + ((long)b * gridDim.x + blockIdx.x) * (d.A + d.A * d.F + d.F * 2 * d.Kc);
    const int G = blockDim.x / d.A;
    const int dd = threadIdx.x % d.A, g = threadIdx.x / d.A;
    float dq_acc = 0.f, dv_acc = 0.f;
    if (g < G) {
        for (int tl = g; tl < CT; tl += G) {
            const int t = t0 + tl;
            float ds = 0.f;
            if (t < d.Ti) {
                const long pi = ((long)b * d.Ti + t) * d.A + dd;
                const float u = tanhf(dense_arg(S, d, tl, dd, pm[pi]));
                const float de = S.de[tl];
                ds = de * S.v[dd] * (1.f - u * u);
                dv_acc += de * u;
                dq_acc += ds;
                dpm[pi] += ds;
            }
            S.buf[tl * LDS_A + dd] = ds;
        }
        atomicAdd(&dq[(long)b * ld_dq + dd], dq_acc);
    }
    __syncthreads();
    // dv: fold the G groups through LDS
    if (threadIdx.x < d.A) S.dvs[threadIdx.x] = 0.f;
    __syncthreads();
    if (g < G) atomicAdd(&S.dvs[dd], dv_acc);
    __syncthreads();
    if (threadIdx.x < d.A) acc_wg[threadIdx.x] += S.dvs[threadIdx.x];

    // dlocf[tl][f] = sum_d ds[tl][d] * Wd[d][f]
    for (int o = threadIdx.x; o < CT * d.F; o += blockDim.x) {
        const int tl = o / d.F, f = o - tl * d.F;
        float a = 0.f;
        for (int i = 0; i < d.A; ++i) a += S.buf[tl * LDS_A + i] * S.wd[i * LDW + f];
        S.dlocf[tl * LDL + f] = a;
    }
    // dWd[d][f] += sum_tl ds[tl][d] * locf[tl][f]
    float* acc_wd = acc_wg + d.A;
    for (int o = threadIdx.x; o < d.A * d.F; o += blockDim.x) {
        const int i = o / d.F, f = o - i * d.F;
        float a = 0.f;
        for (int tl = 0; tl < CT; ++tl) a += S.buf[tl * LDS_A + i] * S.locf[tl * LDL + f];
        acc_wd[o] += a;
    }
    __syncthreads();
    // dWc[f][c][k] += sum_tl dlocf[tl][f] * win[c][tl+k]
    float* acc_wc = acc_wg + d.A + d.A * d.F;
    for (int o = threadIdx.x; o < d.F * 2 * d.Kc; o += blockDim.x) {
        const int f = o / (2 * d.Kc), r = o - f * 2 * d.Kc, c = r / d.Kc, k = r - c * d.Kc;
        float a = 0.f;
        for (int tl = 0; tl < CT; ++tl) a += S.dlocf[tl * LDL + f] * S.win[c][tl + k];
        acc_wc[o] += a;
    }
    // d[w_prev; w_cum][p] = sum_{f,k} dlocf[p-k][f] * Wc[f][c][k]   (transposed conv over the window)
    const int pad = (d.Kc - 1) / 2, ww = CT + d.Kc - 1;
    for (int o = threadIdx.x; o < 2 * ww; o += blockDim.x) {
        const int c = o / ww, p = o - c * ww;
        const int t = t0 + p - pad;
        if (t < 0 || t >= d.Ti) continue;
        float a = 0.f;
        for (int k = 0; k < d.Kc; ++k) {
            const int tl = p - k;
            if (tl < 0 || tl >= CT) continue;
            for (int f = 0; f < d.F; ++f) a += S.dlocf[tl * LDL + f] * S.wc[(f * 2 + c) * d.Kc + k];
        }
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
    GT_REQUIRE(d.A > 0 && d.A <= MAXA && 256 % d.A == 0, GT_E_LIMIT, "%s: attention_dim %d must divide 256 and be <= %d", who, d.A, MAXA);
    GT_REQUIRE(d.F > 0 && d.F <= MAXF && d.Kc > 0 && d.Kc <= MAXK && (d.Kc & 1), GT_E_LIMIT,
               "%s: location filters %d (<=%d) / kernel %d (odd, <=%d)", who, d.F, MAXF, d.Kc, MAXK);
    return 0;
}

}  // namespace

extern "C" int32_t gt_attn_chunks(int32_t Ti) { return cdiv(Ti, CT); }

extern "C" int64_t gt_attn_acc_elems(GtAttnDims d) {
    return (int64_t)d.B * cdiv(d.Ti, CT) * (d.A + d.A * d.F + d.F * 2 * d.Kc);
}

extern "C" int gt_attn_energies_fwd(GtAttnDims d, GtSlabs qpart, const float* w_prev, const float* w_cum,
                                    const float* pm, const float* Wc, const float* Wd, const float* v,
                                    const int32_t* lengths, float* q_out, float* energies, void* stream) {
    int rc = check_dims(d, "gt_attn_energies_fwd");
    if (rc) return rc;
    GT_REQUIRE(qpart.p && w_prev && w_cum && pm && Wc && Wd && v && energies, GT_E_ARG, "gt_attn_energies_fwd: NULL pointer");
    dim3 grid(cdiv(d.Ti, CT), d.B);
    GtProfScope prof(GT_FAM_ATTN_E_FWD, 2.0 * d.B * d.Ti * (double)(d.F * 2 * d.Kc + d.A * d.F + 2 * d.A),
                     4.0 * d.B * d.Ti * (d.A + 3.0), (hipStream_t)stream);
    hipLaunchKernelGGL(attn_energies_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, d, qpart, w_prev, w_cum, pm,
                       Wc, Wd, v, lengths, q_out, energies);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_attn_context_fwd(GtAttnDims d, const float* energies, const float* memory, int64_t ld_mem,
                                   float* w_out, const float* cum_in, float* cum_out, float* c0, int64_t ld0,
                                   float* c1, int64_t ld1, float* c2, int64_t ld2, void* stream) {
    int rc = check_dims(d, "gt_attn_context_fwd");
    if (rc) return rc;
    GT_REQUIRE(energies && memory && w_out && ld_mem >= d.E, GT_E_ARG, "gt_attn_context_fwd: bad argument");
    GT_REQUIRE(cum_out == nullptr || cum_in != nullptr, GT_E_ARG, "gt_attn_context_fwd: cum_in missing");
    dim3 grid(cdiv(d.E, 256), d.B);
    GtProfScope prof(GT_FAM_ATTN_C_FWD, 2.0 * d.B * d.Ti * (double)d.E, 4.0 * d.B * ((double)d.Ti * d.E + 3.0 * d.E + 4.0 * d.Ti),
                     (hipStream_t)stream);
    hipLaunchKernelGGL(attn_context_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, d, energies, memory,
                       (long)ld_mem, w_out, cum_in, cum_out, c0, (long)ld0, c1, (long)ld1, c2, (long)ld2);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_attn_context_bwd(GtAttnDims d, GtSlabs dctx_a, GtSlabs dctx_b, const float* dctx_direct,
                                     int64_t ld_direct, const float* memory, int64_t ld_mem, float* dwcat_in,
                                     float* dcum_acc, const float* dw_extern, float* dctx_out, int64_t ld_dctx_out,
                                     float* dq_zero, int64_t ld_dq, float* dw_out, void* stream) {
    int rc = check_dims(d, "gt_attn_context_bwd");
    if (rc) return rc;
    GT_REQUIRE(memory && dw_out && ld_mem >= d.E, GT_E_ARG, "gt_attn_context_bwd: bad argument");
    GT_REQUIRE((size_t)d.E * 4 <= 64 * 1024, GT_E_LIMIT, "gt_attn_context_bwd: E too large for LDS");
    dim3 grid(cdiv(d.Ti, CT), d.B);
    GtProfScope prof(GT_FAM_ATTN_C_BWD, 2.0 * d.B * d.Ti * (double)d.E, 4.0 * d.B * ((double)d.Ti * d.E + 4.0 * d.E + 6.0 * d.Ti),
                     (hipStream_t)stream);
    hipLaunchKernelGGL(attn_context_bwd_kernel, grid, dim3(256), (size_t)d.E * 4, (hipStream_t)stream, d, dctx_a,
                       dctx_b, dctx_direct, (long)ld_direct, memory, (long)ld_mem, dwcat_in, dcum_acc, dw_extern,
                       dctx_out, (long)ld_dctx_out, dq_zero, (long)ld_dq, dw_out);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_attn_energies_bwd(GtAttnDims d, const float* dw, const float* w, const float* q,
                                      const float* w_prev, const float* w_cum, const float* pm, const float* Wc,
                                      const float* Wd, const float* v, float* dq, int64_t ld_dq, float* dpm,
                                      float* dwcat_out, float* acc, void* stream) {
    int rc = check_dims(d, "gt_attn_energies_bwd");
    if (rc) return rc;
    GT_REQUIRE(dw && w && q && w_prev && w_cum && pm && Wc && Wd && v && dq && dpm && dwcat_out && acc, GT_E_ARG,
               "gt_attn_energies_bwd: NULL pointer");
    dim3 grid(cdiv(d.Ti, CT), d.B);
    GtProfScope prof(GT_FAM_ATTN_E_BWD, 2.0 * d.B * d.Ti * (double)(3 * d.F * 2 * d.Kc + 3 * d.A * d.F + 4 * d.A),
                     4.0 * d.B * d.Ti * (3.0 * d.A + 6.0), (hipStream_t)stream);
    hipLaunchKernelGGL(attn_energies_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, d, dw, w, q, w_prev, w_cum,
                       pm, Wc, Wd, v, dq, (long)ld_dq, dpm, dwcat_out, acc);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int gt_attn_acc_reduce(GtAttnDims d, const float* acc, float* dv, float* dWd, float* dWc, void* stream) {
    int rc = check_dims(d, "gt_attn_acc_reduce");
    if (rc) return rc;
    GT_REQUIRE(acc && dv && dWd && dWc, GT_E_ARG, "gt_attn_acc_reduce: NULL pointer");
    const int per = d.A + d.A * d.F + d.F * 2 * d.Kc;
    const int nwg = d.B * cdiv(d.Ti, CT);
    hipLaunchKernelGGL(attn_acc_reduce_kernel, dim3(cdiv(per, 256)), dim3(256), 0, (hipStream_t)stream, acc, nwg, per,
                       d.A, d.A * d.F, dv, dWd, dWc);
    GT_LAUNCH_CHECK();
    return 0;
}
