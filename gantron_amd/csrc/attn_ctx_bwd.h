// Body of the attention context backward for one (batch row, 32-position chunk) workgroup, shared by
// attn_context_bwd_kernel (attention.hip) and the fused backward step attn_bwd_step_kernel (attention_mfma.hip).
#pragma once
#include "common.h"

namespace {

constexpr int ACB_CT = 32;       // encoder positions per workgroup

// dw[t'] = memory[b][t'] . dctx + carried alignment gradients, for the workgroup's positions t0 .. t0 + 31 (t0 < Tv).
// NE4 > 0 (E == 256 * NE4, VEC): the workgroup's 32 memory rows and the carried per-position values are requested
// at the very top and arrive while the upstream context gradient is being summed from its K-split slabs, so the
// kernel is two memory round trips deep instead of five.
// dctx: E floats of LDS, dots: 32 floats of LDS.  dwcat2 / s_prev (fused step only, else NULL / 0): the carried planes are
// dwcat_in[c] - s_prev * dwcat2[c] (see attn_bwd_step_kernel).  Returns the thread's dw (threads < 32 with t0 + tid < Ti;
// 0 for padded positions), after updating dcum_acc and zeroing the consumed dwcat_in planes.  Contains barriers: every
// thread of the 256-thread workgroup calls it.
template <bool VEC, int NE4>
__device__ __forceinline__ float attn_ctx_bwd_dw(const GtAttnDims& d, const GtSlabs& da, const GtSlabs& db,
                                                 const float* dctx_direct, long ld_direct, const float* memory, long ld_mem,
                                                 int Tv, float* dwcat_in, float* dcum_acc, const float* dw_extern,
                                                 float* dctx_out, long ld_dctx_out, const float* dwcat2, float s_prev,
                                                 float* dctx, float* dots, int b, int chunk) {
    constexpr int CT = ACB_CT;
    const int t0 = chunk * CT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* mb = memory + (long)b * d.Ti * ld_mem;
    constexpr int NR = NE4 > 0 ? NE4 : 1;
    float4 mr[8][NR];
    float pf_cum = 0.f, pf_w0 = 0.f, pf_w1 = 0.f, pf_ext = 0.f;
    const long pi = (long)b * d.Ti + min(t0 + (int)threadIdx.x, d.Ti - 1);
    const long plane = (long)d.B * d.Ti;
    if (NE4 > 0) {
        // wave w owns rows t0 + w + 4*r (r < 8); row indices are clamped so that no load is predicated
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float* row = mb + (long)min(t0 + wave + 4 * r, Tv - 1) * ld_mem + 4 * lane;
#pragma unroll
            for (int i = 0; i < NR; ++i) mr[r][i] = *reinterpret_cast<const float4*>(row + 256 * i);
        }
        if (threadIdx.x < CT) {
            if (dwcat_in != nullptr) {
                pf_cum = dcum_acc[pi]; pf_w0 = dwcat_in[pi]; pf_w1 = dwcat_in[plane + pi];
                if (dwcat2 != nullptr) { pf_w0 -= s_prev * dwcat2[pi]; pf_w1 -= s_prev * dwcat2[plane + pi]; }
            }
            if (dw_extern != nullptr) pf_ext = dw_extern[pi];
        }
    }
    if (NE4 == 1 * 4 && blockDim.x == 256) {
        // E == 1024: one float4 of the context gradient per thread, all slab loads of a thread issued together
        const int e = 4 * threadIdx.x;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        auto add_slabs = [&](const GtSlabs& sl) __attribute__((always_inline)) {
            if (sl.p == nullptr) return;
            const float* p = sl.p + (long)b * sl.ld + e;
#pragma unroll 4
            for (int s2 = 0; s2 < sl.n; ++s2) {
                const float4 t = *reinterpret_cast<const float4*>(p + (long)s2 * sl.slab_stride);
                g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
            }
        };
        add_slabs(da);
        add_slabs(db);
        if (dctx_direct != nullptr) {
            const float4 t = *reinterpret_cast<const float4*>(dctx_direct + (long)b * ld_direct + e);
            g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
        }
        *reinterpret_cast<float4*>(&dctx[e]) = g;
        if (chunk == 0 && dctx_out != nullptr) *reinterpret_cast<float4*>(dctx_out + (long)b * ld_dctx_out + e) = g;
    } else
    for (int e = threadIdx.x; e < d.E; e += blockDim.x) {
        float g = 0.f;
        if (da.p != nullptr) g += slab_sum(da.p, da.n, da.slab_stride, (long)b * da.ld + e);
        if (db.p != nullptr) g += slab_sum(db.p, db.n, db.slab_stride, (long)b * db.ld + e);
        if (dctx_direct != nullptr) g += dctx_direct[(long)b * ld_direct + e];
        dctx[e] = g;
        if (chunk == 0 && dctx_out != nullptr) dctx_out[(long)b * ld_dctx_out + e] = g;
    }
    __syncthreads();
    if (NE4 > 0) {
        const float4* dc4 = reinterpret_cast<const float4*>(dctx);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const float4 dc = dc4[lane + 64 * i];
#pragma unroll
            for (int r = 0; r < 8; ++r)
                acc[r] += mr[r][i].x * dc.x + mr[r][i].y * dc.y + mr[r][i].z * dc.z + mr[r][i].w * dc.w;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float v = wave_sum(acc[r]);
            if (lane == 0) dots[wave + 4 * r] = v;
        }
    } else if (VEC) {
        // wave w owns rows t0 + w + 4*i (i < 8); 4 rows in flight per pass
        const int e4n = d.E / 4;
        const float4* dc4 = reinterpret_cast<const float4*>(dctx);
        for (int pass = 0; pass < 2; ++pass) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            int trow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) trow[r] = min(t0 + wave + 4 * (4 * pass + r), Tv - 1);
            for (int e4 = lane; e4 < e4n; e4 += 64) {
                const float4 dc = dc4[e4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 m = *reinterpret_cast<const float4*>(mb + (long)trow[r] * ld_mem + 4 * e4);
                    acc[r] += m.x * dc.x + m.y * dc.y + m.z * dc.z + m.w * dc.w;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = wave_sum(acc[r]);
                if (lane == 0) dots[wave + 4 * (4 * pass + r)] = v;
            }
        }
    } else {
        for (int tl = wave; tl < CT; tl += 4) {
            const int t = min(t0 + tl, Tv - 1);
            const float* m = mb + (long)t * ld_mem;
            float acc = 0.f;
            for (int e = lane; e < d.E; e += 64) acc += m[e] * dctx[e];
            acc = wave_sum(acc);
            if (lane == 0) dots[tl] = acc;
        }
    }
    __syncthreads();
    float dw = 0.f;
    if (threadIdx.x < CT && t0 + threadIdx.x < d.Ti) {
        const long i = (long)b * d.Ti + t0 + threadIdx.x;
        float carried = 0.f;
        if (dwcat_in != nullptr) {
            float w0 = pf_w0, w1 = pf_w1, cum0 = pf_cum;
            if (NE4 == 0) {
                w0 = dwcat_in[i]; w1 = dwcat_in[plane + i]; cum0 = dcum_acc[i];
                if (dwcat2 != nullptr) { w0 -= s_prev * dwcat2[i]; w1 -= s_prev * dwcat2[plane + i]; }
            }
            const float cum = cum0 + w1;
            dcum_acc[i] = cum;
            carried = w0 + cum;
            dwcat_in[i] = 0.f;
            dwcat_in[plane + i] = 0.f;
        }
        if (dw_extern != nullptr) carried += (NE4 > 0 ? pf_ext : dw_extern[i]);
        dw = (t0 + (int)threadIdx.x < Tv) ? dots[threadIdx.x] + carried : 0.f;
    }
    return dw;
}

}  // namespace
