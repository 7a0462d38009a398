// Persistent, weight-resident LSTM recurrence (gt_plstm_fwd): ONE launch walks a range of time steps of
//   G_s = pre_s + h_{s-1} . W_hh^T,   (i, f, g, o) = act(G_s),   c_s = f c_{s-1} + i g,   h_s = drop(o tanh c_s)
// for `ndir` independent problems (the two directions of the encoder BiLSTM, reference model.py:202-210; the decoder LSTM's
// h_dec recurrence of a block of steps, model.py:406-409, whose input product is hoisted into a block GEMM).
//
// Why: as per-step launches these recurrences re-stream W_hh every step (1 MB x 2 directions x 187 steps for the encoder,
// 16 MB x 870 for the decoder LSTM) and pay a launch, a K-split hand-off and a fill / drain of all 256 CUs per step
// (~20 us for 0.03 - 0.5 GFLOP).  Here a problem is confined to H / 16 workgroups (16 for the encoder, 64 for the decoder
// LSTM), one per CU:
//   * workgroup w owns 16 hidden units = 64 gate columns; its 4 waves split K = H four ways and keep their W_hh block in
//     REGISTERS for the whole launch (64 or 256 per lane: one wave per SIMD owns the whole 512-entry file), as B operands
//     of v_mfma_f32_16x16x4_f32;
//   * the A operand h_{s-1} (64 rows x H) is read straight into registers, float4 along k, from where the owners
//     published it (write-through stores, the exchange form of MI355X_MICROARCH.md: every storing wave drains, one
//     agent-scope add per workgroup on a monotonic counter; readers poll it with one lane, then sc1 loads);
//   * the four K-slices fold through LDS, wave r finishes row tile r: the C layout of the MFMA leaves ALL FOUR gates of
//     (unit = lane & 15, rows 4 (lane >> 4) + 0..3) in one lane, so the cell runs in registers and c never leaves them.
// Residency: every workgroup of a problem waits for its peers, so all of them must be resident together.  The grid is at
// most 64 workgroups on a 256-CU device and no other kernel ever waits on this one while holding a CU, so each workgroup
// is dispatched eventually (the other kernels of the step run as dependent launches: between two of them their CUs are
// empty, and the auxiliary stream this launch sits on has the higher queue priority); every wait is bounded and raises
// `timeout_flag` instead of hanging, and a raised flag ends all further waiting (results are then invalid:
// engine.check_handoffs()).
#include "common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pl_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
constexpr int SC1 = 16;       // cache-policy bit of the raw buffer intrinsics: sc1 (write-through store / L1-bypassing load)

struct PlstmDev {
    const float* Whh; long w_dstride;            // (ndir) x (4H, H), PyTorch gate order [i | f | g | o]
    const float* pre; long pre_dstride;          // (T, Bp, 4H) per direction: input projection + both biases
    float* gates; long gates_dstride;            // (T, Bp, 4H) gate values for the backward pass (may alias pre)
    float* c_hist; long c_dstride;               // (T + 1, Bp, H), slice 0 = initial state
    float* hs; long hs_dstride;                  // (T, Bp, H) emitted h (zero for inactive rows), or NULL
    float* hx; long hx_dstride; long hx_step; long hx_ld; int hx_ring;   // where h_s is published: step slot (s mod ring, or s if
                                                                        // ring == 0) * hx_step, rows of hx_ld
    const uint8_t* keep; float scale;            // (T, Bp, H) dropout keep mask on h, or NULL
    const int* active;                           // (T, Bp): rows with 0 pass their state through (packed sequences), or NULL
    unsigned* cnt;                               // ndir arrival counters, zeroed by the launcher
    int* timeout_flag;
    int B, H, s_lo, s_hi, NW;
};

constexpr int PL_ROWS = 64;                      // Bp (four 16-row MFMA tiles)
constexpr int PL_RED = 4 * 4 * 4 * 64;           // float4 slots of the K-slice fold: [wave][gate][row tile][lane]

// KW = H / 4: the K-slice of a wave (64: encoder H = 256; 256: decoder LSTM H = 1024)
template <int KW>
__global__ __launch_bounds__(256) void plstm_fwd_kernel(PlstmDev a) {
    constexpr int NJ = KW / 16;                  // float4 k-groups of the slice
    __shared__ __attribute__((aligned(16))) float4 red[PL_RED + 256];          // fold buffer + the (64 x 16) h tile
    float4* htile = red + PL_RED;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.x / a.NW, w = blockIdx.x - dir * a.NW;
    const int H = a.H, N = 4 * H;
    const int u0 = 16 * w;                       // first hidden unit of this workgroup
    const int ln = lane & 15, lq = lane >> 4;
    const float* Whh = a.Whh + dir * a.w_dstride;
    const float* pre = a.pre + dir * a.pre_dstride;
    float* gates = a.gates + dir * a.gates_dstride;
    float* c_hist = a.c_hist + dir * a.c_dstride;
    float* hs = a.hs != nullptr ? a.hs + dir * a.hs_dstride : nullptr;
    float* hx = a.hx + dir * a.hx_dstride;
    unsigned* cnt = a.cnt + dir;

    // ---- this wave's block of W_hh, resident: B operand lane (n = ln, k-quad lq) of gate g, k-group j:
    //      W[g H + u0 + ln][wave KW + 16 j + 4 lq + 0..3]
    float4 wreg[4][NJ];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            wreg[g][j] = *reinterpret_cast<const float4*>(Whh + (long)(g * H + u0 + ln) * H + wave * KW + 16 * j + 4 * lq);

    // ---- persistent cell state of (unit u0 + ln, rows 16 wave + 4 lq + r): wave r owns row tile r after the fold
    const int row_base = 16 * wave + 4 * lq;
    float cst[4], hcar[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = row_base + r;
        cst[r] = c_hist[((long)a.s_lo * PL_ROWS + m) * H + u0 + ln];
        hcar[r] = 0.f;
        if (a.active != nullptr && a.s_lo > 0) {
            const long slot = a.hx_ring ? (a.s_lo - 1) % a.hx_ring : a.s_lo - 1;
            hcar[r] = hx[slot * a.hx_step + (long)m * a.hx_ld + u0 + ln];
        }
    }

    for (int s = a.s_lo; s < a.s_hi; ++s) {
        // ---- inputs that do not depend on h_{s-1}: requested first
        float pr[4][4];
        float kp[4];
        int act[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long m = row_base + r;
#pragma unroll
            for (int g = 0; g < 4; ++g) pr[g][r] = pre[((long)s * PL_ROWS + m) * N + g * H + u0 + ln];
            kp[r] = a.keep != nullptr ? (a.keep[((long)s * PL_ROWS + m) * H + u0 + ln] ? a.scale : 0.f) : 1.f;
            act[r] = a.active != nullptr ? a.active[(long)s * PL_ROWS + m] : 1;
        }
        f32x4 acc[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[g][rt] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (s > 0) {
            // ---- wait until every workgroup of this problem has published h_{s-1} (steps before s_lo: earlier launches)
            if (s > a.s_lo) {
                if (tid == 0) {
                    const unsigned want = (unsigned)a.NW * (unsigned)(s - a.s_lo);
                    int spins = 0;
                    // bounded, and STICKY: once any wait of any launch has given up (the flag stays raised until the host
                    // reads it), nobody spins any more -- a run that is already invalid ends quickly instead of paying
                    // one time-out per step
                    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                        __builtin_amdgcn_s_sleep(1);
                        if ((++spins & 1023) == 0 && __hip_atomic_load(a.timeout_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                        if (spins > (1 << 21)) { atomicExch(a.timeout_flag, 1); break; }        // ~ seconds: give up, flag it
                    }
                }
                __syncthreads();
            }
            const long slot = a.hx_ring ? (s - 1) % a.hx_ring : s - 1;
            const __amdgpu_buffer_rsrc_t rd = pl_rsrc(hx + slot * a.hx_step);
            // A operand lane (m = ln, k-quad lq) of row tile rt, k-group j: h[16 rt + ln][wave KW + 16 j + 4 lq + 0..3]
            float4 xa[2][4];
            auto fetch = [&](int j, float4 (&x)[4]) {
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    x[rt] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                        rd, (int)(((long)(16 * rt + ln) * a.hx_ld + wave * KW + 16 * j + 4 * lq) * 4), 0, SC1));
            };
            fetch(0, xa[0]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j + 1 < NJ) fetch(j + 1, xa[(j + 1) & 1]);
                const float4 (&x)[4] = xa[j & 1];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        acc[g][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].x, wreg[g][j].x, acc[g][rt], 0, 0, 0);
                        acc[g][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].y, wreg[g][j].y, acc[g][rt], 0, 0, 0);
                        acc[g][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].z, wreg[g][j].z, acc[g][rt], 0, 0, 0);
                        acc[g][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[rt].w, wreg[g][j].w, acc[g][rt], 0, 0, 0);
                    }
            }
            // ---- fold the four K-slices: wave r ends up with row tile r
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    if (rt != wave) red[((wave * 4 + g) * 4 + rt) * 64 + lane] = __builtin_bit_cast(float4, acc[g][rt]);
            __syncthreads();
        }
        float G[4][4];                           // [gate][r]
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) if (rt == wave) t = acc[g][rt];
            if (s > 0) {
#pragma unroll
                for (int ws = 0; ws < 4; ++ws)
                    if (ws != wave) {
                        const float4 o = red[((ws * 4 + g) * 4 + wave) * 64 + lane];
                        t[0] += o.x; t[1] += o.y; t[2] += o.z; t[3] += o.w;
                    }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) G[g][r] = t[r] + pr[g][r];
        }
        // ---- cell, in registers
        float hraw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ig = sigmoidf_(G[0][r]), fg = sigmoidf_(G[1][r]), gg = tanhf(G[2][r]), og = sigmoidf_(G[3][r]);
            const float cn = fg * cst[r] + ig * gg;
            const float h = og * tanhf(cn) * kp[r];
            const long m = row_base + r;
            if (m < a.B) {
                float* gp = gates + ((long)s * PL_ROWS + m) * N + u0 + ln;
                gp[0] = ig; gp[H] = fg; gp[2 * H] = gg; gp[3 * H] = og;
            }
            if (act[r]) { cst[r] = cn; hcar[r] = h; hraw[r] = h; }
            else hraw[r] = 0.f;                  // packed sequence: state passes through, the emitted output is zero
            if (m >= a.B) { cst[r] = 0.f; hcar[r] = 0.f; hraw[r] = 0.f; }     // pad rows stay zero
            c_hist[((long)(s + 1) * PL_ROWS + m) * H + u0 + ln] = cst[r];
            reinterpret_cast<float*>(htile)[m * 16 + ln] = hcar[r];
        }
        if (hs != nullptr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) hs[((long)s * PL_ROWS + row_base + r) * H + u0 + ln] = hraw[r];
        }
        __syncthreads();
        // ---- publish h_s: one 16-byte write-through store per thread (row tid >> 2, units 4 (tid & 3) ..)
        {
            const long slot = a.hx_ring ? s % a.hx_ring : s;
            const __amdgpu_buffer_rsrc_t wr = pl_rsrc(hx + slot * a.hx_step);
            const float4 v = htile[tid];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), wr,
                                                   (int)(((long)(tid >> 2) * a.hx_ld + u0 + 4 * (tid & 3)) * 4), 0, SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains
        __syncthreads();
        if (tid == 0 && s + 1 < a.s_hi) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

extern "C" int gt_plstm_supported(int32_t Bp, int32_t H) { return Bp == PL_ROWS && (H == 256 || H == 1024) ? 1 : 0; }

extern "C" int gt_plstm_fwd(const GtPlstm* q, void* stream) {
    GT_REQUIRE(q != nullptr, GT_E_ARG, "gt_plstm_fwd: NULL");
    GT_REQUIRE(gt_plstm_supported(q->Bp, q->H), GT_E_LIMIT, "gt_plstm_fwd: Bp = %d, H = %d (needs Bp = 64 and H = 256 or 1024)", q->Bp, q->H);
    GT_REQUIRE(q->B > 0 && q->B <= q->Bp && q->ndir >= 1 && q->ndir <= 2 && q->s_lo >= 0 && q->s_lo < q->s_hi && q->s_hi <= q->T,
               GT_E_ARG, "gt_plstm_fwd: bad sizes B=%d ndir=%d steps [%d,%d) of %d", q->B, q->ndir, q->s_lo, q->s_hi, q->T);
    GT_REQUIRE(q->Whh && q->pre && q->gates && q->c && q->hx && q->counters && q->timeout_flag, GT_E_ARG, "gt_plstm_fwd: NULL pointer");
    GT_REQUIRE(q->hx_ld >= q->H && q->hx_ld % 4 == 0 && aligned16(q->hx) && aligned16(q->Whh) && q->hx_ring >= 0 && q->hx_ring != 1,
               GT_E_ARG, "gt_plstm_fwd: exchange buffer needs 16-byte rows and ring 0 (per-step slots) or >= 2");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(q->counters, 0, sizeof(unsigned) * q->ndir, st);
    GT_REQUIRE(e == hipSuccess, (int)e, "gt_plstm_fwd: memset failed");
    PlstmDev a;
    a.Whh = q->Whh; a.w_dstride = q->w_dstride;
    a.pre = q->pre; a.pre_dstride = q->pre_dstride;
    a.gates = q->gates; a.gates_dstride = q->gates_dstride;
    a.c_hist = q->c; a.c_dstride = q->c_dstride;
    a.hs = q->hs; a.hs_dstride = q->hs_dstride;
    a.hx = q->hx; a.hx_dstride = q->hx_dstride; a.hx_step = q->hx_step; a.hx_ld = q->hx_ld; a.hx_ring = q->hx_ring;
    a.keep = q->keep; a.scale = q->p_drop > 0.f ? 1.0f / (1.0f - q->p_drop) : 1.0f;
    if (q->p_drop <= 0.f) a.keep = nullptr;
    a.active = q->active; a.cnt = q->counters; a.timeout_flag = q->timeout_flag;
    a.B = q->B; a.H = q->H; a.s_lo = q->s_lo; a.s_hi = q->s_hi; a.NW = q->H / 16;
    const dim3 grid(q->ndir * a.NW);
    const double steps = q->s_hi - q->s_lo;
    GtProfScope prof(GT_FAM_REC_CELL, 2.0 * q->ndir * PL_ROWS * 4.0 * q->H * q->H * steps,
                     4.0 * q->ndir * (4.0 * q->H * q->H + steps * PL_ROWS * 10.0 * q->H), st);
    if (q->H == 256) hipLaunchKernelGGL(plstm_fwd_kernel<64>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(plstm_fwd_kernel<256>, grid, dim3(256), 0, st, a);
    GT_LAUNCH_CHECK();
    return 0;
}
