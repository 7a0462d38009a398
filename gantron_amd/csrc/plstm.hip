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
static_assert(sizeof(PlstmDev) <= 1024, "PlstmDev is a by-value kernel argument: keep it well below the 4 KB kernel-argument segment");

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
            const float ig = sigmoidf_(G[0][r]), fg = sigmoidf_(G[1][r]), gg = tanhf_(G[2][r]), og = sigmoidf_(G[3][r]);
            const float cn = fg * cst[r] + ig * gg;
            const float h = og * tanhf_(cn) * kp[r];
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

// ---------------------------------------------------------------------------------------------------------------------
// Backward counterpart (gt_plstm_bwd): ONE launch walks steps s_hi - 1 .. s_lo of
//   dh_s = dh_direct_s + dG_{s+1} . W_hh,   cell backward -> dG_s (gate pre-activation gradients), dc carried on.
// Same decomposition (workgroup w owns 16 hidden units = 64 gate columns = 64 ROWS of W_hh, resident in registers), but the
// product runs the other way round: the workgroup's own dG block (64 batch rows x 64 gate columns, produced locally by the
// cell backward, staged in LDS) times its 64 x H block of W_hh is a PARTIAL of dh for ALL H units, so the exchange is a
// reduce-scatter: the partial is cut into H / 16 tiles of (64 rows x 16 units), tile d goes to workgroup d's inbox
// (xbuf[slot][d][w], 4 KiB, written through), and at the next step every workgroup adds the H / 16 tiles of its inbox.
// The four waves split the OUTPUT columns (H / 4 each, i.e. H / 64 destination workgroups per wave), so no fold is needed.
struct PlstmBwdDev {
    const float* Whh; long w_dstride;
    const float* gates; long gates_dstride;      // (T, Bp, 4H) gate values of the forward pass
    const float* c_hist; long c_dstride;         // (T + 1, Bp, H)
    const float* dh_direct; long dd_dstride; long dd_step; long dd_ld;       // gradient of the emitted h of step s
    float* dG; long dG_dstride;                  // (T, Bp, 4H) out
    float* dc; long dc_dstride;                  // (Bp, H) carried cell-state gradient, read at the start, written at the end
    float* xbuf; long x_dstride;                 // (2 slots, NW dest, NW src, 64, 16) per direction
    const uint8_t* keep; float scale;
    const int* active;
    unsigned* cnt; int* timeout_flag;
    int B, H, s_lo, s_hi, NW, recurrent_in;
};
static_assert(sizeof(PlstmBwdDev) <= 1024, "PlstmBwdDev is a by-value kernel argument: keep it well below the 4 KB kernel-argument segment");

template <int KW>       // KW = H / 4 output columns per wave
__global__ __launch_bounds__(256) void plstm_bwd_kernel(PlstmBwdDev a) {
    constexpr int NCT = KW / 16;                 // destination workgroups (column tiles) per wave
    constexpr int LDA = 68;                      // dG tile row stride in LDS (floats): conflict-free float4 fragment reads
    __shared__ __attribute__((aligned(16))) float dgs[PL_ROWS * LDA];         // [row][64 own gate columns]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.x / a.NW, w = blockIdx.x - dir * a.NW;
    const int H = a.H, N = 4 * H;
    const int u0 = 16 * w;
    const int ln = lane & 15, lq = lane >> 4;
    const float* Whh = a.Whh + dir * a.w_dstride;
    const float* gates = a.gates + dir * a.gates_dstride;
    const float* c_hist = a.c_hist + dir * a.c_dstride;
    const float* dd = a.dh_direct + dir * a.dd_dstride;
    float* dG = a.dG + dir * a.dG_dstride;
    float* dcp = a.dc + dir * a.dc_dstride;
    float* xbuf = a.xbuf + dir * a.x_dstride;
    unsigned* cnt = a.cnt + dir;
    const long tile = (long)PL_ROWS * 16;                            // floats of one (64 x 16) tile
    const long slot_stride = (long)a.NW * a.NW * tile;

    // ---- this wave's block of W_hh, resident: B operand lane (n = ln, k-quad lq) of column tile ct, gate j (= k-group):
    //      k = 16 j + 4 lq + i  <->  row j H + u0 + 4 lq + i of W_hh;  n <-> column wave KW + 16 ct + ln
    float4 wreg[NCT][4];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* wp = Whh + (long)(j * H + u0 + 4 * lq) * H + wave * KW + 16 * ct + ln;
            wreg[ct][j] = make_float4(wp[0], wp[H], wp[2 * (long)H], wp[3 * (long)H]);
        }

    // ---- cell phase layout: thread (row = tid >> 2, units 4 (tid & 3) .. + 3), float4 everywhere
    const int crow = tid >> 2, uq = tid & 3;
    const int cu = u0 + 4 * uq;
    float4 dcv = *reinterpret_cast<const float4*>(dcp + (long)crow * H + cu);

    for (int s = a.s_hi - 1; s >= a.s_lo; --s) {
        // ---- inputs that do not depend on the exchange
        const float* gt = gates + ((long)s * PL_ROWS + crow) * N + cu;
        const float4 gi = *reinterpret_cast<const float4*>(gt), gf = *reinterpret_cast<const float4*>(gt + H);
        const float4 gg = *reinterpret_cast<const float4*>(gt + 2 * (long)H), go = *reinterpret_cast<const float4*>(gt + 3 * (long)H);
        const float4 cn = *reinterpret_cast<const float4*>(c_hist + ((long)(s + 1) * PL_ROWS + crow) * H + cu);
        const float4 cp = *reinterpret_cast<const float4*>(c_hist + ((long)s * PL_ROWS + crow) * H + cu);
        float4 dh = *reinterpret_cast<const float4*>(dd + (long)s * a.dd_step + (long)crow * a.dd_ld + cu);
        uchar4 kp = make_uchar4(1, 1, 1, 1);
        if (a.keep != nullptr) kp = *reinterpret_cast<const uchar4*>(a.keep + ((long)s * PL_ROWS + crow) * H + cu);
        const bool act = crow < a.B && (a.active == nullptr || a.active[(long)s * PL_ROWS + crow] != 0);
        // ---- recurrent part: the H / 16 tiles of this workgroup's inbox, published while step s + 1 ran
        const bool have_in = s + 1 < a.s_hi || a.recurrent_in;
        if (have_in) {
            if (s + 1 < a.s_hi) {
                if (tid == 0) {
                    const unsigned want = (unsigned)a.NW * (unsigned)(a.s_hi - 1 - s);
                    int spins = 0;
                    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                        __builtin_amdgcn_s_sleep(1);
                        if ((++spins & 1023) == 0 && __hip_atomic_load(a.timeout_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                        if (spins > (1 << 21)) { atomicExch(a.timeout_flag, 1); break; }
                    }
                }
                __syncthreads();
            }
            const float* inbox = xbuf + ((s + 1) & 1) * slot_stride + (long)w * a.NW * tile;      // [src][64][16]
            const __amdgpu_buffer_rsrc_t rd = pl_rsrc(inbox);
            for (int s0 = 0; s0 < a.NW; s0 += 16) {               // 16 tiles in flight per thread
                float4 t[16];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    t[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                        rd, (int)(((long)(s0 + j) * tile + crow * 16 + 4 * uq) * 4), 0, SC1));
#pragma unroll
                for (int j = 0; j < 16; ++j) { dh.x += t[j].x; dh.y += t[j].y; dh.z += t[j].z; dh.w += t[j].w; }
            }
        }
        // ---- cell backward (rec_cell.h / lstm_cell_bwd_kernel), thread = (row, 4 units)
        float4 dgi = make_float4(0.f, 0.f, 0.f, 0.f), dgf = dgi, dgg = dgi, dgo = dgi;
        if (act) {
            const float dhv[4] = {dh.x, dh.y, dh.z, dh.w}, giv[4] = {gi.x, gi.y, gi.z, gi.w}, gfv[4] = {gf.x, gf.y, gf.z, gf.w};
            const float ggv[4] = {gg.x, gg.y, gg.z, gg.w}, gov[4] = {go.x, go.y, go.z, go.w};
            const float cnv[4] = {cn.x, cn.y, cn.z, cn.w}, cpv[4] = {cp.x, cp.y, cp.z, cp.w};
            const float dcn[4] = {dcv.x, dcv.y, dcv.z, dcv.w};
            const unsigned char kpv[4] = {kp.x, kp.y, kp.z, kp.w};
            float o[5][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d = dhv[e];
                if (a.keep != nullptr) d *= kpv[e] ? a.scale : 0.f;
                const float tc = tanhf_(cnv[e]);
                const float dcell = dcn[e] + d * gov[e] * (1.f - tc * tc);
                o[0][e] = dcell * ggv[e] * giv[e] * (1.f - giv[e]);
                o[1][e] = dcell * cpv[e] * gfv[e] * (1.f - gfv[e]);
                o[2][e] = dcell * giv[e] * (1.f - ggv[e] * ggv[e]);
                o[3][e] = d * tc * gov[e] * (1.f - gov[e]);
                o[4][e] = dcell * gfv[e];
            }
            dgi = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]); dgf = make_float4(o[1][0], o[1][1], o[1][2], o[1][3]);
            dgg = make_float4(o[2][0], o[2][1], o[2][2], o[2][3]); dgo = make_float4(o[3][0], o[3][1], o[3][2], o[3][3]);
            dcv = make_float4(o[4][0], o[4][1], o[4][2], o[4][3]);
        }
        {
            float* dg = dG + ((long)s * PL_ROWS + crow) * N + cu;
            *reinterpret_cast<float4*>(dg) = dgi; *reinterpret_cast<float4*>(dg + H) = dgf;
            *reinterpret_cast<float4*>(dg + 2 * (long)H) = dgg; *reinterpret_cast<float4*>(dg + 3 * (long)H) = dgo;
            float* ds = dgs + crow * LDA + 4 * uq;                   // k = 16 gate + unit
            *reinterpret_cast<float4*>(ds) = dgi; *reinterpret_cast<float4*>(ds + 16) = dgf;
            *reinterpret_cast<float4*>(ds + 32) = dgg; *reinterpret_cast<float4*>(ds + 48) = dgo;
        }
        if (s == 0) break;                                           // nothing consumes a product behind step 0 (a later launch picks
                                                                     // the tiles of step s_lo up: recurrent_in)
        __syncthreads();
        // ---- partial of dh_{s-1} for this wave's H / 4 units, TRANSPOSED: C^T = W_block^T (units x k) . dG^T (k x rows), so the
        //      C layout (lane (ln, lq) holds rows 4 lq .. 4 lq + 3 of column ln) leaves four consecutive UNITS of batch row ln
        //      in one lane: one 16-byte write-through store per lane per (16 rows x 16 units) patch, no transpose
        float* outbox = xbuf + (s & 1) * slot_stride;
#pragma unroll 1
        for (int rt = 0; rt < 4; ++rt) {
            float4 af[4];                                            // dG^T fragments: row 16 rt + ln, k = 16 j + 4 lq ..
#pragma unroll
            for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const float4*>(dgs + (16 * rt + ln) * LDA + 16 * j + 4 * lq);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[ct][j].x, af[j].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[ct][j].y, af[j].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[ct][j].z, af[j].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[ct][j].w, af[j].w, acc, 0, 0, 0);
                }
                const int d = wave * NCT + ct;                       // destination workgroup (units 16 d .. 16 d + 15)
                const __amdgpu_buffer_rsrc_t wr = pl_rsrc(outbox + ((long)d * a.NW + w) * tile);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), wr, ((16 * rt + ln) * 16 + 4 * lq) * 4, 0, SC1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *reinterpret_cast<float4*>(dcp + (long)crow * H + cu) = dcv;
}

}  // namespace

// Every workgroup of a persistent recurrence spin-waits on its peers: all ndir * H / 16 of them (one per CU, the whole
// register file) must be resident at once, beside the kernels of the attention chain that run meanwhile.  So the CURRENT
// device must have at least twice as many CUs as the largest launch holds (H = 1024: 64 workgroups -> 128 CUs; H = 256, two
// directions: 32 -> 64): a partitioned device (CPX: 32 CUs) or a small part gets the per-step launches instead.  (Streams
// with CU masks are the caller's business: the loops never create any.)  A launch whose workgroups could not all become
// resident does not hang -- the waits are bounded and raise the sticky time-out flag -- but its results are undefined.
static int plstm_device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus[dev] = n > 0 ? n : -1;
    }
    return cus[dev] > 0 ? cus[dev] : 0;
}
extern "C" int gt_plstm_supported(int32_t Bp, int32_t H) {
    if (Bp != PL_ROWS || !(H == 256 || H == 1024)) return 0;
    const int need = 2 * (H == 1024 ? H / 16 : 2 * (H / 16));
    return plstm_device_cus() >= need ? 1 : 0;
}

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
    GtProfScope prof(GT_FAM_PLSTM, 2.0 * q->ndir * PL_ROWS * 4.0 * q->H * q->H * steps,
                     4.0 * q->ndir * (4.0 * q->H * q->H + steps * PL_ROWS * 10.0 * q->H), st);
    if (q->H == 256) hipLaunchKernelGGL(plstm_fwd_kernel<64>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(plstm_fwd_kernel<256>, grid, dim3(256), 0, st, a);
    GT_LAUNCH_CHECK();
    return 0;
}

extern "C" int64_t gt_plstm_bwd_xbuf_elems(int32_t H) { return 2L * (H / 16) * (H / 16) * PL_ROWS * 16; }

extern "C" int gt_plstm_bwd(const GtPlstmBwd* q, void* stream) {
    GT_REQUIRE(q != nullptr, GT_E_ARG, "gt_plstm_bwd: NULL");
    GT_REQUIRE(gt_plstm_supported(q->Bp, q->H), GT_E_LIMIT, "gt_plstm_bwd: Bp = %d, H = %d (needs Bp = 64 and H = 256 or 1024)", q->Bp, q->H);
    GT_REQUIRE(q->B > 0 && q->B <= q->Bp && q->ndir >= 1 && q->ndir <= 2 && q->s_lo >= 0 && q->s_lo < q->s_hi && q->s_hi <= q->T,
               GT_E_ARG, "gt_plstm_bwd: bad sizes B=%d ndir=%d steps [%d,%d) of %d", q->B, q->ndir, q->s_lo, q->s_hi, q->T);
    GT_REQUIRE(q->Whh && q->gates && q->c && q->dh_direct && q->dG && q->dc && q->xbuf && q->counters && q->timeout_flag, GT_E_ARG,
               "gt_plstm_bwd: NULL pointer");
    GT_REQUIRE(q->dd_ld >= q->H && q->dd_ld % 4 == 0 && aligned16(q->dh_direct) && aligned16(q->xbuf) && aligned16(q->dc) &&
                   aligned16(q->dG) && aligned16(q->gates) && aligned16(q->c), GT_E_ALIGN, "gt_plstm_bwd: 16-byte aligned buffers, dd_ld %% 4 == 0");
    GT_REQUIRE(!q->recurrent_in || q->s_hi < q->T, GT_E_ARG, "gt_plstm_bwd: recurrent_in needs a step s_hi whose partials were published");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(q->counters, 0, sizeof(unsigned) * q->ndir, st);
    GT_REQUIRE(e == hipSuccess, (int)e, "gt_plstm_bwd: memset failed");
    PlstmBwdDev a;
    a.Whh = q->Whh; a.w_dstride = q->w_dstride;
    a.gates = q->gates; a.gates_dstride = q->gates_dstride;
    a.c_hist = q->c; a.c_dstride = q->c_dstride;
    a.dh_direct = q->dh_direct; a.dd_dstride = q->dd_dstride; a.dd_step = q->dd_step; a.dd_ld = q->dd_ld;
    a.dG = q->dG; a.dG_dstride = q->dG_dstride;
    a.dc = q->dc; a.dc_dstride = q->dc_dstride;
    a.xbuf = q->xbuf; a.x_dstride = q->x_dstride;
    a.keep = q->p_drop > 0.f ? q->keep : nullptr; a.scale = q->p_drop > 0.f ? 1.0f / (1.0f - q->p_drop) : 1.0f;
    a.active = q->active; a.cnt = q->counters; a.timeout_flag = q->timeout_flag;
    a.B = q->B; a.H = q->H; a.s_lo = q->s_lo; a.s_hi = q->s_hi; a.NW = q->H / 16; a.recurrent_in = q->recurrent_in;
    const dim3 grid(q->ndir * a.NW);
    const double steps = q->s_hi - q->s_lo;
    GtProfScope prof(GT_FAM_PLSTM, 2.0 * q->ndir * PL_ROWS * 4.0 * q->H * q->H * steps,
                     4.0 * q->ndir * (4.0 * q->H * q->H + steps * PL_ROWS * 14.0 * q->H), st);
    if (q->H == 256) hipLaunchKernelGGL(plstm_bwd_kernel<64>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(plstm_bwd_kernel<256>, grid, dim3(256), 0, st, a);
    GT_LAUNCH_CHECK();
    return 0;
}
