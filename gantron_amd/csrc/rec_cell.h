// Epilogue of the fused recurrent product + LSTM cell launches (gt_rec_cell_fwd), shared by the fp32 kernel
// (recurrent.hip) and the bf16-operand kernel (recurrent_bf16.hip): both end their main loop with the folded
// [rows][64 columns] fp32 tile in LDS (row stride REC_EP_LD), columns = the four gates of 16 hidden units.
#pragma once
#include "common.h"

// (plain data: crosses translation units by value)
struct RecCellEpi {
    const float* pre; long ld_pre;        // (rows, 4H) gate pre-activations [i | f | g | o] (PyTorch gate order), added
    const float* c_prev; float* c_out;    // (rows, H)
    const uint8_t* keep; float scale;     // dropout on h (or NULL)
    float* gates;                         // (rows, 4H) gate values saved for the backward pass (may alias `pre`)
    float* h_raw; float* h0; long ld0; float* h1; long ld1; float* h2; long ld2;     // h destinations (NULL = skip)
    const float* Wq; long ldwq;           // query_layer weight (A, H) or NULL
    float* qpart; long q_sstride;         // query slab of column group cg at qpart + cg * q_sstride, rows of Apad
    unsigned* counters;                   // (row blocks, N/64) tickets, zero-initialised ONCE by the caller
    int B, H, A, Apad, S;
    // packed-sequence semantics (encoder BiLSTM): rows with active[m] == 0 pass their state through (c' = c, the emitted
    // h_raw is 0, the carried h is h_prev)
    const int* active; const float* h_prev; long ld_hprev;
    // two independent problems in one launch (the BiLSTM directions): blockIdx.z = dir * nzb + row block; element offsets
    // of direction 1 relative to direction 0
    int nzb;
    long dX, dW, dPart, dPre, dC, dGates, dHraw, dH0, dHprev;
    // diagnostics (GANTRON_DEBUG=rec_stamps): phase stamps go to stamps[8 * workgroup + k] when the struct's `stamp_on` is set; the
    // buffer's address travels in a __device__ variable, not here (the struct is a by-value kernel argument: with 16 more
    // bytes in it rocprofv3's counter collection crashed in its copy of the kernel arguments)
    int stamp_on;
    // Sx > 0: slabs S .. S + Sx - 1 of this column group were written by an EARLIER launch (the part of the product whose
    // operands existed before this step's: GtRecTail, rec_tail.h); the last arriver adds them with the K-slices' own
    int Sx;
};
static_assert(sizeof(RecCellEpi) == 288, "RecCellEpi is a by-value kernel argument (288 bytes: Sx sits in what was padding)");
#define REC_STAMP(e, k) do { if ((e).stamp_on && threadIdx.x == 0) \
    g_rec_stamps_dev[((long)blockIdx.x + (long)gridDim.x * blockIdx.y) * 8 + (k)] = wall_clock64(); } while (0)

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ long long* g_rec_stamps_dev = nullptr;      // (one per translation unit that includes this header)


__device__ __forceinline__ __amdgpu_buffer_rsrc_t rc_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}


constexpr int REC_EP_LD = 68;             // epilogue tile row stride (floats)
constexpr int REC_EP_THREADS = 512;

// direction 1 of a paired launch: offsets of the epilogue's pointers (X / Wp / part are the kernel's business)
__device__ __forceinline__ void rec_cell_dir1(RecCellEpi& e) {
    if (e.pre) e.pre += e.dPre;
    e.c_prev += e.dC; e.c_out += e.dC;
    if (e.gates) e.gates += e.dGates;
    if (e.h_raw) e.h_raw += e.dHraw;
    if (e.h0) e.h0 += e.dH0;
    if (e.h_prev) e.h_prev += e.dHprev;
}

// red: the tile (ROWS x 64, stride REC_EP_LD) followed by >= ROWS * 20 + 4 floats of scratch, all in LDS; part: slabs of
// this (direction, row block); cg / s: column group and K-slice of this workgroup; rb: row block; ticket: this group's
// counter.  Every thread of the workgroup calls it (it contains barriers); returns when the workgroup is done.
template <int MT>
__device__ __forceinline__ void rec_cell_epilogue(float* red, const RecCellEpi& e, float* part, long ps, int N, int cg, int s,
                                                  int rb, unsigned* ticket) {
    constexpr int EP_LD = REC_EP_LD;
    constexpr int REC_THREADS = REC_EP_THREADS;
    constexpr int ROWS = MT * 16;
    constexpr int NV = (ROWS * 16 + REC_THREADS - 1) / REC_THREADS;        // float4 of the tile per thread
    float* lds_h = red + ROWS * EP_LD;             // [ROWS][20] h tile for the query product (inside the staging array)
    unsigned* lds_ticket = reinterpret_cast<unsigned*>(red + ROWS * EP_LD + ROWS * 20);
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = 64 * rb;                      // first batch row of this row block
    float* slab0 = part + (long)cg * 64;                                    // slab 0 of this column group
    if (e.S > 1) {
        // ---- publish this K-slice's tile (write-through), then draw a ticket
        const __amdgpu_buffer_rsrc_t wr = rc_rsrc(slab0 + (long)s * ps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + REC_THREADS * i;
            const int row = f >> 4, c4 = f & 15;
            if (row < ROWS) {
                const float4 v = *reinterpret_cast<const float4*>(&red[row * EP_LD + 4 * c4]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), wr, (int)(((long)row * N + 4 * c4) * 4), 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains its stores
        __syncthreads();
        if (tid == 0)
            *lds_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        REC_STAMP(e, 3);                 // tile published, ticket drawn
        if ((*lds_ticket % (unsigned)e.S) != (unsigned)(e.S - 1)) return;       // not the last K-slice of this group
    }
    // ---- last arriver: inputs of the cell are requested together with the other K-slices' slabs
    const bool cell_thread = tid < ROWS * 4;
    const int crow = tid >> 2, uq = tid & 3;
    const int m = row0 + crow;                     // batch row
    const int u = 16 * cg + 4 * uq;                // first of this thread's 4 hidden units
    const bool live = cell_thread && m < e.B;
    const long cidx = (long)m * e.H + u;
    float4 pre4[4], cp = make_float4(0.f, 0.f, 0.f, 0.f), hp4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uchar4 kp = make_uchar4(1, 1, 1, 1);
    bool act = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) pre4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        if (e.pre != nullptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) pre4[q] = *reinterpret_cast<const float4*>(e.pre + (long)m * e.ld_pre + q * e.H + u);
        }
        cp = *reinterpret_cast<const float4*>(e.c_prev + cidx);
        if (e.keep != nullptr) kp = *reinterpret_cast<const uchar4*>(e.keep + cidx);
        if (e.active != nullptr) {
            act = e.active[m] != 0;
            if (e.h_prev != nullptr) hp4 = *reinterpret_cast<const float4*>(e.h_prev + (long)m * e.ld_hprev + u);
        }
    }
    // query weights of this wave's column blocks: B operand of the 16x16x4 MFMA, lane supplies Wq[n = lane & 15][k..k+3]
    float4 wq[2];
    const int ncb = e.Wq != nullptr ? e.Apad / 16 : 0;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        wq[sl] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int cb = wave + 8 * sl;
        if (cb < ncb) {
            const int ch = min(cb * 16 + (lane & 15), e.A - 1);
            const float4 t = *reinterpret_cast<const float4*>(e.Wq + (long)ch * e.ldwq + 16 * cg + 4 * (lane >> 4));
            if (cb * 16 + (lane & 15) < e.A) wq[sl] = t;
        }
    }
    const int nsl = e.S + e.Sx;                    // the K-slices' slabs, then the earlier launch's
    if (nsl > 1) {
        // The slabs are summed in SLAB order whoever arrived last -- this workgroup's own tile (still in LDS) takes its place s
        // in the sequence -- so the gate sums do not depend on the arrival order (round 4: `the others in order, then my own`
        // made their rounding depend on which K-slice drew the last ticket; the recurrence amplified that over a sequence).
        const __amdgpu_buffer_rsrc_t rd = rc_rsrc(slab0);
        float4 own[NV], acc[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = min(tid + REC_THREADS * i, ROWS * 16 - 1);
            own[i] = *reinterpret_cast<const float4*>(&red[(f >> 4) * EP_LD + 4 * (f & 15)]);
            acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int s0 = 0; s0 < nsl; s0 += 4) {       // four slabs in flight per pass
            float4 t[4][NV];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // slabs beyond the last and this workgroup's own (still in LDS) are replaced by a re-read of slab 0 and masked
                const int sj = s0 + j;
                const bool use = sj < nsl && sj != s;
                // (the re-read must hit written memory: slab 0 is published before any ticket when S > 1; with S == 1 this
                // workgroup's own tile never leaves LDS and the first earlier slab stands in)
                const long base = (long)(use ? sj : (e.Sx > 0 ? e.S : 0)) * ps;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int f = min(tid + REC_THREADS * i, ROWS * 16 - 1);
                    const int row = f >> 4, c4 = f & 15;
                    t[j][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rd, (int)((base + (long)row * N + 4 * c4) * 4), 0, 16));
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sj = s0 + j;
                const bool mine = sj == s && e.S > 1;          // (S == 1: the own tile is slab 0 of the sequence, s == 0)
                const float w = (sj < nsl && sj != s) ? 1.f : 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const float4 v = mine ? own[i] : t[j][i];
                    const float wv = mine ? 1.f : w;
                    acc[i].x += wv * v.x; acc[i].y += wv * v.y; acc[i].z += wv * v.z; acc[i].w += wv * v.w;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + REC_THREADS * i;
            const int row = f >> 4, c4 = f & 15;
            if (row < ROWS) {
                float4* d = reinterpret_cast<float4*>(&red[row * EP_LD + 4 * c4]);
                if (e.S > 1) *d = acc[i];
                else { float4 v = *d; v.x += acc[i].x; v.y += acc[i].y; v.z += acc[i].z; v.w += acc[i].w; *d = v; }
            }
        }
        __syncthreads();
        REC_STAMP(e, 4);                 // the other K-slices' slabs are in
    }
    // ---- cell: thread = (row, 4 units); tile column 16*gate + unit
    float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);      // carried hidden state (what the next step and the query see)
    if (live) {
        float4 gv[4], cn, hraw;
        float* gvf = reinterpret_cast<float*>(&gv[0]);
        const float* cpf = reinterpret_cast<const float*>(&cp);
        float* cnf = reinterpret_cast<float*>(&cn);
        float* hvf = reinterpret_cast<float*>(&hv);
        const unsigned char* kpf = reinterpret_cast<const unsigned char*>(&kp);
        float gi[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(&red[crow * EP_LD + 16 * q + 4 * uq]);
            gi[4 * q + 0] = t.x + pre4[q].x; gi[4 * q + 1] = t.y + pre4[q].y;
            gi[4 * q + 2] = t.z + pre4[q].z; gi[4 * q + 3] = t.w + pre4[q].w;
        }
#pragma unroll
        for (int el = 0; el < 4; ++el) {
            const float ig = sigmoidf_(gi[0 * 4 + el]), fg = sigmoidf_(gi[1 * 4 + el]);
            const float gg = tanhf_(gi[2 * 4 + el]), og = sigmoidf_(gi[3 * 4 + el]);
            gvf[0 * 4 + el] = ig; gvf[1 * 4 + el] = fg; gvf[2 * 4 + el] = gg; gvf[3 * 4 + el] = og;
            const float c = fg * cpf[el] + ig * gg;
            cnf[el] = c;
            float h = og * tanhf_(c);
            if (e.keep != nullptr) h *= kpf[el] ? e.scale : 0.f;
            hvf[el] = h;
        }
        hraw = hv;
        if (!act) { cn = cp; hraw = make_float4(0.f, 0.f, 0.f, 0.f); hv = hp4; }
        if (e.gates != nullptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(e.gates + (long)m * 4 * e.H + q * e.H + u) = gv[q];
        }
        *reinterpret_cast<float4*>(e.c_out + cidx) = cn;
        if (e.h_raw != nullptr) *reinterpret_cast<float4*>(e.h_raw + cidx) = hraw;
        if (e.h0 != nullptr) *reinterpret_cast<float4*>(e.h0 + (long)m * e.ld0 + u) = hv;
        if (e.h1 != nullptr) *reinterpret_cast<float4*>(e.h1 + (long)m * e.ld1 + u) = hv;
        if (e.h2 != nullptr) *reinterpret_cast<float4*>(e.h2 + (long)m * e.ld2 + u) = hv;
    }
    REC_STAMP(e, 5);                     // cell done, state stored
    if (ncb == 0) return;
    // ---- query slab of this column group: [ROWS x 16 units] . [16 x Apad]
    if (cell_thread) *reinterpret_cast<float4*>(&lds_h[crow * 20 + 4 * uq]) = hv;       // rows beyond the batch: zeros
    __syncthreads();
    float* qout = e.qpart + (long)cg * e.q_sstride;
    const __amdgpu_buffer_rsrc_t qr = rc_rsrc(qout);      // (uniform descriptor; the row goes into the per-lane offset)
    constexpr int QAUX = 16;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int cb = wave + 8 * sl;
        if (cb >= ncb) break;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float4 x = *reinterpret_cast<const float4*>(&lds_h[(16 * mt + (lane & 15)) * 20 + 4 * (lane >> 4)]);
            // TRANSPOSED (Wq as the A operand): C^T[query column 4 (lane >> 4) + r][row lane & 15] leaves four consecutive
            // query columns of one batch row in a lane -- one 16-byte write-through store instead of four scattered dwords
            // (the slab reaches the fabric while the launch still runs, not in its end-of-kernel write-back)
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[sl].x, x.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[sl].y, x.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[sl].z, x.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[sl].w, x.w, acc, 0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc), qr,
                                                   (int)(((long)(row0 + 16 * mt + (lane & 15)) * e.Apad + cb * 16 + 4 * (lane >> 4)) * 4), 0, QAUX);
        }
    }
    REC_STAMP(e, 6);                     // query slab issued
}

}  // namespace

// ---- backward counterpart: recurrent input-gradient product + cell backward of the PREVIOUS step ---------------------
// The product of step t, d h_{t-1} = dG_t . W_hh (64 hidden units per workgroup, K-splits meeting as above), is what the
// cell backward of step t - 1 waits for: the last arriver of a column group adds the emitted-state gradient, runs the cell
// backward for its 64 units x all rows and writes dG_{t-1} -- the next launch's operand.  (gt_rec_cellbwd)
struct RecCellBwdEpi {
    const float* dh_direct; long ld_direct;     // (rows, >= H) gradient of the h EMITTED by step t - 1, or NULL
    const float* gates;                         // (rows, 4H) gate values of step t - 1
    const float* c_new; const float* c_prev;    // (rows, H) cell state after / before step t - 1
    const uint8_t* keep; float scale;           // dropout mask of step t - 1's h (or NULL)
    float* dc;                                  // (rows, H) carried cell-state gradient, read and replaced
    float* dgates;                              // (rows, 4H) pre-activation gate gradients of step t - 1
    unsigned* counters;
    int B, H, S;
};

namespace {

template <int MT>
__device__ __forceinline__ void rec_cellbwd_epilogue(float* red, const RecCellBwdEpi& e, float* part, long ps, int N, int cg,
                                                     int s, int rb, unsigned* ticket) {
    constexpr int EP_LD = REC_EP_LD;
    constexpr int REC_THREADS = REC_EP_THREADS;
    constexpr int ROWS = MT * 16;
    constexpr int NV = (ROWS * 16 + REC_THREADS - 1) / REC_THREADS;        // float4 of the tile per thread
    unsigned* lds_ticket = reinterpret_cast<unsigned*>(red + ROWS * EP_LD);
    const int tid = threadIdx.x;
    const int row0 = 64 * rb;
    float* slab0 = part + (long)cg * 64;
    if (e.S > 1) {
        const __amdgpu_buffer_rsrc_t wr = rc_rsrc(slab0 + (long)s * ps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + REC_THREADS * i;
            const int row = f >> 4, c4 = f & 15;
            if (row < ROWS) {
                const float4 v = *reinterpret_cast<const float4*>(&red[row * EP_LD + 4 * c4]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), wr, (int)(((long)row * N + 4 * c4) * 4), 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains its stores
        __syncthreads();
        if (tid == 0) *lds_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if ((*lds_ticket % (unsigned)e.S) != (unsigned)(e.S - 1)) return;       // not the last K-slice of this group
    }
    // ---- last arriver: thread = (row, 4 units) x NV; the cell's inputs are requested together with the other slabs
    float4 gi[NV], gf[NV], gg[NV], go[NV], cn[NV], cp[NV], dcn[NV], dd[NV];
    uchar4 kp[NV];
    bool live[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int f = min(tid + REC_THREADS * i, ROWS * 16 - 1);
        const int row = f >> 4, c4 = f & 15;
        const int m = row0 + row, u = 64 * cg + 4 * c4;
        live[i] = (tid + REC_THREADS * i < ROWS * 16) && m < e.B && u < e.H;
        const int mc = min(m, e.B - 1), uc = min(u, e.H - 4);
        const long idx = (long)mc * e.H + uc;
        const float* gt = e.gates + (long)mc * 4 * e.H + uc;
        gi[i] = *reinterpret_cast<const float4*>(gt); gf[i] = *reinterpret_cast<const float4*>(gt + e.H);
        gg[i] = *reinterpret_cast<const float4*>(gt + 2 * e.H); go[i] = *reinterpret_cast<const float4*>(gt + 3 * e.H);
        cn[i] = *reinterpret_cast<const float4*>(e.c_new + idx);
        cp[i] = *reinterpret_cast<const float4*>(e.c_prev + idx);
        dcn[i] = *reinterpret_cast<const float4*>(e.dc + idx);
        dd[i] = e.dh_direct != nullptr ? *reinterpret_cast<const float4*>(e.dh_direct + (long)mc * e.ld_direct + uc)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
        kp[i] = e.keep != nullptr ? *reinterpret_cast<const uchar4*>(e.keep + idx) : make_uchar4(1, 1, 1, 1);
    }
    float4 acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e.S > 1) {
        const __amdgpu_buffer_rsrc_t rd = rc_rsrc(slab0);
        for (int s0 = 0; s0 < e.S; s0 += 4) {       // four slabs in flight per pass
            float4 t[4][NV];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sj = s0 + j;
                const bool use = sj < e.S && sj != s;
                const long base = (long)(use ? sj : 0) * ps;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int f = min(tid + REC_THREADS * i, ROWS * 16 - 1);
                    const int row = f >> 4, c4 = f & 15;
                    t[j][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rd, (int)((base + (long)row * N + 4 * c4) * 4), 0, 16));
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sj = s0 + j;
                const float w = (sj < e.S && sj != s) ? 1.f : 0.f;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    acc[i].x += w * t[j][i].x; acc[i].y += w * t[j][i].y; acc[i].z += w * t[j][i].z; acc[i].w += w * t[j][i].w;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (!live[i]) continue;
        const int f = tid + REC_THREADS * i;
        const int row = f >> 4, c4 = f & 15;
        const int m = row0 + row, u = 64 * cg + 4 * c4;
        const float4 own = *reinterpret_cast<const float4*>(&red[row * EP_LD + 4 * c4]);
        const float dhv[4] = {acc[i].x + own.x + dd[i].x, acc[i].y + own.y + dd[i].y, acc[i].z + own.z + dd[i].z,
                              acc[i].w + own.w + dd[i].w};
        const float giv[4] = {gi[i].x, gi[i].y, gi[i].z, gi[i].w}, gfv[4] = {gf[i].x, gf[i].y, gf[i].z, gf[i].w};
        const float ggv[4] = {gg[i].x, gg[i].y, gg[i].z, gg[i].w}, gov[4] = {go[i].x, go[i].y, go[i].z, go[i].w};
        const float cnv[4] = {cn[i].x, cn[i].y, cn[i].z, cn[i].w}, cpv[4] = {cp[i].x, cp[i].y, cp[i].z, cp[i].w};
        const float dcv[4] = {dcn[i].x, dcn[i].y, dcn[i].z, dcn[i].w};
        const unsigned char kpv[4] = {kp[i].x, kp[i].y, kp[i].z, kp[i].w};
        float dgi[4], dgf[4], dgg[4], dgo[4], dcp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float dh = dhv[q];
            if (e.keep != nullptr) dh *= (kpv[q] ? e.scale : 0.f);
            const float tc = tanhf_(cnv[q]);
            const float dcell = dcv[q] + dh * gov[q] * (1.f - tc * tc);
            dgi[q] = dcell * ggv[q] * giv[q] * (1.f - giv[q]);
            dgf[q] = dcell * cpv[q] * gfv[q] * (1.f - gfv[q]);
            dgg[q] = dcell * giv[q] * (1.f - ggv[q] * ggv[q]);
            dgo[q] = dh * tc * gov[q] * (1.f - gov[q]);
            dcp[q] = dcell * gfv[q];
        }
        float* dg = e.dgates + (long)m * 4 * e.H + u;
        *reinterpret_cast<float4*>(dg) = make_float4(dgi[0], dgi[1], dgi[2], dgi[3]);
        *reinterpret_cast<float4*>(dg + e.H) = make_float4(dgf[0], dgf[1], dgf[2], dgf[3]);
        *reinterpret_cast<float4*>(dg + 2 * e.H) = make_float4(dgg[0], dgg[1], dgg[2], dgg[3]);
        *reinterpret_cast<float4*>(dg + 3 * e.H) = make_float4(dgo[0], dgo[1], dgo[2], dgo[3]);
        *reinterpret_cast<float4*>(e.dc + (long)m * e.H + u) = make_float4(dcp[0], dcp[1], dcp[2], dcp[3]);
    }
}

}  // namespace
