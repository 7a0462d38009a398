// Internal helpers shared by the HIP translation units of libgantron_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/gantron_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void gt_set_error(const char* fmt, ...);
// value of `key` in the comma list `key[=int]` of environment variable `var` (gantron_amd/switches.py documents the lists);
// a bare key counts as 1, an absent one as `dflt`
int gt_env_item(const char* var, const char* key, int dflt);

#define GT_REQUIRE(cond, code, ...)                 \
    do {                                            \
        if (!(cond)) {                              \
            gt_set_error(__VA_ARGS__);              \
            return (code);                          \
        }                                           \
    } while (0)

#define GT_LAUNCH_CHECK()                                                           \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            gt_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,             \
                         hipGetErrorString(e__));                                   \
            return (int)e__;                                                        \
        }                                                                           \
    } while (0)

__host__ __device__ static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// ---- optional per-launch timing (HIP events on the launch stream), enabled by gt_prof_begin()
enum { GT_FAM_REC_GEMM = 0, GT_FAM_SGEMM = 1, GT_FAM_ATTN_E_FWD = 2, GT_FAM_ATTN_C_FWD = 3, GT_FAM_ATTN_C_BWD = 4,
       GT_FAM_ATTN_E_BWD = 5, GT_FAM_CELL_FWD = 6, GT_FAM_CELL_BWD = 7, GT_FAM_REC_CELL = 8, GT_FAM_PLSTM = 9, GT_FAM_COUNT = 10 };
extern bool g_gt_prof_on;
int gt_absmax_ring(const float* x, long rows, long cols, long ld, const uint64_t** word, hipStream_t st);     // gemm.hip
extern bool g_gt_capturing;          // loops.hip: stream capture in progress (GANTRON_EXPERIMENTAL=graph)
extern double g_gt_prof_ctx_rows;   // valid attention-memory rows of the current batch (0: unknown, charge B*Ti)
bool gt_prof_sample(int family);      // true for every `stride`-th launch of the family while profiling
void gt_prof_count_flops(int family, double flops);      // algorithmic flops of EVERY launch of the family while profiling
void gt_prof_mark(int family, double flops, double bytes, hipStream_t st, bool begin);
void gt_prof_exact(int family, double flops, double bytes, hipEvent_t* e0, hipEvent_t* e1);
// internal (not part of the C ABI): recurrent products with raised wave priority for the loops' critical chain
int gt_rec_gemm_prio(const float* X, int64_t ldx, const float* Wp, float* part, int32_t mtiles, int32_t N, int32_t K,
                     int32_t splits, int64_t part_sstride, int32_t nz, int64_t x_zstride, int64_t w_zstride,
                     int64_t part_zstride, int prio, void* stream);
// planes == 2 (two scaled fp16 planes, recurrent_bf16.hip): the activations' bound for THIS thread's next such launch -- a
// gt_absmax-format device word (NULL: none) and / or a floor known a priori; the launcher takes it and clears it
struct GtRecH2X { const uint64_t* xmax; float xfloor; };
extern thread_local GtRecH2X g_rec_h2x;
int gt_rec_gemm_h2_lean(const float* X, int64_t ldx, const void* Wp, float* part, int32_t mtiles, int32_t N, int32_t K,
                        int32_t splits, int64_t part_sstride, int32_t nz, int64_t x_zstride, int64_t part_zstride,
                        const uint64_t* xmax, float xfloor, int prio, void* stream);      // recurrent.hip
int gt_rec_gemm_bf16_prio(const float* X, int64_t ldx, const void* Wp, float* part, int32_t mtiles, int32_t N, int32_t K,
                          int32_t splits, int64_t part_sstride, int32_t nz, int64_t x_zstride, int64_t w_zstride,
                          int64_t part_zstride, int prio, void* stream, int planes = 1);

// Timing of a sampled launch.  Default: two events recorded on the stream around the launch (a bracket = kernel duration +
// event latency + whatever the dispatch waits for).  exact = true: nothing is recorded here; the launch itself carries the
// events (hipExtLaunchKernelGGL(..., prof.e0, prof.e1, 0, ...)) and they are stamped with the dispatch's own begin / end --
// the duration a profiler reports.  e0 / e1 are NULL when this launch is not sampled (a plain launch then).
struct GtProfScope {
    int fam; hipStream_t st; bool on; bool exact;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    GtProfScope(int family, double flops, double bytes, hipStream_t s, bool exact_ = false)
        : fam(family), st(s), on(g_gt_prof_on && gt_prof_sample(family)), exact(exact_) {
        if (g_gt_prof_on) gt_prof_count_flops(family, flops);
        if (!on) return;
        if (exact) gt_prof_exact(fam, flops, bytes, &e0, &e1);
        else gt_prof_mark(fam, flops, bytes, st, true);
    }
    ~GtProfScope() { if (on && !exact) gt_prof_mark(fam, 0, 0, st, false); }
};

// A launch that carries its own begin / end events when it is one of the profiler's samples (GtProfScope exact = true) and is a
// PLAIN launch otherwise.  (Round 3 launched every recurrent product through hipExtLaunchKernelGGL, events or not; under
// `rocprofv3 --pmc` that path segfaulted now and then inside the tool's copy of the dispatch's kernel arguments -- a read
// running off the end of a mapping, page-aligned fault address, only for the launches with the 288-byte RecCellEpi
// argument and only after several thousand dispatches; the events were NULL in the faulting launches.)
#define GT_LAUNCH_TIMED(kernel, grid, block, lds, st, e0, e1, ...)                                          \
    do {                                                                                                    \
        if ((e0) != nullptr) hipExtLaunchKernelGGL(kernel, grid, block, lds, st, e0, e1, 0, __VA_ARGS__);   \
        else hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                 \
    } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

__device__ __forceinline__ float sigmoidf_(float x) { return __fdividef(1.0f, 1.0f + __expf(-x)); }      // v_exp + v_rcp (1 ulp each)
// tanh as 1 - 2 / (e^{2x} + 1) on the exp / rcp units (|error| < 1e-6 absolute, exact limits at +-inf), used by the loop
// kernels: the attention energies take 4096 of these per workgroup and step, where the library tanhf (a few dozen
// instructions) was 1.2 us of the kernel's 18, the LSTM cells 8 per thread (the fused step 25.5 -> 24.9 us); forward and
// backward kernels use the same function (the backward recomputes tanh(c)).  PostNet / discriminator activations keep tanhf.
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - __fdividef(2.0f, __expf(2.0f * x) + 1.0f); }

// sum of `n` partial slabs (16 per trip) at element index idx (slab s lives at p + s*stride).  The loads are
// issued together (guarded full unroll): a runtime-bounded loop would expose one memory latency per slab.
__device__ __forceinline__ float slab_sum(const float* p, int n, int64_t stride, int64_t idx) {
    float acc = 0.f;
    for (int base = 0; base < n; base += 16) {        // one trip for n <= 16
        float v[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = (base + s < n) ? p[(int64_t)(base + s) * stride + idx] : 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc += v[s];
    }
    return acc;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` needs 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// Counter-based uniform in [0,1): a 64-bit mix (splitmix64 finaliser) of (seed, counter); the
// tests implement the identical integer recipe in numpy (tests/harness.py:u01).
__host__ __device__ __forceinline__ uint32_t gt_hash32(uint64_t seed, uint64_t ctr) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}
__host__ __device__ __forceinline__ float gt_u01(uint64_t seed, uint64_t ctr) {
    return (float)(gt_hash32(seed, ctr) >> 8) * (1.0f / 16777216.0f);
}

// Library-internal variants of gt_lstm_cell_fwd / gt_lstm_cell_bwd (recurrent.hip): nz = 2 runs two
// independent problems in one launch (the two BiLSTM directions, loops.hip); z holds the element offsets of
// the second problem's buffers: fwd {part, pre, c, h_prev, gates, h_raw, h0}, bwd {slabs, dh_direct, dc,
// gates, c, dgates}.
int gt_cell_fwd_z(GtSlabs part, const float* pre, int64_t ld_pre, const float* c_prev, const float* h_prev,
                  int64_t ld_hprev, const uint8_t* keep, float p_drop, const int32_t* active, float* gates, float* c_out,
                  float* h_raw, float* h0, int64_t ld0, float* h1, int64_t ld1, float* h2, int64_t ld2, int32_t B, int32_t H,
                  int nz, const int64_t* z, void* stream);
int gt_cell_bwd_z(GtSlabs dh_a, GtSlabs dh_b, GtSlabs dh_c, const float* dh_direct, int64_t ld_direct,
                  const float* dc_next, const float* gates, const float* c_new, const float* c_prev, const uint8_t* keep,
                  float p_drop, const int32_t* active, float* dgates, float* dc_prev, float* dh_pass, int32_t B, int32_t H,
                  int nz, const int64_t* z, void* stream);

// gt_rec_cell_fwd with packed-sequence semantics (active / h_prev, see lstm_cell_fwd_kernel) and, for ndir == 2, two
// independent problems in one launch (the BiLSTM directions); dz = element offsets of direction 1:
// {X, Wp, part, pre, c (prev and out), gates, h_raw, h0, h_prev}.  Used by gt_bilstm_run.  wbf16: Wp is the bf16 image of
// gt_pack_rec_cell_weight_bf16 (dz[1] then counts 16-bit elements).
int gt_rec_cell_fwd_z(const float* X, int64_t ldx, const void* Wp, int wbf16, float* part, uint32_t* counters, int32_t Bp, int32_t B,
                      int32_t H, int32_t K, int32_t splits, const float* pre, int64_t ld_pre, const float* c_prev,
                      const uint8_t* keep, float p_drop, float* gates, float* c_out, float* h_raw, float* h0, int64_t ld0,
                      float* h1, int64_t ld1, float* h2, int64_t ld2, const float* Wq, int64_t ldwq, float* qpart,
                      int64_t q_sstride, int32_t A, int32_t Apad, const int32_t* active, const float* h_prev, int64_t ld_hprev,
                      int32_t ndir, const int64_t* dz, int32_t prio, void* stream, int32_t k_own = 0, int32_t extra_slabs = 0);
