// Error reporting and version of the C ABI.
#include <stdarg.h>
#include <stdlib.h>
#include "common.h"

static thread_local char g_err[512] = "";

void gt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gt_last_error_string(void) { return g_err; }

// `name[=int]` items separated by ',' or ';'; blanks around the name, the '=' and the value are ignored -- the same rules as
// gantron_amd/switches.py:_parse, which also rejects unknown names when the package is imported
int gt_env_item(const char* var, const char* key, int dflt) {
    const char* e = getenv(var);
    if (e == nullptr) return dflt;
    const size_t n = strlen(key);
    auto blank = [](char c) { return c == ' ' || c == '\t'; };
    for (const char* p = e; *p;) {
        while (*p == ',' || *p == ';' || blank(*p)) ++p;
        const char* q = p;
        while (*q && *q != ',' && *q != ';') ++q;
        const char* ne = p;                                      // end of the name
        while (ne < q && *ne != '=' && !blank(*ne)) ++ne;
        if ((size_t)(ne - p) == n && strncmp(p, key, n) == 0) {
            const char* r = ne;
            while (r < q && blank(*r)) ++r;
            if (r == q) return 1;
            if (*r == '=') {
                ++r;
                while (r < q && blank(*r)) ++r;
                return atoi(r);
            }
        }
        p = q;
    }
    return dflt;
}
extern "C" int gt_version(void) { return GT_ABI_VERSION; }

// sizeof of every struct that crosses the ABI, so a binding can verify its layout at load time
extern "C" int64_t gt_abi_sizeof(int32_t which) {
    switch (which) {
        case 0: return sizeof(GtOperand);
        case 1: return sizeof(GtGemm);
        case 2: return sizeof(GtSlabs);
        case 3: return sizeof(GtAttnDims);
        case 4: return sizeof(GtDecoderFwd);
        case 5: return sizeof(GtDecoderBwd);
        case 6: return sizeof(GtBiLstm);
        case 7: return sizeof(GtDecoderInfer);
        case 8: return sizeof(GtPlstm);
        case 9: return sizeof(GtPlstmBwd);
        case 10: return sizeof(GtRecTail);
        case 11: return sizeof(GtCellBwdTail);
        default: return -1;
    }
}

// ------------------------------------------------------------------ per-launch event timing
#include <vector>
#include <hip/hip_ext.h>
bool g_gt_prof_on = false;
double g_gt_prof_ctx_rows = 0.0;      // valid memory rows of the batch (gt_prof_set_context_rows), 0 = charge B*Ti
namespace {
struct ProfRec { int fam; double flops, bytes; hipEvent_t e0, e1; bool exact; };
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;
hipEvent_t next_event() {
    if (g_pool_next == g_pool.size()) {
        hipEvent_t e;
        hipEventCreate(&e);
        g_pool.push_back(e);
    }
    return g_pool[g_pool_next++];
}
}  // namespace

__global__ void gt_null_kernel() {}
static int g_prof_null_every = 0;
static long g_prof_null_ctr = 0;

void gt_prof_mark(int family, double flops, double bytes, hipStream_t st, bool begin) {
    if (begin) {
        ProfRec r;
        r.fam = family; r.flops = flops; r.bytes = bytes; r.exact = false;
        r.e0 = next_event(); r.e1 = next_event();
        hipEventRecord(r.e0, st);
        g_recs.push_back(r);
    } else if (!g_recs.empty()) {
        hipEventRecord(g_recs.back().e1, st);
        if (g_prof_null_every > 0 && (++g_prof_null_ctr % g_prof_null_every) == 0) {
            // the same bracket around an EMPTY kernel, in the same place of the same busy stream: what the bracket
            // itself costs in situ
            ProfRec r;
            r.fam = GT_FAM_COUNT; r.flops = 0; r.bytes = 0; r.exact = false;
            r.e0 = next_event(); r.e1 = next_event();
            hipEventRecord(r.e0, st);
            hipLaunchKernelGGL(gt_null_kernel, dim3(256), dim3(256), 0, st);
            hipEventRecord(r.e1, st);
            g_recs.push_back(r);
        }
    }
}

void gt_prof_exact(int family, double flops, double bytes, hipEvent_t* e0, hipEvent_t* e1) {
    ProfRec r;
    r.fam = family; r.flops = flops; r.bytes = bytes; r.exact = true;
    r.e0 = next_event(); r.e1 = next_event();
    g_recs.push_back(r);
    *e0 = r.e0; *e1 = r.e1;
}

static int g_prof_stride = 1;
static long g_prof_count[GT_FAM_COUNT];
static double g_prof_flops_all[GT_FAM_COUNT];
void gt_prof_count_flops(int family, double flops) { g_prof_flops_all[family] += flops; }

bool g_gt_capturing = false;          // a decoder loop is being captured into a hipGraph: no timing brackets inside it

bool gt_prof_sample(int family) {
    if (g_gt_capturing) { g_prof_count[family]++; return false; }
    const int stride = family == GT_FAM_SGEMM ? (g_prof_stride >= 16 ? g_prof_stride / 16 : 1) : g_prof_stride;
    return (g_prof_count[family]++ % stride) == 0;
}

extern "C" int gt_prof_set_context_rows(int64_t valid_rows) {
    GT_REQUIRE(valid_rows >= 0, GT_E_ARG, "gt_prof_set_context_rows: negative");
    g_gt_prof_ctx_rows = (double)valid_rows;
    return 0;
}

extern "C" int gt_prof_begin(void) { return gt_prof_begin_sampled(1); }

extern "C" int gt_prof_begin_sampled(int32_t stride) {
    GT_REQUIRE(stride >= 1, GT_E_ARG, "gt_prof_begin_sampled: stride");
    g_recs.clear();
    g_pool_next = 0;
    g_prof_stride = stride;
    g_prof_null_every = stride > 1 ? 4 : 0;        // every 4th bracketed launch is followed by a bracketed empty kernel
    g_prof_null_ctr = 0;
    for (int i = 0; i < GT_FAM_COUNT; ++i) { g_prof_count[i] = 0; g_prof_flops_all[i] = 0.0; }
    g_gt_prof_on = true;
    return 0;
}

// Stops recording, waits for the recorded events and fills per-family totals:
// out[f*4+0] = bracketed launches, +1 = total ms, +2 = total flops, +3 = total bytes  (f < GT_FAM_COUNT);
// out[4*GT_FAM_COUNT + f] = all launches of the family seen while profiling; then the empty-kernel brackets (count, ms);
// out[5*GT_FAM_COUNT + 2 + f] = how many of the family's timed launches carry their own dispatch stamps (no event latency);
// out[6*GT_FAM_COUNT + 2 + f] = algorithmic flops of ALL launches of the family (bracketed or not)
extern "C" int gt_prof_end(double* out) {
    g_gt_prof_on = false;
    GT_REQUIRE(out != nullptr, GT_E_ARG, "gt_prof_end: NULL");
    for (int i = 0; i < GT_FAM_COUNT * 4; ++i) out[i] = 0.0;
    for (int i = 0; i < GT_FAM_COUNT; ++i) out[GT_FAM_COUNT * 4 + i] = (double)g_prof_count[i];
    out[GT_FAM_COUNT * 5] = out[GT_FAM_COUNT * 5 + 1] = 0.0;       // in-situ empty-kernel brackets: count, total ms
    for (int i = 0; i < GT_FAM_COUNT; ++i) out[GT_FAM_COUNT * 5 + 2 + i] = 0.0;     // launches timed by their own dispatch stamps
    for (int i = 0; i < GT_FAM_COUNT; ++i) out[GT_FAM_COUNT * 6 + 2 + i] = g_prof_flops_all[i];       // flops of ALL launches
    for (auto& r : g_recs) {
        hipError_t e = hipEventSynchronize(r.e1);
        GT_REQUIRE(e == hipSuccess, (int)e, "gt_prof_end: event sync failed");
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.e0, r.e1);
        GT_REQUIRE(e == hipSuccess, (int)e, "gt_prof_end: elapsed failed");
        if (r.fam == GT_FAM_COUNT) {
            out[GT_FAM_COUNT * 5] += 1.0;
            out[GT_FAM_COUNT * 5 + 1] += ms;
            continue;
        }
        out[r.fam * 4 + 0] += 1.0;
        out[r.fam * 4 + 1] += ms;
        if (r.exact) out[GT_FAM_COUNT * 5 + 2 + r.fam] += 1.0;
        out[r.fam * 4 + 2] += r.flops;
        out[r.fam * 4 + 3] += r.bytes;
    }
    g_recs.clear();
    return 0;
}

// An event bracket around a kernel measures that kernel's duration (dispatch begin to end, what a profiler reports)
// PLUS the latency of the two event packets.  bench.py brackets empty kernels in situ (gt_prof_end) and needs the
// empty kernel's own duration to separate the two: measured here on the idle stream with the start / stop events of
// hipExtLaunchKernelGGL, which carry the dispatch's own begin / end timestamps (rocprofv3 shows the same ~3.6 us).
extern "C" int gt_prof_calibrate(int32_t n, double* out_us, void* stream) {
    GT_REQUIRE(n > 0 && out_us != nullptr, GT_E_ARG, "gt_prof_calibrate: bad argument");
    hipStream_t st = (hipStream_t)stream;
    std::vector<hipEvent_t> ev(2 * (size_t)n);
    for (auto& e : ev) (void)hipEventCreate(&e);
    for (int i = 0; i < n; ++i) {
        hipExtLaunchKernelGGL(gt_null_kernel, dim3(256), dim3(256), 0, st, ev[2 * i], ev[2 * i + 1], 0);
        (void)hipStreamSynchronize(st);           // idle stream: the start stamp is not delayed by earlier work
    }
    double tot = 0.0;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
        tot += ms;
    }
    for (auto& x : ev) (void)hipEventDestroy(x);
    *out_us = tot * 1e3 / n;
    return 0;
}
