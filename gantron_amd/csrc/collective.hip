// C1 of the hot path behind the C ABI: the data-parallel gradient average (reference distributed.py:137-161: one flat
// all-reduce(SUM) per backward, divided by the world size) for a host that binds this library without PyTorch.  RCCL is not
// linked: its entry points are resolved at the first call from the RCCL the process already has (the one the caller made
// its communicator with), never from a second copy.  The Python package itself keeps using torch.distributed (backend "nccl" = RCCL).
#include <dlfcn.h>
#include <mutex>
#include "common.h"

namespace {

typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_count_fn)(void*, int*);
typedef const char* (*nccl_errstr_fn)(int);
typedef int (*nccl_version_fn)(int*);

struct Rccl {
    nccl_allreduce_fn all_reduce = nullptr;
    nccl_count_fn count = nullptr;
    nccl_errstr_fn errstr = nullptr;
    int version = 0;
    const char* why = "librccl is not loaded in this process";
};
Rccl g_rccl;
std::once_flag g_rccl_once;

// The communicator is the CALLER's: its RCCL is the only one whose ncclAllReduce may be handed that pointer.  So the entry points
// are taken from what the process has ALREADY loaded -- the global symbol scope first (a statically linked RCCL, or one loaded
// RTLD_GLOBAL), then librccl.so.1 / librccl.so with RTLD_NOLOAD -- and never from a copy loaded here by name: a second
// library instance would be handed a communicator it did not create.  A host whose RCCL is not visible that way (loaded
// RTLD_LOCAL under another name, e.g. a framework's bundled copy) passes the two functions itself: gt_allreduce_avg_with.
void rccl_resolve() {
    void* scopes[3] = {RTLD_DEFAULT, dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD),
                       dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD)};
    for (int i = 0; i < 3; ++i) {
        void* h = scopes[i];
        if (i > 0 && h == nullptr) continue;                    // (RTLD_DEFAULT itself is a null handle)
        nccl_allreduce_fn ar = (nccl_allreduce_fn)dlsym(h, "ncclAllReduce");
        nccl_count_fn cnt = (nccl_count_fn)dlsym(h, "ncclCommCount");
        if (ar == nullptr || cnt == nullptr) continue;
        nccl_version_fn ver = (nccl_version_fn)dlsym(h, "ncclGetVersion");
        int v = 0;
        if (ver == nullptr || ver(&v) != 0 || v < 2000) {       // ncclFloat32 = 7 / ncclSum = 0 are the NCCL 2.x API's values
            g_rccl.why = "the RCCL in this process does not report an NCCL 2.x API version (ncclGetVersion)";
            continue;
        }
        g_rccl.all_reduce = ar; g_rccl.count = cnt; g_rccl.version = v;
        g_rccl.errstr = (nccl_errstr_fn)dlsym(h, "ncclGetErrorString");
        return;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(float* p, float s, long n4, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float4 v = reinterpret_cast<float4*>(p)[i];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        reinterpret_cast<float4*>(p)[i] = v;
    }
    if (i == 0) for (long k = 4 * n4; k < n; ++k) p[k] *= s;
}

}  // namespace

static int allreduce_avg(void* comm, float* buf, int64_t n, nccl_allreduce_fn all_reduce, nccl_count_fn count, nccl_errstr_fn errstr,
                         void* stream) {
    int world = 0;
    int rc = count(comm, &world);
    GT_REQUIRE(rc == 0 && world >= 1, rc != 0 ? rc : GT_E_ARG, "gt_allreduce_avg: ncclCommCount failed: %s", errstr ? errstr(rc) : "?");
    hipStream_t st = (hipStream_t)stream;
    rc = all_reduce(buf, buf, (size_t)n, /* ncclFloat32 */ 7, /* ncclSum */ 0, comm, st);
    GT_REQUIRE(rc == 0, rc, "gt_allreduce_avg: ncclAllReduce failed: %s", errstr ? errstr(rc) : "?");
    if (world > 1) {        // sum, then divide: the reference's order (distributed.py:158-161)
        const long n4 = n / 4;
        hipLaunchKernelGGL(scale_kernel, dim3((unsigned)cdiv64(std::max<long>(n4, 1), 256)), dim3(256), 0, st, buf, 1.0f / (float)world, n4, (long)n);
        GT_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int gt_allreduce_avg(void* comm, float* buf, int64_t n, void* stream) {
    GT_REQUIRE(comm != nullptr && buf != nullptr && n > 0, GT_E_ARG, "gt_allreduce_avg: bad argument");
    GT_REQUIRE(aligned16(buf), GT_E_ALIGN, "gt_allreduce_avg: the buffer must be 16-byte aligned");
    std::call_once(g_rccl_once, rccl_resolve);
    GT_REQUIRE(g_rccl.all_reduce != nullptr && g_rccl.count != nullptr, GT_E_ARG,
               "gt_allreduce_avg: %s -- the communicator's own RCCL must be visible (global symbols or librccl.so[.1] already "
               "loaded); otherwise pass its ncclAllReduce / ncclCommCount through gt_allreduce_avg_with", g_rccl.why);
    return allreduce_avg(comm, buf, n, g_rccl.all_reduce, g_rccl.count, g_rccl.errstr, stream);
}

// the same with the caller's own ncclAllReduce and ncclCommCount (the library that created `comm`)
extern "C" int gt_allreduce_avg_with(void* comm, float* buf, int64_t n, void* nccl_all_reduce, void* nccl_comm_count, void* stream) {
    GT_REQUIRE(comm != nullptr && buf != nullptr && n > 0 && nccl_all_reduce != nullptr && nccl_comm_count != nullptr, GT_E_ARG,
               "gt_allreduce_avg_with: bad argument");
    GT_REQUIRE(aligned16(buf), GT_E_ALIGN, "gt_allreduce_avg_with: the buffer must be 16-byte aligned");
    return allreduce_avg(comm, buf, n, (nccl_allreduce_fn)nccl_all_reduce, (nccl_count_fn)nccl_comm_count, nullptr, stream);
}
