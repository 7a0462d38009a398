// C1 of the hot path behind the C ABI: the data-parallel gradient average (reference distributed.py:137-161: one flat
// all-reduce(SUM) per backward, divided by the world size) for a host that binds this library without PyTorch.  RCCL is not
// linked: its entry points are resolved at the first call from the librccl the process already has (the one the caller made
// its communicator with), or loaded by name.  The Python package itself keeps using torch.distributed (backend "nccl" = RCCL).
#include <dlfcn.h>
#include <mutex>
#include "common.h"

namespace {

typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_count_fn)(void*, int*);
typedef const char* (*nccl_errstr_fn)(int);

struct Rccl {
    void* handle = nullptr;
    nccl_allreduce_fn all_reduce = nullptr;
    nccl_count_fn count = nullptr;
    nccl_errstr_fn errstr = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_resolve() {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (int pass = 0; pass < 2 && g_rccl.handle == nullptr; ++pass)          // first the copy that is already loaded
        for (const char* n : names) {
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (g_rccl.handle != nullptr) break;
        }
    if (g_rccl.handle == nullptr) return;
    g_rccl.all_reduce = (nccl_allreduce_fn)dlsym(g_rccl.handle, "ncclAllReduce");
    g_rccl.count = (nccl_count_fn)dlsym(g_rccl.handle, "ncclCommCount");
    g_rccl.errstr = (nccl_errstr_fn)dlsym(g_rccl.handle, "ncclGetErrorString");
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

extern "C" int gt_allreduce_avg(void* comm, float* buf, int64_t n, void* stream) {
    GT_REQUIRE(comm != nullptr && buf != nullptr && n > 0, GT_E_ARG, "gt_allreduce_avg: bad argument");
    GT_REQUIRE(aligned16(buf), GT_E_ALIGN, "gt_allreduce_avg: the buffer must be 16-byte aligned");
    std::call_once(g_rccl_once, rccl_resolve);
    GT_REQUIRE(g_rccl.all_reduce != nullptr && g_rccl.count != nullptr, GT_E_ARG,
               "gt_allreduce_avg: no RCCL in this process (librccl.so.1 / librccl.so could not be loaded)");
    int world = 0;
    int rc = g_rccl.count(comm, &world);
    GT_REQUIRE(rc == 0 && world >= 1, rc != 0 ? rc : GT_E_ARG, "gt_allreduce_avg: ncclCommCount failed: %s",
               g_rccl.errstr ? g_rccl.errstr(rc) : "?");
    hipStream_t st = (hipStream_t)stream;
    rc = g_rccl.all_reduce(buf, buf, (size_t)n, /* ncclFloat32 */ 7, /* ncclSum */ 0, comm, st);
    GT_REQUIRE(rc == 0, rc, "gt_allreduce_avg: ncclAllReduce failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
    if (world > 1) {        // sum, then divide: the reference's order (distributed.py:158-161)
        const long n4 = n / 4;
        hipLaunchKernelGGL(scale_kernel, dim3((unsigned)cdiv64(std::max<long>(n4, 1), 256)), dim3(256), 0, st, buf, 1.0f / (float)world, n4, (long)n);
        GT_LAUNCH_CHECK();
    }
    return 0;
}
