// Which (XCC, SE, CU) do the workgroups of a stream created with hipExtStreamCreateWithCUMask land on?
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/cumask_probe
//   tools/cumask_probe <hex mask words, low word first> ...      e.g.  ffffffff 0 0 0 0 0 0 0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void where(unsigned* out) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    // stay a little, so that the grid spreads over every CU the mask allows
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) {}
}

int main(int argc, char** argv) {
    std::vector<uint32_t> mask;
    for (int i = 1; i < argc; ++i) mask.push_back((uint32_t)strtoul(argv[i], nullptr, 16));
    hipStream_t st;
    if (mask.empty()) { if (hipStreamCreate(&st) != hipSuccess) return 1; }
    else {
        hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); return 1; }
    }
    const int n = 4096;
    unsigned* d; hipMalloc(&d, 8 * n);
    hipLaunchKernelGGL(where, dim3(n), dim3(256), 0, st, d);
    if (hipStreamSynchronize(st) != hipSuccess) { printf("sync failed\n"); return 1; }
    std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, 8 * n, hipMemcpyDeviceToHost);
    std::map<unsigned, std::map<unsigned, int>> per_xcc;       // xcc -> (se, sh, cu) -> count
    for (int i = 0; i < n; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc][(se << 8) | (sh << 4) | cu]++;
    }
    int total = 0;
    for (auto& x : per_xcc) {
        printf("xcc %u: %zu CUs:", x.first, x.second.size());
        for (auto& c : x.second) printf(" %u.%u.%u", c.first >> 8, (c.first >> 4) & 1, c.first & 0xf);
        printf("\n");
        total += (int)x.second.size();
    }
    printf("total distinct CUs: %d\n", total);
    return 0;
}
