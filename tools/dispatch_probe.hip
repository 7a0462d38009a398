// How long does the chip take to START and RETIRE the workgroups of a launch that does nothing?  (dispatch begin -> end of an
// empty kernel per grid / block shape, from the launch's own start / stop stamps; and back-to-back throughput in one stream)
//   hipcc --offload-arch=gfx950 -O2 tools/dispatch_probe.hip -o tools/dispatch_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void nothing(int* p) { if (p != nullptr && threadIdx.x == 4096) *p = 1; }
__global__ void lds_user(int* p) {
    __shared__ int big[16384];        // 64 KB: one or two workgroups per CU
    big[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (p != nullptr && big[(threadIdx.x + 1) % blockDim.x] == -1) *p = 1;
}

template <typename K>
static void run(const char* name, K kern, int grid, int block) {
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float own = 0.f; const int reps = 200;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, (int*)nullptr);
    (void)hipStreamSynchronize(st);
    for (int i = 0; i < reps; ++i) {
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, a, b, 0, (int*)nullptr);
        (void)hipEventSynchronize(b);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, a, b); own += ms;
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, st, (int*)nullptr);
    (void)hipEventRecord(e1, st);
    (void)hipStreamSynchronize(st);
    float tot = 0.f; (void)hipEventElapsedTime(&tot, e0, e1);
    printf("%-10s grid %5d x block %4d (%6d waves): kernel %6.2f us, back-to-back %6.2f us per launch\n", name, grid, block,
           grid * block / 64, own * 1e3 / reps, tot * 1e3 / reps);
}

int main() {
    const int shapes[][2] = {{1, 64}, {64, 256}, {128, 256}, {256, 64}, {256, 256}, {256, 512}, {256, 1024}, {192, 512}, {384, 256},
                             {512, 256}, {1024, 256}, {2048, 256}, {64, 1024}, {128, 1024}};
    for (auto& s : shapes) run("empty", nothing, s[0], s[1]);
    for (auto& s : shapes) if (s[1] >= 256) run("64 KB LDS", lds_user, s[0], s[1]);
    return 0;
}
