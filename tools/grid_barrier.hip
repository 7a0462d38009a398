// Grid-barrier latency probe (development tool for the persistent decoder kernel sketched in DESIGN.md section 7):
// how long does one all-workgroup barrier through a device-scope atomic counter take on this chip (8 XCDs), for
// nwg co-resident workgroups?  Every spin is bounded: a lost arrival sets an error flag and the kernel exits.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
// Measured on MI355X (us per barrier, 256 workgroups): flat counter 14.8, per-XCD counters 10.3, flag slots 13.7;
// with the __threadfence() calls removed: 3.9 / 5.1 / 6.4 -- the fences dominate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void barrier_kernel(int* ctr, int* err, int rounds, int nwg, float* sink, int payload) {
    __shared__ float s;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        // some work and a store that the other workgroups read after the barrier (release / acquire path)
        for (int i = 0; i < payload; ++i) acc += __sinf(acc + i);
        if (threadIdx.x == 0) sink[blockIdx.x] = acc + r;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();                                   // release: publish this workgroup's stores
            atomicAdd(&ctr[r], 1);
            long spins = 0;
            while (__hip_atomic_load(&ctr[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 20000000L) { atomicExch(err, 1); break; }
            }
            __threadfence();                                   // acquire
            s = sink[(blockIdx.x + 1) % nwg];                  // read a neighbour's store
        }
        __syncthreads();
        acc += s * 1e-9f;
        if (*err) return;
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

// Hierarchical variant: workgroups arrive at a per-XCD counter (blockIdx.x & 7: round-robin placement), the last
// arrival of each XCD arrives at the global counter, waits for all 8 and then releases its XCD through a per-XCD flag
// that the other workgroups of the XCD poll.
__global__ void barrier2_kernel(int* xc, int* gc, int* xf, int* err, int rounds, int nwg, float* sink) {
    __shared__ float s;
    float acc = 0.f;
    const int x = blockIdx.x & 7, per = nwg / 8;
    for (int r = 0; r < rounds; ++r) {
        if (threadIdx.x == 0) sink[blockIdx.x] = acc + r;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const int prev = atomicAdd(&xc[r * 8 + x], 1);
            long spins = 0;
            if (prev == per - 1) {
                atomicAdd(&gc[r], 1);
                while (__hip_atomic_load(&gc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 20000000L) { atomicExch(err, 1); break; }
                }
                __hip_atomic_store(&xf[r * 8 + x], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(&xf[r * 8 + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > 20000000L) { atomicExch(err, 1); break; }
                }
            }
            __threadfence();
            s = sink[(blockIdx.x + 1) % nwg];
        }
        __syncthreads();
        acc += s * 1e-9f;
        if (*err) return;
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

// Flag-array variant (no contended atomics): every workgroup publishes its epoch in its own slot; the first
// workgroup of each XCD (blockIdx.x < 8) gathers its XCD's slots with one wave-wide load per poll, publishes the
// XCD's epoch, gathers the 8 XCD epochs the same way and releases its XCD through a flag the others poll.
__global__ void barrier3_kernel(int* flags, int* gflags, int* rel, int* err, int rounds, int nwg, float* sink) {
    __shared__ float s;
    float acc = 0.f;
    const int x = blockIdx.x & 7, per = nwg / 8, lane = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        const int e = r + 1;
        if (threadIdx.x == 0) sink[blockIdx.x] = acc + r;
        __syncthreads();
        if (threadIdx.x < 64) {                                   // wave 0
            long spins = 0;
            if (lane == 0) {
                __threadfence();
                __hip_atomic_store(&flags[blockIdx.x], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (blockIdx.x < 8) {
                for (;;) {                                         // gather this XCD's arrivals (per <= 64)
                    const int v = lane < per ? __hip_atomic_load(&flags[x + 8 * lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : e;
                    if (__all(v >= e)) break;
                    if (++spins > 20000000L) { if (lane == 0) atomicExch(err, 1); break; }
                }
                if (lane == 0) __hip_atomic_store(&gflags[x], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                for (;;) {                                         // gather the 8 XCDs
                    const int v = lane < 8 ? __hip_atomic_load(&gflags[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : e;
                    if (__all(v >= e)) break;
                    if (++spins > 20000000L) { if (lane == 0) atomicExch(err, 1); break; }
                }
                if (lane == 0) __hip_atomic_store(&rel[x], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else if (lane == 0) {
                while (__hip_atomic_load(&rel[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < e) {
                    if (++spins > 20000000L) { atomicExch(err, 1); break; }
                }
            }
            if (lane == 0) {
                __threadfence();
                s = sink[(blockIdx.x + 1) % nwg];
            }
        }
        __syncthreads();
        acc += s * 1e-9f;
        if (*err) return;
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

int main() {
    const int rounds = 2000;
    int *ctr, *err;
    float* sink;
    hipMalloc(&ctr, rounds * sizeof(int));
    hipMalloc(&err, sizeof(int));
    hipMalloc(&sink, 4096 * sizeof(float));
    for (int nwg : {32, 64, 128, 256, 512}) {
        for (int threads : {64, 512}) {
            hipMemset(ctr, 0, rounds * sizeof(int));
            hipMemset(err, 0, sizeof(int));
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(barrier_kernel, dim3(nwg), dim3(threads), 0, 0, ctr, err, rounds, nwg, sink, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            int herr = 0;
            hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
            printf("nwg %4d x %3d threads: %.2f us per barrier%s\n", nwg, threads, ms * 1e3 / rounds, herr ? "  (ABORTED: lost arrival)" : "");
        }
    }
    int *xc, *gc, *xf;
    hipMalloc(&xc, rounds * 8 * sizeof(int));
    hipMalloc(&gc, rounds * sizeof(int));
    hipMalloc(&xf, rounds * 8 * sizeof(int));
    for (int nwg : {64, 128, 256, 512}) {
        hipMemset(xc, 0, rounds * 8 * sizeof(int));
        hipMemset(gc, 0, rounds * sizeof(int));
        hipMemset(xf, 0, rounds * 8 * sizeof(int));
        hipMemset(err, 0, sizeof(int));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(barrier2_kernel, dim3(nwg), dim3(64), 0, 0, xc, gc, xf, err, rounds, nwg, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        int herr = 0;
        hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
        printf("hierarchical, nwg %4d: %.2f us per barrier%s\n", nwg, ms * 1e3 / rounds, herr ? "  (ABORTED: lost arrival)" : "");
    }
    int *flags, *gflags, *rel;
    hipMalloc(&flags, 512 * sizeof(int));
    hipMalloc(&gflags, 8 * sizeof(int));
    hipMalloc(&rel, 8 * sizeof(int));
    for (int nwg : {64, 128, 256, 512}) {
        hipMemset(flags, 0, 512 * sizeof(int));
        hipMemset(gflags, 0, 8 * sizeof(int));
        hipMemset(rel, 0, 8 * sizeof(int));
        hipMemset(err, 0, sizeof(int));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(barrier3_kernel, dim3(nwg), dim3(64), 0, 0, flags, gflags, rel, err, rounds, nwg, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        int herr = 0;
        hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
        printf("flag array,   nwg %4d: %.2f us per barrier%s\n", nwg, ms * 1e3 / rounds, herr ? "  (ABORTED: lost arrival)" : "");
    }
    return 0;
}
