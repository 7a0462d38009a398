// All-gather hand-off probe (development tool for the persistent decoder loop sketched in DESIGN.md section 7).
// NWG co-resident workgroups repeat: each writes its 1 KB slice of a 256 KB state vector, all wait until every slice of
// this step is published, every workgroup reads the WHOLE vector (as a decoder step would read h / ctx of all units).
// Three publication protocols, microseconds per step:
//   0  plain stores + __threadfence() (release) + counter, consumer: poll + __threadfence() (acquire) + plain loads
//   1  write-through stores (agent-scope relaxed atomic stores = sc1) + s_waitcnt + counter, consumer: poll +
//      acquire fence (buffer_inv sc1) + plain loads          (the documented hand-off recipe)
//   2  as 1, but the consumer reads with agent-scope relaxed atomic loads (sc1) and no fence at all
// Every spin is bounded: a lost arrival sets an error flag and all workgroups leave.
//   hipcc --offload-arch=gfx950 -O3 -o handoff_probe handoff_probe.hip && ./handoff_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

constexpr int THREADS = 256;
constexpr int SLICE = 256;                 // floats per workgroup (1 KB)

template <int MODE>
__global__ __launch_bounds__(THREADS) void probe(float* state /* 2 x nwg*SLICE */, int* ctr, int* err, int steps, int nwg,
                                                 float* sink, int* stale) {
    __shared__ float red[THREADS / 64];
    const int tid = threadIdx.x, wg = blockIdx.x;
    const int n = nwg * SLICE;
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        float* cur = state + (long)(s & 1) * n;                    // double buffered: step s+1 writes the other half
        const float v = (float)((s % 7) * 3 + (wg % 5));             // small integers: sums are exact in fp32
        if (MODE == 0) {
            cur[wg * SLICE + tid] = v;
        } else {
            __hip_atomic_store(&cur[wg * SLICE + tid], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (tid == 0) {
            if (MODE == 0) __threadfence();
            __hip_atomic_fetch_add(&ctr[s], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long spins = 0;
            while (__hip_atomic_load(&ctr[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000L) { atomicExch(err, 1); break; }
            }
            if (MODE == 0) __threadfence();
            if (MODE == 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        // read the whole state (nwg*SLICE floats) with all threads
        float sum = 0.f;
        if (MODE == 2) {
            for (int i = tid; i < n; i += THREADS) sum += __hip_atomic_load(&cur[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const float4* c4 = reinterpret_cast<const float4*>(cur);
            for (int i = tid; i < n / 4; i += THREADS) {
                const float4 t = c4[i];
                sum += t.x + t.y + t.z + t.w;
            }
        }
        acc += sum;
        // every slice of this step must be visible: the exact sum is known
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if ((tid & 63) == 0) red[tid >> 6] = sum;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int w = 0; w < THREADS / 64; ++w) tot += red[w];
            float expect = 0.f;
            for (int g = 0; g < nwg; ++g) expect += (float)SLICE * (float)((s % 7) * 3 + (g % 5));
            if (tot != expect) atomicAdd(stale, 1);
        }
        __syncthreads();
    }
    if (tid == 0) sink[wg] = acc;
}

template <int MODE>
static void run(int nwg, int steps, float* state, int* ctr, int* err, float* sink, int* stale) {
    hipMemset(ctr, 0, steps * sizeof(int));
    hipMemset(err, 0, sizeof(int));
    hipMemset(stale, 0, sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(THREADS), 0, 0, state, ctr, err, steps, nwg, sink, stale);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    int herr = 0;
    hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost);
        int hst = 0;
    hipMemcpy(&hst, stale, sizeof(int), hipMemcpyDeviceToHost);
    printf("mode %d  nwg %4d: %.2f us per step%s  stale reads in %d of %d (workgroup, step) pairs\n", MODE, nwg, ms * 1e3 / steps,
           herr ? "  (ABORTED: lost arrival)" : "", hst, nwg * steps);
}

int main() {
    const int steps = 1000, maxwg = 256;
    float *state, *sink;
    int *ctr, *err, *stale;
    hipMalloc(&state, 2L * maxwg * SLICE * sizeof(float));
    hipMalloc(&sink, maxwg * sizeof(float));
    hipMalloc(&ctr, steps * sizeof(int));
    hipMalloc(&err, sizeof(int));
    hipMalloc(&stale, sizeof(int));
    hipMemset(state, 0, 2L * maxwg * SLICE * sizeof(float));
    for (int nwg : {64, 128, 256}) {
        run<0>(nwg, steps, state, ctr, err, sink, stale);
        run<1>(nwg, steps, state, ctr, err, sink, stale);
        run<2>(nwg, steps, state, ctr, err, sink, stale);
    }
    return 0;
}
