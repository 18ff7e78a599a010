// Cost of a device-wide barrier between co-resident workgroups on MI355X (8 XCDs, non-coherent L2s):
// monotonically increasing arrival counter, agent-scope relaxed atomics, bounded spin (never hangs).
// Also: barrier + 4 KB sc1 payload per block written before / read after (what a fused stage chain would do).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000) { *err = 1; ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

__global__ void k_barrier(unsigned* ctr, int nbar, int* err, float* payload, int with_payload) {
    const int G = gridDim.x;
    float acc = 0.f;
    for (int i = 0; i < nbar; ++i) {
        if (with_payload) {   // each block writes 4 KB (sc1), then after the barrier reads its neighbour's 4 KB
            for (int e = threadIdx.x; e < 1024; e += blockDim.x)
                __hip_atomic_store(payload + (size_t)blockIdx.x * 1024 + e, (float)i + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!grid_barrier(ctr, (unsigned)(G * (i + 1)), err)) return;
        if (with_payload) {
            const int nb = (blockIdx.x + 1) % G;
            for (int e = threadIdx.x; e < 1024; e += blockDim.x)
                acc += __hip_atomic_load(payload + (size_t)nb * 1024 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (acc == 12345.678f) payload[0] = acc;
}

int main() {
    unsigned* ctr; int* err; float* payload;
    hipMalloc(&ctr, 4); hipMalloc(&err, 4); hipMalloc(&payload, 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int with_payload = 0; with_payload < 2; ++with_payload)
        for (int G : {8, 32, 64, 128, 256}) {
            const int nbar = 200;
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(ctr, 0, 4); hipMemset(err, 0, 4);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_barrier, dim3(G), dim3(256), 0, 0, ctr, nbar, err, payload, with_payload);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            int h_err = 0; hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost);
            printf("blocks %3d payload %d: %.2f us per barrier%s\n", G, with_payload, best * 1e3 / nbar, h_err ? "  (SPIN TIMEOUT)" : "");
        }
    return 0;
}
