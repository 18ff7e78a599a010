// Same-XCD variant of grid_barrier.hip: launch 8*G workgroups, only those the hardware placed on XCD 0 take part
// (workgroups are dealt round-robin over the 8 XCDs; the XCC_ID hardware register is checked, not assumed).
// Participants share one L2, so payloads can use plain stores + loads that bypass only the per-CU L1 (sc0), and the
// barrier counter can be a workgroup... no: agent-scope atomic executed in that L2.
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}

__device__ __forceinline__ bool barrier(unsigned* ctr, unsigned target, int* err) {
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

// mode 0: barrier only; 1: 4 KB payload with sc1 stores / sc1 loads (cross-XCD-safe); 2: plain stores + glc (sc0) loads
__global__ void k(unsigned* ctr, int nbar, int* err, float* payload, int mode, int G, unsigned* xcc_seen) {
    if ((blockIdx.x & 7) != 0) return;
    const int me = blockIdx.x >> 3;
    if (threadIdx.x == 0) xcc_seen[me] = xcc_id();
    float acc = 0.f;
    for (int i = 0; i < nbar; ++i) {
        if (mode == 1) {
            for (int e = threadIdx.x; e < 1024; e += blockDim.x)
                __hip_atomic_store(payload + (size_t)me * 1024 + e, (float)i + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (mode == 2) {
            for (int e = threadIdx.x; e < 1024; e += blockDim.x) payload[(size_t)me * 1024 + e] = (float)i + e;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!barrier(ctr, (unsigned)(G * (i + 1)), err)) return;
        const int nb = (me + 1) % G;
        if (mode == 1) {
            for (int e = threadIdx.x; e < 1024; e += blockDim.x)
                acc += __hip_atomic_load(payload + (size_t)nb * 1024 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (mode == 2) {
            for (int e = threadIdx.x; e < 1024; e += blockDim.x)
                acc += __hip_atomic_load(payload + (size_t)nb * 1024 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (mode == 2 && i == nbar - 1 && acc == -1.f) payload[0] = acc;
    }
    if (acc == 12345.678f) payload[0] = acc;
}

int main() {
    unsigned *ctr, *seen; int* err; float* payload;
    (void)hipMalloc(&ctr, 4); (void)hipMalloc(&err, 4); (void)hipMalloc(&payload, 64 * 4096); (void)hipMalloc(&seen, 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int G : {8, 16, 32}) {
            const int nbar = 200;
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipMemset(ctr, 0, 4); (void)hipMemset(err, 0, 4);
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(8 * G), dim3(256), 0, 0, ctr, nbar, err, payload, mode, G, seen);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            int h_err = 0; unsigned h_seen[64];
            (void)hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h_seen, seen, G * 4, hipMemcpyDeviceToHost);
            int same = 1; for (int i = 1; i < G; ++i) same &= (h_seen[i] == h_seen[0]);
            printf("participants %2d mode %d: %.2f us per stage  (all on XCC %u: %s)%s\n", G, mode, best * 1e3 / nbar, h_seen[0],
                   same ? "yes" : "NO", h_err ? "  SPIN TIMEOUT" : "");
        }
    return 0;
}
