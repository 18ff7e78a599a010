// poison.hip -- a foreign load for the multi-queue probes: a kernel that fills its workgroup's 64 KiB of LDS with a pattern
// (NaNs by default) and spins for a while, launched back to back on its own stream.  If a fit's results change while ONLY this
// runs next to it, some kernel of the library reads LDS it did not write (what an earlier workgroup on that CU left behind);
// if they do not, uninitialised LDS is not what makes concurrent fits nondeterministic.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/poison.hip -o tools/probes/libpoison.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void poison_kernel(unsigned pattern, int spins, unsigned* sink) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = pattern ^ (unsigned)(i * 2654435761u & 0xFFFF);
    __syncthreads();
    unsigned acc = 0;
    for (int s = 0; s < spins; ++s) acc += lds[(threadIdx.x * 33 + s * 7) & 16383];
    if (acc == 0x12345678u) sink[0] = acc;          // keeps the loop alive
}

extern "C" int poison_launch(void* stream, unsigned pattern, int blocks, int spins, unsigned* sink) {
    static bool init = false;
    if (!init) {
        if (hipFuncSetAttribute((const void*)poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess) return 1;
        init = true;
    }
    hipLaunchKernelGGL(poison_kernel, dim3(blocks), dim3(256), 65536, (hipStream_t)stream, pattern, spins, sink);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
