// bpermute_repro.hip -- library-free reproducer attempt for the finding of round 3: kernels that reduce
// across lanes with ds_bpermute_b32 (what __shfl_xor compiles to on gfx950) return wrong sums while kernels of ANOTHER
// hardware queue keep the same CUs' LDS path busy.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/bpermute_repro.hip -o tools/probes/bpermute_repro
//   tools/probes/bpermute_repro <aggressor> <seconds>
//
// victim     (stream 0): every wave sums 64 lane values (exact small integers as floats) with the xor butterfly of __shfl_xor, a few
//            thousand times per launch, and counts every result that is not the exact sum; a second victim kernel does the same
//            butterfly with DPP row operations + v_readlane (no LDS-unit instruction) as the control.
// aggressor  (stream 1), back to back:  0 = none   1 = LDS-DMA ring (global_load_lds into 64 KiB, like the plane GEMM)
//            2 = plain LDS traffic (ds_write / ds_read over 64 KiB)   3 = streaming global loads / stores only
// Exit code 1 when a victim saw a wrong sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e__ = (x);                                                                          \
        if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(2); } \
    } while (0)

typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(1))) const void* glb_vp;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float bcast(float v, int lane) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane)); }

// MODE 0: __shfl_xor butterfly (ds_bpermute_b32)   MODE 1: DPP + readlane
template <int MODE>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ in, int iters, unsigned* __restrict__ bad, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        // lane value: small exact integers, different per wave and iteration; loaded so that the wave also has memory traffic
        const float x = in[(gw * 64 + lane + it * 4096) & 0xFFFFF];
        float expect = 0.f;                       // what the sum must be: every lane recomputes it from the same table row
        float v = x;
        if (MODE == 0) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        } else {
            v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
            v = (bcast(v, 0) + bcast(v, 16)) + (bcast(v, 32) + bcast(v, 48));
        }
        // the exact sum, lane by lane through readlane (VALU / SALU only, never the LDS unit)
#pragma unroll 8
        for (int l = 0; l < 64; ++l) expect += bcast(x, l);
        if (v != expect) atomicAdd(bad, 1u);
        acc += v;
    }
    if (acc == -1.f) sink[0] = acc;
}

template <int KIND>
__global__ __launch_bounds__(512) void aggressor(const unsigned short* __restrict__ src, float* __restrict__ out, int steps) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        if (KIND == 1) {         // 32 KiB by LDS-DMA per step: 4 pieces of 1 KiB per wave
            for (int p = 0; p < 4; ++p) {
                const unsigned short* g = src + ((((long)blockIdx.x * 37 + s) * 16384 + (p * 8 + wave) * 512 + lane * 8) & 0xFFFFF8);
                __builtin_amdgcn_global_load_lds((glb_vp)g, (lds_vp)(smem + (s & 1) * 16384 + (p * 8 + wave) * 512), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            acc += (float)smem[(s & 1) * 16384 + ((tid * 33) & 16383)];
        } else if (KIND == 2) {  // plain LDS stores and loads
            for (int p = 0; p < 8; ++p) smem[(p * 512 + tid) * 8 & 32767] = (unsigned short)(s + p);
            __syncthreads();
            for (int p = 0; p < 8; ++p) acc += (float)smem[((p * 512 + tid) * 8 + 3) & 32767];
            __syncthreads();
        } else {                 // streaming global traffic
            const uint4 v = *reinterpret_cast<const uint4*>(src + ((((long)blockIdx.x * 37 + s) * 4096 + tid * 8) & 0xFFFFF8));
            acc += (float)v.x;
        }
    }
    if (acc == -1.f) out[0] = acc;
}

int main(int argc, char** argv) {
    const int kind = argc > 1 ? atoi(argv[1]) : 1;
    const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    float *in, *sink;
    unsigned short* src;
    unsigned *bad0, *bad1;
    CK(hipMalloc(&in, 4 << 20)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&src, 2 << 20));
    CK(hipMalloc(&bad0, 4)); CK(hipMalloc(&bad1, 4));
    float* h = (float*)malloc(4 << 20);
    for (int i = 0; i < (1 << 20); ++i) h[i] = (float)((i * 2654435761u >> 20) & 0x3FF);      // 0 .. 1023: a 64-lane sum is exact
    CK(hipMemcpy(in, h, 4 << 20, hipMemcpyHostToDevice));
    CK(hipMemset(src, 1, 2 << 20)); CK(hipMemset(bad0, 0, 4)); CK(hipMemset(bad1, 0, 4));
    CK(hipFuncSetAttribute((const void*)aggressor<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)aggressor<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)aggressor<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int k = 0; k < 20; ++k) {
            hipLaunchKernelGGL(victim<0>, dim3(600), dim3(256), 0, sv, in, 200, bad0, sink);
            hipLaunchKernelGGL(victim<1>, dim3(600), dim3(256), 0, sv, in, 200, bad1, sink);
            if (kind == 1) hipLaunchKernelGGL(aggressor<1>, dim3(496), dim3(512), 65536, sa, src, sink, 24);
            if (kind == 2) hipLaunchKernelGGL(aggressor<2>, dim3(496), dim3(512), 65536, sa, src, sink, 24);
            if (kind == 3) hipLaunchKernelGGL(aggressor<3>, dim3(496), dim3(512), 65536, sa, src, sink, 24);
            ++launches;
        }
        CK(hipStreamSynchronize(sv)); CK(hipStreamSynchronize(sa));
    }
    unsigned b0 = 0, b1 = 0;
    CK(hipMemcpy(&b0, bad0, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b1, bad1, 4, hipMemcpyDeviceToHost));
    printf("aggressor %d: %ld victim launches x 2400 waves x 200 sums: wrong sums via ds_bpermute %u, via DPP + readlane %u\n", kind, launches, b0, b1);
    return (b0 || b1) ? 1 : 0;
}
