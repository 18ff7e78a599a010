// pk_victim.hip -- the synthetic packed-fp32 victim of packed_fp32_repro.hip as a tiny shared library, so that Python can run it
// beside the LIBRARY's kernels (tools/probes/probe_pk_victim.py):  hipcc --offload-arch=gfx950 -O3 -shared -fPIC pk_victim.hip -o libpkvictim.so
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(1))) const void* glb_vp;

// the forms the compiler emitted in the library kernel: the (mean, rstd) pair is ONE 64-bit operand and op_sel picks which half
// both lanes of the packed instruction use -- a - mean: v_pk_add_f32 ... op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1];  * rstd: op_sel:[0,1]
__device__ __forceinline__ f32x2 pk_sub_lo(f32x2 a, f32x2 mean_rstd) {
    f32x2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(mean_rstd));
    return r;
}
__device__ __forceinline__ f32x2 pk_mul_hi(f32x2 a, f32x2 mean_rstd) {
    f32x2 r;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(mean_rstd));
    return r;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float s_sub(float a, float m) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(m)); return r; }
__device__ __forceinline__ float s_mul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_fma(float a, float b, float c) { float r; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// rows x 512 floats; a 512-thread workgroup handles 64 rows x 16 columns like the library kernel's write-out.  The packed results
// go STRAIGHT into the 16-byte store (as in the library kernel's ISA: v_pk_fma_f32 x2, global_store_dwordx4), nothing in between;
// a second kernel on the same stream recomputes every element with scalar instructions and compares what is in memory.
__global__ __launch_bounds__(512) void victim(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                              float* __restrict__ y, int rows, int iters) {
    __shared__ float2 st[64];
    if (threadIdx.x < 64) st[threadIdx.x] = make_float2(0.01f * threadIdx.x, 1.f + 0.001f * threadIdx.x);
    __syncthreads();
    const int c0 = blockIdx.x * 16;
    for (int it = 0; it < iters; ++it) {
        for (int idx = threadIdx.x; idx < 64 * 4; idx += 512) {
            const int row = idx >> 2, c = c0 + ((idx & 3) << 2);
            if (row >= rows) continue;
            const f32x4 x4 = *reinterpret_cast<const f32x4*>(x + (long)row * 512 + c);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
            const float2 s = st[row];
            f32x2 lo = {x4.x, x4.y}, hi = {x4.z, x4.w};
            const f32x2 mr = {s.x, s.y};
            lo = pk_sub_lo(lo, mr); hi = pk_sub_lo(hi, mr);
            lo = pk_mul_hi(lo, mr); hi = pk_mul_hi(hi, mr);
            lo = pk_fma(f32x2{gm.x, gm.y}, lo, f32x2{bt.x, bt.y});
            hi = pk_fma(f32x2{gm.z, gm.w}, hi, f32x2{bt.z, bt.w});
            *reinterpret_cast<f32x4*>(y + ((long)it * 64 + row) * 512 + c) = f32x4{lo.x, lo.y, hi.x, hi.y};
        }
    }
}
__global__ __launch_bounds__(512) void check(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                             const float* __restrict__ y, int rows, int iters, unsigned* __restrict__ bad, unsigned* __restrict__ bad_lane) {
    const int c0 = blockIdx.x * 16;
    for (int it = 0; it < iters; ++it)
        for (int idx = threadIdx.x; idx < 64 * 4; idx += 512) {
            const int row = idx >> 2, c = c0 + ((idx & 3) << 2);
            if (row >= rows) continue;
            const f32x4 x4 = *reinterpret_cast<const f32x4*>(x + (long)row * 512 + c);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
            const float sx = 0.01f * row, sy = 1.f + 0.001f * row;
            const f32x4 got = *reinterpret_cast<const f32x4*>(y + ((long)it * 64 + row) * 512 + c);
            const float r0 = s_fma(gm.x, s_mul(s_sub(x4.x, sx), sy), bt.x), r1 = s_fma(gm.y, s_mul(s_sub(x4.y, sx), sy), bt.y);
            const float r2 = s_fma(gm.z, s_mul(s_sub(x4.z, sx), sy), bt.z), r3 = s_fma(gm.w, s_mul(s_sub(x4.w, sx), sy), bt.w);
            const bool ok = __float_as_uint(got.x) == __float_as_uint(r0) && __float_as_uint(got.y) == __float_as_uint(r1) &&
                            __float_as_uint(got.z) == __float_as_uint(r2) && __float_as_uint(got.w) == __float_as_uint(r3);
            if (!ok) { atomicAdd(bad, 1u); atomicAdd(bad_lane + (threadIdx.x & 63), 1u); }
        }
}


static float *g_x, *g_gm, *g_bt, *g_y;
static unsigned *g_bad, *g_lane;
extern "C" int pkv_init() {
    const int n = 64 * 512;
    float* h = (float*)malloc(n * 4);
    if (hipMalloc(&g_x, n * 4) || hipMalloc(&g_gm, 512 * 4) || hipMalloc(&g_bt, 512 * 4) || hipMalloc(&g_y, 4 * n * 4) || hipMalloc(&g_bad, 4) ||
        hipMalloc(&g_lane, 256)) return 1;
    for (int i = 0; i < n; ++i) h[i] = 0.001f * (float)((i * 2654435761u >> 12) & 0xFFF) - 2.f;
    hipMemcpy(g_x, h, n * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < 512; ++i) h[i] = 1.f + 0.0007f * i;
    hipMemcpy(g_gm, h, 512 * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < 512; ++i) h[i] = 0.1f - 0.0003f * i;
    hipMemcpy(g_bt, h, 512 * 4, hipMemcpyHostToDevice);
    hipMemset(g_bad, 0, 4); hipMemset(g_lane, 0, 256);
    free(h);
    return 0;
}
// `launches` victim + check pairs on `stream`; returns the running total of mismatching float4
extern "C" unsigned pkv_run(void* stream, int launches) {
    hipStream_t st = (hipStream_t)stream;
    for (int k = 0; k < launches; ++k) {
        hipMemsetAsync(g_y, 0xff, 4 * 64 * 512 * 4, st);
        hipLaunchKernelGGL(victim, dim3(32), dim3(512), 0, st, g_x, g_gm, g_bt, g_y, 50, 4);
        hipLaunchKernelGGL(check, dim3(32), dim3(512), 0, st, g_x, g_gm, g_bt, g_y, 50, 4, g_bad, g_lane);
    }
    hipStreamSynchronize(st);
    unsigned b = 0;
    hipMemcpy(&b, g_bad, 4, hipMemcpyDeviceToHost);
    return b;
}
