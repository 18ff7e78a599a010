// packed_fp32_repro.hip -- LIBRARY-FREE REPRODUCER for DESIGN.md section 6 (MI355X / gfx950, ROCm 7.2, hipcc 7.2):
// packed fp32 VALU instructions with an op_sel half-select (v_pk_add_f32 ... op_sel_hi:[1,0], v_pk_mul_f32 ... op_sel:[0,1] -- the
// forms the compiler emits when one operand is a scalar held in half of a 64-bit register pair) return WRONG results in lanes
// 48-63 of a wave while waves of ANOTHER kernel execute MFMA instructions on the same CU; alone they are always right.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/packed_fp32_repro.hip -o tools/probes/packed_fp32_repro
//   tools/probes/packed_fp32_repro <aggressor mask> <seconds>
//
// victim (stream 0): the instruction sequence of the library kernel in which the wrong bits were pinned down -- three 16-byte
//   global loads + one LDS read, v_pk_add_f32 x2, v_pk_mul_f32 x2 (both with op_sel), v_pk_fma_f32 x2, one 16-byte store (inline
//   asm, so the forms are exactly these).  A second kernel on the same stream recomputes every element with scalar v_sub / v_mul /
//   v_fma -- IEEE operations on the same inputs, so the bits must agree -- and counts every float4 that differs, by 16-lane group.
// aggressors (streams 1, 2), bit mask:  1 = MFMA loop (v_mfma_f32_16x16x32_bf16)   2 = LDS-DMA ring (global_load_lds into 64 KiB)
//   4 = streaming global loads / stores   8 = a second packed-fp32 kernel   16 = transposing LDS reads (ds_read_b64_tr_b16)   0 = none
// Measured (profiles/r03_determinism_probes.txt, sessions 48-53), 4-5 s each:
//   mask  0:      0 of 6.5e5 launches        mask 1 (MFMA only): 64 float4, ALL in lanes 48-63      mask 2 / 4 / 8 / 16: 0
//   mask 15: 135 792 float4 in 40 500 launches, ALL in lanes 48-63          mask 31: 48, all in lanes 48-63
//   the same victim beside the library's own kernels (probe_pk_victim.py): 1.4 - 6.2 MILLION float4 in 4 s per aggressor kind.
//   With the packed instructions in their plain form (full 64-bit operands, no op_sel) the same program shows 0 everywhere.
// The library is built without packed fp32 instructions for that reason (sign-language-nlp_amd/Makefile).
// Exit code 1 when the victim saw a mismatch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e__ = (x);                                                                          \
        if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(2); } \
    } while (0)

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

__global__ __launch_bounds__(512) void agg_mfma(float* __restrict__ out, int steps) {
    typedef __attribute__((ext_vector_type(4))) float f4;
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x & 15) + i); b[i] = (__bf16)(0.02f * i); }
    for (int s = 0; s < steps; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == -1.f) out[0] = 1.f;
}
__global__ __launch_bounds__(512) void agg_dma(const unsigned short* __restrict__ src, float* __restrict__ out, int steps) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        for (int p = 0; p < 4; ++p) {
            const unsigned short* g = src + ((((long)blockIdx.x * 37 + s) * 16384 + (p * 8 + wave) * 512 + lane * 8) & 0xFFFF8);
            __builtin_amdgcn_global_load_lds((glb_vp)g, (lds_vp)(smem + (s & 1) * 16384 + (p * 8 + wave) * 512), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += (float)smem[(s & 1) * 16384 + ((tid * 33) & 16383)];
    }
    if (acc == -1.f) out[0] = acc;
}
__global__ __launch_bounds__(256) void agg_stream(const float4* __restrict__ src, float4* __restrict__ dst, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float4 v = src[i];
        v.x += 1.f;
        dst[i] = v;
    }
}
__global__ __launch_bounds__(512) void agg_tr(float* __restrict__ out, int steps) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    __shared__ __attribute__((aligned(16))) unsigned short img[64 * 72];       // a [k][row] bf16 image like the library's m-major tiles
    for (int i = threadIdx.x; i < 64 * 72; i += 512) img[i] = (unsigned short)(0x3f80 + (i & 63));
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 15;
    int acc = 0;
    for (int s = 0; s < steps; ++s) {
        const int kb = ((s & 1) * 32) + ((lane >> 4) << 3) + (i >> 2);
        const unsigned short* p0 = img + kb * 72 + ((s >> 1) & 3) * 16 + ((i & 3) << 2);
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0 + 4 * 72));
        acc += v0[0] + v0[3] + v1[1] + v1[2];
    }
    if (acc == -1) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void agg_pk(float* __restrict__ out, int steps) {
    f32x2 a = {0.5f + threadIdx.x, 1.5f}, b = {1.0001f, 0.9999f}, c = {0.f, 0.f};
    for (int s = 0; s < steps; ++s) c = pk_fma(a, b, c);
    if (c.x + c.y == -1.f) out[0] = c.x;
}

int main(int argc, char** argv) {
    const int mask = argc > 1 ? atoi(argv[1]) : 15;
    const double seconds = argc > 2 ? atof(argv[2]) : 5.0;
    hipStream_t sv, s1, s2;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int rows = 50;
    float *x, *gm, *bt, *y, *sink;
    float4 *big0, *big1;
    unsigned short* src;
    unsigned *bad, *bad_lane;
    const long nbig = 16L << 20;      // 256 MiB per streaming buffer
    CK(hipMalloc(&x, 64 * 512 * 4)); CK(hipMalloc(&gm, 512 * 4)); CK(hipMalloc(&bt, 512 * 4)); CK(hipMalloc(&y, 4 * 64 * 512 * 4));
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&src, 2 << 20)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&bad_lane, 64 * 4));
    CK(hipMalloc(&big0, nbig * 16)); CK(hipMalloc(&big1, nbig * 16));
    float* h = (float*)malloc(64 * 512 * 4);
    for (int i = 0; i < 64 * 512; ++i) h[i] = 0.001f * (float)((i * 2654435761u >> 12) & 0xFFF) - 2.f;
    CK(hipMemcpy(x, h, 64 * 512 * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 512; ++i) h[i] = 1.f + 0.0007f * i;
    CK(hipMemcpy(gm, h, 512 * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 512; ++i) h[i] = 0.1f - 0.0003f * i;
    CK(hipMemcpy(bt, h, 512 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(src, 1, 2 << 20)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(bad_lane, 0, 256)); CK(hipMemset(big0, 0, nbig * 16));
    CK(hipFuncSetAttribute((const void*)agg_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int k = 0; k < 50; ++k) {
            CK(hipMemsetAsync(y, 0xff, 4 * 64 * 512 * 4, sv));
            hipLaunchKernelGGL(victim, dim3(32), dim3(512), 0, sv, x, gm, bt, y, rows, 4);
            hipLaunchKernelGGL(check, dim3(32), dim3(512), 0, sv, x, gm, bt, y, rows, 4, bad, bad_lane);
            if (mask & 1) hipLaunchKernelGGL(agg_mfma, dim3(512), dim3(512), 0, s1, sink, 400);
            if (mask & 2) hipLaunchKernelGGL(agg_dma, dim3(496), dim3(512), 65536, s2, src, sink, 24);
            if (mask & 4) hipLaunchKernelGGL(agg_stream, dim3(2048), dim3(256), 0, s1, big0, big1, nbig / 16);
            if (mask & 8) hipLaunchKernelGGL(agg_pk, dim3(1024), dim3(256), 0, s2, sink, 2000);
            if (mask & 16) hipLaunchKernelGGL(agg_tr, dim3(512), dim3(512), 0, s1, sink, 2000);
            ++launches;
        }
        CK(hipStreamSynchronize(sv)); CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
    }
    unsigned b = 0, lanes[64];
    CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(lanes, bad_lane, 256, hipMemcpyDeviceToHost));
    printf("aggressors %2d: %ld victim launches (32 workgroups x 4 passes x 200 float4): packed != scalar in %u float4", mask, launches, b);
    if (b) {
        printf("; by 16-lane group:");
        for (int q = 0; q < 4; ++q) { unsigned t = 0; for (int l = 0; l < 16; ++l) t += lanes[q * 16 + l]; printf(" %u", t); }
    }
    printf("\n");
    return b ? 1 : 0;
}
