// What bounds one K-step of the plane GEMM's operand ring on MI355X?  A workgroup of 8 waves streams STEPS stages
// of PIECES KiB-per-wave pieces through an S-deep LDS ring with the GEMM's synchronisation (counted vmcnt + one
// barrier per step) but no MFMA work: per-step time as a function of ring depth, stage size, workgroups per CU and
// where the data lives (a footprint that fits the L2s / MALL / only HBM).  Also the same ring through VGPRs
// (global_load_dwordx4 + ds_write_b128) instead of LDS-DMA.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_ring.hip -o tools/micro/dma_ring
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <initializer_list>

typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(1))) const void* glb_vp;

// every workgroup walks its own contiguous run of the buffer (wrapping inside `foot` bytes)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// CONSUME: 0 = touch one word per lane, 1 = the GEMM's fragment reads (8 x ds_read_b128 per wave and 32 KiB stage),
//          2 = reads + the 12 MFMAs they feed (split-bf16, 3 passes, 2x2 quadrant)
// ORDER: 0 = refill the ring, then consume; 1 = consume, then refill; 2 = one DMA piece after every MFMA group
template <int S, int PIECES, bool DMA, int CONSUME = 0, int ORDER = 0>
__global__ __launch_bounds__(512) void ring_kernel(const char* __restrict__ buf, size_t foot, int steps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = PIECES * 8 * 1024;       // bytes per stage (8 waves x PIECES KiB)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = ((size_t)blockIdx.x * steps * STAGE) % foot;
    float acc = 0.f;
    f32x4 macc[2][2] = {};
    float4 regs[S][PIECES];
    auto issue = [&](int t, int stage, int slot) {
        size_t off = base + (size_t)t * STAGE;
        if (off + STAGE > foot) off %= (foot - STAGE + 1), off &= ~(size_t)1023;
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const char* src = buf + off + (size_t)(p * 8 + wave) * 1024 + lane * 16;
            char* dst = smem + stage * STAGE + (p * 8 + wave) * 1024;
            if (DMA) __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)dst, 16, 0, 0);
            else regs[slot][p] = *reinterpret_cast<const float4*>(src);
        }
    };
    for (int t = 0; t < S - 1; ++t) issue(t, t, t);
#pragma unroll 1
    for (int t0 = 0; t0 < steps; t0 += S) {
#pragma unroll
        for (int u = 0; u < S; ++u) {            // unrolled by the ring depth so that stage / slot indices are static
            const int t = t0 + u;
            if (DMA) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * PIECES) : "memory");
            } else {                              // the register path: the oldest slot has landed -> write it to LDS
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * PIECES) : "memory");
#pragma unroll
                for (int p = 0; p < PIECES; ++p)
                    *reinterpret_cast<float4*>(smem + u * STAGE + (p * 8 + wave) * 1024 + lane * 16) = regs[u][p];
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // the stage consumed in the previous step is free once every wave has passed this barrier: it may be
            // refilled before, after or in between this step's fragment reads / MFMAs
            const int tn = t + S - 1, sn = (u + S - 1) % S;
            auto issue_piece = [&](int p) {
                const int tt = tn < steps ? tn : 0;
                size_t off = base + (size_t)tt * STAGE;
                if (off + STAGE > foot) off %= (foot - STAGE + 1), off &= ~(size_t)1023;
                const char* src = buf + off + (size_t)(p * 8 + wave) * 1024 + lane * 16;
                char* dst = smem + sn * STAGE + (p * 8 + wave) * 1024;
                __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)dst, 16, 0, 0);
            };
            if (ORDER == 0) issue(tn < steps ? tn : 0, sn, sn);
            if (CONSUME == 0) {
                acc += *reinterpret_cast<const float*>(smem + u * STAGE + ((threadIdx.x * 16) % STAGE));
            } else {
                bf16x8 f[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)      // conflict-free: a wave reads 1 KiB contiguous
                    f[i] = *reinterpret_cast<const bf16x8*>(smem + u * STAGE + ((i * 4096 + wave * 1024 + lane * 16) % STAGE));
                if (CONSUME == 1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(f[i]));
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            macc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[4 + i], f[2 + j], macc[i][j], 0, 0, 0);
                            macc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i], f[6 + j], macc[i][j], 0, 0, 0);
                            macc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i], f[2 + j], macc[i][j], 0, 0, 0);
                            if (ORDER == 2 && i * 2 + j < PIECES) issue_piece(i * 2 + j);
                        }
                }
            }
            if (ORDER == 1) issue(tn < steps ? tn : 0, sn, sn);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (CONSUME == 2) acc += macc[0][0][0] + macc[0][1][1] + macc[1][0][2] + macc[1][1][3];
    if (acc == 1.2345f) sink[0] = acc;
}

// Wave specialisation: NPROD extra waves do nothing but refill the ring (they sit blocked in the memory pipeline's issue
// queue), the 8 consumer waves never issue a global load: after the step's barrier they go straight to the fragment
// reads and MFMAs.  Same 32 KiB stage, same single barrier per step.
template <int S, int NPROD>
__global__ __launch_bounds__(512 + 64 * NPROD) void ring_spec_kernel(const char* __restrict__ buf, size_t foot, int steps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 32 * 1024, NPIECE = 32 / NPROD;     // KiB pieces per producer wave and stage
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = ((size_t)blockIdx.x * steps * STAGE) % foot;
    float acc = 0.f;
    f32x4 macc[2][2] = {};
    if (wave >= 8) {                                            // ---- producer
        const int pw = wave - 8;
        auto issue = [&](int t, int stage) {
            size_t off = base + (size_t)t * STAGE;
            if (off + STAGE > foot) off %= (foot - STAGE + 1), off &= ~(size_t)1023;
#pragma unroll
            for (int p = 0; p < NPIECE; ++p) {
                const int piece = pw * NPIECE + p;
                __builtin_amdgcn_global_load_lds((glb_vp)(buf + off + (size_t)piece * 1024 + lane * 16),
                                                 (lds_vp)(smem + stage * STAGE + piece * 1024), 16, 0, 0);
            }
        };
        for (int t = 0; t < S - 1; ++t) issue(t, t);
#pragma unroll 1
        for (int t0 = 0; t0 < steps; t0 += S) {
#pragma unroll
            for (int u = 0; u < S; ++u) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * NPIECE) : "memory");
                __builtin_amdgcn_s_barrier();
                const int tn = t0 + u + S - 1;
                issue(tn < steps ? tn : 0, (u + S - 1) % S);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
#pragma unroll 1
    for (int t0 = 0; t0 < steps; t0 += S) {                    // ---- consumers
#pragma unroll
        for (int u = 0; u < S; ++u) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            bf16x8 f[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                f[i] = *reinterpret_cast<const bf16x8*>(smem + u * STAGE + ((i * 4096 + wave * 1024 + lane * 16) % STAGE));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    macc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[4 + i], f[2 + j], macc[i][j], 0, 0, 0);
                    macc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i], f[6 + j], macc[i][j], 0, 0, 0);
                    macc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i], f[2 + j], macc[i][j], 0, 0, 0);
                }
        }
    }
    acc += macc[0][0][0] + macc[0][1][1] + macc[1][0][2] + macc[1][1][3];
    if (acc == 1.2345f) sink[0] = acc;
}

template <int S, int NPROD>
static void run_spec(const char* buf, size_t foot, int grid, float* sink, const char* where) {
    const int nsteps = 48;
    const size_t lds = (size_t)S * 32 * 1024;
    hipFuncSetAttribute((const void*)ring_spec_kernel<S, NPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((ring_spec_kernel<S, NPROD>), dim3(grid), dim3(512 + 64 * NPROD), lds, 0, buf, foot, nsteps, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double us = best * 1e3, bytes = (double)grid * nsteps * 32768.0;
    printf("%-4s dma consume 2 %d producer waves ring %d x 32 KiB  grid %4d : %7.2f us  %6.3f us/step  %7.1f GB/s  (%5.1f GB/s per workgroup)\n",
           where, NPROD, S, grid, us, us / nsteps, bytes / us * 1e-3, bytes / us * 1e-3 / grid);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

// "B direct": a 64x64x64 tile-step where only A goes through LDS (16 KiB stage, LDS-DMA) and every wave loads the B
// fragments of its OWN columns straight from global memory into VGPRs (fragment-major weight planes: one coalesced
// 1 KiB load per fragment, no sharing between waves -> no LDS traffic for B).  NW waves per workgroup, each owning all
// 64 rows and 64/NW columns; B for the next step is prefetched into a second register set.
template <int NW>
__global__ __launch_bounds__(64 * NW) void ring_bdirect_kernel(const char* __restrict__ abuf, const char* __restrict__ bbuf,
                                                               size_t foot, int steps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = 16 * 1024, NPIECE = 16 / NW, NF = 4 / NW;    // A pieces per wave; n-fragments (16 columns) per wave
    constexpr int NB = NF * 4;                                         // B loads per wave and step: NF x 2 kk x (hi, lo)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t abase = ((size_t)blockIdx.x * steps * STAGE) % foot, bbase = ((size_t)blockIdx.x * 7919 * STAGE) % foot;
    f32x4 macc[4][NF] = {};
    float4 b0[NB], b1[NB];
    auto issue_a = [&](int t, int stage) {
        size_t off = abase + (size_t)t * STAGE;
        if (off + STAGE > foot) off %= (foot - STAGE + 1), off &= ~(size_t)1023;
#pragma unroll
        for (int p = 0; p < NPIECE; ++p) {
            const int piece = wave * NPIECE + p;
            __builtin_amdgcn_global_load_lds((glb_vp)(abuf + off + (size_t)piece * 1024 + lane * 16),
                                             (lds_vp)(smem + stage * STAGE + piece * 1024), 16, 0, 0);
        }
    };
    auto load_b = [&](int t, float4 (&b)[NB]) {
        size_t off = bbase + (size_t)t * STAGE;
        if (off + STAGE > foot) off %= (foot - STAGE + 1), off &= ~(size_t)1023;
#pragma unroll
        for (int i = 0; i < NB; ++i) b[i] = *reinterpret_cast<const float4*>(bbuf + off + (size_t)(wave * NB + i) * 1024 + lane * 16);
    };
    auto consume = [&](int stage, const float4 (&b)[NB]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 ah[4], al[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8*>(smem + stage * STAGE + (kk * 4 + i) * 1024 + lane * 16);
                al[i] = *reinterpret_cast<const bf16x8*>(smem + stage * STAGE + 8192 + (kk * 4 + i) * 1024 + lane * 16);
            }
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, b[(j * 2 + kk) * 2]), bl = __builtin_bit_cast(bf16x8, b[(j * 2 + kk) * 2 + 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    macc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh, macc[i][j], 0, 0, 0);
                    macc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl, macc[i][j], 0, 0, 0);
                    macc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh, macc[i][j], 0, 0, 0);
                }
            }
        }
    };
    issue_a(0, 0);
    load_b(0, b0);
#pragma unroll 1
    for (int t = 0; t < steps; t += 2) {
        // vmcnt counts loads in issue order: A(t) pieces, then B(t): both must have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_a(t + 1 < steps ? t + 1 : 0, 1);
        load_b(t + 1 < steps ? t + 1 : 0, b1);
        consume(0, b0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        issue_a(t + 2 < steps ? t + 2 : 0, 0);
        load_b(t + 2 < steps ? t + 2 : 0, b0);
        consume(1, b1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc += macc[i][j][0] + macc[i][j][3];
    if (acc == 1.2345f) sink[0] = acc;
}

template <int NW>
static void run_bdirect(const char* buf, size_t foot, int grid, float* sink, const char* where) {
    const int nsteps = 48;
    const size_t lds = 2 * 16 * 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((ring_bdirect_kernel<NW>), dim3(grid), dim3(64 * NW), lds, 0, buf, buf + (foot >> 1 & ~(size_t)1023), foot >> 1, nsteps, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double us = best * 1e3;
    printf("%-4s B-direct %d waves/workgroup, A ring 2 x 16 KiB  grid %4d : %7.2f us  %6.3f us/step  -> %6.3f us per 64x64x64 tile-step and CU\n",
           where, NW, grid, us, us / nsteps, us / nsteps / (grid / 256.0 < 1 ? 1 : grid / 256.0));
    hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int S, int PIECES, bool DMA, int CONSUME = 0, int ORDER = 0>
static void run(const char* buf, size_t foot, int grid, float* sink, const char* where) {
    const int nsteps = 48;                       // a multiple of every ring depth used below
    const size_t lds = (size_t)S * PIECES * 8 * 1024;
    hipFuncSetAttribute((const void*)ring_kernel<S, PIECES, DMA, CONSUME, ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((ring_kernel<S, PIECES, DMA, CONSUME, ORDER>), dim3(grid), dim3(512), lds, 0, buf, foot, nsteps, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double us = best * 1e3, bytes = (double)grid * nsteps * PIECES * 8192.0;
    printf("%-4s %s consume %d order %d ring %d x %2d KiB  grid %4d : %7.2f us  %6.3f us/step  %7.1f GB/s  (%5.1f GB/s per workgroup)\n", where,
           DMA ? "dma" : "reg", CONSUME, ORDER, S, PIECES * 8, grid, us, us / nsteps, bytes / us * 1e-3, bytes / us * 1e-3 / grid);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char**) {
    const size_t cap = (size_t)1 << 30;
    char* buf; float* sink;
    if (hipMalloc(&buf, cap) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, cap);
    struct { const char* name; size_t foot; } places[] = {{"L2", (size_t)2 << 20}, {"MALL", (size_t)96 << 20}, {"HBM", cap}};
    const int nplaces = argc > 1 ? 3 : 2;
    const bool full = argc > 1;
    for (int ip = 0; ip < nplaces; ++ip) {
        auto& pl = places[ip];
        for (int grid : {64, 256, 512}) {
            run<2, 4, true, 0>(buf, pl.foot, grid, sink, pl.name);
            run<2, 4, true, 2, 0>(buf, pl.foot, grid, sink, pl.name);
            if (full) run<2, 4, true, 2, 2>(buf, pl.foot, grid, sink, pl.name);
            run_bdirect<2>(buf, pl.foot, grid, sink, pl.name);
            run_bdirect<4>(buf, pl.foot, grid, sink, pl.name);
            if (grid == 512) { run_bdirect<2>(buf, pl.foot, 1024, sink, pl.name); run_bdirect<4>(buf, pl.foot, 1024, sink, pl.name); run_bdirect<2>(buf, pl.foot, 1280, sink, pl.name); }
            run_spec<2, 1>(buf, pl.foot, grid, sink, pl.name);
            run_spec<2, 2>(buf, pl.foot, grid, sink, pl.name);
            run_spec<2, 4>(buf, pl.foot, grid, sink, pl.name);
            run_spec<3, 2>(buf, pl.foot, grid, sink, pl.name);
            run<3, 4, true, 2, 0>(buf, pl.foot, grid, sink, pl.name);
            if (full) run<3, 4, true, 2, 1>(buf, pl.foot, grid, sink, pl.name);
            if (full) run<3, 4, true, 2, 2>(buf, pl.foot, grid, sink, pl.name);
            if (full) run<4, 4, true, 2, 1>(buf, pl.foot, grid, sink, pl.name);
            if (full) run<4, 4, true, 2, 2>(buf, pl.foot, grid, sink, pl.name);
            if (!full) continue;
            run<4, 4, true>(buf, pl.foot, grid, sink, pl.name);
            run<4, 2, true>(buf, pl.foot, grid, sink, pl.name);
            run<6, 2, true>(buf, pl.foot, grid, sink, pl.name);
            run<2, 4, false>(buf, pl.foot, grid, sink, pl.name);
            run<3, 4, false>(buf, pl.foot, grid, sink, pl.name);
        }
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
