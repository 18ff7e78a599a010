// Would a 256 x 256 tile with ONE workgroup per CU beat the shipped 128 x 128 x 2-per-CU plane GEMM?  (the
// memory side costs the shipped kernel 23-38 %; a 256-wide tile moves half the operand bytes per FLOP, but with one workgroup per
// CU nothing covers its DMA-issue / fragment-read phases -- unless the two waves of a SIMD run half a step apart.)
//
// This micro-kernel is that K loop and nothing else: three-pass split-bf16, k-major hi / lo planes, 32-k stages, TWO stages of
// 64 KiB, 8 waves as 2 (M) x 4 (N), each 128 x 64 (32 accumulator tiles); wave group A (waves 0-3, the tile's upper half) and
// group B (waves 4-7) alternate: while one group issues its 96 MFMAs the other reads its fragments (group A also issues the
// ring's DMA), one s_barrier per half step.  The epilogue is a plain accumulator-layout store (so results can be checked).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pingpong_tile.hip -o tools/micro/pingpong_tile && tools/micro/pingpong_tile
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(1))) const void* glb_vp;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short u16;

constexpr int PT = 64, BKS = 32, IMG = PT * BKS;          // a plane image: 64 rows x 32 k = 4 KiB
constexpr int TM = 256, TN = 256;
constexpr int STAGE = 16 * IMG;                            // A_hi | A_lo | B_hi | B_lo, 4 images each: 64 KiB

// k-major 32-k image: two 64-byte rows share a 128-byte line; 16-byte slots XOR-swizzled by the line (gemm_planes.hip img32_off)
__device__ __forceinline__ int img32_off(int row, int k) {
    const int line = row >> 1, sl = ((row & 1) << 2) | (k >> 3);
    return line * PT + (((sl ^ (line & 7)) << 3) | (k & 7));
}
// lane offset (bytes) of DMA piece p (0..7) of a 128-row slab starting at row0: image p >> 2, lines 8 (p & 3) ...
__device__ __forceinline__ unsigned slab_lane_off(long ld, int row0, int p, int lane) {
    const int h = p >> 2, line = 8 * (p & 3) + (lane >> 3), ps = lane & 7;
    const int sl = ps ^ (line & 7);
    const long e = (long)(row0 + PT * h + 2 * line + (sl >> 2)) * ld + ((sl & 3) << 3);
    return (unsigned)(e * 2);
}
__device__ __forceinline__ bf16x8 frag(const u16* img, int r0, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + img32_off(r0 + (lane & 15), (lane >> 4) << 3));
}

// MODE 0: ping-pong (groups half a step apart).  MODE 1: all eight waves in step (one barrier per step, every wave issues DMA).
template <int MODE>
__global__ __launch_bounds__(512) void pp_kernel(const u16* __restrict__ Ahi, const u16* __restrict__ Alo, const u16* __restrict__ Bhi,
                                                  const u16* __restrict__ Blo, long lda, long ldb, int tiles_x, int tiles_y, int ksteps,
                                                  float* __restrict__ C, long ldc) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wi = wave & 3;
    int bx, by;
    {   // XCD-aware order: each XCD a contiguous run of tiles, column by column inside groups of 2 tile rows
        const int nwg = tiles_x * tiles_y, lid = blockIdx.x, xcd = lid & 7, q = nwg >> 3, r = nwg & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lid >> 3);
        constexpr int GR = 2;
        const int g = t / (GR * tiles_x), rem = t - g * (GR * tiles_x), rows_here = min(GR, tiles_y - g * GR);
        bx = rem / rows_here;
        by = g * GR + (rem - bx * rows_here);
    }
    const int bm0 = by * TM, bn0 = bx * TN;
    const int wm0 = grp * 128, wn0 = wi * 64;                 // this wave's 128 x 64 sub-tile

    // DMA pieces: a stage is 8 slabs (A_hi, A_lo, B_hi, B_lo x two 128-row slabs) of 8 pieces.  MODE 0: the four waves of group A
    // issue everything (pieces wi and wi + 4 of every slab: 16 per wave); MODE 1: every wave issues piece `wave` of every slab (8).
    constexpr int NPW = MODE == 0 ? 2 : 1;
    unsigned aoff[2][NPW], boff[2][NPW];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < NPW; ++q) {
            const int p = MODE == 0 ? wi + 4 * q : wave;
            aoff[s][q] = slab_lane_off(lda, bm0 + 128 * s, p, lane);
            boff[s][q] = slab_lane_off(ldb, bn0 + 128 * s, p, lane);
        }
    auto issue = [&](int k, int stage, int s_lo = 0, int s_hi = 2) {       // slabs [s_lo, s_hi) of every plane
        u16* st = smem + stage * STAGE;
        const long kt = (long)k * BKS;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < NPW; ++q) {
                if (s < s_lo || s >= s_hi) continue;
                const int p = MODE == 0 ? wi + 4 * q : wave;
                u16* d = st + (2 * s) * IMG + p * 512;
                __builtin_amdgcn_global_load_lds((glb_vp)(reinterpret_cast<const char*>(Ahi + kt) + aoff[s][q]), (lds_vp)(d), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_vp)(reinterpret_cast<const char*>(Alo + kt) + aoff[s][q]), (lds_vp)(d + 4 * IMG), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_vp)(reinterpret_cast<const char*>(Bhi + kt) + boff[s][q]), (lds_vp)(d + 8 * IMG), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((glb_vp)(reinterpret_cast<const char*>(Blo + kt) + boff[s][q]), (lds_vp)(d + 12 * IMG), 16, 0, 0);
            }
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // a wave's 128 rows are walked as two HALF steps of 64 rows: only 4 A tiles (hi, lo) + 4 B tiles (hi, lo) = 64 registers of
    // fragments live beside the 128 accumulator registers (all 8 A tiles at once spill: 255 VGPRs + 432 spilled)
    bf16x8 ah[4], al[4], bh[4], bl[4];
    auto read_half = [&](int stage, int h) {
        const u16* st = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wm0 + 64 * h + 16 * i;
            ah[i] = frag(st + (row / PT) * IMG, row % PT, lane);
            al[i] = frag(st + (4 + row / PT) * IMG, row % PT, lane);
        }
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn0 + 16 * j;
                bh[j] = frag(st + (8 + row / PT) * IMG, row % PT, lane);
                bl[j] = frag(st + (12 + row / PT) * IMG, row % PT, lane);
            }
        }
    };
    auto mfma_half = [&](int h) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 c = h == 0 ? acc[i][j] : acc[4 + i][j];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], c, 0, 0, 0);
                if (h == 0) acc[i][j] = c; else acc[4 + i][j] = c;
            }
    };
    auto bar = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); };
    auto drain = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    if constexpr (MODE == 0) {
        // half steps hs = 2k + h.  Group A: segment 2hs reads half hs, segment 2hs + 1 computes it; group B one segment later.
        // Slice k sits in buffer k % 2 and is read in segments 4k .. 4k + 3; slice k + 2 is issued into it by group A in its two
        // read segments 4k + 4 / 4k + 6 (one 128-row slab of every plane each) and is first read in segment 4k + 8.
        // the two groups run their own copies of the loop (same number of barriers): no merge points between the roles
        if (grp == 0) {
            issue(0, 0);
            if (ksteps > 1) { issue(1, 1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16) : "memory"); } else drain();
            bar();
            read_half(0, 0);
            bar();
            for (int k = 0; k < ksteps; ++k) {
                const int cur = k & 1, nxt = cur ^ 1;
                mfma_half(0);                                                    // segment 4k + 1
                bar();
                if (k >= 1 && k + 1 < ksteps) issue(k + 1, nxt, 1, 2);           // segment 4k + 2: second slab of slice k + 1
                read_half(cur, 1);
                bar();
                mfma_half(1);                                                    // segment 4k + 3
                drain();                                                         // slice k + 1 has landed
                bar();
                if (k + 2 < ksteps) issue(k + 2, cur, 0, 1);                     // segment 4k + 4: first slab of slice k + 2
                if (k + 1 < ksteps) read_half(nxt, 0);
                bar();
            }
        } else {
            bar();
            bar();
            for (int k = 0; k < ksteps; ++k) {
                const int cur = k & 1;
                read_half(cur, 0);                                               // segment 4k + 1
                bar();
                mfma_half(0);                                                    // segment 4k + 2
                bar();
                read_half(cur, 1);                                               // segment 4k + 3
                bar();
                mfma_half(1);                                                    // segment 4k + 4
                bar();
            }
        }
        drain();
    } else {
        issue(0, 0);
        for (int k = 0; k < ksteps; ++k) {
            drain();
            bar();
            if (k + 1 < ksteps) issue(k + 1, (k + 1) & 1);
            read_half(k & 1, 0);
            mfma_half(0);
            read_half(k & 1, 1);
            mfma_half(1);
        }
    }
    // accumulator-layout store: lane -> rows 4 (lane >> 4) .. + 3, column lane & 15 of each 16 x 16 tile
    const int crow = (lane >> 4) << 2, ccol = lane & 15;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(long)(bm0 + wm0 + 16 * i + crow + r) * ldc + bn0 + wn0 + 16 * j + ccol] = acc[i][j][r];
}

static u16 f2bf(float x) { unsigned u; memcpy(&u, &x, 4); const unsigned r = u + 0x7FFF + ((u >> 16) & 1); return (u16)(r >> 16); }
static float bf2f(u16 h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
static double run(const u16* Ahi, const u16* Alo, const u16* Bhi, const u16* Blo, int M, int N, int K, float* C, int reps) {
    const int tx = N / TN, ty = M / TM;
    CK(hipFuncSetAttribute((const void*)pp_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE * 2));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(pp_kernel<MODE>, dim3(tx * ty), dim3(512), 2 * STAGE * 2, 0, Ahi, Alo, Bhi, Blo, (long)K, (long)K, tx, ty, K / BKS, C, (long)N);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(pp_kernel<MODE>, dim3(tx * ty), dim3(512), 2 * STAGE * 2, 0, Ahi, Alo, Bhi, Blo, (long)K, (long)K, tx, ty, K / BKS, C, (long)N);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
    // ---- correctness on a small problem
    {
        const int M = 512, N = 512, K = 256;
        std::vector<float> A((size_t)M * K), B((size_t)N * K);
        srand(1);
        for (auto& v : A) v = (float)rand() / RAND_MAX - 0.5f;
        for (auto& v : B) v = (float)rand() / RAND_MAX - 0.5f;
        std::vector<u16> ah(A.size()), al(A.size()), bh(B.size()), bl(B.size());
        for (size_t i = 0; i < A.size(); ++i) { ah[i] = f2bf(A[i]); al[i] = f2bf(A[i] - bf2f(ah[i])); }
        for (size_t i = 0; i < B.size(); ++i) { bh[i] = f2bf(B[i]); bl[i] = f2bf(B[i] - bf2f(bh[i])); }
        u16 *dah, *dal, *dbh, *dbl; float* dC;
        CK(hipMalloc(&dah, ah.size() * 2)); CK(hipMalloc(&dal, al.size() * 2)); CK(hipMalloc(&dbh, bh.size() * 2)); CK(hipMalloc(&dbl, bl.size() * 2));
        CK(hipMalloc(&dC, (size_t)M * N * 4));
        CK(hipMemcpy(dah, ah.data(), ah.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dal, al.data(), al.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbh, bh.data(), bh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dbl, bl.data(), bl.size() * 2, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 2; ++mode) {
            CK(hipMemset(dC, 0, (size_t)M * N * 4));
            if (mode == 0) run<0>(dah, dal, dbh, dbl, M, N, K, dC, 1); else run<1>(dah, dal, dbh, dbl, M, N, K, dC, 1);
            std::vector<float> C((size_t)M * N);
            CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0, scale = 0;
            for (int m = 0; m < M; m += 7)
                for (int n = 0; n < N; n += 5) {
                    double r = 0;
                    for (int k = 0; k < K; ++k) r += (double)A[(size_t)m * K + k] * B[(size_t)n * K + k];
                    worst = fmax(worst, fabs(C[(size_t)m * N + n] - r)); scale = fmax(scale, fabs(r));
                }
            printf("mode %d: max err %.3e of %.3e (%s)\n", mode, worst, scale, worst < 1e-4 * scale ? "ok" : "WRONG");
        }
        hipFree(dah); hipFree(dal); hipFree(dbh); hipFree(dbl); hipFree(dC);
    }
    // ---- timing at the configs[4] in_proj shape and two others (random planes)
    const int shapes[][3] = {{16384, 3072, 1024}, {36096, 512, 512}, {16384, 1024, 3072}, {65536, 1024, 1024}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        u16 *ah, *al, *bh, *bl; float* C;
        CK(hipMalloc(&ah, (size_t)M * K * 2)); CK(hipMalloc(&al, (size_t)M * K * 2)); CK(hipMalloc(&bh, (size_t)N * K * 2)); CK(hipMalloc(&bl, (size_t)N * K * 2));
        CK(hipMalloc(&C, (size_t)M * N * 4));
        std::vector<u16> h((size_t)M * K);
        for (auto& v : h) v = f2bf((float)rand() / RAND_MAX - 0.5f);
        CK(hipMemcpy(ah, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        for (auto& v : h) v = f2bf(((float)rand() / RAND_MAX - 0.5f) * 0.004f);
        CK(hipMemcpy(al, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        h.resize((size_t)N * K);
        for (auto& v : h) v = f2bf((float)rand() / RAND_MAX - 0.5f);
        CK(hipMemcpy(bh, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        for (auto& v : h) v = f2bf(((float)rand() / RAND_MAX - 0.5f) * 0.004f);
        CK(hipMemcpy(bl, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        const double fl = 2.0 * M * N * K;
        const double t0 = run<0>(ah, al, bh, bl, M, N, K, C, 30), t1 = run<1>(ah, al, bh, bl, M, N, K, C, 30);
        printf("%6d x %5d x %5d  (%4d tiles of 256 x 256): ping-pong %8.1f us %6.1f TFLOP/s   in-step %8.1f us %6.1f TFLOP/s\n", M, N, K,
               (M / TM) * (N / TN), t0, fl / t0 / 1e6, t1, fl / t1 / 1e6);
        hipFree(ah); hipFree(al); hipFree(bh); hipFree(bl); hipFree(C);
    }
    return 0;
}
