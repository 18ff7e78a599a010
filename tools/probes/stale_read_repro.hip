// stale_read_repro.hip -- library-free probe for the multi-queue stale read.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stale_read_repro.hip -o tools/probes/stale_read_repro
//   tools/probes/stale_read_repro <store_mode> <load_mode> <seconds> [streams]        (run 1..3 copies at once)
//
// One producer / consumer pair per iteration on ONE stream, no host synchronisation in between -- the situation of a
// GEMM epilogue followed by the LayerNorm-backward that reads it:
//   producer(it):  X[r][c] = f(it, r, c) for a [2400 x 512] fp32 buffer, written tile by tile (64 x 64 tiles, one per
//                  workgroup) with the store pattern of a GEMM epilogue,
//   consumer(it):  one wave per row reads X and compares every element with f(it, r, c); mismatches are counted and
//                  the first few recorded (a stale element shows the value of an EARLIER iteration).
// The buffer is re-used every iteration (like the plan's workspace is every step), so a consumer that hits a stale L1 /
// L2 line or overtakes its producer reads the previous iteration's value.
//
// store_mode 0: row-major 16-byte stores (the vec_out epilogue)      1: MFMA accumulator layout, dword stores (64-byte row
//            segments; the fp32-operand GEMM's epilogue)              2: as 0 plus 8-byte bf16 "plane" stores to a 2nd buffer,
//            which the consumer checks too (partial-line stores)
// load_mode  0: plain 16-byte global loads                           1: global_load_lds (LDS-DMA) then ds_read
// streams    1 (default): everything on one stream; n > 1: n independent producer/consumer chains on n streams of this
//            process (each with its own buffers) -- the in-process version of running n copies.
// rows       rows of the buffer (default 2400)
// rmw        n > 0: n read-modify-write kernels (X += 1, ONE workgroup per row, like the d-memory accumulation over the decoder
//            layers) between producer and consumer; the consumer expects f + n.
//
// Exit code 0 and "mismatches 0" = this stack keeps a producer's output visible to the next kernel of the same stream while
// other queues are busy; anything else is a reproducer that needs none of this library's kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e__ = (x);                                                          \
        if (e__ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e__)); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

constexpr int COLS = 512, TILE = 64;
static int ROWS_H = 2400;                 // rows of X (argv[5]): 2400 = the cfg2 activations (4.9 MB, cycles through the 4 MiB L2s);
                                          // 128 = 256 KB, stays resident in every XCD's L2 between iterations
__constant__ int ROWS;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(1))) const void* glb_vp;

__host__ __device__ inline unsigned fval(unsigned it, unsigned r, unsigned c) {
    return (it * 2654435761u + r * 40503u + c * 97u + 12345u) & 0x7FFFFFu;      // exact as a float
}

struct Bad { unsigned it, r, c, got, want; };

// one 64 x 64 tile per workgroup of 512 threads (the plane GEMM's geometry); bx fastest
template <int MODE>
__global__ __launch_bounds__(512) void producer(float* __restrict__ X, unsigned short* __restrict__ P, unsigned it) {
    const int tiles_x = COLS / TILE;
    // XCD-aware order like the library: blocks b and b+8 share an XCD; spread tiles so neighbours land on different XCDs
    const int nwg = gridDim.x, lid = blockIdx.x, xcd = lid & 7, q = nwg >> 3, r8 = nwg & 7;
    const int t = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (lid >> 3);
    const int by = t / tiles_x, bx = t - by * tiles_x;
    const int bm0 = by * TILE, bn0 = bx * TILE, tid = threadIdx.x;
    if (MODE == 1) {
        // accumulator layout: wave w owns a 16 x 32 sub-tile ((w>>1)*16, (w&1)*32); lane: col = lane & 15, rows 4*(lane>>4) + r
        const int lane = tid & 63, wave = tid >> 6, wm0 = (wave >> 1) * 16, wn0 = (wave & 1) * 32;
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 4; ++r) {
                const int gm = bm0 + wm0 + ((lane >> 4) << 2) + r, gn = bn0 + wn0 + j * 16 + (lane & 15);
                if (gm < ROWS) X[(long)gm * COLS + gn] = (float)fval(it, gm, gn);
            }
        return;
    }
    for (int pass = 0; pass < 2; ++pass) {
        const int row = pass * 32 + (tid >> 4), c4 = (tid & 15) << 2, gm = bm0 + row, gn = bn0 + c4;
        if (gm >= ROWS) continue;
        float4 v = make_float4((float)fval(it, gm, gn), (float)fval(it, gm, gn + 1), (float)fval(it, gm, gn + 2), (float)fval(it, gm, gn + 3));
        *reinterpret_cast<float4*>(X + (long)gm * COLS + gn) = v;
        if (MODE == 2) {   // 8-byte "plane" store: the low 16 bits of each value
            uint2 w;
            w.x = (fval(it, gm, gn) & 0xFFFFu) | ((fval(it, gm, gn + 1) & 0xFFFFu) << 16);
            w.y = (fval(it, gm, gn + 2) & 0xFFFFu) | ((fval(it, gm, gn + 3) & 0xFFFFu) << 16);
            *reinterpret_cast<uint2*>(P + (long)gm * COLS + gn) = w;
        }
    }
}

// 4 waves per workgroup, one row per wave (the LayerNorm kernels' geometry)
template <int LOAD, bool PLANES>
__global__ __launch_bounds__(256) void consumer(const float* __restrict__ X, const unsigned short* __restrict__ P, unsigned it,
                                                unsigned* __restrict__ count, Bad* __restrict__ bad, float* __restrict__ sink, unsigned add) {
    __shared__ __attribute__((aligned(16))) float stage[4][COLS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = blockIdx.x * 4 + wave;
    if (row >= ROWS) return;
    float acc = 0.f;
    for (int c0 = 0; c0 < COLS; c0 += 256) {
        const int c = c0 + lane * 4;
        float4 v;
        if (LOAD == 1) {
            __builtin_amdgcn_global_load_lds((glb_vp)(X + (long)row * COLS + c), (lds_vp)(&stage[wave][c0]), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            v = *reinterpret_cast<const float4*>(&stage[wave][c]);
        } else {
            v = *reinterpret_cast<const float4*>(X + (long)row * COLS + c);
        }
        const float f[4] = {v.x, v.y, v.z, v.w};
        for (int e = 0; e < 4; ++e) {
            const unsigned want = fval(it, row, c + e) + add, got = (unsigned)f[e];
            acc += f[e];
            if (got != want || f[e] != (float)want) {
                const unsigned k = atomicAdd(count, 1u);
                if (k < 64) bad[k] = Bad{it, (unsigned)row, (unsigned)(c + e), __float_as_uint(f[e]), want};
            }
        }
        if (PLANES) {
            const uint2 w = *reinterpret_cast<const uint2*>(P + (long)row * COLS + c);
            const unsigned g[4] = {w.x & 0xFFFFu, w.x >> 16, w.y & 0xFFFFu, w.y >> 16};
            for (int e = 0; e < 4; ++e) {
                const unsigned want = fval(it, row, c + e) & 0xFFFFu;
                if (g[e] != want) {
                    const unsigned k = atomicAdd(count, 1u);
                    if (k < 64) bad[k] = Bad{it, (unsigned)row, (unsigned)(c + e) | 0x80000000u, g[e], want};
                }
            }
        }
    }
    if (lane == 0) sink[row] = acc;     // a store of the consumer's own (like dx), so it is not a pure reader
}

// one workgroup per row: X[row][:] += 1 (and the planes' low halves follow): a different row -> CU / XCD map than the producer's
__global__ __launch_bounds__(256) void rmw(float* __restrict__ X) {
    const int row = blockIdx.x, c = threadIdx.x * 2;
    float2 v = *reinterpret_cast<float2*>(X + (long)row * COLS + c);
    v.x += 1.f; v.y += 1.f;
    *reinterpret_cast<float2*>(X + (long)row * COLS + c) = v;
}

struct Chain {
    hipStream_t st;
    float *X, *sink;
    unsigned short* P;
    unsigned* count;
    Bad* bad;
};

int main(int argc, char** argv) {
    const int store_mode = argc > 1 ? atoi(argv[1]) : 0, load_mode = argc > 2 ? atoi(argv[2]) : 0;
    const double seconds = argc > 3 ? atof(argv[3]) : 5.0;
    const int nstreams = argc > 4 ? atoi(argv[4]) : 1;
    ROWS_H = argc > 5 ? atoi(argv[5]) : 2400;
    const int nrmw = argc > 6 ? atoi(argv[6]) : 0;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(ROWS), &ROWS_H, sizeof(int)));
    const int ROWS = ROWS_H;
    std::vector<Chain> ch(nstreams);
    for (Chain& c : ch) {
        CK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
        CK(hipMalloc(&c.X, (size_t)ROWS * COLS * 4));
        CK(hipMalloc(&c.P, (size_t)ROWS * COLS * 2));
        CK(hipMalloc(&c.sink, ROWS * 4));
        CK(hipMalloc(&c.count, 4));
        CK(hipMalloc(&c.bad, 64 * sizeof(Bad)));
        CK(hipMemset(c.X, 0, (size_t)ROWS * COLS * 4));
        CK(hipMemset(c.P, 0, (size_t)ROWS * COLS * 2));
        CK(hipMemset(c.count, 0, 4));
    }
    CK(hipDeviceSynchronize());
    const int pgrid = ((ROWS + TILE - 1) / TILE) * (COLS / TILE), cgrid = (ROWS + 3) / 4;
    unsigned it = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        for (int k = 0; k < 50; ++k) {
            ++it;
            for (Chain& c : ch) {
                if (store_mode == 0) hipLaunchKernelGGL(producer<0>, dim3(pgrid), dim3(512), 0, c.st, c.X, c.P, it);
                else if (store_mode == 1) hipLaunchKernelGGL(producer<1>, dim3(pgrid), dim3(512), 0, c.st, c.X, c.P, it);
                else hipLaunchKernelGGL(producer<2>, dim3(pgrid), dim3(512), 0, c.st, c.X, c.P, it);
                for (int k2 = 0; k2 < nrmw; ++k2) hipLaunchKernelGGL(rmw, dim3(ROWS), dim3(256), 0, c.st, c.X);
                if (store_mode == 2) {
                    if (load_mode == 1) hipLaunchKernelGGL((consumer<1, true>), dim3(cgrid), dim3(256), 0, c.st, c.X, c.P, it, c.count, c.bad, c.sink, (unsigned)nrmw);
                    else hipLaunchKernelGGL((consumer<0, true>), dim3(cgrid), dim3(256), 0, c.st, c.X, c.P, it, c.count, c.bad, c.sink, (unsigned)nrmw);
                } else {
                    if (load_mode == 1) hipLaunchKernelGGL((consumer<1, false>), dim3(cgrid), dim3(256), 0, c.st, c.X, c.P, it, c.count, c.bad, c.sink, (unsigned)nrmw);
                    else hipLaunchKernelGGL((consumer<0, false>), dim3(cgrid), dim3(256), 0, c.st, c.X, c.P, it, c.count, c.bad, c.sink, (unsigned)nrmw);
                }
            }
        }
        for (Chain& c : ch) CK(hipStreamSynchronize(c.st));
    }
    CK(hipDeviceSynchronize());
    unsigned total = 0;
    for (int s = 0; s < nstreams; ++s) {
        unsigned n = 0;
        CK(hipMemcpy(&n, ch[s].count, 4, hipMemcpyDeviceToHost));
        total += n;
        if (n) {
            Bad b[64];
            CK(hipMemcpy(b, ch[s].bad, sizeof(b), hipMemcpyDeviceToHost));
            for (unsigned k = 0; k < n && k < 8; ++k)
                printf("  stream %d: it %u row %u col %u%s got 0x%08x want %u\n", s, b[k].it, b[k].r, b[k].c & 0x7FFFFFFFu,
                       (b[k].c >> 31) ? " (plane)" : "", b[k].got, b[k].want);
        }
    }
    printf("store_mode %d load_mode %d streams %d rows %d rmw %d: %u iterations, mismatches %u\n", store_mode, load_mode, nstreams, ROWS, nrmw, it, total);
    return total ? 1 : 0;
}
