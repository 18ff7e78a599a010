// gemm.hip -- split-bf16 MFMA GEMM for gfx950 (MI355X).
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) )
//
// All tensors are fp32 in HBM.  Tiles are converted to bf16 on the way into
// LDS; with precision 3 each fp32 value x is split into hi = bf16(x) and
// lo = bf16(x - hi) and the product is accumulated as Alo*Bhi + Ahi*Blo +
// Ahi*Bhi on v_mfma_f32_16x16x32_bf16 with fp32 accumulators (~2e-5 relative
// error: what the reference-parity bar of 1e-3 on logits needs; a single bf16
// pass measures 5e-3 and flips argmaxes).
//
// Shapes here are small (M = 2400 or 50 tokens, K,N <= 1536) and every launch
// sits on a dependent chain, so the kernel is built for per-block LATENCY:
//  * 64 x BN tile (BN = 64, or 16 to spread a skinny problem over more CUs),
//    K-step 64: K = 512 is 8 steps;
//  * the fp32 tiles of steps t+1 AND t+2 are in flight in registers while step
//    t is consumed from LDS (HBM/L2 latency is ~1 us, a step's MFMA work ~0.1 us);
//  * k-major operands are stored [row][k] and read as 16-byte fragments;
//    m-major operands (dgrad's W, both wgrad operands) are stored [k][row]
//    with 8-byte vector writes and read with ds_read_b64_tr_b16, the CDNA4
//    transposing LDS read -- no 2-byte scatter, no transposed copies in HBM.
//
// Replaces the matmuls inside nn.Linear / MultiheadAttention in/out
// projections / nn.LSTM / nn.GRU that the reference reaches at
// /root/reference/model/transformer.py:40-48 and
// /root/reference/model/base/encoder_decoder_attn_bkp.py:95-100,186-200.
#include <atomic>
#include <type_traits>

#include "common.hpp"
#include "gemm_jobs.hpp"

namespace slnlp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#if SLNLP_PROBE_FENCES == 256
// timeline probe build (tools/probes/probe_gemm_timeline.py): every workgroup records 100 MHz timestamps of its phases
constexpr int GTS_MAX = 1 << 14, GTS_W = 6;   // words: entry, first K tile in LDS, K loop done, image written, end, {grid, block}
__device__ unsigned long long g_gts[GTS_MAX][GTS_W];
__device__ unsigned g_gts_n;
#define GTS_MARK(slot) do { if (threadIdx.x == 0) gts[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define GTS_MARK(slot) do { } while (0)
#endif

constexpr int BM = 64, BKT = 64;
constexpr int KLD = BKT;      // [row][k] image: 64 bf16 = 128-B rows, 16-B slots XOR-swizzled by (row & 7)

// element offset of (row, k) in the k-major image.  The mixed-row lane groups of ds_read_b128
// ({0-3,12-15,20-27}, ...) hit 16 distinct 16-B slots of the 256-B bank row with this swizzle;
// a padded stride cannot do that (the g=1 slots are the g=0 slots shifted by one).
__device__ __forceinline__ int kmaj_off(int row, int k) { return row * KLD + ((((k >> 3) ^ (row & 7)) << 3) | (k & 7)); }


__device__ __forceinline__ unsigned short f2bf(float x) {
    __bf16 b = (__bf16)x;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// One operand tile = ROWS x 64(k) fp32.  NV float4 per thread.
//  KMAJOR: element (row,k) at P + row*ld + k; float4 runs along k; 16 float4 per row.
// !KMAJOR: element (row,k) at P + k*ld + row; float4 runs along row; ROWS/4 float4 per k.
template <bool KMAJOR, int ROWS>
struct TileIO {
    static constexpr int NV = ROWS * BKT / 4 / 256;       // 4 (ROWS=64), 2 (ROWS=32) or 1 (ROWS=16)
    static constexpr int MLD = ROWS + 8;                  // [k][row] image row stride (bf16)
    static constexpr int PLANE = KMAJOR ? ROWS * KLD : BKT * MLD;

    __device__ static __forceinline__ void coords(int idx, int& row, int& k) {
        if (KMAJOR) { row = idx >> 4; k = (idx & 15) << 2; }
        else { k = idx / (ROWS / 4); row = (idx % (ROWS / 4)) << 2; }
    }

    // Issue the loads of one K-tile.  Branch-free and with NO use of the loaded values: any use here
    // (even zeroing a tail lane) makes hipcc wait vmcnt(0) right behind each load and serialises the
    // whole prefetch.  Out-of-range coordinates are clamped to a valid address; stash() zeroes them.
    template <bool VEC>
    __device__ static __forceinline__ void fetch(const float* __restrict__ P, long ld, int row0, int nrows,
                                                 int k0, int K, int tid, float4 (&r)[NV]) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            int row, k;
            coords(tid + 256 * u, row, k);
            row += row0;
            k += k0;
            if (VEC) {   // compile-time: the hot kernel has no control flow around its loads
                const int rc = row < nrows ? row : 0, kc = k < K ? k : 0;
                r[u] = *reinterpret_cast<const float4*>(KMAJOR ? P + (long)rc * ld + kc : P + (long)kc * ld + rc);
            } else {
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int rr = KMAJOR ? row : row + e, kk = KMAJOR ? k + e : k;
                    rr = rr < nrows ? rr : 0;
                    kk = kk < K ? kk : 0;
                    x[e] = KMAJOR ? P[(long)rr * ld + kk] : P[(long)kk * ld + rr];
                }
                r[u] = make_float4(x[0], x[1], x[2], x[3]);
            }
        }
    }

    // fp32 -> bf16 hi (+ lo) and store into the LDS image.  hi is the TRUNCATED upper half of the
    // fp32 word (1 VALU op instead of a round-to-nearest convert); x - hi is exact in fp32 and
    // lo = rne_bf16(x - hi) absorbs the truncation, so hi + lo still represents x to ~2^-16.
    // EDGE = false: interior tile, no bounds masks at all.
    template <int NSPLIT, bool EDGE>
    __device__ static __forceinline__ void stash(unsigned short* __restrict__ T, int tid, const float4 (&r)[NV],
                                                 int row0, int nrows, int k0, int K) {
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            int row, k;
            coords(tid + 256 * u, row, k);
            float x[4] = {r[u].x, r[u].y, r[u].z, r[u].w};
            if (EDGE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int rr = row0 + (KMAJOR ? row : row + e), kk = k0 + (KMAJOR ? k + e : k);
                    if (!(rr < nrows && kk < K)) x[e] = 0.f;      // edge / K-tail zero fill (v_cndmask)
                }
            }
            const int off = KMAJOR ? kmaj_off(row, k) : k * MLD + row;   // both 8-B aligned
            unsigned ub[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) ub[e] = __float_as_uint(x[e]);
            uint2 w;
            if (NSPLIT == 3) {
                w.x = (ub[0] >> 16) | (ub[1] & 0xFFFF0000u);
                w.y = (ub[2] >> 16) | (ub[3] & 0xFFFF0000u);
                *reinterpret_cast<uint2*>(T + off) = w;
                float lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) lo[e] = x[e] - __uint_as_float(ub[e] & 0xFFFF0000u);
                const bf16x2 l01 = __builtin_convertvector(f32x2{lo[0], lo[1]}, bf16x2);
                const bf16x2 l23 = __builtin_convertvector(f32x2{lo[2], lo[3]}, bf16x2);
                w.x = __builtin_bit_cast(unsigned, l01);
                w.y = __builtin_bit_cast(unsigned, l23);
                *reinterpret_cast<uint2*>(T + PLANE + off) = w;
            } else {   // single pass: round to nearest
                const bf16x2 h01 = __builtin_convertvector(f32x2{x[0], x[1]}, bf16x2);
                const bf16x2 h23 = __builtin_convertvector(f32x2{x[2], x[3]}, bf16x2);
                w.x = __builtin_bit_cast(unsigned, h01);
                w.y = __builtin_bit_cast(unsigned, h23);
                *reinterpret_cast<uint2*>(T + off) = w;
            }
        }
    }

    // MFMA 16x16x32 operand fragment of tile rows [r0, r0+16), k in [kk*32, kk*32+32):
    // lane l holds (row r0 + (l&15), k = kk*32 + 8*(l>>4) + j), j = 0..7.
    __device__ static __forceinline__ bf16x8 frag(const unsigned short* __restrict__ T, int r0, int kk, int lane) {
        if (KMAJOR) {
            return *reinterpret_cast<const bf16x8*>(T + kmaj_off(r0 + (lane & 15), kk * 32 + ((lane >> 4) << 3)));
        } else {
            // transposing read: lane (i = l&15; q = i>>2, p = i&3) addresses k-row q, columns 4p..4p+3 of a
            // 4(k) x 16(row) block and receives the 4 k-values of column i.
            const int i = lane & 15, kb = kk * 32 + ((lane >> 4) << 3) + (i >> 2);
            const unsigned short* p0 = T + kb * MLD + r0 + ((i & 3) << 2);
            typedef __attribute__((address_space(3))) s16x4* lds_p;
            const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
            const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0 + 4 * MLD));
            const s16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            return __builtin_bit_cast(bf16x8, v);
        }
    }

    // bf16(hi)+bf16(lo) value of tile element (row, k) -- for the fused bias-gradient row sums
    template <int NSPLIT>
    __device__ static __forceinline__ float value(const unsigned short* __restrict__ T, int row, int k) {
        const int off = KMAJOR ? kmaj_off(row, k) : k * MLD + row;
        float v = bf2f(T[off]);
        if (NSPLIT == 3) v += bf2f(T[PLANE + off]);
        return v;
    }
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also emits s_waitcnt vmcnt(0),
// which would drain the register prefetch of the next two K-tiles at every step.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// LDS of one tile job: A planes, B planes, 4 x 64 row-sum scratch
template <int NSPLIT, bool AK, bool BK, int BNT>
constexpr int tile_lds_elems() {
    return (NSPLIT == 3 ? 2 : 1) * (TileIO<AK, BM>::PLANE + TileIO<BK, BNT>::PLANE) + 4 * BM * 2;
}

// One workgroup's tile: block (bid_x, bid_y) of a (grid_x, grid_y) grid over C.
// The K sum of every output element is DEFINED as two halves -- K tiles [0, T) and [T, ktiles), T = ceil(ktiles / 2), each
// accumulated in tile order from zero -- added at the end (first + second).  How the halves are computed is then a scheduling
// choice that does not touch the result:
//   KS = 2: the workgroup has two groups of 256 threads; group g runs the K loop over its own half of the K tiles with its own
//           stage images (all loads of both halves are in flight together, the dependent chain of K steps is half as long),
//           group 1 hands its accumulators over through LDS and group 0 adds them and runs the epilogue.  For launches that
//           cannot fill the chip anyway (the decoder's 50-row GEMMs: 32 ... 96 workgroups of 8 K tiles, whose duration IS the
//           K chain);
//   KS = 1: one group walks both halves and parks the first half's accumulators when it reaches tile T -- half the threads and a
//           quarter less registers per workgroup: the merged launches of a lockstep group (throughput-bound), which therefore
//           need not run the kernel a solo fit runs to return its bits.
template <int NSPLIT, bool AK, bool BK, int BNT, bool VEC, int KS = 1>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, int bid_x, int bid_y, int grid_x, int grid_y,
                                          unsigned short* __restrict__ smem_base) {
#if SLNLP_PROBE_FENCES == 256
    unsigned long long gts[GTS_W] = {0, 0, 0, 0, 0, 0};
    struct GtsFlush {
        unsigned long long* t; int gx, bx;
        __device__ ~GtsFlush() {
            if (threadIdx.x == 0) {
                t[4] = __builtin_amdgcn_s_memrealtime();
                t[5] = ((unsigned long long)gx << 32) | (unsigned)bx;
                const unsigned i = atomicAdd(&g_gts_n, 1u) & (unsigned)(GTS_MAX - 1);
                for (int k = 0; k < GTS_W; ++k) g_gts[i][k] = t[k];
            }
        }
    } gts_flush{gts, grid_x * grid_y, bid_y * grid_x + bid_x};
    GTS_MARK(0);
#endif
    const int grp = KS == 2 ? (int)(threadIdx.x >> 8) : 0;
    unsigned short* __restrict__ smem = smem_base + grp * tile_lds_elems<NSPLIT, AK, BK, BNT>();
    using TA = TileIO<AK, BM>;
    using TB = TileIO<BK, BNT>;
    constexpr int NP = (NSPLIT == 3) ? 2 : 1;
    constexpr int MT = (BNT == 16) ? 1 : 2;  // 16x16 tiles per wave along M
    constexpr int NT = (BNT == 64) ? 2 : 1;  // ... along N
    unsigned short* As = smem;
    unsigned short* Bs = smem + NP * TA::PLANE;
    float (*rsum)[BM] = reinterpret_cast<float (*)[BM]>(smem + NP * (TA::PLANE + TB::PLANE));

    const slnlp_gemm_args& g = p.a;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    // BNT=64: waves 2x2, each 32x32.  BNT=32: waves 2x2, each 32x16.  BNT=16: waves 4x1, each 16x16.
    const int wm0 = (BNT == 16) ? wave * 16 : (wave >> 1) * 32;
    const int wn0 = (BNT == 64) ? (wave & 1) * 32 : (BNT == 32) ? (wave & 1) * 16 : 0;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private 4 MiB L2 each);
    // remap so each XCD owns a contiguous run of tiles (all column blocks of a few row blocks): its
    // L2 then holds the whole B operand plus a few A panels instead of every panel of both.
    int bx, by;
    {
        const int nwg = grid_x * grid_y, id = bid_y * grid_x + bid_x;
        const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        by = t / grid_x;
        bx = t - by * grid_x;
    }
    const int bm0 = by * BM, bn0 = bx * BNT;
    const int M = g.M, N = g.N, K = g.K;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool do_rowsum = (g.rowsum_a != nullptr) && (bx == 0);
    float rowsum = 0.f;

    const int ktiles = (K + BKT - 1) / BKT;
    // this group's K tiles [k0, k1); both groups make `trips` K steps (the barriers are the workgroup's), group 1 idles in its
    // last one when the number of tiles is odd
    const int half = (ktiles + 1) / 2;           // T: the second half of the K sum starts at this tile
    const int trips = KS == 2 ? half : ktiles;
    const int k0 = grp * trips, k1 = (KS == 2 && grp == 0) ? trips : ktiles;
    f32x4 acc_first[MT][NT];                     // KS = 1: the first half's sums, parked at tile T
    float rowsum_first = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc_first[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // DEPTH K tiles in flight in registers.  (Four for the 16-wide tile of the decoder's 50-row products -- with KS = 2 all eight
    // tiles of a K = 512 product requested at kernel entry -- was measured with per-workgroup timelines, tools/probes/
    // probe_gemm_timeline.py: the K loop of 4 steps stayed at 2.6 us, it is the steps' own convert / barrier / MFMA chain and not a
    // wait for loads, and the first tile arrived 0.6 us (warm) to 1.5 us (cold) LATER behind the twenty requests.)
    // Prefetches are UNCONDITIONAL (past-the-end tiles read a clamped, valid address and are never
    // stashed): a guard would add a join point and make hipcc fall back to conservative vmcnt counts.
    constexpr int DEPTH = 2;
    // (Requesting the 16-wide tile's bias / gate / residual values here, in front of the K tiles, was measured too: the epilogue got
    // 0.15 - 0.3 us shorter and the first tile arrived 0.4 us later behind the nine extra requests -- a net loss.)
    float4 ra[DEPTH][TA::NV], rb[DEPTH][TB::NV];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        TA::template fetch<VEC>(g.A, g.lda, bm0, M, (k0 + d) * BKT, K, tid, ra[d]);
        TB::template fetch<VEC>(g.B, g.ldb, bn0, N, (k0 + d) * BKT, K, tid, rb[d]);
    }

    auto consume = [&]() {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ah[i] = TA::frag(As, wm0 + 16 * i, kk, lane);
                if (NSPLIT == 3) al[i] = TA::frag(As + TA::PLANE, wm0 + 16 * i, kk, lane);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bh[j] = TB::frag(Bs, wn0 + 16 * j, kk, lane);
                if (NSPLIT == 3) bl[j] = TB::frag(Bs + TB::PLANE, wn0 + 16 * j, kk, lane);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (NSPLIT == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (do_rowsum) {  // thread owns row (tid & 63), k-quarter (tid >> 6)
            const int row = tid & 63, kq = (tid >> 6) * 16;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += TA::template value<NSPLIT>(As, row, kq + k);
            rowsum += s;
        }
    };

    // DEPTH K-steps per trip so the prefetch registers keep compile-time names
    auto mainloop = [&](auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        for (int it = 0; it < trips; it += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (d > 0 && it + d >= trips) break;
                const int kt = k0 + it + d;
                if (KS == 1 && kt == half) {     // (block-uniform; never taken when there is one tile)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) { acc_first[i][j] = acc[i][j]; acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                    rowsum_first = rowsum;
                    rowsum = 0.f;
                }
                lds_barrier();
                if (KS == 1 || kt < k1) {
                    TA::template stash<NSPLIT, EDGE>(As, tid, ra[d], bm0, M, kt * BKT, K);
                    TB::template stash<NSPLIT, EDGE>(Bs, tid, rb[d], bn0, N, kt * BKT, K);
                }
                lds_barrier();
#if SLNLP_PROBE_FENCES == 256
                if (it + d == 0) GTS_MARK(1);
#endif
                TA::template fetch<VEC>(g.A, g.lda, bm0, M, (kt + DEPTH) * BKT, K, tid, ra[d]);
                TB::template fetch<VEC>(g.B, g.ldb, bn0, N, (kt + DEPTH) * BKT, K, tid, rb[d]);
                if (KS == 1 || kt < k1) consume();
            }
        }
    };
    // interior tile (block-uniform): no bounds masks in the conversion
    if (bm0 + BM <= M && bn0 + BNT <= N && (K % BKT) == 0) mainloop(std::false_type{});
    else mainloop(std::true_type{});
    GTS_MARK(2);
    if (KS == 1 && ktiles <= half) {             // a single tile: it IS the first half (the loop never reached tile T)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) { acc_first[i][j] = acc[i][j]; acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        rowsum_first = rowsum;
        rowsum = 0.f;
    }
    if (do_rowsum) {
        // row sums of A: per half the four k-quarters of the threads, (q0 + q1) + (q2 + q3); first half + second half
        if (KS == 2) {
            rsum[tid >> 6][tid & 63] = rowsum;
            __syncthreads();
            float (*rs1)[BM] = reinterpret_cast<float (*)[BM]>(smem_base + tile_lds_elems<NSPLIT, AK, BK, BNT>() + NP * (TA::PLANE + TB::PLANE));
            float (*rs0)[BM] = reinterpret_cast<float (*)[BM]>(smem_base + NP * (TA::PLANE + TB::PLANE));
            if (grp == 0 && tid < 64 && bm0 + tid < M)
                g.rowsum_a[bm0 + tid] = ((rs0[0][tid] + rs0[1][tid]) + (rs0[2][tid] + rs0[3][tid])) + ((rs1[0][tid] + rs1[1][tid]) + (rs1[2][tid] + rs1[3][tid]));
        } else {
            lds_barrier();                       // (the last K step's fragment reads: rsum may overlap nothing, but keep the phases apart)
            rsum[tid >> 6][tid & 63] = rowsum_first;
            __syncthreads();
            float s_first = 0.f;
            if (tid < 64) s_first = (rsum[0][tid] + rsum[1][tid]) + (rsum[2][tid] + rsum[3][tid]);
            __syncthreads();
            rsum[tid >> 6][tid & 63] = rowsum;
            __syncthreads();
            if (tid < 64 && bm0 + tid < M) g.rowsum_a[bm0 + tid] = s_first + ((rsum[0][tid] + rsum[1][tid]) + (rsum[2][tid] + rsum[3][tid]));
        }
    }
    if (KS == 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = acc_first[i][j] + acc[i][j];
    }
    if (KS == 2) {   // group 1's half of the K sum -> LDS -> group 0 (its stage images are free now)
        float* red = reinterpret_cast<float*>(smem_base + tile_lds_elems<NSPLIT, AK, BK, BNT>());
        __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[((i * NT + j) * 4 + r) * 256 + tid] = acc[i][j][r];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += red[((i * NT + j) * 4 + r) * 256 + tid];
    }

    // ---- epilogue: +bias -> relu -> gate -> dropout -> +resid, through an fp32 image of the tile in LDS (the stage images are free).
    // Straight out of the accumulator layout -- MT x NT tiles x 4 rows unrolled, a Philox per tile (per ROW with per-head dropout) and
    // a tanh per element inlined 16 times -- the epilogue was three quarters of this kernel's code (5100 of 6900 instructions at the
    // 64-wide tile): straight-line code every workgroup walks once, cold in the instruction cache.  Now the accumulators go to the
    // image as they are and ONE rolled loop takes "quads" -- 4 consecutive rows x 1 column, the unit one Philox call serves -- through
    // the chain; consecutive lanes hold consecutive columns, so the stores are whole 64-float rows instead of 16-float segments.
    // Same operations in the same order per element: the bits do not change.
    constexpr int ILD = BNT + 4;
    static_assert(tile_lds_elems<NSPLIT, AK, BK, BNT>() * 2 >= BM * ILD * 4, "the tile's image must fit the stage memory");
    float* img = reinterpret_cast<float*>(smem_base);
    if (KS == 1) lds_barrier();                  // (KS = 2: the hand-over above already put a barrier behind every wave's last fragment read)
    {
        const int crow = (lane >> 4) << 2, ccol = lane & 15;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) img[(wm0 + i * 16 + crow + r) * ILD + wn0 + j * 16 + ccol] = acc[i][j][r];
    }
    lds_barrier();                               // (group 1 of a KS = 2 workgroup has ended: the barrier counts the waves that are left)
    GTS_MARK(3);
    const bool per_head = g.drop_head_dim > 0;
    DropKey dkey = {};                           // the step's Threefry key, made ONCE in front of the loop (common.hpp: dropout_key)
    if (g.drop_p > 0.f) dkey = dropout_key(g.rng, g.drop_site);
#pragma unroll 1
    for (int q = tid; q < (BM / 4) * BNT; q += 256) {
        const int col = q % BNT, gm0 = bm0 + 4 * (q / BNT), gn = bn0 + col;
        if (gn >= N || gm0 >= M) continue;
        const float bias = g.bias ? g.bias[gn] : 0.f;
        unsigned lot[4] = {0u, 0u, 0u, 0u};       // the quad's four 16-bit lots (common.hpp: one call serves 4 rows x columns {c, c ^ 16})
        if (g.drop_p > 0.f) {
            if (!per_head) {
                const uint4 bits = dropout_bits8(dkey, (unsigned)gm0 >> 2, drop_cc((unsigned)gn));
                const int half = drop_half((unsigned)gn);
#pragma unroll
                for (int r = 0; r < 4; ++r) lot[r] = pick_lot(bits, half, r);
            } else {                              // one keep/drop decision per (row, head), see slnlp.h
#pragma unroll 1
                for (int r = 0; r < 4; ++r) {
                    const unsigned rh = (unsigned)(gm0 + r) * (unsigned)(N / g.drop_head_dim) + (unsigned)(gn / g.drop_head_dim);
                    const uint4 hb = dropout_bits8(dkey, rh >> 2, 0u);
                    const unsigned wsel = pick_lot(hb, 0, (int)(rh & 3u));
                    if (r == 0) lot[0] = wsel; else if (r == 1) lot[1] = wsel; else if (r == 2) lot[2] = wsel; else lot[3] = wsel;
                }
            }
        }
        const float* src = img + (gm0 - bm0) * ILD + col;
        // the quad's gate and residual values are requested before its first store: resid may alias C (an in-place add), so a load
        // written behind a store has to stay there -- four dependent round trips per quad instead of one
        float gt[4], rs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm = gm0 + r;
            gt[r] = (g.gate && gm < M) ? g.gate[(long)gm * g.ldg + gn] : 0.f;
            rs[r] = (g.resid && gm < M) ? g.resid[(long)gm * g.ldr + gn] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm = gm0 + r;
            if (gm >= M) break;
            float v = src[r * ILD] + bias;
            if (g.relu == 1) v = fmaxf(v, 0.f);
            else if (g.relu == 2) v = tanhf(v);
            if (g.gate) v = g.gate_mode == 1 ? v * (1.f - gt[r] * gt[r]) : (gt[r] > 0.f ? v * g.gate_scale : 0.f);
            if (g.drop_p > 0.f) v = (lot[r] >= p.drop_thr) ? v * p.drop_scale : 0.f;
            if (g.resid) v += rs[r];
            g.C[(long)gm * g.ldc + gn] = v;
            if (g.C_hi) {                        // also as planes: the operand of a B-row product (gemm_rows.hip)
                unsigned short hh, ll;
                split_bf16(v, hh, ll);
                g.C_hi[(long)gm * g.ldc_p + gn] = hh;
                if (g.C_lo) g.C_lo[(long)gm * g.ldc_p + gn] = ll;
            }
        }
    }
}

// (two 256-thread workgroups per CU must fit: at most 256 registers a lane -- the scalar-load build of the 64-wide tile sits at the edge)
template <int NSPLIT, bool AK, bool BK, int BNT, bool VEC, int KS = 1>
__global__ __launch_bounds__(256 * KS, KS == 1 ? 2 : 1) void gemm_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned short smem[KS * tile_lds_elems<NSPLIT, AK, BK, BNT>()];
    probe_kernel_begin();
    gemm_tile<NSPLIT, AK, BK, BNT, VEC, KS>(p, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, smem);
    probe_kernel_end();
}

// Several independent fp32-operand GEMMs in ONE launch (e.g. the data- and weight-gradient of one dY in the
// decoder, whose B-row GEMMs are pure launch latency): workgroups [block_begin, block_begin + gx*gy) run job j.
// `tab` != nullptr: a merged (lockstep) launch -- the jobs of K fits in a device-resident table, blockmap[block] = job
template <int NSPLIT, int KS>
__global__ __launch_bounds__(256 * KS) void gemm_group_kernel(const GemmGroupParams P, const GemmJob* __restrict__ tab,
                                                              const int* __restrict__ blockmap) {
    __shared__ __attribute__((aligned(16))) unsigned short smem[KS * tile_lds_elems<NSPLIT, false, false, 64>()];   // the largest variant
    GemmParams p;
    int variant, gx, gy, lid;
    if (tab) {
        const GemmJob& J = tab[blockmap[blockIdx.x]];
        p = J.p; variant = J.variant; gx = J.gx; gy = J.gy; lid = blockIdx.x - J.block_begin;
    } else {
        int j = 0;
        for (int t = 1; t < P.njobs; ++t)
            if ((int)blockIdx.x >= P.block_begin[t]) j = t;
        p = P.job[j]; variant = P.variant[j]; gx = P.gx[j]; gy = P.gy[j]; lid = blockIdx.x - P.block_begin[j];
    }
    launder(p.a);
    if (p.a.batch > 1) {                    // `batch` GEMMs of one shape: this block belongs to GEMM z
        const int z = lid / (gx * gy);
        lid -= z * gx * gy;
        p.a.A += (long)z * p.a.batch_stride_a;
        p.a.B += (long)z * p.a.batch_stride_b;
        p.a.C += (long)z * p.a.batch_stride_c;
        if (p.a.resid) p.a.resid += (long)z * p.a.batch_stride_c;
        if (p.a.C_hi) p.a.C_hi += (long)z * p.a.batch_stride_c;     // (plane outputs share C's column layout: ldc_p = ldc)
        if (p.a.C_lo) p.a.C_lo += (long)z * p.a.batch_stride_c;
    }
    const int bx = lid % gx, by = lid / gx;
    probe_kernel_begin();
    switch (variant) {
        case 0: gemm_tile<NSPLIT, true, true, 64, true, KS>(p, bx, by, gx, gy, smem); break;
        case 1: gemm_tile<NSPLIT, true, true, 16, true, KS>(p, bx, by, gx, gy, smem); break;
        case 2: gemm_tile<NSPLIT, true, false, 64, true, KS>(p, bx, by, gx, gy, smem); break;
        case 3: gemm_tile<NSPLIT, true, false, 16, true, KS>(p, bx, by, gx, gy, smem); break;
        case 4: gemm_tile<NSPLIT, false, false, 64, true, KS>(p, bx, by, gx, gy, smem); break;
        default: gemm_tile<NSPLIT, false, false, 16, true, KS>(p, bx, by, gx, gy, smem); break;
    }
    probe_kernel_end();
}

// KS = 2 (the two K halves on two thread groups) for launches that cannot fill the chip and whose duration is the chain of K steps.
// Results do not depend on KS (gemm_tile: the K sum is two halves either way), so the rule may look at whatever it likes -- a merged
// lockstep launch picks again for its own size (gemm_group_ks, lockstep.hip).  -1 = automatic; 1 / 2 forced (slnlp_set_gemm_ks: tests).
static std::atomic<int> g_gemm_ks{[] { const char* e = getenv("SLNLP_GEMM_KS"); const int v = e ? atoi(e) : 0; return v == 1 || v == 2 ? v : -1; }()};
int gemm_group_ks(int blocks, int longest_ktiles) {
    const int forced = g_gemm_ks.load(std::memory_order_relaxed);
    if (forced > 0) return forced;
    // <= 256 workgroups: the decoder's 50-row products (32 ... 128) and the RNN's recurrent data-gradient group (256: a solo LSTM step
    // gains 2.6 %); a merged launch of 4 Transformer fits is past that and throughput-bound (two thread groups per workgroup cost
    // it 3 %, 15 fits 2 %, 16 LSTM fits 15 %).  The longest K loop decides: a decoder weight gradient (K = the batch's 50 rows, one
    // tile -- its second thread group idles) shares its launch with the data gradient (K = 512: 8 steps -> 4), and the launch lasts
    // as long as that chain (cfg2 step 2.78 -> 2.75 ms).
    return blocks <= 256 && longest_ktiles >= 4 ? 2 : 1;
}
static int pick_ks(const slnlp_gemm_args* jobs, int njobs, int blocks) {
    int longest = 0;
    for (int i = 0; i < njobs; ++i) longest = std::max(longest, ceil_div(jobs[i].K, BKT));
    return gemm_group_ks(blocks, longest);
}

template <int NSPLIT, bool AK, bool BK, bool VEC>
static void launch2(const GemmParams& p, hipStream_t s) {
    // Skinny problems (one row-block) get 16-column tiles: 4x the workgroups, so a
    // [50 x 512] x [512 x 512] decoder GEMM runs on 32 CUs instead of 8.
    const bool narrow = p.a.M <= BM && p.a.rowsum_a == nullptr;
    const dim3 grid(ceil_div(p.a.N, narrow ? 16 : 64), ceil_div(p.a.M, BM));
    const int ks = VEC ? pick_ks(&p.a, 1, (int)(grid.x * grid.y)) : 1;
    if (narrow) {
        if (ks == 2) hipLaunchKernelGGL((gemm_kernel<NSPLIT, AK, BK, 16, VEC, VEC ? 2 : 1>), grid, dim3(VEC ? 512 : 256), 0, s, p);
        else hipLaunchKernelGGL((gemm_kernel<NSPLIT, AK, BK, 16, VEC>), grid, dim3(256), 0, s, p);
    } else {   // (64x32 tiles for ~1-block-per-CU grids were measured: no gain, 17.6 vs 16.0 us)
        if (ks == 2) hipLaunchKernelGGL((gemm_kernel<NSPLIT, AK, BK, 64, VEC, VEC ? 2 : 1>), grid, dim3(VEC ? 512 : 256), 0, s, p);
        else hipLaunchKernelGGL((gemm_kernel<NSPLIT, AK, BK, 64, VEC>), grid, dim3(256), 0, s, p);
    }
}

template <int NSPLIT, bool AK, bool BK>
static void launch(const GemmParams& p, hipStream_t s) {
    // 16-byte vector loads need both operands 16-B aligned with ld % 4 == 0; anything else
    // (e.g. an unpadded [B, 202] matrix) takes the scalar-load build of the same kernel.
    if (p.a_vec && p.b_vec) launch2<NSPLIT, AK, BK, true>(p, s);
    else launch2<NSPLIT, AK, BK, false>(p, s);
}

static bool vec_ok(const float* ptr, long ld) {
    return (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0);
}

static int fill_params(const slnlp_gemm_args& a, GemmParams& p) {
    SLNLP_CHECK_ARG(a.A && a.B && a.C, "gemm: null operand");
    SLNLP_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "gemm: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    SLNLP_CHECK_ARG(a.precision == 1 || a.precision == 3, "gemm: precision must be 1 or 3, got %d", a.precision);
    SLNLP_CHECK_ARG(a.lda >= (a.a_kmajor ? a.K : a.M), "gemm: lda %ld too small", (long)a.lda);
    SLNLP_CHECK_ARG(a.ldb >= (a.b_kmajor ? a.K : a.N), "gemm: ldb %ld too small", (long)a.ldb);
    SLNLP_CHECK_ARG(a.ldc >= a.N, "gemm: ldc %ld < N %d", (long)a.ldc, a.N);
    SLNLP_CHECK_ARG(a.drop_p >= 0.f && a.drop_p < 1.f, "gemm: dropout p=%f out of [0,1)", a.drop_p);
    SLNLP_CHECK_ARG(a.drop_p == 0.f || a.rng, "gemm: dropout needs rng state");
    SLNLP_CHECK_ARG(!a.gate || a.ldg >= a.N, "gemm: ldg too small");
    SLNLP_CHECK_ARG(!a.resid || a.ldr >= a.N, "gemm: ldr too small");
    SLNLP_CHECK_ARG(!(a.a_kmajor == 0 && a.b_kmajor != 0), "gemm: layout (A m-major, B k-major) not built");
    SLNLP_CHECK_ARG(a.drop_head_dim >= 0 && (a.drop_head_dim == 0 || a.N % a.drop_head_dim == 0),
                    "gemm: drop_head_dim %d does not divide N %d", a.drop_head_dim, a.N);
    SLNLP_CHECK_ARG(!a.C_hi || a.ldc_p >= a.N, "gemm: ldc_p < N");
    p.a = a;
    p.drop_thr = dropout_threshold(a.drop_p);
    p.drop_scale = 1.f / (1.f - a.drop_p);
    p.a_vec = vec_ok(a.A, a.lda);
    p.b_vec = vec_ok(a.B, a.ldb);
    return 0;
}

int gemm(const slnlp_gemm_args& a, hipStream_t s) {
    if (a.A_hi || a.B_hi) return gemm_planes(a, s);   // pre-split operands: LDS-DMA kernel (gemm_planes.hip)
    if (recording()) return gemm_group(&a, 1, s);     // lockstep: every GEMM is a job of a grouped launch (same tile code)
    GemmParams p;
    SLNLP_TRY(fill_params(a, p));
    const bool ak = a.a_kmajor != 0, bk = a.b_kmajor != 0;
    if (a.precision == 3) {
        if (ak && bk) launch<3, true, true>(p, s);
        else if (ak) launch<3, true, false>(p, s);
        else launch<3, false, false>(p, s);
    } else {
        if (ak && bk) launch<1, true, true>(p, s);
        else if (ak) launch<1, true, false>(p, s);
        else launch<1, false, false>(p, s);
    }
    SLNLP_CHECK_LAUNCH("gemm");
    return SLNLP_OK;
}

// fp32-operand jobs in one launch; a job with operands that cannot take 16-B vector loads makes the whole group
// fall back to one launch per job (same results, just not fused)
// wide_mask bit i: job i takes 64-column tiles although it has one block of rows (16-column tiles spread a skinny product over 4 x
// the CUs, which pays while its K loop is long; a product cut into short K-slices has its workgroups from the slices, and wide tiles
// quarter the re-reads of its A rows).  Same K order: same bits either way.
int gemm_group(const slnlp_gemm_args* jobs, int njobs, hipStream_t s, unsigned wide_mask) {
    SLNLP_CHECK_ARG(jobs && njobs >= 1 && njobs <= GEMM_GROUP_MAX, "gemm_group: 1..%d jobs", GEMM_GROUP_MAX);
    GemmGroupParams P;
    P.njobs = njobs;
    int blocks = 0;
    bool fusable = true;
    for (int i = 0; i < njobs; ++i) {
        const slnlp_gemm_args& a = jobs[i];
        SLNLP_CHECK_ARG(!a.A_hi && !a.B_hi, "gemm_group: fp32 and pre-split jobs cannot share a launch");
        SLNLP_CHECK_ARG(a.precision == jobs[0].precision, "gemm_group: jobs of one launch share the precision");
        SLNLP_TRY(fill_params(a, P.job[i]));
        fusable = fusable && P.job[i].a_vec && P.job[i].b_vec;
        const bool narrow = a.M <= BM && a.rowsum_a == nullptr && !((wide_mask >> i) & 1u);
        P.variant[i] = (a.a_kmajor && a.b_kmajor ? 0 : a.a_kmajor ? 2 : 4) + (narrow ? 1 : 0);
        P.gx[i] = ceil_div(a.N, narrow ? 16 : 64);
        P.gy[i] = ceil_div(a.M, BM);
        P.block_begin[i] = blocks;
        const int nb = a.batch > 1 ? a.batch : 1;
        SLNLP_CHECK_ARG(nb == 1 || (!a.bias && !a.gate && !a.rowsum_a && a.drop_p == 0.f), "gemm_group: a batched job takes a residual only");
        SLNLP_CHECK_ARG(nb == 1 || (a.batch_stride_a % 4 == 0 && a.batch_stride_b % 4 == 0), "gemm_group: batch strides must keep 16-byte alignment");
        blocks += P.gx[i] * P.gy[i] * nb;
    }
    const int ks = pick_ks(jobs, njobs, blocks);
    if (recording()) {
        SLNLP_CHECK_ARG(fusable, "gemm_group: an operand that cannot take 16-byte loads cannot join a lockstep launch");
        return record_op(gemm_group_kernel_ptr(jobs[0].precision, ks), dim3(blocks), dim3(256 * ks), 0, REC_GEMM_GROUP, &P, sizeof(P), "gemm_group");
    }
    bool batched = false;
    for (int i = 0; i < njobs; ++i) batched = batched || jobs[i].batch > 1;
    if (!fusable || (njobs == 1 && !batched)) {
        for (int i = 0; i < njobs; ++i) {
            const int nb = jobs[i].batch > 1 ? jobs[i].batch : 1;
            for (int z = 0; z < nb; ++z) {                     // (operands without 16-byte alignment: one launch per GEMM)
                slnlp_gemm_args a = jobs[i];
                a.batch = 0;
                a.A += (long)z * jobs[i].batch_stride_a; a.B += (long)z * jobs[i].batch_stride_b; a.C += (long)z * jobs[i].batch_stride_c;
                if (a.resid) a.resid += (long)z * jobs[i].batch_stride_c;
                if (a.C_hi) a.C_hi += (long)z * jobs[i].batch_stride_c;
                if (a.C_lo) a.C_lo += (long)z * jobs[i].batch_stride_c;
                SLNLP_TRY(gemm(a, s));
            }
        }
        return SLNLP_OK;
    }
    void* args[3] = {(void*)&P, nullptr, nullptr};
    const GemmJob* no_tab = nullptr;
    const int* no_map = nullptr;
    args[1] = (void*)&no_tab; args[2] = (void*)&no_map;
    if (hipLaunchKernel(gemm_group_kernel_ptr(jobs[0].precision, ks), dim3(blocks), dim3(256 * ks), args, 0, s) != hipSuccess) {
        set_error("gemm_group: launch failed: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    SLNLP_CHECK_LAUNCH("gemm_group");
    return SLNLP_OK;
}

const void* gemm_group_kernel_ptr(int precision, int ks) {
    if (precision == 3) return ks == 2 ? (const void*)gemm_group_kernel<3, 2> : (const void*)gemm_group_kernel<3, 1>;
    return ks == 2 ? (const void*)gemm_group_kernel<1, 2> : (const void*)gemm_group_kernel<1, 1>;
}

// ------------------------------------------------------------- fused recurrent step ---
// One forward timestep of an LSTM / GRU layer (up to two directions) in ONE launch: the recurrent GEMM
// h_{t-1} W_hh^T and the point-wise cell (rnn.hip rnn_cell_fwd_kernel) that consumes it.  A workgroup owns 16 hidden
// units: it computes their G gate pre-activations for all B rows (G B-tiles of 16 weight rows, one shared A tile per
// K-step, same split-bf16 K order as gemm_tile, so results are bit-identical to GEMM + cell) and applies the cell in
// the accumulator layout -- every lane holds all G gates of its (row, unit) pairs.  The new state goes to a DIFFERENT
// buffer than the one read (other workgroups still read h_{t-1}): the caller chains the per-timestep `hprev` slots.
struct RnnStepParams {
    slnlp_rnn_step_dir d[2];
    int B, Hd, ndir;
    const long* lengths;
    float fill;
    long ld_out;
    float drop_p;
    unsigned drop_thr;
    int drop_site;
    const unsigned long long* rng;
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ slnlp_rnn_step_dir as_global(slnlp_rnn_step_dir d) {
    d.h_in = as_global(d.h_in); d.h_out = as_global(d.h_out); d.w_hh = as_global(d.w_hh); d.b_hh = as_global(d.b_hh);
    d.xproj = as_global(d.xproj); d.c = as_global(d.c); d.cprev_save = as_global(d.cprev_save); d.acts = as_global(d.acts);
    d.hn_save = as_global(d.hn_save); d.out = as_global(d.out);
    return d;
}

// ------------------------------------------------------------------------------------------ fused backward timestep ---
// One launch per backward timestep (rounds 1-3: a cell kernel + a grouped K-sliced GEMM launch, 192 + 192 launches per cfg3 step):
//   dh(t) = dgh(t+1) W_hh + carry(t+1)            [B, Hd]     recurrent data gradient of the step processed just before
//   cell backward of step t (rnn.hip, rnn_cell_bwd_body: same arithmetic, same order)  ->  dgx(t), dgh(t), dc, carry(t)
// A workgroup owns 16 hidden units (output columns of the GEMM) of one direction and 64 batch rows.  The contraction runs over
// the G * Hd gate columns: G groups of 256 threads, group g contracting gate g's Hd columns with its own stage images (all G K
// loops in flight together: the dependent chain is Hd / 64 steps, as in the K-sliced launch it replaces); the groups' partial
// sums meet in LDS and are added in gate order -- ((P0 + carry) + P1) + P2 (+ P3), the order of the unfused path -- and group 0
// applies the cell in the accumulator layout.  dgh_next == NULL: first step of a layer, dh = dh_state (no product).
struct RnnStepBwdParams {
    slnlp_rnn_step_bwd_dir d[2];
    int B, Hd, ndir;
    const long* lengths;
    long ld_dout;
    float drop_p;
    unsigned drop_thr;
    int drop_site;
    const unsigned long long* rng;
};
__device__ __forceinline__ slnlp_rnn_step_bwd_dir as_global(slnlp_rnn_step_bwd_dir d) {
    d.cell = as_global(d.cell);
    d.dgh_next = as_global(d.dgh_next); d.w_hh = as_global(d.w_hh);
    return d;
}
template <int NSPLIT, bool LSTM>
constexpr int rnn_step_bwd_group_elems() {
    return (NSPLIT == 3 ? 2 : 1) * (TileIO<true, BM>::PLANE + TileIO<false, 16>::PLANE);
}
template <int NSPLIT, bool LSTM>
constexpr size_t rnn_step_bwd_lds() {
    constexpr int G = LSTM ? 4 : 3;
    return (size_t)G * rnn_step_bwd_group_elems<NSPLIT, LSTM>() * sizeof(unsigned short) + (size_t)(G - 1) * 256 * sizeof(f32x4);
}

template <int NSPLIT, bool LSTM>
__global__ __launch_bounds__(LSTM ? 1024 : 768) void rnn_step_bwd_kernel(const RnnStepBwdParams P0, const RnnStepBwdParams* __restrict__ tab) {
    extern __shared__ __attribute__((aligned(16))) unsigned short bsm[];
    RnnStepBwdParams P;
    if (tab) P = tab[blockIdx.z];
    else P = P0;
    P.lengths = as_global(P.lengths);
    P.rng = as_global(P.rng);
    constexpr int G = LSTM ? 4 : 3;
    constexpr int NP = NSPLIT == 3 ? 2 : 1;
    using TA = TileIO<true, BM>;
    using TB = TileIO<false, 16>;
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    unsigned short* As = bsm + grp * rnn_step_bwd_group_elems<NSPLIT, LSTM>();
    unsigned short* Bs = As + NP * TA::PLANE;
    f32x4* red = reinterpret_cast<f32x4*>(bsm + G * rnn_step_bwd_group_elems<NSPLIT, LSTM>());
    const int dir = blockIdx.y % P.ndir;
    const slnlp_rnn_step_bwd_dir sd = as_global(dir == 0 ? P.d[0] : P.d[1]);
    const slnlp_rnn_cell_bwd_dir& d = sd.cell;
    const int B = P.B, Hd = P.Hd, GH = G * Hd, j0 = blockIdx.x * 16, bm0 = (blockIdx.y / P.ndir) * BM;
    const bool product = sd.dgh_next != nullptr;       // (launch-uniform per direction)

    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (product) {
        const int ktiles = Hd / BKT, kg = grp * Hd;    // this group's gate columns [kg, kg + Hd)
        float4 ra0[TA::NV], ra1[TA::NV], rb0[TB::NV], rb1[TB::NV];
        auto fetch = [&](int kt, float4 (&ra)[TA::NV], float4 (&rb)[TB::NV]) {
            const int kc = kt < ktiles ? kt : 0;           // past-the-end prefetch: a valid tile, never stashed
            TA::template fetch<true>(sd.dgh_next, GH, bm0, B, kg + kc * BKT, GH, tid, ra);
            TB::template fetch<true>(sd.w_hh, Hd, j0, Hd, kg + kc * BKT, GH, tid, rb);
        };
        auto stash = [&](int kt, const float4 (&ra)[TA::NV], const float4 (&rb)[TB::NV]) {
            // rows >= B hold a clamped row's data and only feed accumulator rows that are never used (no masks: Hd % 64 == 0)
            TA::template stash<NSPLIT, false>(As, tid, ra, bm0, B, kg + kt * BKT, GH);
            TB::template stash<NSPLIT, false>(Bs, tid, rb, j0, Hd, kg + kt * BKT, GH);
        };
        auto consume = [&]() {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 ah = TA::frag(As, wave * 16, kk, lane);
                const bf16x8 bh = TB::frag(Bs, 0, kk, lane);
                if (NSPLIT == 3) {
                    const bf16x8 al = TA::frag(As + TA::PLANE, wave * 16, kk, lane);
                    const bf16x8 bl = TB::frag(Bs + TB::PLANE, 0, kk, lane);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
            }
        };
        fetch(0, ra0, rb0);
        fetch(1, ra1, rb1);
        for (int kt = 0; kt < ktiles; kt += 2) {
            lds_barrier();
            stash(kt, ra0, rb0);
            lds_barrier();
            fetch(kt + 2, ra0, rb0);
            consume();
            if (kt + 1 >= ktiles) break;
            lds_barrier();
            stash(kt + 1, ra1, rb1);
            lds_barrier();
            fetch(kt + 3, ra1, rb1);
            consume();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the dummy prefetches
        if (grp > 0) red[(grp - 1) * 256 + tid] = acc;
        __syncthreads();
    }
    if (grp != 0) return;

    // ---- the cell backward of this timestep in the accumulator layout (rnn.hip rnn_cell_bwd_body: same arithmetic and order)
    const int j = j0 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = bm0 + wave * 16 + ((lane >> 4) << 2) + r;
        if (b >= B) break;
        const long idx = (long)b * Hd + j;
        const bool valid = P.lengths ? (d.t < P.lengths[b]) : true;
        float dh;
        if (product) {
            dh = acc[r] + d.carry[idx];                      // (the unfused path adds carry as the first job's residual)
#pragma unroll
            for (int e = 0; e < G - 1; ++e) dh += red[e * 256 + tid][r];
        } else {
            dh = d.dh_state[idx];
        }
        float* gx = d.dgx + (long)b * GH;
        float* gh = LSTM ? gx : d.dgh + (long)b * GH;
        if (!valid) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                gx[g * Hd + j] = 0.f;
                if (!LSTM) gh[g * Hd + j] = 0.f;
            }
            d.carry[idx] = dh;
            continue;
        }
        if (d.dout) {
            float g = d.dout[(long)b * P.ld_dout + j];
            if (P.drop_p > 0.f)
                g = dropout_keep(P.rng, P.drop_site, (unsigned)(d.out_row0 + b), (unsigned)(d.out_col0 + j), P.drop_thr)
                        ? g / (1.f - P.drop_p) : 0.f;
            dh += g;
        }
        const float* a = d.acts + (long)b * GH;
        if constexpr (LSTM) {
            const float gi = a[j], gf = a[Hd + j], gg = a[2 * Hd + j], go = a[3 * Hd + j];
            const float cprev = d.cprev_save[idx];
            const float tc = tanhf(gf * cprev + gi * gg);
            const float dc = d.dc_state[idx] + dh * go * (1.f - tc * tc);
            gx[j] = dc * gg * gi * (1.f - gi);
            gx[Hd + j] = dc * cprev * gf * (1.f - gf);
            gx[2 * Hd + j] = dc * gi * (1.f - gg * gg);
            gx[3 * Hd + j] = dh * tc * go * (1.f - go);
            d.dc_state[idx] = dc * gf;
            d.carry[idx] = 0.f;
        } else {
            const float rr = a[j], z = a[Hd + j], nn = a[2 * Hd + j];
            const float hprev = d.hprev_save[idx], hn = d.hn_save[idx];
            const float dn_pre = dh * (1.f - z) * (1.f - nn * nn);
            const float dr_pre = dn_pre * hn * rr * (1.f - rr);
            const float dz_pre = dh * (hprev - nn) * z * (1.f - z);
            gx[j] = dr_pre; gx[Hd + j] = dz_pre; gx[2 * Hd + j] = dn_pre;
            gh[j] = dr_pre; gh[Hd + j] = dz_pre; gh[2 * Hd + j] = dn_pre * rr;
            d.carry[idx] = dh * z;
        }
    }
}

template <int NSPLIT, bool LSTM>
static const void* rnn_step_bwd_fn() { return (const void*)rnn_step_bwd_kernel<NSPLIT, LSTM>; }

// raise the kernels' dynamic LDS limit once per device (plan creation: never inside a graph capture)
int rnn_step_bwd_init() {
    static DeviceOnce once;
    return once.run([]() -> int {
        const bool ok =
            hipFuncSetAttribute(rnn_step_bwd_fn<3, true>(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rnn_step_bwd_lds<3, true>()) == hipSuccess &&
            hipFuncSetAttribute(rnn_step_bwd_fn<3, false>(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rnn_step_bwd_lds<3, false>()) == hipSuccess &&
            hipFuncSetAttribute(rnn_step_bwd_fn<1, true>(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rnn_step_bwd_lds<1, true>()) == hipSuccess &&
            hipFuncSetAttribute(rnn_step_bwd_fn<1, false>(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rnn_step_bwd_lds<1, false>()) == hipSuccess;
        if (!ok) {
            set_error("rnn_step_bwd_init: cannot raise dynamic LDS limit: %s", hipGetErrorString(hipGetLastError()));
            return SLNLP_ERR_LAUNCH;
        }
        return 0;
    });
}

bool rnn_step_bwd_covers(int B, int Hd) { return Hd % 64 == 0 && B > 0; }

int rnn_step_bwd(int lstm, const slnlp_rnn_step_bwd_dir* dirs, int ndir, int B, int Hd, const int64_t* lengths, int64_t ld_dout,
                 float drop_p, int drop_site, const unsigned long long* rng, int precision, hipStream_t st) {
    SLNLP_CHECK_ARG(dirs && (ndir == 1 || ndir == 2) && rnn_step_bwd_covers(B, Hd), "rnn_step_bwd: bad args (Hd %% 64 == 0)");
    SLNLP_CHECK_ARG(precision == 1 || precision == 3, "rnn_step_bwd: precision must be 1 or 3");
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "rnn_step_bwd: bad dropout args");
    const int G = lstm ? 4 : 3;
    RnnStepBwdParams P;
    for (int k = 0; k < ndir; ++k) {
        const slnlp_rnn_step_bwd_dir& d = dirs[k];
        SLNLP_CHECK_ARG(d.cell.dc_state || !lstm, "rnn_step_bwd: dc_state missing in direction %d", k);
        SLNLP_CHECK_ARG(d.cell.acts && d.cell.dgx && d.cell.carry && (lstm ? d.cell.cprev_save != nullptr : (d.cell.hprev_save && d.cell.hn_save && d.cell.dgh)),
                        "rnn_step_bwd: null pointer in direction %d", k);
        SLNLP_CHECK_ARG(d.dgh_next ? (d.w_hh && vec_ok(d.dgh_next, (long)G * Hd) && vec_ok(d.w_hh, Hd)) : d.cell.dh_state != nullptr,
                        "rnn_step_bwd: direction %d needs {dgh_next, w_hh} (16-byte aligned) or dh_state", k);
        SLNLP_CHECK_ARG((dirs[0].dgh_next != nullptr) == (d.dgh_next != nullptr), "rnn_step_bwd: the directions of a launch are both first steps or both not");
        P.d[k] = d;
    }
    if (ndir == 1) P.d[1] = P.d[0];
    P.B = B; P.Hd = Hd; P.ndir = ndir; P.lengths = (const long*)lengths; P.ld_dout = ld_dout;
    P.drop_p = drop_p; P.drop_thr = dropout_threshold(drop_p); P.drop_site = drop_site; P.rng = rng;
    SLNLP_TRY(rnn_step_bwd_init());
    const dim3 grid(Hd / 16, ndir * ceil_div(B, BM));
    const dim3 block(G * 256);
    const void* fn = precision == 3 ? (lstm ? rnn_step_bwd_fn<3, true>() : rnn_step_bwd_fn<3, false>())
                                    : (lstm ? rnn_step_bwd_fn<1, true>() : rnn_step_bwd_fn<1, false>());
    const size_t lds = precision == 3 ? (lstm ? rnn_step_bwd_lds<3, true>() : rnn_step_bwd_lds<3, false>())
                                      : (lstm ? rnn_step_bwd_lds<1, true>() : rnn_step_bwd_lds<1, false>());
    if (recording()) return record_op(fn, grid, block, lds, REC_Z, &P, sizeof(P), "rnn_step_bwd");
    const RnnStepBwdParams* tab = nullptr;
    void* args[2] = {&P, &tab};
    if (hipLaunchKernel(fn, grid, block, args, lds, st) != hipSuccess) {
        set_error("rnn_step_bwd: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    return SLNLP_OK;
}

// grid (Hd / 16, ndir x row tiles, fit): `tab` != nullptr is a lockstep launch, fit z takes tab[z] (launch.hpp)
// KS = 2: two groups of 256 threads, one half of the K tiles each (gemm_tile's scheme: the K sum is two halves by definition, so the
// bits do not depend on KS) -- a solo fit's 64-workgroup launch is a chain of 8 K steps, a merged lockstep launch takes KS = 1.
template <int NSPLIT, bool LSTM, bool EDGE, int KS = 1>
__global__ __launch_bounds__(256 * KS) void rnn_step_fwd_kernel(const RnnStepParams P0, const RnnStepParams* __restrict__ tab) {
    RnnStepParams P;
    if (tab) P = tab[blockIdx.z];
    else P = P0;
    P.lengths = as_global(P.lengths);
    P.rng = as_global(P.rng);
    constexpr int G = LSTM ? 4 : 3;
    constexpr int NP = NSPLIT == 3 ? 2 : 1;
    using TA = TileIO<true, BM>;
    using TB = TileIO<true, 16>;
    __shared__ __attribute__((aligned(16))) unsigned short As_all[KS * NP * TA::PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs_all[KS * G * NP * TB::PLANE];
    __shared__ f32x4 red[KS == 2 ? G * 256 : 1];               // group 1's half of the K sum on its way to group 0
    const int grp = KS == 2 ? (int)(threadIdx.x >> 8) : 0;
    unsigned short* As = As_all + grp * NP * TA::PLANE;
    unsigned short* Bs = Bs_all + grp * G * NP * TB::PLANE;
    const int dir = blockIdx.y % P.ndir;
    const slnlp_rnn_step_dir d = as_global(dir == 0 ? P.d[0] : P.d[1]);
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int B = P.B, Hd = P.Hd, j0 = blockIdx.x * 16, bm0 = (blockIdx.y / P.ndir) * BM;
    const int K = Hd, ktiles = (K + BKT - 1) / BKT;

    f32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ra0[TA::NV], ra1[TA::NV], rb0[G][TB::NV], rb1[G][TB::NV];
    auto fetch = [&](int kt, float4 (&ra)[TA::NV], float4 (&rb)[G][TB::NV]) {
        TA::template fetch<true>(d.h_in, Hd, bm0, B, kt * BKT, K, tid, ra);
#pragma unroll
        for (int g = 0; g < G; ++g) TB::template fetch<true>(d.w_hh + (long)g * Hd * Hd, Hd, j0, Hd, kt * BKT, K, tid, rb[g]);
    };
    auto stash = [&](int kt, const float4 (&ra)[TA::NV], const float4 (&rb)[G][TB::NV]) {
        // EDGE = false (Hd % 64 == 0): no masks at all -- rows >= B hold a clamped row's data and only feed accumulator
        // rows that are never stored
        TA::template stash<NSPLIT, EDGE>(As, tid, ra, bm0, B, kt * BKT, K);
#pragma unroll
        for (int g = 0; g < G; ++g) TB::template stash<NSPLIT, EDGE>(Bs + g * NP * TB::PLANE, tid, rb[g], j0, Hd, kt * BKT, K);
    };
    auto consume = [&]() {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 ah = TA::frag(As, wave * 16, kk, lane);
            bf16x8 al = ah;
            if (NSPLIT == 3) al = TA::frag(As + TA::PLANE, wave * 16, kk, lane);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const unsigned short* bt = Bs + g * NP * TB::PLANE;
                const bf16x8 bh = TB::frag(bt, 0, kk, lane);
                if (NSPLIT == 3) {
                    const bf16x8 bl = TB::frag(bt + TB::PLANE, 0, kk, lane);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[g], 0, 0, 0);
                }
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[g], 0, 0, 0);
            }
        }
    };
    // the cell's own operands are requested first, so their latency hides behind the K loop
    const int j = j0 + (lane & 15), jj = j < Hd ? j : 0;
    float bh[G], xpv[4][G], hpv[4], cpv[4];
#pragma unroll
    for (int g = 0; g < G; ++g) bh[g] = d.b_hh ? d.b_hh[g * Hd + jj] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = bm0 + wave * 16 + ((lane >> 4) << 2) + r, bb = b < B ? b : 0;
#pragma unroll
        for (int g = 0; g < G; ++g) xpv[r][g] = d.xproj[(long)bb * G * Hd + g * Hd + jj];
        hpv[r] = d.h_in[(long)bb * Hd + jj];
        cpv[r] = LSTM ? d.c[(long)bb * Hd + jj] : 0.f;
    }
    // the K sum in two halves, tiles [0, T) and [T, ktiles), first + second: gemm_tile's definition (bit-identical to GEMM + cell)
    const int half = (ktiles + 1) / 2;
    if constexpr (KS == 1) {
        f32x4 acc_first[G];
        auto park = [&]() {
#pragma unroll
            for (int g = 0; g < G; ++g) { acc_first[g] = acc[g]; acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        };
        fetch(0, ra0, rb0);
        fetch(1, ra1, rb1);
        for (int kt = 0; kt < ktiles; kt += 2) {
            if (kt == half) park();
            lds_barrier();
            stash(kt, ra0, rb0);
            lds_barrier();
            fetch(kt + 2, ra0, rb0);
            consume();
            if (kt + 1 >= ktiles) break;
            if (kt + 1 == half) park();
            lds_barrier();
            stash(kt + 1, ra1, rb1);
            lds_barrier();
            fetch(kt + 3, ra1, rb1);
            consume();
        }
        if (ktiles <= half) park();
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = acc_first[g] + acc[g];
    } else {
        // group g: tiles [k0, k1); both groups make `half` steps (the barriers are the workgroup's), group 1 idles in its last one
        // when the number of tiles is odd; past-the-end prefetches read a clamped, valid tile and are never stashed
        const int k0 = grp * half, k1 = grp == 0 ? half : ktiles;
        fetch(k0, ra0, rb0);
        fetch(k0 + 1, ra1, rb1);
        for (int it = 0; it < half; it += 2) {
            const int kt = k0 + it;
            lds_barrier();
            if (kt < k1) stash(kt, ra0, rb0);
            lds_barrier();
            fetch(kt + 2, ra0, rb0);
            if (kt < k1) consume();
            if (it + 1 >= half) break;
            lds_barrier();
            if (kt + 1 < k1) stash(kt + 1, ra1, rb1);
            lds_barrier();
            fetch(kt + 3, ra1, rb1);
            if (kt + 1 < k1) consume();
        }
        if (grp == 1) {
#pragma unroll
            for (int g = 0; g < G; ++g) red[g * 256 + tid] = acc[g];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = acc[g] + red[g * 256 + tid];
    }

    // ---- cell (same arithmetic and order as rnn_cell_fwd_kernel)
    if (j >= Hd) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = bm0 + wave * 16 + ((lane >> 4) << 2) + r;
        if (b >= B) break;
        const long idx = (long)b * Hd + j;
        const bool valid = P.lengths ? (d.t < P.lengths[b]) : true;
        float* a = d.acts + (long)b * G * Hd;
        const float hprev = hpv[r];
        float hnew;
        if constexpr (LSTM) {
            const float cprev = cpv[r];
            const float gi = sigm(xpv[r][0] + (acc[0][r] + bh[0]));
            const float gf = sigm(xpv[r][1] + (acc[1][r] + bh[1]));
            const float gg = tanhf(xpv[r][2] + (acc[2][r] + bh[2]));
            const float go = sigm(xpv[r][3] + (acc[3][r] + bh[3]));
            const float cnew = gf * cprev + gi * gg;
            hnew = go * tanhf(cnew);
            a[j] = gi; a[Hd + j] = gf; a[2 * Hd + j] = gg; a[3 * Hd + j] = go;
            d.cprev_save[idx] = cprev;
            d.c[idx] = valid ? cnew : cprev;
        } else {
            const float hn = acc[2][r] + bh[2];
            const float rr = sigm(xpv[r][0] + (acc[0][r] + bh[0]));
            const float z = sigm(xpv[r][1] + (acc[1][r] + bh[1]));
            const float nn = tanhf(xpv[r][2] + rr * hn);
            hnew = (1.f - z) * nn + z * hprev;
            a[j] = rr; a[Hd + j] = z; a[2 * Hd + j] = nn;
            d.hn_save[idx] = hn;
        }
        d.h_out[idx] = valid ? hnew : hprev;
        if (d.out) {
            float o = valid ? hnew : P.fill;
            if (P.drop_p > 0.f && valid)
                o = dropout_keep(P.rng, P.drop_site, (unsigned)(d.out_row0 + b), (unsigned)(d.out_col0 + j), P.drop_thr)
                        ? o / (1.f - P.drop_p) : 0.f;
            d.out[(long)b * P.ld_out + j] = o;
        }
    }
}

// The same timestep RE-TILED for a solo fit's launch: a workgroup owns 16 batch rows x 16 hidden units x all G gates (grid Hd / 16 x
// ndir x row tiles of 16: 256 workgroups at B = 50, Hd = 512, two directions -- every CU -- instead of 64), WAVE g computes gate g's
// 16 x 16 tile, and wave 0 applies the cell once the gates have met in LDS.  Why: such a launch lasts as long as one workgroup takes
// to LOAD its operands (gemm_rows.hip measured the same for the decoder's products: ~33 GB/s per compute unit), and the 64-row
// tile above pulls 128 KB of h beside its 128 KB of W_hh per workgroup where this one pulls 32 + 128.  Same K order, same halves,
// same cell arithmetic per element: bit-identical to rnn_step_fwd_kernel (tests/test_rnn_gpu.py), so a merged lockstep launch --
// which pays for total bytes, not for one workgroup's -- keeps the 64-row kernel (rnn_step_fwd_for_blocks).
template <int NSPLIT, bool LSTM, bool EDGE, int KS = 1>
__global__ __launch_bounds__(256 * KS) void rnn_step_fwd_rt_kernel(const RnnStepParams P0, const RnnStepParams* __restrict__ tab) {
    RnnStepParams P;
    if (tab) P = tab[blockIdx.z];
    else P = P0;
    P.lengths = as_global(P.lengths);
    P.rng = as_global(P.rng);
    constexpr int G = LSTM ? 4 : 3;
    constexpr int NP = NSPLIT == 3 ? 2 : 1;
    using TA = TileIO<true, 16>;
    using TB = TileIO<true, 16>;
    __shared__ __attribute__((aligned(16))) unsigned short As_all[KS * NP * TA::PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs_all[KS * G * NP * TB::PLANE];
    __shared__ f32x4 red[(KS == 2 ? 4 : 0) * 64 + 4 * 64];     // group 1's half of the K sum; then the gates on their way to wave 0
    const int grp = KS == 2 ? (int)(threadIdx.x >> 8) : 0;
    unsigned short* As = As_all + grp * NP * TA::PLANE;
    unsigned short* Bs = Bs_all + grp * G * NP * TB::PLANE;
    const int dir = blockIdx.y % P.ndir;
    const slnlp_rnn_step_dir d = as_global(dir == 0 ? P.d[0] : P.d[1]);
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // wave = gate
    const int B = P.B, Hd = P.Hd, j0 = blockIdx.x * 16, bm0 = (blockIdx.y / P.ndir) * 16;
    const int K = Hd, ktiles = (K + BKT - 1) / BKT;

    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ra0[TA::NV], ra1[TA::NV], rb0[G][TB::NV], rb1[G][TB::NV];
    auto fetch = [&](int kt, float4 (&ra)[TA::NV], float4 (&rb)[G][TB::NV]) {
        TA::template fetch<true>(d.h_in, Hd, bm0, B, kt * BKT, K, tid, ra);
#pragma unroll
        for (int g = 0; g < G; ++g) TB::template fetch<true>(d.w_hh + (long)g * Hd * Hd, Hd, j0, Hd, kt * BKT, K, tid, rb[g]);
    };
    auto stash = [&](int kt, const float4 (&ra)[TA::NV], const float4 (&rb)[G][TB::NV]) {
        TA::template stash<NSPLIT, EDGE>(As, tid, ra, bm0, B, kt * BKT, K);
#pragma unroll
        for (int g = 0; g < G; ++g) TB::template stash<NSPLIT, EDGE>(Bs + g * NP * TB::PLANE, tid, rb[g], j0, Hd, kt * BKT, K);
    };
    auto consume = [&]() {
        if (wave >= G) return;                                     // (GRU: three gates, the fourth wave only stages)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 ah = TA::frag(As, 0, kk, lane);
            bf16x8 al = ah;
            if (NSPLIT == 3) al = TA::frag(As + TA::PLANE, 0, kk, lane);
            const unsigned short* bt = Bs + wave * NP * TB::PLANE;
            const bf16x8 bh = TB::frag(bt, 0, kk, lane);
            if (NSPLIT == 3) {
                const bf16x8 bl = TB::frag(bt + TB::PLANE, 0, kk, lane);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
        }
    };
    // the cell's own operands are requested first (by the wave that will apply it), so their latency hides behind the K loop
    const int j = j0 + (lane & 15), jj = j < Hd ? j : 0;
    float bh[G], xpv[4][G], hpv[4], cpv[4];
    if (grp == 0 && wave == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) bh[g] = d.b_hh ? d.b_hh[g * Hd + jj] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = bm0 + ((lane >> 4) << 2) + r, bb = b < B ? b : 0;
#pragma unroll
            for (int g = 0; g < G; ++g) xpv[r][g] = d.xproj[(long)bb * G * Hd + g * Hd + jj];
            hpv[r] = d.h_in[(long)bb * Hd + jj];
            cpv[r] = LSTM ? d.c[(long)bb * Hd + jj] : 0.f;
        }
    }
    // the K sum in two halves, tiles [0, T) and [T, ktiles), first + second: gemm_tile's definition
    const int half = (ktiles + 1) / 2;
    if constexpr (KS == 1) {
        f32x4 acc_first = f32x4{0.f, 0.f, 0.f, 0.f};
        auto park = [&]() { acc_first = acc; acc = f32x4{0.f, 0.f, 0.f, 0.f}; };
        fetch(0, ra0, rb0);
        fetch(1, ra1, rb1);
        for (int kt = 0; kt < ktiles; kt += 2) {
            if (kt == half) park();
            lds_barrier();
            stash(kt, ra0, rb0);
            lds_barrier();
            fetch(kt + 2, ra0, rb0);
            consume();
            if (kt + 1 >= ktiles) break;
            if (kt + 1 == half) park();
            lds_barrier();
            stash(kt + 1, ra1, rb1);
            lds_barrier();
            fetch(kt + 3, ra1, rb1);
            consume();
        }
        if (ktiles <= half) park();
        acc = acc_first + acc;
    } else {
        const int k0 = grp * half, k1 = grp == 0 ? half : ktiles;
        fetch(k0, ra0, rb0);
        fetch(k0 + 1, ra1, rb1);
        for (int it = 0; it < half; it += 2) {
            const int kt = k0 + it;
            lds_barrier();
            if (kt < k1) stash(kt, ra0, rb0);
            lds_barrier();
            fetch(kt + 2, ra0, rb0);
            if (kt < k1) consume();
            if (it + 1 >= half) break;
            lds_barrier();
            if (kt + 1 < k1) stash(kt + 1, ra1, rb1);
            lds_barrier();
            fetch(kt + 3, ra1, rb1);
            if (kt + 1 < k1) consume();
        }
        if (grp == 1) red[4 * 64 + wave * 64 + lane] = acc;
        __syncthreads();
        if (grp == 1) return;
        acc = acc + red[4 * 64 + wave * 64 + lane];
    }
    // the gates meet: wave g -> LDS -> wave 0
    red[wave * 64 + lane] = acc;
    lds_barrier();                                 // (KS = 2: group 1 has left; the barrier counts the waves that remain)
    if (wave != 0) return;
    f32x4 ga[G];
#pragma unroll
    for (int g = 0; g < G; ++g) ga[g] = red[g * 64 + lane];

    // ---- cell (same arithmetic and order as rnn_cell_fwd_kernel)
    if (j >= Hd) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = bm0 + ((lane >> 4) << 2) + r;
        if (b >= B) break;
        const long idx = (long)b * Hd + j;
        const bool valid = P.lengths ? (d.t < P.lengths[b]) : true;
        float* a = d.acts + (long)b * G * Hd;
        const float hprev = hpv[r];
        float hnew;
        if constexpr (LSTM) {
            const float cprev = cpv[r];
            const float gi = sigm(xpv[r][0] + (ga[0][r] + bh[0]));
            const float gf = sigm(xpv[r][1] + (ga[1][r] + bh[1]));
            const float gg = tanhf(xpv[r][2] + (ga[2][r] + bh[2]));
            const float go = sigm(xpv[r][3] + (ga[3][r] + bh[3]));
            const float cnew = gf * cprev + gi * gg;
            hnew = go * tanhf(cnew);
            a[j] = gi; a[Hd + j] = gf; a[2 * Hd + j] = gg; a[3 * Hd + j] = go;
            d.cprev_save[idx] = cprev;
            d.c[idx] = valid ? cnew : cprev;
        } else {
            const float hn = ga[2][r] + bh[2];
            const float rr = sigm(xpv[r][0] + (ga[0][r] + bh[0]));
            const float z = sigm(xpv[r][1] + (ga[1][r] + bh[1]));
            const float nn = tanhf(xpv[r][2] + rr * hn);
            hnew = (1.f - z) * nn + z * hprev;
            a[j] = rr; a[Hd + j] = z; a[2 * Hd + j] = nn;
            d.hn_save[idx] = hn;
        }
        d.h_out[idx] = valid ? hnew : hprev;
        if (d.out) {
            float o = valid ? hnew : P.fill;
            if (P.drop_p > 0.f && valid)
                o = dropout_keep(P.rng, P.drop_site, (unsigned)(d.out_row0 + b), (unsigned)(d.out_col0 + j), P.drop_thr)
                        ? o / (1.f - P.drop_p) : 0.f;
            d.out[(long)b * P.ld_out + j] = o;
        }
    }
}

// The kernel a MERGED launch of `blocks` workgroups runs in place of the recorded forward-step kernel `fn` (nullptr: `fn` is not one
// of them, or is the right one already): same results, the thread-group count of the merged size.
// kernel table: [precision 3 / 1][LSTM / GRU][EDGE][16-row re-tiled?][KS - 1]
static const void* rnn_step_kernel(int ns, bool lstm, bool edge, bool rt, int ks) {
#define SLNLP_RS(NS, L, E) (rt ? (ks == 2 ? (const void*)rnn_step_fwd_rt_kernel<NS, L, E, 2> : (const void*)rnn_step_fwd_rt_kernel<NS, L, E, 1>) \
                               : (ks == 2 ? (const void*)rnn_step_fwd_kernel<NS, L, E, 2> : (const void*)rnn_step_fwd_kernel<NS, L, E, 1>))
    if (ns == 3) {
        if (lstm) return edge ? SLNLP_RS(3, true, true) : SLNLP_RS(3, true, false);
        return edge ? SLNLP_RS(3, false, true) : SLNLP_RS(3, false, false);
    }
    if (lstm) return edge ? SLNLP_RS(1, true, true) : SLNLP_RS(1, true, false);
    return edge ? SLNLP_RS(1, false, true) : SLNLP_RS(1, false, false);
#undef SLNLP_RS
}
// slnlp_set_rnn_step_tile / SLNLP_RNN_STEP_RT=0: the 64-row tile for solo launches too (tests, A / B measurements; same bits)
static std::atomic<int> g_rnn_step_rt{[] { const char* e = getenv("SLNLP_RNN_STEP_RT"); return (e && atoi(e) == 0) ? 0 : 1; }()};
static bool rnn_step_rt_enabled() { return g_rnn_step_rt.load(std::memory_order_relaxed) != 0; }
// which tiling / thread groups a launch of `fits` timesteps [B x Hd, ndir directions] takes: the 16-row tile while it still fits the
// chip about twice over (a launch-latency chain: one fit), the 64-row tile (a third of the operand bytes in total) beyond
static void rnn_step_shape(int B, int Hd, int ndir, int fits, bool* rt, int* ks, dim3* grid) {
    const int gx = ceil_div(Hd, 16), rt_blocks = gx * ndir * ceil_div(B, 16) * fits;
    *rt = rnn_step_rt_enabled() && B > 16 && rt_blocks <= 512;
    *grid = dim3(gx, ndir * ceil_div(B, *rt ? 16 : BM));
    *ks = gemm_group_ks((int)(grid->x * grid->y) * fits, ceil_div(Hd, BKT));
}
// The kernel a MERGED launch of `fits` fits runs in place of the recorded forward-step kernel `fn` (nullptr: `fn` is not one of them):
// same results, the tiling and thread-group count of the merged size.  `grid`: in = the recorded (x, y), out = the merged one.
const void* rnn_step_fwd_for_blocks(const void* fn, const void* recorded_args, int fits, int* threads, dim3* grid) {
    for (int ns = 1; ns <= 3; ns += 2)
        for (int l = 0; l < 2; ++l)
            for (int e = 0; e < 2; ++e)
                for (int rt = 0; rt < 2; ++rt)
                    for (int ks = 1; ks <= 2; ++ks)
                        if (fn == rnn_step_kernel(ns, l != 0, e != 0, rt != 0, ks)) {
                            const RnnStepParams& P = *static_cast<const RnnStepParams*>(recorded_args);
                            bool mrt;
                            int mks;
                            rnn_step_shape(P.B, P.Hd, P.ndir, fits, &mrt, &mks, grid);
                            *threads = 256 * mks;
                            return rnn_step_kernel(ns, l != 0, e != 0, mrt, mks);
                        }
    return nullptr;
}

int rnn_step_fwd(int lstm, const slnlp_rnn_step_dir* dirs, int ndir, int B, int Hd, const int64_t* lengths, float fill,
                 int64_t ld_out, float drop_p, int drop_site, const unsigned long long* rng, int precision, hipStream_t st) {
    SLNLP_CHECK_ARG(dirs && (ndir == 1 || ndir == 2) && B > 0 && Hd > 0 && Hd % 4 == 0, "rnn_step_fwd: bad args (Hd %% 4 == 0)");
    SLNLP_CHECK_ARG(precision == 1 || precision == 3, "rnn_step_fwd: precision must be 1 or 3");
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "rnn_step_fwd: bad dropout args");
    RnnStepParams P;
    for (int k = 0; k < ndir; ++k) {
        const slnlp_rnn_step_dir& d = dirs[k];
        SLNLP_CHECK_ARG(d.h_in && d.h_out && d.h_in != d.h_out && d.w_hh && d.xproj && d.acts &&
                            (lstm ? (d.c && d.cprev_save) : (d.hn_save != nullptr)),
                        "rnn_step_fwd: null pointer (or h_in == h_out) in direction %d", k);
        SLNLP_CHECK_ARG(vec_ok(d.h_in, Hd) && vec_ok(d.w_hh, Hd), "rnn_step_fwd: h_in / w_hh must be 16-byte aligned");
        P.d[k] = d;
    }
    if (ndir == 1) P.d[1] = P.d[0];
    P.B = B; P.Hd = Hd; P.ndir = ndir; P.lengths = (const long*)lengths; P.fill = fill; P.ld_out = ld_out;
    P.drop_p = drop_p; P.drop_thr = dropout_threshold(drop_p); P.drop_site = drop_site; P.rng = rng;
    const bool edge = (Hd % BKT) != 0;
    // tile and thread groups for ONE fit's launch (a merged lockstep launch picks again for its size: rnn_step_fwd_for_blocks)
    bool rt;
    int ks;
    dim3 grid;
    rnn_step_shape(B, Hd, ndir, 1, &rt, &ks, &grid);
    const void* fn = rnn_step_kernel(precision, lstm != 0, edge, rt, ks);
    if (recording()) return record_op(fn, grid, dim3(256 * ks), 0, REC_Z, &P, sizeof(P), "rnn_step_fwd");
    const RnnStepParams* no_tab = nullptr;
    void* args[2] = {&P, &no_tab};
    if (hipLaunchKernel(fn, grid, dim3(256 * ks), args, 0, st) != hipSuccess) {
        set_error("rnn_step_fwd: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    return SLNLP_OK;
}

// ------------------------------------------------------- persistent recurrent layer ---
// ALL S timesteps of one bidirectional LSTM / GRU layer in ONE launch.  The per-timestep kernel above is ~5 us of
// launch latency plus a K loop that re-reads and re-converts the same W_hh slice 48 times; here a workgroup keeps its
// slice of W_hh (the G x 16 rows of its 16 hidden units, all K) in LDS as bf16 hi/lo for the whole sequence
// (128 KiB at Hd = 512) and only streams h_{t-1} per step.  The Hd/16 x ndir co-resident workgroups meet at a
// device-wide barrier between steps (sense-reversing counter, agent-scope atomics, bounded spin: 1.35 us for 64
// workgroups, tools/micro/grid_barrier.hip); the new state is written with sc1 (write-through) stores and drained
// before the barrier, and every h slot is written once and read only afterwards, so no workgroup can see a stale
// L1 / L2 line -- no fences (an agent-scope fence is a whole-L2 write-back on this part).
// Same K order and split as rnn_step_fwd_kernel -> bit-identical results.
// Measured (round 1): 13.9 us per timestep, no faster than the per-timestep launches (13.4 us) -- with one wave per SIMD
// the K loop (convert h_{t-1}, two block barriers per K tile, LDS fragment reads exposed in front of every MFMA group)
// costs ~8 us, and h_{t-1} arrives from the memory side.  It is therefore OPT-IN (slnlp_rnn_set_persistent); the plan
// for it: 8 waves (two per SIMD, gates split over wave pairs) and the state exchanged as bf16 planes via LDS-DMA.
struct RnnLayerParams {
    slnlp_rnn_layer_dir d[2];
    int B, Hd, S;
    const long* lengths;
    float fill;
    long ld_out;
    float drop_p;
    unsigned drop_thr;
    int drop_site;
    const unsigned long long* rng;
    unsigned* bar;      // {count, generation}
    int* err;
};

__device__ __forceinline__ void grid_barrier_sr(unsigned* bar, int* err, unsigned nblocks) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's sc1 stores have reached the memory side
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned gen = __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the reset lands before anyone is released
            __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            long spins = 0;
            while (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000) { *err = 1; break; }     // never hang: flag the step as invalid and move on
            }
        }
    }
    __syncthreads();
}

template <int NSPLIT, bool LSTM>
__global__ __launch_bounds__(256) void rnn_layer_fwd_kernel(const RnnLayerParams P) {
    constexpr int G = LSTM ? 4 : 3;
    constexpr int NP = NSPLIT == 3 ? 2 : 1;
    using TA = TileIO<true, BM>;
    using TB = TileIO<true, 16>;
    extern __shared__ __attribute__((aligned(16))) unsigned short lsm[];
    const slnlp_rnn_layer_dir& d = P.d[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = P.B, Hd = P.Hd, S = P.S, j0 = blockIdx.x * 16, GH = G * Hd;
    const int K = Hd, ktiles = K / BKT;                          // host guarantees Hd % 64 == 0, B <= 64
    unsigned short* Wl = lsm;                                    // [ktile][gate][plane][16 x 64]
    unsigned short* As = lsm + (size_t)ktiles * G * NP * TB::PLANE;
    const unsigned nblocks = gridDim.x * gridDim.y;

    // ---- resident weight slice: rows g*Hd + j0 .. +15 of W_hh, every K tile, split once
    for (int kt = 0; kt < ktiles; ++kt)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float4 rw[TB::NV];
            TB::template fetch<true>(d.w_hh + (long)g * Hd * Hd, Hd, j0, Hd, kt * BKT, K, tid, rw);
            TB::template stash<NSPLIT, false>(Wl + ((size_t)kt * G + g) * NP * TB::PLANE, tid, rw, j0, Hd, kt * BKT, K);
        }
    const int j = j0 + (lane & 15);
    float bh[G];
#pragma unroll
    for (int g = 0; g < G; ++g) bh[g] = d.b_hh ? d.b_hh[g * Hd + j] : 0.f;
    const bool rev = d.reverse != 0;

    for (int step = 0; step < S; ++step) {
        const int t = rev ? S - 1 - step : step, tn = rev ? t - 1 : t + 1;
        const float* h_in = d.hprev + (long)t * B * Hd;
        float* h_out = step + 1 < S ? d.hprev + (long)tn * B * Hd : d.h_final;
        const float* xproj = d.xproj + (long)t * B * GH;
        // the cell's own operands first: their latency hides behind the K loop
        float xpv[4][G], hpv[4], cpv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = wave * 16 + ((lane >> 4) << 2) + r, bb = b < B ? b : 0;
#pragma unroll
            for (int g = 0; g < G; ++g) xpv[r][g] = xproj[(long)bb * GH + g * Hd + j];
            hpv[r] = h_in[(long)bb * Hd + j];
            cpv[r] = LSTM ? d.c[(long)bb * Hd + j] : 0.f;
        }
        f32x4 acc[G];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 ra0[TA::NV], ra1[TA::NV];
        TA::template fetch<true>(h_in, Hd, 0, B, 0, K, tid, ra0);
        TA::template fetch<true>(h_in, Hd, 0, B, BKT, K, tid, ra1);
        auto consume = [&](int kt) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 ah = TA::frag(As, wave * 16, kk, lane);
                bf16x8 al = ah;
                if (NSPLIT == 3) al = TA::frag(As + TA::PLANE, wave * 16, kk, lane);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const unsigned short* bt = Wl + ((size_t)kt * G + g) * NP * TB::PLANE;
                    const bf16x8 bhf = TB::frag(bt, 0, kk, lane);
                    if (NSPLIT == 3) {
                        const bf16x8 blf = TB::frag(bt + TB::PLANE, 0, kk, lane);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bhf, acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, blf, acc[g], 0, 0, 0);
                    }
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bhf, acc[g], 0, 0, 0);
                }
            }
        };
        // the K sum in two halves, tiles [0, T) and [T, ktiles), first + second: gemm_tile's definition
        const int half = (ktiles + 1) / 2;
        f32x4 acc_first[G];
        auto park = [&]() {
#pragma unroll
            for (int g = 0; g < G; ++g) { acc_first[g] = acc[g]; acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        };
        for (int kt = 0; kt < ktiles; kt += 2) {
            if (kt == half) park();
            lds_barrier();
            TA::template stash<NSPLIT, false>(As, tid, ra0, 0, B, kt * BKT, K);
            lds_barrier();
            TA::template fetch<true>(h_in, Hd, 0, B, (kt + 2) * BKT, K, tid, ra0);
            consume(kt);
            if (kt + 1 >= ktiles) break;
            if (kt + 1 == half) park();
            lds_barrier();
            TA::template stash<NSPLIT, false>(As, tid, ra1, 0, B, (kt + 1) * BKT, K);
            lds_barrier();
            TA::template fetch<true>(h_in, Hd, 0, B, (kt + 3) * BKT, K, tid, ra1);
            consume(kt + 1);
        }
        if (ktiles <= half) park();
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = acc_first[g] + acc[g];
        // ---- cell (same arithmetic and order as rnn_cell_fwd_kernel)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = wave * 16 + ((lane >> 4) << 2) + r;
            if (b >= B) break;
            const long idx = (long)b * Hd + j;
            const bool valid = P.lengths ? (t < P.lengths[b]) : true;
            float* a = d.acts + (long)t * B * GH + (long)b * GH;
            const float hprev = hpv[r];
            float hnew;
            if constexpr (LSTM) {
                const float cprev = cpv[r];
                const float gi = sigm(xpv[r][0] + (acc[0][r] + bh[0]));
                const float gf = sigm(xpv[r][1] + (acc[1][r] + bh[1]));
                const float gg = tanhf(xpv[r][2] + (acc[2][r] + bh[2]));
                const float go = sigm(xpv[r][3] + (acc[3][r] + bh[3]));
                const float cnew = gf * cprev + gi * gg;
                hnew = go * tanhf(cnew);
                a[j] = gi; a[Hd + j] = gf; a[2 * Hd + j] = gg; a[3 * Hd + j] = go;
                d.cprev[(long)t * B * Hd + idx] = cprev;
                d.c[idx] = valid ? cnew : cprev;
            } else {
                const float hn = acc[2][r] + bh[2];
                const float rr = sigm(xpv[r][0] + (acc[0][r] + bh[0]));
                const float z = sigm(xpv[r][1] + (acc[1][r] + bh[1]));
                const float nn = tanhf(xpv[r][2] + rr * hn);
                hnew = (1.f - z) * nn + z * hprev;
                a[j] = rr; a[Hd + j] = z; a[2 * Hd + j] = nn;
                d.hn[(long)t * B * Hd + idx] = hn;
            }
            // the next step's workgroups (other XCDs) read this: write-through store
            __hip_atomic_store(h_out + idx, valid ? hnew : hprev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d.out) {
                float o = valid ? hnew : P.fill;
                if (P.drop_p > 0.f && valid)
                    o = dropout_keep(P.rng, P.drop_site, (unsigned)(t * B + b), (unsigned)(d.out_col0 + j), P.drop_thr)
                            ? o / (1.f - P.drop_p) : 0.f;
                d.out[((long)t * B + b) * P.ld_out + j] = o;
            }
        }
        if (step + 1 < S) grid_barrier_sr(P.bar, P.err, nblocks);
    }
}

// one-time opt-in to > 64 KiB dynamic LDS; called from plan creation so it never lands inside a graph capture
int rnn_layer_init() {
    static DeviceOnce once;                     // hipFuncSetAttribute applies per device
    return once.run([]() -> int {
    const int lim = 156 * 1024;
    const bool ok =
        hipFuncSetAttribute((const void*)rnn_layer_fwd_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess &&
        hipFuncSetAttribute((const void*)rnn_layer_fwd_kernel<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess &&
        hipFuncSetAttribute((const void*)rnn_layer_fwd_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess &&
        hipFuncSetAttribute((const void*)rnn_layer_fwd_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess;
    if (!ok) {
        set_error("rnn_layer_init: cannot raise dynamic LDS limit: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    return 0;
    });
}

static size_t rnn_layer_lds(int G, int Hd, int precision) {
    const int NP = precision == 3 ? 2 : 1;
    return ((size_t)(Hd / BKT) * G * NP * TileIO<true, 16>::PLANE + (size_t)NP * TileIO<true, BM>::PLANE) * sizeof(unsigned short);
}

// 0 = launched; 1 = shape not covered by the persistent kernel (caller uses the per-timestep path)
int rnn_layer_fwd(int lstm, const slnlp_rnn_layer_dir* dirs, int ndir, int B, int Hd, int S, const int64_t* lengths,
                  float fill, int64_t ld_out, float drop_p, int drop_site, const unsigned long long* rng, int precision,
                  unsigned* bar, int* err, int* launched, hipStream_t st) {
    SLNLP_CHECK_ARG(dirs && (ndir == 1 || ndir == 2) && B > 0 && Hd > 0 && S > 0 && bar && err && launched, "rnn_layer_fwd: bad args");
    SLNLP_CHECK_ARG(precision == 1 || precision == 3, "rnn_layer_fwd: precision must be 1 or 3");
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "rnn_layer_fwd: bad dropout args");
    const int G = lstm ? 4 : 3;
    const size_t lds = rnn_layer_lds(G, Hd, precision);
    *launched = 0;
    if (B > BM || Hd % BKT != 0 || lds > 156 * 1024 || (Hd / 16) * ndir > 128) return SLNLP_OK;   // not covered
    RnnLayerParams P;
    for (int k = 0; k < ndir; ++k) {
        const slnlp_rnn_layer_dir& d = dirs[k];
        SLNLP_CHECK_ARG(d.hprev && d.h_final && d.w_hh && d.xproj && d.acts && (lstm ? (d.c && d.cprev) : (d.hn != nullptr)),
                        "rnn_layer_fwd: null pointer in direction %d", k);
        SLNLP_CHECK_ARG(vec_ok(d.hprev, Hd) && vec_ok(d.w_hh, Hd) && ((long)B * Hd) % 4 == 0, "rnn_layer_fwd: hprev / w_hh must be 16-byte aligned");
        P.d[k] = d;
    }
    if (ndir == 1) P.d[1] = P.d[0];
    P.B = B; P.Hd = Hd; P.S = S; P.lengths = (const long*)lengths; P.fill = fill; P.ld_out = ld_out;
    P.drop_p = drop_p; P.drop_thr = dropout_threshold(drop_p); P.drop_site = drop_site; P.rng = rng;
    P.bar = bar; P.err = err;
    const dim3 grid(Hd / 16, ndir);
    SLNLP_TRY(rnn_layer_init());
#define SLNLP_LAYER(NS, L) hipLaunchKernelGGL((rnn_layer_fwd_kernel<NS, L>), grid, dim3(256), lds, st, P)
    if (precision == 3) { if (lstm) SLNLP_LAYER(3, true); else SLNLP_LAYER(3, false); }
    else { if (lstm) SLNLP_LAYER(1, true); else SLNLP_LAYER(1, false); }
#undef SLNLP_LAYER
    SLNLP_CHECK_LAUNCH("rnn_layer_fwd");
    *launched = 1;
    return SLNLP_OK;
}

}  // namespace slnlp

extern "C" int slnlp_rnn_layer_fwd(int lstm, const slnlp_rnn_layer_dir* dirs, int ndir, int B, int Hd, int S,
                                   const int64_t* lengths, float fill, int64_t ld_out, float drop_p, int drop_site,
                                   const unsigned long long* rng, int precision, uint32_t* sync, int* launched, void* stream) {
    if (!sync) {
        slnlp::set_error("slnlp_rnn_layer_fwd: sync words required");
        return SLNLP_ERR_INVALID_ARG;
    }
    return slnlp::rnn_layer_fwd(lstm, dirs, ndir, B, Hd, S, lengths, fill, ld_out, drop_p, drop_site, rng, precision, sync,
                                reinterpret_cast<int*>(sync + 2), launched, (hipStream_t)stream);
}

extern "C" int slnlp_rnn_step_bwd(int lstm, const slnlp_rnn_step_bwd_dir* dirs, int ndir, int B, int Hd, const int64_t* lengths,
                                  int64_t ld_dout, float drop_p, int drop_site, const unsigned long long* rng, int precision,
                                  void* stream) {
    return slnlp::rnn_step_bwd(lstm, dirs, ndir, B, Hd, lengths, ld_dout, drop_p, drop_site, rng, precision, (hipStream_t)stream);
}

extern "C" int slnlp_rnn_step_fwd(int lstm, const slnlp_rnn_step_dir* dirs, int ndir, int B, int Hd, const int64_t* lengths,
                                  float fill, int64_t ld_out, float drop_p, int drop_site, const unsigned long long* rng,
                                  int precision, void* stream) {
    return slnlp::rnn_step_fwd(lstm, dirs, ndir, B, Hd, lengths, fill, ld_out, drop_p, drop_site, rng, precision,
                               (hipStream_t)stream);
}

extern "C" int slnlp_gemm(const slnlp_gemm_args* args, void* stream) {
    if (!args) {
        slnlp::set_error("slnlp_gemm: null args");
        return SLNLP_ERR_INVALID_ARG;
    }
    return slnlp::gemm(*args, (hipStream_t)stream);
}

#if SLNLP_PROBE_FENCES == 256
// probe build only: copy the recorded workgroup timelines to the host and reset the recorder; returns the number recorded
extern "C" int slnlp_probe_gemm_ts(unsigned long long* dst, int max_entries) {
    unsigned n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(slnlp::g_gts_n), sizeof(n)) != hipSuccess) return -1;
    if (n > (unsigned)slnlp::GTS_MAX) n = slnlp::GTS_MAX;
    if ((int)n > max_entries) n = max_entries;
    if (n && hipMemcpyFromSymbol(dst, HIP_SYMBOL(slnlp::g_gts), (size_t)n * slnlp::GTS_W * sizeof(unsigned long long)) != hipSuccess) return -1;
    const unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(slnlp::g_gts_n), &zero, sizeof(zero)) != hipSuccess) return -1;
    return (int)n;
}
#endif

extern "C" int slnlp_set_gemm_ks(int ks) {
    if (ks != 0 && ks != 1 && ks != 2) {
        slnlp::set_error("set_gemm_ks: %d (0 = automatic, 1 or 2 thread groups per workgroup)", ks);
        return SLNLP_ERR_INVALID_ARG;
    }
    slnlp::g_gemm_ks.store(ks == 0 ? -1 : ks, std::memory_order_relaxed);
    return 0;
}

extern "C" int slnlp_set_rnn_step_tile(int rows16) {
    slnlp::g_rnn_step_rt.store(rows16 ? 1 : 0, std::memory_order_relaxed);
    return 0;
}
