// gemm_planes.hip -- split-bf16 MFMA GEMM over PRE-SPLIT operands, staged by LDS-DMA.
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ),  A = Ahi + Alo, B = Bhi + Blo  (bf16 planes)
//
// gemm.hip converts fp32 -> bf16 hi/lo inside its K-loop; measured on MI355X that loop is ~90 %
// address / convert / ds_write / barrier work and ~10 % MFMA, and every element is converted again
// by each tile that reads it.  Here every operand arrives as two bf16 planes written ONCE by its
// producer (GEMM / LayerNorm / attention epilogues, the weight splitter), zero-padded to multiples
// of 64 in both dimensions, so a K-step is: 4 x global_load_lds_dwordx4 per wave (global -> LDS
// DMA, no VGPRs, no VALU, no ds_write), one counted s_waitcnt, two barriers, fragment reads, MFMA.
//
//  * tile 64x64x64, 8 waves (4x2, each 16x32), two LDS stages of {Ahi, Alo, Bhi, Blo} x 8 KiB;
//  * LDS-DMA writes lane-linearly, so the bank-conflict swizzle is applied to the SOURCE address
//    and undone with the same XOR on the fragment read (both are involutions):
//      k-major image [row][64 k]  : 16-B slot ^= (row & 7)            -> conflict-free ds_read_b128
//      m-major image [k][64 rows] : 16-B slot ^= 2*(k>>1 & 1 | k>>3 & 1 << 1) -> ds_read_b64_tr_b16
//  * padding makes every tile interior: no bounds logic anywhere in the loop.
#include <algorithm>
#include <vector>

#include "common.hpp"
#include "gemm_jobs.hpp"

namespace slnlp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(1))) const void* glb_vp;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int PT = 64;                 // tile edge (M, N and K)
constexpr int PTHREADS = 512;

__device__ __forceinline__ int mswz(int k) { return (((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1; }

// element offset of logical (row, k) inside a plane image
template <bool KMAJOR>
__device__ __forceinline__ int img_off(int row, int k) {
    if (KMAJOR) return row * PT + ((((k >> 3) ^ (row & 7)) << 3) | (k & 7));
    return k * PT + ((((row >> 3) ^ mswz(k)) << 3) | (row & 7));
}

// One wave copies 8 image lines (1 KiB) of one plane: lane -> (line 8*wave + lane/8, physical slot lane%8),
// source = the logical slot that the swizzle maps there.  The source address is  plane + K-STEP TERM (uniform: k0 elements for a
// k-major plane, k0 * ld for an m-major one) + LANE TERM (loop-invariant, below): the K loop then advances scalar registers only
// and the DMA takes its address as {SGPR base, 32-bit VGPR offset} -- no 64-bit vector arithmetic per piece and step.
template <bool KMAJOR>
__device__ __forceinline__ unsigned plane_lane_off(long ld, int row0, int wave, int lane) {       // bytes
    const int line = 8 * wave + (lane >> 3), ps = lane & 7;
    const long e = KMAJOR ? (long)(row0 + line) * ld + ((ps ^ (line & 7)) << 3) : (long)line * ld + row0 + ((ps ^ mswz(line)) << 3);
    return (unsigned)(e * 2);
}
__device__ __forceinline__ void dma_piece(const unsigned short* __restrict__ plane, long kterm, unsigned lane_off, unsigned short* dst) {
    const char* src = reinterpret_cast<const char*>(plane + kterm) + lane_off;
    __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)dst, 16, 0, 0);
}

template <bool KMAJOR>
__device__ __forceinline__ bf16x8 pfrag(const unsigned short* __restrict__ img, int r0, int kk, int lane) {
    if (KMAJOR) {
        return *reinterpret_cast<const bf16x8*>(img + img_off<true>(r0 + (lane & 15), kk * 32 + ((lane >> 4) << 3)));
    } else {
        const int i = lane & 15, kb = kk * 32 + ((lane >> 4) << 3) + (i >> 2), col = r0 + ((i & 3) << 2);
        typedef __attribute__((address_space(3))) s16x4* lds_p;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + img_off<false>(col, kb)));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + img_off<false>(col, kb + 4)));
        const s16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

// ---- 32-k stages (deeper rings / larger tiles in the same LDS): a plane image is 64 rows x 32 k = 4 KiB.
//   k-major: 64-byte rows would put a ds_read_b128 lane group on two 16-B slots of the bank row, so TWO rows share a 128-B
//            line -- line L = rows 2L, 2L+1; logical slot s = (row & 1) * 4 + k / 8; physical slot = s ^ (L & 7): conflict-free
//            fragment reads, and the DMA still writes lane-linearly (source address = the logical slot the swizzle maps there);
//   m-major: [k][64 rows] lines of 128 B as above, 32 of them.
// One DMA instruction of a wave moves 1 KiB = 8 lines; a SLAB (128 rows x 32 k of one plane = two images) is one instruction
// of each of the 8 waves: wave -> image (wave >> 2), lines 8 * (wave & 3) ...
template <bool KMAJOR>
__device__ __forceinline__ int img32_off(int row, int k) {
    if (KMAJOR) {
        const int line = row >> 1, sl = ((row & 1) << 2) | (k >> 3);
        return line * PT + (((sl ^ (line & 7)) << 3) | (k & 7));
    }
    return img_off<false>(row, k);
}
template <bool KMAJOR>
__device__ __forceinline__ unsigned slab32_lane_off(long ld, int row0, int row_last, int wave, int lane) {   // bytes
    const int h = wave >> 2, line = 8 * (wave & 3) + (lane >> 3), ps = lane & 7;
    const int r_img = min(row0 + PT * h, row_last);       // (planes are padded to 64 rows: never read past the last padded block)
    long e;
    if (KMAJOR) {
        const int sl = ps ^ (line & 7);
        e = (long)(r_img + 2 * line + (sl >> 2)) * ld + ((sl & 3) << 3);
    } else {
        e = (long)line * ld + r_img + ((ps ^ mswz(line)) << 3);
    }
    return (unsigned)(e * 2);
}
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 pfrag32(const unsigned short* __restrict__ img, int r0, int lane) {
    if (KMAJOR) return *reinterpret_cast<const bf16x8*>(img + img32_off<true>(r0 + (lane & 15), (lane >> 4) << 3));
    return pfrag<false>(img, r0, 0, lane);
}

__device__ __forceinline__ float pbf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

// 8-byte agent-scope (sc1) store / load: write-through to / fetched from the memory side, never a stale XCD-L2 line
__device__ __forceinline__ void st_agent2(float* p, float a, float b) {
    const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_agent2(const float* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}

#if SLNLP_PROBE_FENCES == 128
// timeline probe build (tools/probes/probe_tile_timeline.py): every workgroup records 100 MHz timestamps of its phases
constexpr int TS_MAX = 1 << 16;
constexpr int TS_W = 8;                // words per record: 5 stamps (100 MHz), {XCC id, block}, shader-clock stamps at marks 1 and 2
__device__ unsigned long long g_ts[TS_MAX][TS_W];
__device__ unsigned g_ts_n;
#ifdef SLNLP_PROBE_EPI
#define TS_CLK(slot) do { } while (0)
#else
#define TS_CLK(slot) do { if (slot == 1) ts[6] = __builtin_amdgcn_s_memtime(); if (slot == 2) ts[7] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#define TS_MARK(slot) do { if (threadIdx.x == 0) { ts[slot] = __builtin_amdgcn_s_memrealtime(); \
        TS_CLK(slot); } } while (0)
#else
#define TS_MARK(slot) do { } while (0)
#endif
#if SLNLP_PROBE_FENCES == 128 && defined(SLNLP_PROBE_EPI)
// epilogue phases instead of the two shader-clock stamps: words 6 / 7 = 100 MHz stamps behind (1b) and behind the barrier in front of (2)
#define TS_EPI(slot) do { if (threadIdx.x == 0) ts[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TS_EPI(slot) do { } while (0)
#endif

// The K loop of one (passes, operand layouts) variant: accumulators (and A's row sums) of output tile (bm0, bn0) over ring steps
// [kt0, kt1).  Inlined once per variant into plane_tile, whose tile decoding, split-K meeting and epilogue are the same code for all.
//
// Tile geometry (template): BM x BN output tile, 8 waves as 4 (M) x 2 (N), each wave a (BM/4) x (BN/2) sub-tile = MT x NT MFMA
// tiles of 16 x 16; K-step BKS (64 or 32) through an NST-deep LDS ring.  An operand panel of BM rows is BM/64 plane images,
// so the DMA pattern, the swizzles and the fragment reads are those of the 64 x 64 tile whatever BM / BN:
//   64 x 64 x 64, 2 stages   : 16 x 32 per wave -- 6 fragment reads for 6 MFMAs per 32-k; 32 KiB of operand planes per K-step for
//                              64 x 64 x 64 products; two workgroups per CU.  Launches that need every CU for a few hundred tiles.
//   128 x 128 x 64, 2 stages : 32 x 64 per wave -- 12 fragment reads for 24 MFMAs per 32-k, half the L2 -> LDS bytes per FLOP
//                              (the measured wall of the 64 x 64 tile); 128 KiB, one workgroup per CU.
//   128 x 128 x 32, 2 stages : the same tile in 64 KiB -- two workgroups per CU again, each other's cover during prologue / epilogue.
//   256 x 256 x 32, 2 stages : 64 x 128 per wave -- 24 fragment reads for 96 MFMAs, half the operand bytes per FLOP again; 128 KiB,
//                              one workgroup per CU: launches of many K-steps per tile and several tiles per CU.
// The K partition (in units of 64) does not depend on the geometry, so every geometry accumulates every output element in the
// same order: identical bits.
template <int NSPLIT, bool AK, bool BK, int BM, int BN, int BKS, int NST>
__device__ __forceinline__ void plane_kloop(const slnlp_gemm_args& g, unsigned short* smem, int bm0, int bn0, int kt0, int kt1,
                                            bool do_rowsum, f32x4 (&acc)[BM / 64][BN / 32], float (&rowsum)[BM / PT],
                                            unsigned long long* ts) {
    // planes per operand: NSPLIT 3 = A_lo B_hi + A_hi B_lo + A_hi B_hi; NSPLIT 2 = A_hi (B_hi + B_lo) -- the A operand (the dY of a
    // gradient product) contributes its bf16 head only, a third less MFMA work and a quarter less staging; NSPLIT 1 = A_hi B_hi
    constexpr int NPA = NSPLIT == 3 ? 2 : 1, NPB = NSPLIT >= 2 ? 2 : 1;
    constexpr int SUBM = BM / PT, SUBN = BN / PT;            // 64-row plane images per operand panel
    constexpr int MT = BM / 64, NT = BN / 32;                // 16 x 16 MFMA tiles per wave: (BM/4)/16 x (BN/2)/16
    constexpr int IMG_E = PT * BKS;                          // elements of one image (8 KiB at 64 k, 4 KiB at 32 k)
    constexpr int A_IMGS = NPA * SUBM, B_IMGS = NPB * SUBN;
    constexpr int STAGE = (A_IMGS + B_IMGS) * IMG_E;         // A planes then B planes
    constexpr int PIECES = (A_IMGS + B_IMGS) * BKS / 64;     // DMA instructions per wave and stage
    static_assert(BKS == 64 || (BM % 128 == 0 && BN % 128 == 0), "32-k stages move 128-row slabs");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * (BM / 4), wn0 = (wave & 1) * (BN / 2);
    // planes are zero-padded to multiples of 64 rows; a wider panel may reach further: read the last padded block again
    // instead (its products land in rows >= M / columns >= N, which no epilogue stores)
    const int am_last = ((g.M + PT - 1) / PT - 1) * PT, bn_last = ((g.N + PT - 1) / PT - 1) * PT;

    // this wave's DMA pieces of a stage: per operand panel piece a loop-invariant lane offset (bytes inside the plane; planes
    // are far below 4 GiB); the hi and the lo plane of an operand share it
    constexpr int APC = BKS == 64 ? SUBM : SUBM / 2, BPC = BKS == 64 ? SUBN : SUBN / 2;      // pieces per plane and stage
    unsigned aoff[APC], boff[BPC];
#pragma unroll
    for (int q = 0; q < APC; ++q) {
        if constexpr (BKS == 64) aoff[q] = plane_lane_off<AK>(g.lda_p, BM > PT ? min(bm0 + q * PT, am_last) : bm0, wave, lane);
        else aoff[q] = slab32_lane_off<AK>(g.lda_p, bm0 + 2 * q * PT, am_last, wave, lane);
    }
#pragma unroll
    for (int q = 0; q < BPC; ++q) {
        if constexpr (BKS == 64) boff[q] = plane_lane_off<BK>(g.ldb_p, BN > PT ? min(bn0 + q * PT, bn_last) : bn0, wave, lane);
        else boff[q] = slab32_lane_off<BK>(g.ldb_p, bn0 + 2 * q * PT, bn_last, wave, lane);
    }
    constexpr int PSTEP = BKS == 64 ? 1 : 2;                 // images a piece spans (a 32-k slab = two images)
    auto issue = [&](int kt, int stage) {
        const int k0 = (kt < kt1 ? kt : kt0) * BKS;       // past-the-end prefetch re-reads a valid tile (never consumed)
        const long ka = AK ? (long)k0 : (long)k0 * g.lda_p, kb = BK ? (long)k0 : (long)k0 * g.ldb_p;    // uniform
        unsigned short* s = smem + stage * STAGE + wave * 512;
#pragma unroll
        for (int q = 0; q < APC; ++q) {
            dma_piece(g.A_hi, ka, aoff[q], s + q * PSTEP * IMG_E);
            if (NPA == 2) dma_piece(g.A_lo, ka, aoff[q], s + (SUBM + q * PSTEP) * IMG_E);
        }
#pragma unroll
        for (int q = 0; q < BPC; ++q) {
            dma_piece(g.B_hi, kb, boff[q], s + (A_IMGS + q * PSTEP) * IMG_E);
            if (NPB == 2) dma_piece(g.B_lo, kb, boff[q], s + (A_IMGS + SUBN + q * PSTEP) * IMG_E);
        }
    };

    // NST-deep LDS ring: steps kt+1 .. kt+NST-1 are in flight while step kt is consumed (a K-step's MFMA work is
    // ~0.2 - 0.6 us, one DMA round trip ~1 us).  ONE barrier per step: the stage refilled at step kt was consumed at
    // step kt-1, which every wave has finished once it passes this step's barrier.
    for (int t = 0; t < NST - 1; ++t) issue(kt0 + t, t);
    for (int kt = kt0; kt < kt1; ++kt) {
        const int it = kt - kt0;
        // this wave's DMA of step kt has landed once only the (NST-2) newer steps' pieces are outstanding
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"((NST - 2) * PIECES) : "memory");
        __builtin_amdgcn_s_barrier();                      // ... and so has every other wave's part
        asm volatile("" ::: "memory");
#if SLNLP_PROBE_FENCES == 128
        if (kt == kt0) TS_MARK(1);
#endif
        issue(kt + NST - 1, (it + NST - 1) % NST);
        const unsigned short* s = smem + (it % NST) * STAGE;
#pragma unroll
        for (int kk = 0; kk < BKS / 32; ++kk) {
            // (B fragments JB tiles at a time: the 256-wide tile's 4 + 8 tiles x (hi, lo) at once would not leave room for its 128 accumulators)
            constexpr int JB = NT > 4 ? 4 : NT;
            bf16x8 ah[MT], al[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = wm0 + 16 * i;              // a 16-row MFMA tile never straddles two 64-row images
                if constexpr (BKS == 64) {
                    ah[i] = pfrag<AK>(s + (row / PT) * IMG_E, row % PT, kk, lane);
                    if (NPA == 2) al[i] = pfrag<AK>(s + (SUBM + row / PT) * IMG_E, row % PT, kk, lane);
                } else {
                    ah[i] = pfrag32<AK>(s + (row / PT) * IMG_E, row % PT, lane);
                    if (NPA == 2) al[i] = pfrag32<AK>(s + (SUBM + row / PT) * IMG_E, row % PT, lane);
                }
            }
#pragma unroll
            for (int j0 = 0; j0 < NT; j0 += JB) {
                bf16x8 bh[JB], bl[JB];
#pragma unroll
                for (int j = 0; j < JB; ++j) {
                    const int row = wn0 + 16 * (j0 + j);
                    if constexpr (BKS == 64) {
                        bh[j] = pfrag<BK>(s + (A_IMGS + row / PT) * IMG_E, row % PT, kk, lane);
                        if (NPB == 2) bl[j] = pfrag<BK>(s + (A_IMGS + SUBN + row / PT) * IMG_E, row % PT, kk, lane);
                    } else {
                        bh[j] = pfrag32<BK>(s + (A_IMGS + row / PT) * IMG_E, row % PT, lane);
                        if (NPB == 2) bl[j] = pfrag32<BK>(s + (A_IMGS + SUBN + row / PT) * IMG_E, row % PT, lane);
                    }
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
                        f32x4& c = acc[i][j0 + j];
                        if (NPA == 2) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], c, 0, 0, 0);
                        if (NPB == 2) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], c, 0, 0, 0);
                    }
            }
        }
        if (do_rowsum) {
            // 64 k: thread = row (tid & 63) of every image, k-octet tid >> 6 of the step.  32 k: a step holds 4 octets, so the
            // waves cover two images per pass (image pair p: sm + wave / 4, octet wave % 4) and a thread keeps the sums of the
            // even and the odd steps apart -- octets o and o + 4 of the 64-k tile -- so the eight octet sums, and their final
            // order, are those of the 64-k path: the bias gradient keeps its bits across geometries.
            constexpr int OCT = BKS / 8;                      // k-octets per image and step
            const int row = tid & 63, kq = ((tid >> 6) % OCT) * 8, sub0 = (tid >> 6) / OCT;
#pragma unroll
            for (int sm = 0; sm < SUBM; sm += 8 / OCT) {
                const int im = sm + sub0;
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int off = BKS == 64 ? img_off<AK>(row, kq + k) : img32_off<AK>(row, kq + k);
                    t += pbf2f(s[im * IMG_E + off]);
                    if (NPA == 2) t += pbf2f(s[(SUBM + im) * IMG_E + off]);
                }
                if (BKS == 64) rowsum[sm] += t;
                else if (kt & 1) rowsum[sm + 1] += t;          // (kt0 is even: a split starts on a 64-k tile)
                else rowsum[sm] += t;
            }
        }
    }
}

// One block's work: output tile (bx, by) of the job, K-tiles [kt0, kt1) -- the K loop of the job's variant, then the split-K
// meeting and the epilogue, which do not depend on the variant and exist ONCE per kernel.  (The code a workgroup walks through
// after its K loop is straight-line and cold in the instruction cache -- 64 KiB per two CUs: when every variant carried its own
// fully unrolled epilogues a 128 x 128 kernel was 316 KiB of code and its workgroups spent 6 us of a 19 us life there, a
// 256 x 256 one 61 us.  Hence the rolled loops below: only what needs a STATIC accumulator register index is unrolled.)
template <int NSPLIT, int BM, int BN, int BKS, int NST>
__device__ __forceinline__ void plane_tile(const PlaneJob& job, int lid, bool placed, unsigned short* smem) {
    unsigned long long* tsp = nullptr;
#if SLNLP_PROBE_FENCES == 128
    unsigned long long ts[TS_W] = {0, 0, 0, 0, 0, 0, 0, 0};
    struct TsFlush {
        unsigned long long* t;
        __device__ ~TsFlush() {
            if (threadIdx.x == 0) {
                t[4] = __builtin_amdgcn_s_memrealtime();
                unsigned hw;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hw));
                t[5] = ((unsigned long long)hw << 32) | blockIdx.x;
                const unsigned i = atomicAdd(&g_ts_n, 1u) & (unsigned)(TS_MAX - 1);   // a ring: the last TS_MAX workgroups
                for (int k = 0; k < TS_W; ++k) g_ts[i][k] = t[k];
            }
        }
    } ts_flush{ts};
    tsp = ts;
    TS_MARK(0);
#endif
    constexpr int SUBM = BM / PT;                            // 64-row plane images per A panel
    constexpr int MT = BM / 64, NT = BN / 32;                // 16 x 16 MFMA tiles per wave: (BM/4)/16 x (BN/2)/16
    constexpr int KSUB = PT / BKS;                           // ring steps per 64-k tile
    const slnlp_gemm_args& g = job.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * (BM / 4), wn0 = (wave & 1) * (BN / 2);
    const int nks = job.nks;
    int bx, by, ks, tile;
    {   // XCD-aware order (see gemm.hip): each XCD owns a contiguous run of (tile, split) units
        const int nwg = job.tiles_x * job.tiles_y * nks;
        const int xcd = lid & 7, q = nwg >> 3, r = nwg & 7;
        // (a merged launch: `lid` IS the unit -- the host laid the units of all jobs out over the XCDs)
        const int t = placed ? lid : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lid >> 3);
        // split-major unit order: the units of one XCD share a K-slice, so its 4 MiB L2 holds that slice of a few
        // A / B panels instead of whole panels (measured 5x over-fetch with tile-major order on the weight gradient)
        const int ntiles = job.tiles_x * job.tiles_y;
        ks = t / ntiles;
        tile = t - ks * ntiles;
        // tiles in groups of GR tile rows, column by column inside a group: the workgroups resident on an XCD at one time
        // (32 at one per CU, 64 at two) then form a GR x (32 / GR) block of the tile grid whatever tiles_x is, and share its
        // GR + 32 / GR operand panels through the XCD's L2 instead of fetching 32 + 1
        constexpr int GR = BM >= 256 ? 2 : BM >= 128 ? 4 : 8;
        const int grp = tile / (GR * job.tiles_x), rem = tile - grp * (GR * job.tiles_x);
        const int rows_here = min(GR, job.tiles_y - grp * GR);
        bx = rem / rows_here;
        by = grp * GR + (rem - bx * rows_here);
        tile = by * job.tiles_x + bx;                          // (the id the partial tiles and arrival counters use)
    }
    const int bm0 = by * BM, bn0 = bx * BN;
    const int M = g.M, N = g.N, K = g.K;
    const int ktiles = (K + PT - 1) / PT;
    const int kt0 = (int)((long)ktiles * ks / nks) * KSUB, kt1 = (int)((long)ktiles * (ks + 1) / nks) * KSUB;   // ring steps

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_rowsum = (g.rowsum_a != nullptr) && (bx == 0);
    float rowsum[SUBM];                                      // (64 k: one per image; 32 k: image pair p -> [2p] even steps, [2p + 1] odd steps)
#pragma unroll
    for (int i = 0; i < SUBM; ++i) rowsum[i] = 0.f;

    // variants 3, 4 (launches of the split-bf16 family only): the gradient products of one dY with its bf16 head alone (precision 2)
    // (a two-pass stage is 3/4 of a three-pass one: THREE of them fit where two of those do -- a prefetch distance of two K-steps)
    if (job.variant == 0) plane_kloop<NSPLIT, true, true, BM, BN, BKS, NST>(g, smem, bm0, bn0, kt0, kt1, do_rowsum, acc, rowsum, tsp);
    else if (job.variant == 1) plane_kloop<NSPLIT, true, false, BM, BN, BKS, NST>(g, smem, bm0, bn0, kt0, kt1, do_rowsum, acc, rowsum, tsp);
    else if (job.variant == 2) plane_kloop<NSPLIT, false, false, BM, BN, BKS, NST>(g, smem, bm0, bn0, kt0, kt1, do_rowsum, acc, rowsum, tsp);
    else if constexpr (NSPLIT == 3) {
        if (job.variant == 3) plane_kloop<2, true, false, BM, BN, BKS, NST + 1>(g, smem, bm0, bn0, kt0, kt1, do_rowsum, acc, rowsum, tsp);
        else plane_kloop<2, false, false, BM, BN, BKS, NST + 1>(g, smem, bm0, bn0, kt0, kt1, do_rowsum, acc, rowsum, tsp);
    }
    TS_MARK(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // drain the dummy prefetches before LDS is reused / freed
    __builtin_amdgcn_s_barrier();
    float* rs = reinterpret_cast<float*>(smem);
    int* flag = reinterpret_cast<int*>(smem) + SUBM * PTHREADS;   // (behind the rs table: SUBM * 8 * 64 floats at most)
    float rs_row = 0.f;                                      // tid < BM: this block's row sum of A row bm0 + tid
    if (do_rowsum) {
        // rs[image][octet of the 64-k tile][row]
        if (BKS == 64) {
#pragma unroll
            for (int sm = 0; sm < SUBM; ++sm) rs[(sm * 8 + (tid >> 6)) * PT + (tid & 63)] = rowsum[sm];
        } else {
#pragma unroll
            for (int sm = 0; sm < SUBM; sm += 2) {
                const int im = sm + (tid >> 8), o = (tid >> 6) & 3;
                rs[(im * 8 + o) * PT + (tid & 63)] = rowsum[sm];
                rs[(im * 8 + o + 4) * PT + (tid & 63)] = rowsum[sm + 1];
            }
        }
        __syncthreads();
        if (tid < BM) {
#pragma unroll
            for (int w = 0; w < 8; ++w) rs_row += rs[((tid / PT) * 8 + w) * PT + (tid % PT)];
        }
    }
    if (nks > 1) {
        // Split-K meeting point.  The L2s of the 8 XCDs are not coherent with each other and an agent-scope fence
        // costs a whole-L2 write-back per wave (measured: 8x slower kernel), so the partial tiles never live in
        // L2: they are written and read with agent-scope (sc1, write-through / bypass) accesses, and the only
        // ordering needed is "my stores have completed (vmcnt 0, block barrier) before my arrival is counted".
        // 16-byte sc0 sc1 buffer accesses (8-byte agent atomics before: a scalar sc1 store is one fabric write each, 2.7x the
        // time per byte of a dwordx4, and 8-byte sc1 loads run at 0.54-0.70x the 16-byte rate -- MI355X guide, cache-policy table)
        constexpr int SC = 17;                               // cache policy bits: sc0 | sc1
        constexpr int TILE_FLOATS = BM * BN;                 // a partial tile in the accumulator layout: [MT * NT][PTHREADS][4]
        const __amdgpu_buffer_rsrc_t mine =
            __builtin_amdgcn_make_buffer_rsrc(job.part + ((long)tile * nks + ks) * TILE_FLOATS, 0, TILE_FLOATS * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), mine, tid * 16, (i * NT + j) * PTHREADS * 16, SC);
        if (do_rowsum && tid < BM)
            __hip_atomic_store(job.part_rs + ((long)by * nks + ks) * BM + tid, rs_row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (job.hand_off == 4) return;                       // the K-slices meet in plane_splitk_reduce_kernel, the next launch on the stream
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (job.hand_off == 3) {                         // probe: agent-scope release in front of the arrival count
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const bool last = __hip_atomic_fetch_add(job.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nks - 1;
            if (last && job.hand_off >= 2) {                 // probe: agent-scope acquire before the partials are read
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        rs_row = 0.f;
        // fixed order: the result does not depend on arrival order.  (The widest tile adds its partials half a tile at a time: all 32
        // loads of a split in flight beside the 128 accumulators would spill.)
        constexpr int HALVES = MT * NT > 16 ? 2 : 1, MH = MT / HALVES;
#pragma unroll
        for (int half = 0; half < HALVES; ++half) {
#pragma unroll 1
            for (int s = 0; s < nks; ++s) {
                const __amdgpu_buffer_rsrc_t q =
                    __builtin_amdgcn_make_buffer_rsrc(job.part + ((long)tile * nks + s) * TILE_FLOATS, 0, TILE_FLOATS * 4, 0x00020000);
#pragma unroll
                for (int i = half * MH; i < (half + 1) * MH; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const u32x4 p = __builtin_amdgcn_raw_buffer_load_b128(q, tid * 16, (i * NT + j) * PTHREADS * 16, SC);
                        acc[i][j] += __builtin_bit_cast(f32x4, p);
                    }
                if (half == 0 && do_rowsum && tid < BM)
                    rs_row += __hip_atomic_load(job.part_rs + ((long)by * nks + s) * BM + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tid == 0) __hip_atomic_store(job.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    if (do_rowsum && tid < BM && bm0 + tid < M) g.rowsum_a[bm0 + tid] = rs_row;
    TS_MARK(3);

    // ---- epilogue: +bias -> activation -> gate -> dropout -> +resid ; fp32 store (+ optional bf16 planes), through an fp32 image
    // of the tile in LDS, ER = 128 rows at a time (a 256 x 260 image would not fit): chunk `ch` is written by the waves that own its
    // rows and stored by all eight.
    //  (1a) the accumulators go to the image as they are (unrolled: static register indices, four ds_write each);
    //  (1b) jobs with dropout (or tanh) only: in the accumulator layout still -- a lane holds 4 rows x 1 column of a 16 x 16 tile,
    //       and one Philox call serves those 4 rows -- every lane takes its own values through bias, activation, gate and dropout
    //       IN the image, one ROLLED loop over its MT x NT tiles (its own words: no barrier between 1a and 1b; the bias comes from
    //       an LDS copy, a global load per iteration would put its latency on every one of them);
    //  (2)  row-major: a thread takes 4 consecutive columns of a row -- bias, ReLU and gate here for the jobs that skipped 1b,
    //       residual -- and writes ONE 16-byte fp32 store and two 8-byte plane stores (scalar pieces when the job's pointers /
    //       strides do not allow vectors); gate and residual pieces of eight passes are requested up front.
    // The element-wise arithmetic and its order are those of a plain per-element epilogue: results do not depend on the route.
    constexpr int SLD = BN + 4, ER = BM > 128 ? 128 : BM, ECH = BM / ER;
    const int crow = (lane >> 4) << 2, ccol = lane & 15;
    float* stg = reinterpret_cast<float*>(smem);
    float* sbias = stg + ER * SLD;                           // [BN] behind the image (and behind the row-sum table / flag above)
    const bool early = g.drop_p > 0.f || g.relu == 2;        // block-uniform
    if (early && tid < BN) sbias[tid] = (g.bias && bn0 + tid < N) ? g.bias[bn0 + tid] : 0.f;
    DropKey dkey = {};                                       // the step's Threefry key, made ONCE (common.hpp: dropout_key)
    if (g.drop_p > 0.f) dkey = dropout_key(g.rng, g.drop_site);
#pragma unroll 1
    for (int ch = 0; ch < ECH; ++ch) {
        __syncthreads();                                     // every thread is done with the K-loop stages / the split-K flag / the last chunk
        if (ECH == 1 || wm0 / ER == ch) {                    // (wave-uniform)
            const int lr0 = wm0 - ch * ER + crow, lc0 = wn0 + ccol;     // this lane's rows / column inside the chunk's image, tile (0, 0)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg[(lr0 + 16 * i + r) * SLD + lc0 + 16 * j] = acc[i][j][r];
#if defined(SLNLP_PROBE_EPI) && SLNLP_PROBE_EPI == 2
            if (ch == 0) TS_EPI(6);
#endif
            if (early) {
                // a lane's column tiles come in pairs 16 columns apart (wn0 and bn0 are multiples of 32): the 4 rows x 2 columns ONE
                // Philox call serves (common.hpp).  The pair's eight values (and gate values) are READ FIRST, then taken through the
                // chain, then written: element by element the compiler must keep every image read behind the previous image write
                // (same array, runtime indices) -- eight dependent LDS round trips per pair, which, not the Philox, was two thirds
                // of the dropout epilogue's 12 us (r05 timeline with a one-round Philox: 8.8 us).  Values of rows >= M / columns >= N
                // are computed like the others and never stored by phase 2.
                static_assert(NT % 2 == 0, "column tiles in pairs");
#pragma unroll 1
                for (int t = 0; t < MT * (NT / 2); ++t) {
                    const int lm0 = lr0 + 16 * (t / (NT / 2)), ln0 = lc0 + 32 * (t % (NT / 2));
                    const int gm0 = bm0 + ch * ER + lm0, gn0 = bn0 + ln0;
                    if (gn0 >= N || gm0 >= M) continue;      // (never stored)
                    float v[2][4], gt[2][4];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[h][r] = stg[(lm0 + r) * SLD + ln0 + 16 * h];
                            gt[h][r] = 1.f;
                            if (g.gate && gm0 + r < M && gn0 + 16 * h < N) gt[h][r] = g.gate[(long)(gm0 + r) * g.ldg + gn0 + 16 * h];
                        }
                    const float bias[2] = {sbias[ln0], sbias[ln0 + 16]};
                    uint4 bits = make_uint4(0, 0, 0, 0);
                    if (g.drop_p > 0.f) bits = dropout_bits8(dkey, (unsigned)gm0 >> 2, drop_cc((unsigned)gn0));
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float x = v[h][r] + bias[h];
                            if (g.relu == 1) x = fmaxf(x, 0.f);
                            else if (g.relu == 2) x = tanhf(x);
                            if (g.gate) x = g.gate_mode == 1 ? x * (1.f - gt[h][r] * gt[h][r]) : (gt[h][r] > 0.f ? x * g.gate_scale : 0.f);
                            if (g.drop_p > 0.f) x = (pick_lot(bits, h, r) >= job.drop_thr) ? x * job.drop_scale : 0.f;
                            v[h][r] = x;
                        }
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int r = 0; r < 4; ++r) stg[(lm0 + r) * SLD + ln0 + 16 * h] = v[h][r];
                }
            }
        }
#if defined(SLNLP_PROBE_EPI) && SLNLP_PROBE_EPI == 2
        if (ch == 0) TS_EPI(7);
#else
        if (ch == 0) TS_EPI(6);
#endif
        const int gmc = bm0 + ch * ER;                       // first row of the chunk
        const bool late_gate = !early && g.gate;
        if (job.vec_out) {
            // the residual (and gate) pieces of eight passes are requested before they are needed: resid may alias C (an in-place
            // add), so inside the store loop the compiler has to keep every load behind the previous pass's stores -- eight
            // serialised round trips for a 128 x 128 tile; a thread only ever reads the elements it is about to write, so reading
            // them up front is safe.  (Eight at a time: 64 registers beside the accumulators the other chunk's waves still hold.)
            constexpr int NPASS = ER * BN / 4 / PTHREADS, PB = NPASS > 8 ? 8 : NPASS;
            static_assert(PTHREADS % (BN / 4) == 0, "a thread keeps its four columns over the passes");
            const int c4 = (tid % (BN / 4)) << 2, gn = bn0 + c4, row0 = tid / (BN / 4);
            constexpr int RSTEP = PTHREADS / (BN / 4);       // rows between a thread's passes
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!early && g.bias && gn < N) b4 = *reinterpret_cast<const float4*>(g.bias + gn);
#pragma unroll 1
            for (int pb = 0; pb < NPASS; pb += PB) {
                float4 rr[PB], gg[PB];
#pragma unroll
                for (int pass = 0; pass < PB; ++pass) {
                    const int gm = gmc + row0 + (pb + pass) * RSTEP;
                    rr[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
                    gg[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (g.resid && gm < M && gn < N) rr[pass] = *reinterpret_cast<const float4*>(g.resid + (long)gm * g.ldr + gn);
                    if (late_gate && gm < M && gn < N) gg[pass] = *reinterpret_cast<const float4*>(g.gate + (long)gm * g.ldg + gn);
                }
                if (pb == 0) __syncthreads();
#if !defined(SLNLP_PROBE_EPI) || SLNLP_PROBE_EPI != 2
                if (pb == 0 && ch == 0) TS_EPI(7);
#endif
#pragma unroll
                for (int pass = 0; pass < PB; ++pass) {
                    const int row = row0 + (pb + pass) * RSTEP, gm = gmc + row;
                    if (gm >= M || gn >= N) continue;         // N % 4 == 0 (vec_out): a live piece is 4 live columns
                    float4 v = *reinterpret_cast<const float4*>(stg + row * SLD + c4);
                    if (!early) {
                        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
                        if (g.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        if (late_gate) {
                            const float4 t = gg[pass];
                            // (__fmul_rn: the product is rounded before the residual is added -- never contracted into an FMA with
                            // it -- as on the route through the accumulator layout, where an LDS round trip separates the two)
                            if (g.gate_mode == 1) {
                                v.x = __fmul_rn(v.x, 1.f - t.x * t.x); v.y = __fmul_rn(v.y, 1.f - t.y * t.y);
                                v.z = __fmul_rn(v.z, 1.f - t.z * t.z); v.w = __fmul_rn(v.w, 1.f - t.w * t.w);
                            } else {
                                v.x = t.x > 0.f ? __fmul_rn(v.x, g.gate_scale) : 0.f; v.y = t.y > 0.f ? __fmul_rn(v.y, g.gate_scale) : 0.f;
                                v.z = t.z > 0.f ? __fmul_rn(v.z, g.gate_scale) : 0.f; v.w = t.w > 0.f ? __fmul_rn(v.w, g.gate_scale) : 0.f;
                            }
                        }
                    }
                    if (g.resid) { v.x += rr[pass].x; v.y += rr[pass].y; v.z += rr[pass].z; v.w += rr[pass].w; }
                    if (g.C) *reinterpret_cast<float4*>(g.C + (long)gm * g.ldc + gn) = v;
                    if (g.C_hi) {
                        PlaneOut po;
                        po.hi = g.C_hi;
                        po.lo = g.C_lo;
                        if (g.C_lo) store_planes4(po, (long)gm * g.ldc_p + gn, v);
                        else {
                            uint2 w;
                            w.x = head_bf16(v.x) | ((unsigned)head_bf16(v.y) << 16);
                            w.y = head_bf16(v.z) | ((unsigned)head_bf16(v.w) << 16);
                            *reinterpret_cast<uint2*>(g.C_hi + (long)gm * g.ldc_p + gn) = w;
                        }
                    }
                }
            }
        } else {
            __syncthreads();
#pragma unroll 1
            for (int e = tid; e < ER * BN; e += PTHREADS) {
                const int row = e / BN, col = e % BN, gm = gmc + row, gn = bn0 + col;
                if (gm >= M || gn >= N) continue;
                float v = stg[row * SLD + col];
                if (!early) {
                    if (g.bias) v += g.bias[gn];
                    if (g.relu == 1) v = fmaxf(v, 0.f);
                    if (g.gate) {
                        const float gt = g.gate[(long)gm * g.ldg + gn];
                        v = g.gate_mode == 1 ? __fmul_rn(v, 1.f - gt * gt) : (gt > 0.f ? __fmul_rn(v, g.gate_scale) : 0.f);
                    }
                }
                if (g.resid) v += g.resid[(long)gm * g.ldr + gn];
                if (g.C) g.C[(long)gm * g.ldc + gn] = v;
                if (g.C_hi) {
                    unsigned short h, l;
                    split_bf16(v, h, l);
                    g.C_hi[(long)gm * g.ldc_p + gn] = h;
                    if (g.C_lo) g.C_lo[(long)gm * g.ldc_p + gn] = l;
                }
            }
        }
    }
}

// `tab` != nullptr: a merged (lockstep) launch -- the jobs of K fits in a device-resident table, blockmap[block] = job
// ---------------------------------------------------------------------------------------------- fp8 forward tile ---
// precision 8: C[M,N] = epilogue( col_scale[n] * sum_k A8(m,k) B8(n,k) ), both operands OCP e4m3 byte planes, k-major.
// Same skeleton as plane_tile -- 64x64 output tile, 8 waves (16x32 each), LDS-DMA ring, one barrier per K-step -- with
// one byte per element: a K-step is 128 k (the 128-byte image row and its swizzle are those of the bf16 k-major image,
// so the DMA pattern is unchanged), a stage is 16 KiB (A + B) instead of 32, the ring is 4 deep (3 tiles in flight in the
// same 64 KiB), and a 16-byte fragment read feeds TWO v_mfma_f32_16x16x32_fp8_fp8 (its low and high 8 bytes; the k order
// inside a K-step is permuted the same way for A and B, which a contraction does not see).  Per algorithmic FLOP that is a
// quarter of the operand bytes and a third of the MFMA issue slots of the three-pass split-bf16 kernel.
constexpr int Q8_BK = 128;             // k (= bytes) per K-step
constexpr int Q8_IMG = PT * Q8_BK;     // bytes per 64-row operand image (8 KiB)
typedef __attribute__((ext_vector_type(8))) int i32x8;

__device__ __forceinline__ void dma_q8(const unsigned char* __restrict__ plane, long ld, int row0, int k0, unsigned char* img,
                                       int wave, int lane) {
    const int line = 8 * wave + (lane >> 3), ps = lane & 7;
    const unsigned char* src = plane + (long)(row0 + line) * ld + k0 + ((ps ^ (line & 7)) << 4);
    __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(img + wave * 1024), 16, 0, 0);
}

// fragment of v_mfma_scale_f32_16x16x128_f8f6f4: lane (row = lane & 15, quarter = lane >> 4) holds 32 consecutive k-bytes of its
// row -- 16-byte slots 2q and 2q + 1 of the 128-byte image row (each under the row's swizzle).  A and B use the same map, so
// whatever order the instruction gives the 128 k inside a step, both operands agree on it.
__device__ __forceinline__ i32x8 q8_frag(const unsigned char* __restrict__ img, int r0, int lane) {
    const int row = r0 + (lane & 15), q = lane >> 4;
    const uint4 lo = *reinterpret_cast<const uint4*>(img + row * Q8_BK + (((2 * q) ^ (row & 7)) << 4));
    const uint4 hi = *reinterpret_cast<const uint4*>(img + row * Q8_BK + (((2 * q + 1) ^ (row & 7)) << 4));
    return i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
}

// precision 8 tile: BM x BN outputs, 128-k steps through an NST-deep LDS-DMA ring, 8 waves as 4 (M) x 2 (N), ONE block-scaled
// MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 x e4m3, every block scale 2^0) per 16 x 16 tile and step: twice the MFMA rate
// of the plain fp8 forms (MI355X guide, matrix cores) -- the instruction behind the 5 PFLOP/s dense fp8 peak.  Per algorithmic
// FLOP a 256 x 256 tile moves 1/16 of the operand bytes of the three-pass split-bf16 64 x 64 tile.
template <int BM, int BN, int NST>
__device__ __forceinline__ void q8_tile(const PlaneJob& job, int lid, unsigned char* smem) {
    constexpr int SUBM = BM / PT, SUBN = BN / PT, MT = BM / 64, NT = BN / 32;
    constexpr int STAGE = (SUBM + SUBN) * Q8_IMG, PIECES = SUBM + SUBN;
    constexpr int E8M0_ONE = 0x7F7F7F7F;                        // four block scales of 2^0
    const slnlp_gemm_args& g = job.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * (BM / 4), wn0 = (wave & 1) * (BN / 2);
    int bx, by;
    {
        const int nwg = job.tiles_x * job.tiles_y;
        const int xcd = lid & 7, q = nwg >> 3, r = nwg & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lid >> 3);
        constexpr int GR = BM >= 128 ? 4 : 8;                   // tiles in groups of GR tile rows (see plane_tile)
        const int grp = t / (GR * job.tiles_x), rem = t - grp * (GR * job.tiles_x);
        const int rows_here = min(GR, job.tiles_y - grp * GR);
        bx = rem / rows_here;
        by = grp * GR + (rem - bx * rows_here);
    }
    const int bm0 = by * BM, bn0 = bx * BN, M = g.M, N = g.N;
    const int ktiles = (g.K + Q8_BK - 1) / Q8_BK;
    const int am_last = ((M + PT - 1) / PT - 1) * PT, bn_last = ((N + PT - 1) / PT - 1) * PT;
    const unsigned char* A8 = reinterpret_cast<const unsigned char*>(g.A_hi);
    const unsigned char* B8 = reinterpret_cast<const unsigned char*>(g.B_hi);
    auto issue = [&](int kt, int stage) {
        const int k0 = (kt < ktiles ? kt : 0) * Q8_BK;          // past-the-end prefetch re-reads a valid tile (never consumed)
        unsigned char* st = smem + stage * STAGE;
#pragma unroll
        for (int sm = 0; sm < SUBM; ++sm) dma_q8(A8, g.lda_p, BM > PT ? min(bm0 + sm * PT, am_last) : bm0, k0, st + sm * Q8_IMG, wave, lane);
#pragma unroll
        for (int sn = 0; sn < SUBN; ++sn) dma_q8(B8, g.ldb_p, BN > PT ? min(bn0 + sn * PT, bn_last) : bn0, k0, st + (SUBM + sn) * Q8_IMG, wave, lane);
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < NST - 1; ++t) issue(t, t);
    for (int kt = 0; kt < ktiles; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"((NST - 2) * PIECES) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(kt + NST - 1, (kt + NST - 1) % NST);
        const unsigned char* sa = smem + (kt % NST) * STAGE;
        const unsigned char* sb = sa + SUBM * Q8_IMG;
        i32x8 af[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = wm0 + 16 * i;
            af[i] = q8_frag(sa + (row / PT) * Q8_IMG, row % PT, lane);
        }
        constexpr int JB = NT > 4 ? 4 : NT;                     // B fragments live at a time (register pressure of the widest tile)
#pragma unroll
        for (int j0 = 0; j0 < NT; j0 += JB) {
            i32x8 bf[JB];
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                const int row = wn0 + 16 * (j0 + j);
                bf[j] = q8_frag(sb + (row / PT) * Q8_IMG, row % PT, lane);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < JB; ++j)
                    acc[i][j0 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af[i], bf[j], acc[i][j0 + j], 0, 0, 0, E8M0_ONE, 0, E8M0_ONE);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // drain the dummy prefetches before the ring is reused
    __builtin_amdgcn_s_barrier();
    // ---- epilogue: * col_scale -> +bias -> relu -> dropout -> +resid ; fp32 store (+ optional bf16 / fp8 planes)
    const int crow = (lane >> 4) << 2, ccol = lane & 15;
    if (job.vec_out) {
        // 64 tile rows at a time through an fp32 LDS image (the whole 256 x 256 tile would not fit): the waves that own the
        // rows apply the element-wise part in the accumulator layout, then every thread stores 16-byte row pieces.
        constexpr int SLD = BN + 4, CH = BM / PT;
        float* stg = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            if (ch) __syncthreads();
            if (wm0 / PT == ch) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int lm0 = wm0 % PT + 16 * i + crow, ln = wn0 + 16 * j + ccol;
                        const int gm0 = bm0 + ch * PT + lm0, gn = bn0 + ln;
                        const bool live = gn < N && gm0 < M;
                        const float cs = (live && g.col_scale) ? g.col_scale[gn] : 1.f, bias = (live && g.bias) ? g.bias[gn] : 0.f;
                        uint4 bits = make_uint4(0, 0, 0, 0);     // (the call of an odd column tile is its left neighbour's: the compiler merges them)
                        if (live && g.drop_p > 0.f) bits = dropout_bits8(dropout_key(g.rng, g.drop_site), (unsigned)gm0 >> 2, drop_cc((unsigned)gn));
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = acc[i][j][r] * cs + bias;
                            if (g.relu == 1) v = fmaxf(v, 0.f);
                            else if (g.relu == 2) v = tanhf(v);
                            if (live && gm0 + r < M && g.drop_p > 0.f) v = (pick_lot(bits, j & 1, r) >= job.drop_thr) ? v * job.drop_scale : 0.f;
                            stg[(lm0 + r) * SLD + ln] = v;
                        }
                    }
            }
            __syncthreads();
#pragma unroll
            for (int pass = 0; pass < PT * BN / 4 / PTHREADS; ++pass) {
                const int piece = pass * PTHREADS + tid, row = piece / (BN / 4), c4 = (piece % (BN / 4)) << 2;
                const int gm = bm0 + ch * PT + row, gn = bn0 + c4;
                if (gm >= M || gn >= N) continue;
                float4 v = *reinterpret_cast<const float4*>(stg + row * SLD + c4);
                if (g.resid) {
                    const float4 rr = *reinterpret_cast<const float4*>(g.resid + (long)gm * g.ldr + gn);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                if (g.C) *reinterpret_cast<float4*>(g.C + (long)gm * g.ldc + gn) = v;
                if (g.C_hi && g.C_lo) {
                    PlaneOut po;
                    po.hi = g.C_hi; po.lo = g.C_lo; po.q8 = g.C_q8;
                    store_planes4(po, (long)gm * g.ldc_p + gn, v);
                } else if (g.C_q8) {
                    *reinterpret_cast<unsigned*>(g.C_q8 + (long)gm * g.ldc_p + gn) = pack_fp8x4(v.x, v.y, v.z, v.w);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int gm0 = bm0 + wm0 + 16 * i + crow, gn = bn0 + wn0 + j * 16 + ccol;
            if (gn >= N || gm0 >= M) continue;
            const float cs = g.col_scale ? g.col_scale[gn] : 1.f, bias = g.bias ? g.bias[gn] : 0.f;
            uint4 bits = make_uint4(0, 0, 0, 0);
            if (g.drop_p > 0.f) bits = dropout_bits8(dropout_key(g.rng, g.drop_site), (unsigned)gm0 >> 2, drop_cc((unsigned)gn));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = gm0 + r;
                if (gm >= M) break;
                float v = acc[i][j][r] * cs + bias;
                if (g.relu == 1) v = fmaxf(v, 0.f);
                else if (g.relu == 2) v = tanhf(v);
                if (g.drop_p > 0.f) v = (pick_lot(bits, j & 1, r) >= job.drop_thr) ? v * job.drop_scale : 0.f;
                if (g.resid) v += g.resid[(long)gm * g.ldr + gn];
                if (g.C) g.C[(long)gm * g.ldc + gn] = v;
                if (g.C_hi) {
                    unsigned short h, l;
                    split_bf16(v, h, l);
                    g.C_hi[(long)gm * g.ldc_p + gn] = h;
                    if (g.C_lo) g.C_lo[(long)gm * g.ldc_p + gn] = l;
                }
                if (g.C_q8) g.C_q8[(long)gm * g.ldc_p + gn] = (unsigned char)(pack_fp8x4(v, 0.f, 0.f, 0.f) & 0xFFu);
            }
        }
}

// block map entry of a merged launch: job index | unit << PLACE_JOB_BITS (-1: padding)
constexpr int PLACE_JOB_BITS = 12, PLACE_JOBS = 1 << PLACE_JOB_BITS;

// GEO: the launch's tile geometry (table below); every job of a launch uses it (the host picks it per launch: plane_geo_for()).
struct GeoInfo { int bm, bn, bks, nst; };
constexpr int NGEO = 4;
constexpr GeoInfo GEO[NGEO] = {{64, 64, 64, 2}, {128, 128, 64, 2}, {128, 128, 32, 2}, {256, 256, 32, 2}};

template <int NSPLIT, int G>
__global__ __launch_bounds__(PTHREADS) void gemm_planes_kernel(const PlaneGroupParams P, const PlaneJob* __restrict__ tab,
                                                               const int* __restrict__ blockmap) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];   // the LDS ring (+ scratch after the loop)
    PlaneJob job;
    int lid;
    bool placed = false;
    if (tab) {
        // a merged launch's block map holds (job, unit of the job) per block, units placed on the XCDs by the host (plane_merge_place)
        const int m = blockmap[blockIdx.x];
        if (m < 0) return;                                                  // padding block (the XCDs' unit lists differ in length)
        job = tab[m & (PLACE_JOBS - 1)];
        lid = m >> PLACE_JOB_BITS;
        placed = true;
    } else {
        int j = 0;
        for (int t = 1; t < P.njobs; ++t)
            if ((int)blockIdx.x >= P.job[t].block_begin) j = t;             // block-uniform
        job = P.job[j];
        lid = blockIdx.x - job.block_begin;
    }
    launder(job.a);
    job.part = as_global(job.part); job.part_rs = as_global(job.part_rs); job.counters = as_global(job.counters);
    if (lid >= job.tiles_x * job.tiles_y * job.nks) return;
    constexpr GeoInfo g = GEO[G];
    probe_kernel_begin();
    plane_tile<NSPLIT, g.bm, g.bn, g.bks, g.nst>(job, lid, placed, smem);
    probe_kernel_end();
}

// ring bytes: stages x (A + B panels) x (hi, lo) x bks k x 2 B -- 64 KiB, 128 KiB, 64 KiB for three-pass jobs; 72 / 144 / 72 KiB for
// two-pass jobs (three stages of A_hi + B_hi + B_lo); the epilogue's fp32 image of the tile (bm x (bn + 4) floats: 17 KiB / 66 KiB)
// lives in the same memory.  A launch asks for the largest (one workgroup size per launch): two workgroups per CU still fit
constexpr size_t plane_lds(int geo) {
    const size_t ring3 = (size_t)GEO[geo].nst * 2 * (GEO[geo].bm + GEO[geo].bn) * GEO[geo].bks * sizeof(unsigned short);
    const size_t ring2 = (size_t)(GEO[geo].nst + 1) * (GEO[geo].bm + 2 * GEO[geo].bn) * GEO[geo].bks * sizeof(unsigned short);   // two-pass jobs: one stage more
    const size_t ring = ring3 > ring2 ? ring3 : ring2;
    const size_t image = ((size_t)(GEO[geo].bm > 128 ? 128 : GEO[geo].bm) * (GEO[geo].bn + 4) + GEO[geo].bn) * sizeof(float);   // (128 rows at a time, + the bias row)
    return ring > image ? ring : image;
}
constexpr int GROUP_COUNTERS = 4096;                       // ints at the head of the scratch buffer

// ---- precision 8 launches: their own kernel (all jobs of a launch are fp8 jobs), four geometries
struct Q8GeoInfo { int bm, bn, nst; };
constexpr int NQGEO = 2;
constexpr Q8GeoInfo QGEO[NQGEO] = {{64, 64, 4}, {128, 128, 2}};

// ---- the K-slices of a LARGE weight gradient meet in a launch of their own.  Inside the GEMM launch the last workgroup to arrive
// adds a tile's nks partials and stores it -- 48 workgroups (configs[4] in_proj: 48 tiles of 256 x 256, 5 slices) each pulling 1.25 MB
// through one CU and writing 256 KB while the other 208 CUs are done: 19.5 us of meeting + 19.3 us of epilogue behind a 142 us K
// loop (r05 timeline, `PROBE_PAIR=1 tools/probes/probe_tile_timeline.py`).  When that job has a launch to itself (slnlp_gemm_wd, both
// gradients large) its workgroups only write their partial tiles (hand_off 4) and this kernel adds them, slice order, from zero --
// the same sums, bit for bit -- on every CU: one (tile, 16 x 16-per-wave slab) per workgroup, all slices' loads in flight at once.
template <int G>
__global__ __launch_bounds__(PTHREADS) void plane_splitk_reduce_kernel(const PlaneJob job_in) {
    constexpr int BM = GEO[G].bm, BN = GEO[G].bn, MT = BM / 64, NT = BN / 32, TILE_FLOATS = BM * BN, SC = 17;
    PlaneJob job = job_in;
    launder(job.a);
    const slnlp_gemm_args& g = job.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x / (MT * NT), slab = blockIdx.x - tile * (MT * NT), i = slab / NT, j = slab - i * NT;
    const int by = tile / job.tiles_x, bx = tile - by * job.tiles_x, nks = job.nks;
    const int wm0 = (wave >> 1) * (BM / 4), wn0 = (wave & 1) * (BN / 2);
    u32x4 p[WD_MAX_SPLITK];
#pragma unroll
    for (int s = 0; s < WD_MAX_SPLITK; ++s) {
        if (s < nks) {
            const __amdgpu_buffer_rsrc_t q =
                __builtin_amdgcn_make_buffer_rsrc(job.part + ((long)tile * nks + s) * TILE_FLOATS, 0, TILE_FLOATS * 4, 0x00020000);
            p[s] = __builtin_amdgcn_raw_buffer_load_b128(q, tid * 16, slab * PTHREADS * 16, SC);
        }
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < WD_MAX_SPLITK; ++s)
        if (s < nks) acc += __builtin_bit_cast(f32x4, p[s]);
    const int gm0 = by * BM + wm0 + 16 * i + ((lane >> 4) << 2), gn = bx * BN + wn0 + 16 * j + (lane & 15);
    if (gn < g.N) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (gm0 + r < g.M) g.C[(long)(gm0 + r) * g.ldc + gn] = acc[r];
    }
    if (g.rowsum_a && bx == 0 && slab == 0 && tid < BM) {      // the bias gradient's slices (first column tile's workgroups wrote them)
        float rs_row = 0.f;
        for (int s = 0; s < nks; ++s) rs_row += __hip_atomic_load(job.part_rs + ((long)by * nks + s) * BM + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (by * BM + tid < g.M) g.rowsum_a[by * BM + tid] = rs_row;
    }
}
static const void* splitk_reduce_kernel_ptr(int geo) {
    switch (geo) {
        case 0: return (const void*)plane_splitk_reduce_kernel<0>;
        case 1: return (const void*)plane_splitk_reduce_kernel<1>;
        case 2: return (const void*)plane_splitk_reduce_kernel<2>;
        default: return (const void*)plane_splitk_reduce_kernel<3>;
    }
}

template <int G>
__global__ __launch_bounds__(PTHREADS) void gemm_q8_kernel(const PlaneGroupParams P, const PlaneJob* __restrict__ tab,
                                                           const int* __restrict__ blockmap) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    PlaneJob job;
    if (tab) {
        job = tab[blockmap[blockIdx.x]];
    } else {
        int j = 0;
        for (int t = 1; t < P.njobs; ++t)
            if ((int)blockIdx.x >= P.job[t].block_begin) j = t;
        job = P.job[j];
    }
    launder(job.a);
    const int lid = blockIdx.x - job.block_begin;
    if (lid >= job.tiles_x * job.tiles_y) return;
    q8_tile<QGEO[G].bm, QGEO[G].bn, QGEO[G].nst>(job, lid, reinterpret_cast<unsigned char*>(smem));
}
constexpr size_t q8_lds(int geo) {
    const size_t ring = (size_t)QGEO[geo].nst * (QGEO[geo].bm + QGEO[geo].bn) * Q8_BK;
    const size_t image = (size_t)PT * (QGEO[geo].bn + 4) * sizeof(float);
    return ring > image ? ring : image;
}
static const void* q8_kernel_ptr(int geo) {
    switch (geo) {
        case 1: return (const void*)gemm_q8_kernel<1>;
        default: return (const void*)gemm_q8_kernel<0>;
    }
}
static int q8_geo_of_knob(int knob) { return knob == 64 ? 0 : knob == 128 ? 1 : -1; }
static std::atomic<int> g_q8_geo{[] { const char* e = getenv("SLNLP_Q8_TILE"); return q8_geo_of_knob(e ? atoi(e) : 0); }()};
// 128 x 128 (two stages in 64 KiB: two workgroups per CU) once the launch has at least two of them per CU -- measured
// (tools/bench_fp8_tiles.py): configs[4] in_proj 618 -> 890 TFLOP/s, a 16-fit merged launch 415 -> 589; a single cfg2 launch
// (228 tiles) 303 -> 265.  Four-stage rings and 256-wide tiles (one workgroup per CU) were built and lost by 30-50 %.
static int q8_geo_auto(const PlaneJob* jobs, int njobs) {
    const int forced = g_q8_geo.load(std::memory_order_relaxed);
    if (forced >= 0) return forced;
    long units = 0;
    for (int i = 0; i < njobs; ++i) units += (long)ceil_div(jobs[i].a.M, 128) * ceil_div(jobs[i].a.N, 128);
    return units >= 512 ? 1 : 0;
}

// Probe switch for the split-K meeting point (tools/probes): env SLNLP_SPLITK_MODE = 1: never split K;
// 2: the last workgroup runs an agent-scope acquire before it reads the partials; 3: 2 + an agent-scope release in front of
// every arrival count.  Read once per process; 0 / unset = the shipped protocol.
static int splitk_mode() {
    static const int mode = [] { const char* e = getenv("SLNLP_SPLITK_MODE"); return e ? atoi(e) : 0; }();
    return mode;
}

// ---- which geometry a launch uses.  -1 = automatic (plane_geo_for), 0 .. NGEO-1 = forced (slnlp_set_plane_tile: tuning, tests)
static int geo_of_knob(int knob) { return knob == 64 ? 0 : knob == 128 ? 1 : knob == 12832 ? 2 : knob == 256 ? 3 : -1; }
static std::atomic<int> g_plane_geo{[] { const char* e = getenv("SLNLP_PLANE_TILE"); return geo_of_knob(e ? atoi(e) : 0); }()};
constexpr int BIG_TILE_MIN_UNITS = 200;    // a launch takes 128 x 128 tiles when it still has at least this many of them

// 128 x 128 tiles move half the operand bytes per FLOP through the L2 -> LDS path and issue half the LDS fragment reads per MFMA;
// they pay once a launch has about a workgroup per CU (measured, tools/bench_plane_tiles.py: 220 tiles +10 %, 124 tiles -60 %).
// Which ring (per-workgroup timelines, tools/probes/probe_tile_timeline.py: a 128 x 128 x 64 workgroup spends 1.6 us filling the
// ring, 11 us in the K loop of K = 512 and 6 us in its epilogue -- with one workgroup per CU nothing covers the first and the last):
//   * 32-k x 2 stages (64 KiB, two workgroups per CU, each the other's cover) once every K loop is at least 1024 long
//     (configs[4] gradient groups: +10 %), and from about 300 tiles up whatever K: forward launches (15 merged fits
//     [36000 x 512] x [512 x 512] 88 -> 74 us, configs[4] in_proj 342 -> 293 us) and the merged gradient groups of a lockstep
//     unit (a 15-fit cfg2 step 20.2 -> 19.5 ms with every launch on this ring; 21.3 on the 64-k ring, 21.8 on 64 x 64 tiles);
//   * 64-k x 2 stages (128 KiB, one workgroup per CU) for launches of about one tile per CU (cfg2 in_proj, 228 tiles: 20.6 against
//     21.3 us; at K = 512 the 32-k ring pays twice the barriers and has no second workgroup to hide them behind).
constexpr int TWO_PER_CU_UNITS = 300;    // (a 4-fit lockstep step, 496- and 320-tile launches: 6.83 -> 6.36 ms on the 32-k ring; 2 fits, 248 tiles: equal)
//   * 256 x 256 x 32 (128 KiB, one workgroup per CU, half the operand bytes per FLOP of the 128-wide tiles) for FORWARD launches --
//     both operands k-major, no split-K -- whose K loops are at least 1024 long and whose tiles fill the 256 CUs' rounds to 85 %:
//     configs[4] in_proj [16384 x 1024] x [1024 x 3072], 768 tiles: 285 -> 255 us (405 TFLOP/s); FFN2 [16384 x 3072] x [3072 x 1024],
//     256 tiles: 290 -> 236 us (436 TFLOP/s).  A single round of them needs K >= 2048: configs[4]'s out_proj (256 tiles, K = 1024) wins
//     back to back (88 against 102 us) and loses inside the train step, where nothing covers its one workgroup per CU behind the
//     previous kernel (step 15.00 -> 14.90 ms without it).  Not below: at K = 512 the tile's longer fill / drain is not paid back (846 tiles:
//     185 against 183 us; 282 tiles: 88 against 67), half-empty rounds lose outright (128 tiles: 77 against 52 us), and the merged
//     gradient groups (split-K partial tiles of 256 KiB, two-pass rings) stay 7 % ahead on the 128 x 128 x 32 ring.
struct LaunchShape {
    long units128 = 0, units256 = 0;     // workgroups the launch would have at 128 x 128 / 256 x 256 tiles
    int min_k = 1 << 30;                 // shortest K loop (per split)
    bool forward = true;                 // every job: both operands k-major, no split-K
    int njobs = 0;
};
static void shape_add(LaunchShape& s, const slnlp_gemm_args& a, int nks) {
    if (nks < 1) nks = 1;
    s.units128 += (long)ceil_div(a.M, 128) * ceil_div(a.N, 128) * nks;
    s.units256 += (long)ceil_div(a.M, 256) * ceil_div(a.N, 256) * nks;
    s.min_k = std::min(s.min_k, a.K / nks);
    s.forward = s.forward && a.a_kmajor && a.b_kmajor && nks == 1 && a.precision != 2;
    ++s.njobs;
}
static int plane_geo_auto(const LaunchShape& s) {
    const int forced = g_plane_geo.load(std::memory_order_relaxed);
    if (forced >= 0) return forced;
    if (s.units128 < BIG_TILE_MIN_UNITS) return 0;
    if (s.forward && s.min_k >= 1024 && (s.units256 >= 512 || s.min_k >= 2048) && s.units256 * 100 >= (s.units256 + 255) / 256 * 256 * 85) return 3;
    // ... and for a gradient product launched ALONE (gemm_planes_wd below: the data and the weight gradient of a large dY take a
    // launch each) with long K loops and a filled round of 256-wide tiles: configs[4] in_proj, data gradient 256 tiles x K 3072 --
    // 212 -> 192 us; weight gradient 48 tiles x 5 K slices -- 253 -> 236 us (profiles/r05_plane_pair_sweep.txt)
    if (s.njobs == 1 && !s.forward && s.min_k >= 2048 && s.units256 >= 192 && s.units256 * 100 >= (s.units256 + 255) / 256 * 256 * 85) return 3;
    return (s.min_k >= 1024 || s.units128 >= TWO_PER_CU_UNITS) ? 2 : 1;
}
int plane_geo_for(const slnlp_gemm_args* jobs, const int* split_k, int njobs) {
    LaunchShape s;
    for (int i = 0; i < njobs; ++i) {
        if (jobs[i].precision == 8) return 0;                // (fp8 launches have their own geometries: q8_geo_auto)
        shape_add(s, jobs[i], split_k ? split_k[i] : 1);
    }
    return plane_geo_auto(s);
}

// set the kernels' LDS attribute up front (plan creation) so it never lands inside a graph capture
int gemm_planes_init() {
    static DeviceOnce once;
    return once.run([]() -> int {
        bool ok = true;
        for (int prec = 1; prec <= 3; prec += 2)
            for (int geo = 0; geo < NGEO; ++geo)
                ok = ok && hipFuncSetAttribute(gemm_planes_kernel_ptr(prec, geo), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plane_lds(geo)) == hipSuccess;
        for (int geo = 0; geo < NQGEO; ++geo)
            ok = ok && hipFuncSetAttribute(q8_kernel_ptr(geo), hipFuncAttributeMaxDynamicSharedMemorySize, (int)q8_lds(geo)) == hipSuccess;
        if (!ok) {
            set_error("gemm_planes_init: cannot raise dynamic LDS limit: %s", hipGetErrorString(hipGetLastError()));
            return SLNLP_ERR_LAUNCH;
        }
        return 0;
    });
}

static int check_plane_job(const slnlp_gemm_args& a) {
    SLNLP_CHECK_ARG(a.A_hi && a.B_hi, "gemm_planes: operand planes required");
    SLNLP_CHECK_ARG(a.C || a.C_hi, "gemm_planes: no output");
    SLNLP_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "gemm_planes: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    if (a.precision == 8) {
        SLNLP_CHECK_ARG(a.a_kmajor && a.b_kmajor, "gemm_planes: precision 8 (fp8) is built for k-major operands (forward products)");
        SLNLP_CHECK_ARG(a.lda_p % 128 == 0 && a.ldb_p % 128 == 0 && a.lda_p >= a.K && a.ldb_p >= a.K,
                        "gemm_planes: fp8 plane row strides must be multiples of 128 bytes and cover K");
        SLNLP_CHECK_ARG(!a.gate && !a.rowsum_a && a.drop_head_dim == 0, "gemm_planes: fp8 jobs take bias / relu / dropout / residual only");
        SLNLP_CHECK_ARG((((uintptr_t)a.A_hi | (uintptr_t)a.B_hi) & 15) == 0, "gemm_planes: planes must be 16-byte aligned");
        SLNLP_CHECK_ARG(!a.C || a.ldc >= a.N, "gemm_planes: ldc < N");
        SLNLP_CHECK_ARG(a.drop_p >= 0.f && a.drop_p < 1.f && (a.drop_p == 0.f || a.rng), "gemm_planes: bad dropout args");
        return 0;
    }
    SLNLP_CHECK_ARG(a.precision == 1 || (a.precision == 3 && a.A_lo && a.B_lo) || (a.precision == 2 && a.B_lo),
                    "gemm_planes: precision 3 needs both lo planes, precision 2 the lo plane of B");
    SLNLP_CHECK_ARG(a.precision != 2 || !a.b_kmajor, "gemm_planes: precision 2 is built for the gradient layouts (B m-major)");
    SLNLP_CHECK_ARG(a.lda_p % 64 == 0 && a.ldb_p % 64 == 0, "gemm_planes: plane row strides must be multiples of 64");
    SLNLP_CHECK_ARG((((uintptr_t)a.A_hi | (uintptr_t)a.B_hi | (uintptr_t)a.A_lo | (uintptr_t)a.B_lo) & 15) == 0,
                    "gemm_planes: planes must be 16-byte aligned");
    SLNLP_CHECK_ARG(!a.C || a.ldc >= a.N, "gemm_planes: ldc < N");
    SLNLP_CHECK_ARG(!a.C_hi || a.ldc_p >= a.N, "gemm_planes: ldc_p < N");
    SLNLP_CHECK_ARG(a.drop_p >= 0.f && a.drop_p < 1.f && (a.drop_p == 0.f || a.rng), "gemm_planes: bad dropout args");
    SLNLP_CHECK_ARG(!(a.a_kmajor == 0 && a.b_kmajor != 0), "gemm_planes: layout (A m-major, B k-major) not built");
    SLNLP_CHECK_ARG(a.drop_head_dim == 0, "gemm_planes: per-head dropout is only built for fp32-operand GEMMs");
    return 0;
}

// split-K scratch: arrival counters, then per job its partial tiles [tile][split][bm x bn] and row sums [tile_y][split][bm].
// Sized for whichever geometry the launch may take (M and N rounded up to 256 cover all of them).
size_t gemm_group_scratch_bytes(const slnlp_gemm_args* jobs, const int* split_k, int njobs) {
    size_t floats = 0;
    for (int i = 0; i < njobs; ++i) {
        const int nks = split_k ? split_k[i] : 1;
        if (nks <= 1) continue;
        const size_t n256 = (size_t)ceil_div(jobs[i].N, 256) * 256, m256 = (size_t)ceil_div(jobs[i].M, 256) * 256;
        floats += n256 * m256 * nks + m256 * nks;
    }
    return GROUP_COUNTERS * sizeof(int) + floats * sizeof(float);
}

int gemm_planes_group(const slnlp_gemm_args* jobs, const int* split_k, int njobs, void* scratch, size_t scratch_bytes,
                      hipStream_t s, bool defer_reduce) {
    SLNLP_CHECK_ARG(jobs && njobs >= 1 && njobs <= MAX_JOBS, "gemm_group: 1..%d jobs", MAX_JOBS);
    PlaneGroupParams P;
    P.njobs = njobs;
    const int geo = plane_geo_for(jobs, split_k, njobs);
    int blocks = 0, ctr = 0;
    size_t off = GROUP_COUNTERS * sizeof(int);
    for (int i = 0; i < njobs; ++i) {
        const slnlp_gemm_args& a = jobs[i];
        SLNLP_TRY(check_plane_job(a));
        // one launch = one kernel family: fp8, single-pass bf16, or split-bf16 (precision 3 and 2 jobs may share a launch)
        auto family = [](int p) { return p == 2 ? 3 : p; };
        SLNLP_CHECK_ARG(family(a.precision) == family(jobs[0].precision), "gemm_group: jobs of one launch share the precision family");
        PlaneJob& j = P.job[i];
        j.a = a;
        j.drop_thr = dropout_threshold(a.drop_p);
        j.drop_scale = 1.f / (1.f - a.drop_p);
        j.variant = a.precision == 8 ? 3 : (a.a_kmajor && a.b_kmajor) ? 0 : a.a_kmajor ? 1 : 2;
        if (a.precision == 2) j.variant = a.a_kmajor ? 3 : 4;
        j.hand_off = splitk_mode() >= 2 ? splitk_mode() : 0;
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        j.vec_out = a.N % 4 == 0 && (!a.C || (a.ldc % 4 == 0 && al16(a.C))) && (!a.resid || (a.ldr % 4 == 0 && al16(a.resid))) &&
                    (!a.C_hi || (a.ldc_p % 4 == 0 && (reinterpret_cast<uintptr_t>(a.C_hi) & 7) == 0 &&
                                 (!a.C_lo || (reinterpret_cast<uintptr_t>(a.C_lo) & 7) == 0)));
        if (a.precision == 8) SLNLP_CHECK_ARG(!split_k || split_k[i] <= 1, "gemm_group: fp8 jobs do not split K");
        j.tiles_x = ceil_div(a.N, GEO[geo].bn);
        j.tiles_y = ceil_div(a.M, GEO[geo].bm);
        const int ktiles = ceil_div(a.K, PT);
        int nks = split_k ? split_k[i] : 1;
        if (nks < 1 || splitk_mode() == 1) nks = 1;          // (probe mode 1: no split-K at all)
        if (nks > ktiles) nks = ktiles;
        j.nks = nks;
        j.block_begin = blocks;
        j.part = nullptr;
        j.part_rs = nullptr;
        j.counters = nullptr;
        const int tiles = j.tiles_x * j.tiles_y;
        if (nks > 1) {
            SLNLP_CHECK_ARG(scratch && (((uintptr_t)scratch) & 15) == 0, "gemm_group: split-K needs a 16-byte aligned scratch buffer");
            SLNLP_CHECK_ARG(ctr + tiles <= GROUP_COUNTERS, "gemm_group: more than %d split-K tiles in one launch", GROUP_COUNTERS);
            j.counters = reinterpret_cast<int*>(scratch) + ctr;
            ctr += tiles;
            j.part = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + off);
            // region sizes do not depend on the geometry (M and N rounded up to 256): a merged launch may re-tile the job
            const size_t n256 = (size_t)ceil_div(a.N, 256) * 256, m256 = (size_t)ceil_div(a.M, 256) * 256;
            off += n256 * m256 * nks * sizeof(float);
            j.part_rs = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + off);
            off += m256 * nks * sizeof(float);
            SLNLP_CHECK_ARG(off <= scratch_bytes, "gemm_group: scratch too small (%zu > %zu bytes)", off, scratch_bytes);
        }
        blocks += tiles * nks;
    }
    SLNLP_TRY(gemm_planes_init());
    const void* fn = gemm_planes_kernel_ptr(jobs[0].precision, geo);
    size_t lds = plane_lds(geo);
    if (jobs[0].precision == 8) {        // fp8 launch: its own kernel and geometries; re-tile the jobs for the one it takes
        const int qg = q8_geo_auto(P.job, njobs);
        blocks = 0;
        for (int i = 0; i < njobs; ++i) {
            PlaneJob& j = P.job[i];
            j.tiles_x = ceil_div(j.a.N, QGEO[qg].bn);
            j.tiles_y = ceil_div(j.a.M, QGEO[qg].bm);
            j.block_begin = blocks;
            blocks += j.tiles_x * j.tiles_y;
        }
        fn = q8_kernel_ptr(qg);
        lds = q8_lds(qg);
    }
    // a split-K job with a launch to itself and a plain store for an epilogue: its slices meet in a second launch (see
    // plane_splitk_reduce_kernel).  Not under a lockstep recorder: a merged launch re-tiles its jobs and keeps the in-kernel meeting
    const slnlp_gemm_args& a0 = jobs[0];
    const bool deferred = defer_reduce && njobs == 1 && P.job[0].nks > 1 && P.job[0].nks <= WD_MAX_SPLITK && !recording() && a0.precision != 8 &&
                          splitk_mode() == 0 && a0.C && !a0.bias && !a0.relu && !a0.gate && a0.drop_p == 0.f && !a0.resid && !a0.C_hi && !a0.C_q8 &&
                          !a0.col_scale;
    if (deferred) P.job[0].hand_off = 4;
    if (recording()) return record_op(fn, dim3(blocks), dim3(PTHREADS), lds, REC_PLANE_GROUP, &P, sizeof(P), "gemm_planes");
    void* args[3];
    const PlaneJob* tab = nullptr;
    const int* bmap = nullptr;
    args[0] = &P; args[1] = &tab; args[2] = &bmap;
    const int timed = launch_timer_begin(s);                 // (bench.py: this kernel's launches timed INSIDE a train step)
    if (hipLaunchKernel(fn, dim3(blocks), dim3(PTHREADS), args, lds, s) != hipSuccess) {
        set_error("gemm_planes: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    if (deferred) {
        PlaneJob j = P.job[0];
        void* rargs[1] = {&j};
        const int rblocks = j.tiles_x * j.tiles_y * (GEO[geo].bm / 64) * (GEO[geo].bn / 32);
        if (hipLaunchKernel(splitk_reduce_kernel_ptr(geo), dim3(rblocks), dim3(PTHREADS), rargs, 0, s) != hipSuccess) {
            set_error("gemm_planes (split-K reduce): %s", hipGetErrorString(hipGetLastError()));
            return SLNLP_ERR_LAUNCH;
        }
    }
    if (timed >= 0) launch_timer_end(timed, s, blocks, njobs, geo);
    return 0;
}

// ---- the gradient pair of one dY: dW = dY^T x (split-K over the tokens) and dX = dY W.
// Split factor of the weight gradient (K loop = tokens): the plane GEMM keeps 2 workgroups per CU resident (512 slots) and a K-step
// costs about the same in every workgroup, so estimate   time ~ rounds(total workgroups / 512) x longest K loop   (+1 step for the
// split-K meeting) and take the best split; more workgroups than slots only adds a second, mostly empty round.
static int wd_split_grouped(const slnlp_gemm_args& wg, const slnlp_gemm_args& dg) {
    auto cd = [](int a, int b) { return (a + b - 1) / b; };
    const int tw = cd(wg.M, 64) * cd(wg.N, 64), td = cd(dg.M, 64) * cd(dg.N, 64), kw = cd(wg.K, 64), kd = cd(dg.K, 64);
    int best = 1, best_cost = 1 << 30;
    for (int n = 1; n <= WD_MAX_SPLITK && n <= kw; ++n) {
        const int len = cd(kw, n) + (n > 1 ? 1 : 0);
        const int cost = cd(tw * n + td, 512) * (len > kd ? len : kd);
        if (cost < best_cost) { best_cost = cost; best = n; }
    }
    return best;
}
// ONE launch for both -- the weight gradient's workgroups fill the CUs the data gradient leaves idle, and there is no second
// dispatch to pay for -- unless both are large: then neither has idle CUs to offer, sharing a launch only mixes their working
// sets in the L2s (1.70 GB fetched per launch together against 0.48 + 0.70 GB alone at configs[4]'s in_proj,
// profiles/r05_pmc_plane_pair.txt), and alone each takes the tile and the split that suit it: the data gradient one full round of
// 256-wide tiles, the weight gradient as many K slices as fill one round of them (48 tiles x 5).  Measured on one box
// (profiles/r05_plane_pair_sweep.txt): 480 us grouped (128-wide tiles, 6 slices) -> 192 + 236 us.
WdPlan gemm_planes_wd_plan(const slnlp_gemm_args& wg, const slnlp_gemm_args& dg) {
    WdPlan p;
    p.separate = 0;
    p.split = wd_split_grouped(wg, dg);
    const long d128 = (long)ceil_div(dg.M, 128) * ceil_div(dg.N, 128), w256 = (long)ceil_div(wg.M, 256) * ceil_div(wg.N, 256);
    static const bool allow_separate = [] { const char* e = getenv("SLNLP_WD_SEPARATE"); return !(e && atoi(e) == 0); }();   // (A / B measurements)
    if (allow_separate && g_plane_geo.load(std::memory_order_relaxed) < 0 && wg.precision != 8 && d128 >= 1024 && dg.K >= 1024 && w256 >= 24 && w256 <= 256 && wg.K >= 4096) {
        p.separate = 1;
        int n = (int)(256 / w256);                          // K slices that fill one round of 256-wide tiles
        n = std::min(n, std::min(WD_MAX_SPLITK, ceil_div(wg.K, 2048)));
        p.split = std::max(n, 1);
    }
    return p;
}
int gemm_planes_wd(const slnlp_gemm_args& wg, const slnlp_gemm_args& dg, void* scratch, size_t scratch_bytes, hipStream_t st) {
    const WdPlan p = gemm_planes_wd_plan(wg, dg);
    if (p.separate) {
        const int one = 1;
        static const bool defer = [] { const char* e = getenv("SLNLP_WD_DEFER"); return !(e && atoi(e) == 0); }();   // (A / B measurements)
        SLNLP_TRY(gemm_planes_group(&wg, &p.split, 1, scratch, scratch_bytes, st, defer));
        return gemm_planes_group(&dg, &one, 1, scratch, scratch_bytes, st);
    }
    const slnlp_gemm_args jobs[2] = {wg, dg};
    const int split[2] = {p.split, 1};
    return gemm_planes_group(jobs, split, 2, scratch, scratch_bytes, st);
}

// A recorded job re-tiled for a merged launch (lockstep.hip): only the tile grid changes -- the K partition (nks) and the
// scratch regions stay, so every output element is accumulated in the same order and the result keeps its bits.
void plane_job_retile(PlaneJob& j, int geo) {
    j.tiles_x = ceil_div(j.a.N, GEO[geo].bn);
    j.tiles_y = ceil_div(j.a.M, GEO[geo].bm);
}
// The concatenated job list of a merged launch (lockstep.hip): pick the geometry for ALL its tiles and re-tile every job.
// `recorded_fn` is the kernel one fit's own launch recorded (it tells the family: fp8, split-bf16 or single-pass bf16).
void plane_merge_geometry(const void* recorded_fn, PlaneJob* jobs, int njobs, const void** fn, size_t* lds) {
    bool q8 = false;
    for (int geo = 0; geo < NQGEO; ++geo) q8 = q8 || recorded_fn == q8_kernel_ptr(geo);
    if (q8) {
        const int qg = q8_geo_auto(jobs, njobs);
        for (int i = 0; i < njobs; ++i) {
            jobs[i].tiles_x = ceil_div(jobs[i].a.N, QGEO[qg].bn);
            jobs[i].tiles_y = ceil_div(jobs[i].a.M, QGEO[qg].bm);
        }
        *fn = q8_kernel_ptr(qg);
        *lds = q8_lds(qg);
        return;
    }
    int prec = 1;
    for (int geo = 0; geo < NGEO; ++geo)
        if (recorded_fn == gemm_planes_kernel_ptr(3, geo)) prec = 3;
    LaunchShape shape;
    for (int i = 0; i < njobs; ++i) shape_add(shape, jobs[i].a, jobs[i].nks);
    const int geo = plane_geo_auto(shape);
    for (int i = 0; i < njobs; ++i) plane_job_retile(jobs[i], geo);
    *fn = gemm_planes_kernel_ptr(prec, geo);
    *lds = plane_lds(geo);
}

// The block map of a merged plane-GEMM launch (lockstep.hip): entry b = (job, unit of the job) of workgroup b, which runs on XCD
// b % 8 (each with a private 4 MiB L2) in order b / 8.  A solo launch gives every XCD a contiguous run of ITS job's units
// (plane_tile); the K fits of a merged launch are as many small jobs -- a fit's weight gradient is 16 tiles x 3 K-slices -- and
// dealing each job's units over all eight XCDs made every XCD fetch a part of every job's operand panels (the weight-gradient
// workgroups of a 15-fit launch ran 2.1 us per K-step on HBM traffic: profiles/r04_lockstep_plane_timeline.txt).  With at least
// four jobs and 256 units the units of ALL jobs, in job order, are cut into eight contiguous runs of equal WORK (units x K-steps)
// instead: an XCD then holds a few whole jobs and a panel is fetched by one L2 (15-fit gradient groups 168 -> 142 us; a lockstep
// step of 3 / 4 / 8 / 15 fits 4.97 / 5.71 / 9.68 / 16.5 -> 4.87 / 5.63 / 9.18 / 15.9 ms).  One fit's own two jobs keep the per-job
// layout (2.80 ms against 2.90: a long weight-gradient slice per XCD beside XCDs with short tiles only).
// Which workgroup computes a unit has no bearing on its result.
constexpr long PLACE_MIN_UNITS = 256;
constexpr int PLACE_MIN_JOBS = 4;
int plane_merge_place(const void* merged_fn, const PlaneJob* jobs, int njobs, std::vector<int>& map) {
    int geo = -1;
    for (int g = 0; g < NGEO; ++g)
        if (merged_fn == gemm_planes_kernel_ptr(1, g) || merged_fn == gemm_planes_kernel_ptr(3, g)) geo = g;
    if (geo < 0) return 0;                                  // (fp8 launches keep their own layout)
    if (njobs > PLACE_JOBS) return -1;
    long total = 0, work = 0;
    std::vector<long> ksteps(njobs);
    for (int j = 0; j < njobs; ++j) {
        const PlaneJob& J = jobs[j];
        const long units = (long)J.tiles_x * J.tiles_y * J.nks;
        if (units >= (1L << (31 - PLACE_JOB_BITS))) return -1;
        ksteps[j] = (J.a.K / (J.nks > 0 ? J.nks : 1) + GEO[geo].bks - 1) / GEO[geo].bks + 6;      // (+ fill, meeting, epilogue)
        total += units;
        work += units * ksteps[j];
    }
    map.clear();
    if (total < PLACE_MIN_UNITS || njobs < PLACE_MIN_JOBS) {
        // per job, as a solo launch lays it out: jobs start on multiples of 8, block lid of a job -> XCD lid % 8 -> the unit plane_tile's
        // own mapping would give it
        for (int j = 0; j < njobs; ++j) {
            const int units = jobs[j].tiles_x * jobs[j].tiles_y * jobs[j].nks, padded = (units + 7) & ~7, q = units >> 3, r = units & 7;
            for (int lid = 0; lid < padded; ++lid) {
                const int xcd = lid & 7, t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lid >> 3);
                map.push_back(lid < units ? (j | (t << PLACE_JOB_BITS)) : -1);
            }
        }
        return 1;
    }
    std::vector<int> units[8];
    long done = 0;                                          // work placed so far
    int x = 0;
    for (int j = 0; j < njobs; ++j) {
        const int n = jobs[j].tiles_x * jobs[j].tiles_y * jobs[j].nks;
        for (int t = 0; t < n; ++t) {
            while (x < 7 && done * 8 >= work * (x + 1)) ++x;
            units[x].push_back(j | (t << PLACE_JOB_BITS));
            done += ksteps[j];
        }
    }
    // inside an XCD's run the units with the longest K loops go first (a fit's weight-gradient slices before its data-gradient tiles):
    // the launch ends when the last long workgroup does, so none of them should start late
    for (int k = 0; k < 8; ++k)
        std::stable_sort(units[k].begin(), units[k].end(),
                         [&](int a, int b) { return ksteps[a & (PLACE_JOBS - 1)] > ksteps[b & (PLACE_JOBS - 1)]; });
    size_t longest = 0;
    for (int k = 0; k < 8; ++k) longest = std::max(longest, units[k].size());
    map.assign(longest * 8, -1);
    for (int k = 0; k < 8; ++k)
        for (size_t i = 0; i < units[k].size(); ++i) map[i * 8 + k] = units[k][i];
    return 1;
}

template <int NSPLIT>
static const void* kernel_of(int geo) {
    switch (geo) {
        case 1: return (const void*)gemm_planes_kernel<NSPLIT, 1>;
        case 2: return (const void*)gemm_planes_kernel<NSPLIT, 2>;
        case 3: return (const void*)gemm_planes_kernel<NSPLIT, 3>;
        default: return (const void*)gemm_planes_kernel<NSPLIT, 0>;
    }
}
const void* gemm_planes_kernel_ptr(int precision, int geo) { return precision == 1 ? kernel_of<1>(geo) : kernel_of<3>(geo); }

int gemm_planes(const slnlp_gemm_args& a, hipStream_t s) { return gemm_planes_group(&a, nullptr, 1, nullptr, 0, s); }

// fp32 [R, C] (row stride ld) -> bf16 hi / lo planes with row stride ldp (valid region only; the padding of the
// planes stays zero from their one-time memset).  Used for the weights once per step and by tests.
__device__ __forceinline__ void split_planes_body(const float* __restrict__ x, long ld, int R, int C, unsigned short* __restrict__ hi,
                                                  unsigned short* __restrict__ lo, long ldp) {
    const long n4 = (long)R * (C >> 2);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const int r = (int)(i / (C >> 2)), c = (int)(i % (C >> 2)) << 2;
        const float4 v = *reinterpret_cast<const float4*>(x + (long)r * ld + c);
        const float f[4] = {v.x, v.y, v.z, v.w};
        unsigned short h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split_bf16(f[e], h[e], l[e]);
        uint2 w;
        w.x = h[0] | ((unsigned)h[1] << 16); w.y = h[2] | ((unsigned)h[3] << 16);
        *reinterpret_cast<uint2*>(hi + (long)r * ldp + c) = w;
        if (lo) {
            w.x = l[0] | ((unsigned)l[1] << 16); w.y = l[2] | ((unsigned)l[3] << 16);
            *reinterpret_cast<uint2*>(lo + (long)r * ldp + c) = w;
        }
    }
}

SLNLP_ZKERNEL(split_planes_kernel, 256, split_planes_body)

int split_planes(const float* x, int64_t ld, int R, int C, unsigned short* hi, unsigned short* lo, int64_t ldp, hipStream_t st) {
    SLNLP_CHECK_ARG(x && hi && R > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ldp % 4 == 0 && ldp >= C,
                    "split_planes: bad args (C, ld, ldp must be multiples of 4)");
    int grid = ceil_div((long)R * (C / 4), 256);
    if (grid > 4096) grid = 4096;
    return zlaunch(split_planes_kernel, dim3(grid), 256, 0, st, "split_planes", x, (long)ld, R, C, hi, lo, (long)ldp);
}

// fp32 rows -> e4m3 rows + one scale per row (one wave per row).  The weight operand of precision 8: scale = max|row| / 448.
// `rows` != nullptr: a table of {source offset (floats), K} entries over one arena (q plane at the same offsets, scale[entry]).
__device__ __forceinline__ void quant_rows_fp8_body(const float* __restrict__ x, long ld, int R, int K, unsigned char* __restrict__ q,
                                                    long ldq, float* __restrict__ scale, const QuantRow* __restrict__ rows) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* src = rows ? x + rows[r].off : x + (long)r * ld;
    unsigned char* dst = rows ? q + rows[r].off : q + (long)r * ldq;
    const int k_len = rows ? rows[r].K : K;
    float m = 0.f;
    for (int k = lane * 4; k < k_len; k += 256) {
        const float4 v = *reinterpret_cast<const float4*>(src + k);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    m = wave_max(m);
    const float sc = m > 0.f ? m * (1.f / 448.f) : 1.f, inv = 1.f / sc;
    if (lane == 0) scale[r] = sc;
    for (int k = lane * 4; k < k_len; k += 256) {
        const float4 v = *reinterpret_cast<const float4*>(src + k);
        *reinterpret_cast<unsigned*>(dst + k) = pack_fp8x4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    }
}
SLNLP_ZKERNEL(quant_rows_fp8_kernel, 256, quant_rows_fp8_body)

int quant_rows_fp8(const float* x, int64_t ld, int R, int K, unsigned char* q, int64_t ldq, float* scale, const void* row_table,
                   hipStream_t st) {
    SLNLP_CHECK_ARG(x && q && scale && R > 0, "quant_rows_fp8: bad arguments");
    SLNLP_CHECK_ARG(row_table || (K > 0 && K % 4 == 0 && ld % 4 == 0 && ldq % 4 == 0 && ldq >= K), "quant_rows_fp8: K, ld, ldq must be multiples of 4");
    return zlaunch(quant_rows_fp8_kernel, dim3(ceil_div(R, 4)), 256, 0, st, "quant_rows_fp8", x, (long)ld, R, K, q, (long)ldq, scale,
                   (const QuantRow*)row_table);
}

}  // namespace slnlp

extern "C" int slnlp_quant_rows_fp8(const float* x, int64_t ld, int R, int K, uint8_t* q, int64_t ldq, float* scale, void* stream) {
    return slnlp::quant_rows_fp8(x, ld, R, K, q, ldq, scale, nullptr, (hipStream_t)stream);
}

extern "C" int slnlp_set_plane_tile(int tile) {
    if (tile != 0 && slnlp::geo_of_knob(tile) < 0) {
        slnlp::set_error("set_plane_tile: %d (0 = automatic, 64, 128, 12832 = 128 x 128 with 32-k stages, 256 = 256 x 256 with 32-k stages)", tile);
        return SLNLP_ERR_INVALID_ARG;
    }
    slnlp::g_plane_geo.store(slnlp::geo_of_knob(tile), std::memory_order_relaxed);
    return 0;
}

extern "C" int slnlp_set_fp8_tile(int tile) {
    if (tile != 0 && slnlp::q8_geo_of_knob(tile) < 0) {
        slnlp::set_error("set_fp8_tile: %d (0 = automatic, 64 or 128)", tile);
        return SLNLP_ERR_INVALID_ARG;
    }
    slnlp::g_q8_geo.store(slnlp::q8_geo_of_knob(tile), std::memory_order_relaxed);
    return 0;
}

extern "C" int64_t slnlp_gemm_group_scratch_bytes(const slnlp_gemm_args* jobs, const int32_t* split_k, int njobs) {
    if (!jobs || njobs < 1) return 0;
    return (int64_t)slnlp::gemm_group_scratch_bytes(jobs, split_k, njobs);
}
extern "C" int slnlp_gemm_group(const slnlp_gemm_args* jobs, const int32_t* split_k, int njobs, void* scratch,
                                int64_t scratch_bytes, void* stream) {
    if (jobs && njobs >= 1 && !jobs[0].A_hi && !jobs[0].B_hi) return slnlp::gemm_group(jobs, njobs, (hipStream_t)stream);
    return slnlp::gemm_planes_group(jobs, split_k, njobs, scratch, (size_t)(scratch_bytes < 0 ? 0 : scratch_bytes),
                                    (hipStream_t)stream);
}

extern "C" int slnlp_gemm_wd(const slnlp_gemm_args* wgrad, const slnlp_gemm_args* dgrad, void* scratch, int64_t scratch_bytes, void* stream) {
    if (!wgrad || !dgrad) {
        slnlp::set_error("gemm_wd: null args");
        return SLNLP_ERR_INVALID_ARG;
    }
    return slnlp::gemm_planes_wd(*wgrad, *dgrad, scratch, (size_t)(scratch_bytes < 0 ? 0 : scratch_bytes), (hipStream_t)stream);
}
extern "C" int slnlp_gemm_wd_plan(const slnlp_gemm_args* wgrad, const slnlp_gemm_args* dgrad, int32_t* split, int32_t* separate, int32_t* geo_wgrad,
                                  int32_t* geo_dgrad) {
    if (!wgrad || !dgrad) {
        slnlp::set_error("gemm_wd_plan: null args");
        return SLNLP_ERR_INVALID_ARG;
    }
    const slnlp::WdPlan p = slnlp::gemm_planes_wd_plan(*wgrad, *dgrad);
    if (split) *split = p.split;
    if (separate) *separate = p.separate;
    const slnlp_gemm_args jobs[2] = {*wgrad, *dgrad};
    const int sk[2] = {p.split, 1}, one = 1;
    const int g = p.separate ? -1 : slnlp::plane_geo_for(jobs, sk, 2);
    if (geo_wgrad) *geo_wgrad = p.separate ? slnlp::plane_geo_for(wgrad, &p.split, 1) : g;
    if (geo_dgrad) *geo_dgrad = p.separate ? slnlp::plane_geo_for(dgrad, &one, 1) : g;
    return 0;
}

extern "C" int slnlp_split_planes(const float* x, int64_t ld, int R, int C, uint16_t* hi, uint16_t* lo, int64_t ldp,
                                  void* stream) {
    return slnlp::split_planes(x, ld, R, C, hi, lo, ldp, (hipStream_t)stream);
}

#if SLNLP_PROBE_FENCES == 128
// probe build only: copy the recorded workgroup timelines to the host and reset the recorder; returns the number recorded
extern "C" int slnlp_probe_ts(unsigned long long* dst, int max_entries) {
    unsigned n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(slnlp::g_ts_n), sizeof(n)) != hipSuccess) return -1;
    if (n > (unsigned)slnlp::TS_MAX) n = slnlp::TS_MAX;
    if ((int)n > max_entries) n = max_entries;
    if (n && hipMemcpyFromSymbol(dst, HIP_SYMBOL(slnlp::g_ts), (size_t)n * slnlp::TS_W * sizeof(unsigned long long)) != hipSuccess) return -1;
    const unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(slnlp::g_ts_n), &zero, sizeof(zero)) != hipSuccess) return -1;
    return (int)n;
}
#endif
