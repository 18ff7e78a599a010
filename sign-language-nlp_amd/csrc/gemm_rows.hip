// gemm_rows.hip -- the decoder's B-row products on PRE-SPLIT operands, register-direct.
//
//   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) ),  A = Ahi + Alo, B = Bhi + Blo  (bf16 planes, both k-major)
//
// The reference decodes ONE target position (transformer.py:82-87: tgt of length 1), so every Linear of its decoder is a
// [batch rows x E] x [E x E]^T product -- 50 rows at the headline config -- on a dependent chain of ~150 launches per step:
// 2 % of the step's FLOPs, 29 % of its time (r04).  Such a launch lasts as long as ONE workgroup does, and gemm.hip's workgroup
// spends its 5.8 us converting fp32 tiles to bf16 hi / lo on their way into LDS: two barriers and a convert / ds_write pass per
// 64-k step, four steps deep even with the K sum on two thread groups (profiles/r04_gemm_timeline.txt).  Here nothing is
// converted and nothing is staged:
//  * operands arrive as bf16 hi / lo planes written by their producers (the optimizer kernels for the weights; LayerNorm, the
//    embedding and this kernel's own epilogue for the activations), zero-padded to 64 rows / 64 k;
//  * a k-major plane row IS the MFMA fragment layout (lane (row l & 15, k-octet l >> 4) reads 16 contiguous bytes), so every
//    wave loads its own fragments straight into registers and issues its MFMAs as they land: no LDS staging, no barrier, no
//    VALU in the K loop; one memory round trip per launch;
//  * ONE MFMA tile (16 rows x 16 columns) per workgroup, 512 threads = 8 waves, wave w = the 64-k tiles w, w + 8, ...: every
//    byte of the workgroup's panels is loaded by one wave, each tile's partial product is computed from zero, and the partials
//    are added in tile order through LDS -- C = ((P_0 + P_1) + P_2) + ... -- this kernel's definition of the K sum
//    (gemm_rows_body: why the tile is this small);
//  * the epilogue (bias -> activation -> gate -> dropout -> residual; fp32 and / or planes out) runs in the accumulator layout;
//    its bias / gate / residual values and the dropout key are requested behind the operand loads, before the first MFMA.
// A SLNLP_ZKERNEL: K fits in lockstep share one launch through grid.z (launch.hpp).
#include <algorithm>
#include <atomic>

#include "common.hpp"
#include "gemm_jobs.hpp"
#include "launch.hpp"

namespace slnlp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct RowsParams {
    slnlp_gemm_args a;
    unsigned drop_thr;
    float drop_scale;
};
__device__ __forceinline__ RowsParams as_global(RowsParams p) {
    launder(p.a);
    return p;
}

#if SLNLP_PROBE_FENCES == 256
// timeline probe build (tools/probes/probe_gemm_timeline.py): wave 0 of every workgroup records 100 MHz timestamps of its phases
constexpr int RTS_MAX = 1 << 14, RTS_W = 6;   // words: entry, first tile's operands landed, partials stored, K sum done, end, {grid, block}
__device__ unsigned long long g_rts[RTS_MAX][RTS_W];
__device__ unsigned g_rts_n;
#define RTS_MARK(slot) do { if (threadIdx.x == 0) rts[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RTS_MARK(slot) do { } while (0)
#endif

constexpr int RT_THREADS = 512, RT_WAVES = 8;
constexpr int RT_MAX_KTILES = 16;             // K <= 1024

// Work split.  Such a launch is bound by what ONE compute unit can load (about 33 GB/s of 16-byte fragment loads): a workgroup
// of a 64 x 16 tiling pulls the whole activation panel plus its weight columns -- 160 KB at K = 512, in 4.8 us whatever the kernel
// does with them (the fp32-operand kernel and a register-direct plane kernel with that tiling both measured 5.8 us per workgroup:
// profiles/r05_gemm_timeline.txt).  So the tile of a solo fit's launch is ONE MFMA tile, 16 rows x 16 columns, on 4 x N / 16
// workgroups (128 for a [50 x 512] product instead of 32): 64 KB and 3.1 us per workgroup.  K fits in lockstep fill the chip anyway
// and pay for bytes instead: their merged launch takes 64 x 16 or 64 x 32 tiles (5 / 3 MB of panel reads per product instead of 8).
// Geometry (MT x NT MFMA tiles per workgroup) is picked per launch (rows_geo) and NEVER changes a result:
// WAVE w takes the 64-k tiles w, w + 8, ... and computes each tile's partial product P_t from zero -- per 32-k step MT A and NT B
// fragments, hi and lo, every byte of the workgroup's panels loaded by exactly one wave.  The partials meet in LDS and are added
// in TILE ORDER -- C = ((P_0 + P_1) + P_2) + ... -- which is the DEFINITION of this kernel's K sum: one value per product however
// many waves, workgroups or fits share the launch, whatever the tile.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef __attribute__((address_space(1))) const void* glb_vp;

// ---- m-major operands (the data gradient's W [k][n], both operands of a weight gradient): a wave stages ITS 64-k x 16-row
// image [64 k][16 rows] of a plane (2 KiB, 32-byte rows) by LDS-DMA and reads MFMA fragments with the transposing LDS read.  The
// image is private to the wave, so the only ordering needed is the wave's own s_waitcnt vmcnt (no barrier).
constexpr int RT_IMG = 64 * 16;               // elements of one image
__device__ __forceinline__ void rows_dma_image(const unsigned short* __restrict__ plane, long ld, int k0, int r0, unsigned short* img, int lane) {
    // two instructions of 64 lanes x 16 B: lane -> k-row 32 q + lane / 2, half lane % 2; the destination is wave-uniform base + lane x 16
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const unsigned short* src = plane + (long)(k0 + 32 * q + (lane >> 1)) * ld + r0 + 8 * (lane & 1);
        __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(img + q * 512), 16, 0, 0);
    }
}
// fragment of rows 0 .. 15 of the image, k in [32 kk, 32 kk + 32): lane (i = l & 15) addresses k-row kb + i / 4, rows 4 (i % 4) .. + 3
// of a 4 (k) x 16 (row) block and receives the 4 k-values of row i (ds_read_b64_tr_b16)
__device__ __forceinline__ bf16x8 rows_tr_frag(const unsigned short* img, int kk, int lane) {
    const int i = lane & 15, kb = kk * 32 + ((lane >> 4) << 3) + (i >> 2);
    const unsigned short* p0 = img + kb * 16 + ((i & 3) << 2);
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0 + 4 * 16));
    const s16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// ---- the WEIGHT operand of a product arrives as fp32 (row stride ldb) and is split in registers: hi = bf16(x) rounded to nearest,
// lo = bf16(x - hi) -- exactly split_bf16, i.e. the bits a plane of the weight would hold -- so the optimizer kernels need not keep
// bf16 planes of the decoder's weights (20 instead of 24 B per parameter and step for 58 % of the cfg2 parameters: a 15-fit lockstep
// step's sgd_kernel 1.87 -> 1.57 ms).  Same bytes per fragment as hi + lo planes (8 x 4 B).
template <int NSPLIT>
__device__ __forceinline__ void rows_split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)x[e];
        hi[e] = h;
        if (NSPLIT == 3) lo[e] = (__bf16)(x[e] - (float)h);
    }
}
// k-major weight rows ([n][k], the forward products): the lane's 8 consecutive k of its row, two 16-byte loads
__device__ __forceinline__ void rows_load8(const float* __restrict__ p, float (&x)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}
// m-major weight ([k][n], the data gradients): a wave-private fp32 image [64 k][16 n] (4 KiB) filled by four LDS-DMA instructions
// (lane -> k-row 16 q + lane / 4, 16-byte piece lane % 4; a piece never starts past column N - 4) ...
constexpr int RT_FIMG = 64 * 16;              // floats of one fp32 image
__device__ __forceinline__ void rows_dma_image_f32(const float* __restrict__ W, long ld, int k0, int c0, int N, float* img, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = min(c0 + 4 * (lane & 3), N - 4);
        const float* src = W + (long)(k0 + 16 * q + (lane >> 2)) * ld + c;
        __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(img + q * 256), 16, 0, 0);
    }
}
// ... and read back transposed: lane (column l & 15, k-octet l >> 4) takes its 8 k-values with 8 ds_read_b32
__device__ __forceinline__ void rows_img_col8(const float* img, int kk, int lane, float (&x)[8]) {
    const float* p = img + (kk * 32 + ((lane >> 4) << 3)) * 16 + (lane & 15);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = p[e * 16];
}

// One product unit: output tile (bx, by) of 16 MT rows x 16 NT columns.  BK: the B operand is k-major (forward products: fragments
// straight from the plane) or m-major (data gradients: the wave's image through LDS, above).  smem: [partials][wave images].
template <int NSPLIT, int MT, int NT, bool BK>
__device__ __forceinline__ void rows_tile(const RowsParams& p, int bx, int by, float* part) {
    static_assert(MT * NT <= RT_WAVES, "one epilogue tile per wave");
    const slnlp_gemm_args& g = p.a;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if SLNLP_PROBE_FENCES == 256
    unsigned long long rts[RTS_W] = {0, 0, 0, 0, 0, 0};
    struct RtsFlush {
        unsigned long long* t;
        __device__ ~RtsFlush() {
            if (threadIdx.x == 0) {
                t[4] = __builtin_amdgcn_s_memrealtime();
                t[5] = ((unsigned long long)(gridDim.x * gridDim.y) << 32) | (unsigned)(blockIdx.y * gridDim.x + blockIdx.x);
                const unsigned i = atomicAdd(&g_rts_n, 1u) & (unsigned)(RTS_MAX - 1);
                for (int k = 0; k < RTS_W; ++k) g_rts[i][k] = t[k];
            }
        }
    } rts_flush{rts};
    RTS_MARK(0);
#endif
    const int bn0 = bx * 16 * NT, bm0 = by * 16 * MT;
    const int M = g.M, N = g.N;
    const int ktiles = (g.K + 63) >> 6;
    // this lane's fragment rows and its k-octet inside a 32-k step.  The planes are zero-padded to multiples of 64 rows and weight
    // planes are followed by more of the arena; a tile that starts past the last row block re-reads the last valid tile instead
    // (its products belong to rows >= M / columns >= N, which nothing stores)
    const int am_last = ((M + 15) / 16 - 1) * 16, bn_last = ((N + 15) / 16 - 1) * 16;
    const long koct = 8 * (lane >> 4);
    const unsigned short *ah[MT], *al[MT];
    const float* bw[NT];                                   // k-major weight: this lane's row of column tile j, at its k-octet
    int bcol[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const long off = (long)(min(bm0 + 16 * i, am_last) + (lane & 15)) * g.lda_p + koct;
        ah[i] = g.A_hi + off; al[i] = g.A_lo + off;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        bcol[j] = min(bn0 + 16 * j, bn_last);
        bw[j] = g.B + (long)min(bcol[j] + (lane & 15), N - 1) * g.ldb + koct;      // (a row past N is a repeat of row N - 1: never stored)
    }
    // (m-major B: this wave's fp32 images, behind the partial tiles: [NT column tiles][64 k][16 n])
    float* bimg = part + (size_t)ktiles * MT * NT * 256 + wave * NT * RT_FIMG;

    // what the epilogue will want: wave w < MT * NT runs it for MFMA tile (w / NT, w % NT); the lane holds rows gm0 .. gm0 + 3 of column gn
    const int ei = wave / NT, ej = wave % NT;
    const int gm0 = bm0 + 16 * ei + ((lane >> 4) << 2), gn = bn0 + 16 * ej + (lane & 15);
    const bool live = wave < MT * NT && gn < N && gm0 < M;
    float bias = 0.f, gt[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
    DropKey dkey = {};
    auto epilogue_loads = [&]() {
        if (!live) return;
        if (g.bias) bias = g.bias[gn];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gm = gm0 + r < M ? gm0 + r : gm0;
            if (g.gate) gt[r] = g.gate[(long)gm * g.ldg + gn];
            if (g.resid) rs[r] = g.resid[(long)gm * g.ldr + gn];
        }
        if (g.drop_p > 0.f) dkey = dropout_key(g.rng, g.drop_site);
    };

    for (int t = wave; t < ktiles; t += RT_WAVES) {
        bf16x8 fa[2][MT], la[2][MT], fb[2][NT], lb[2][NT];
        if (!BK) {
            if (t != wave) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the images' last fragment reads: LDS operations of a wave retire in order)
#pragma unroll
            for (int j = 0; j < NT; ++j) rows_dma_image_f32(g.B, g.ldb, t * 64, bcol[j], N, bimg + j * RT_FIMG, lane);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k = t * 64 + kk * 32;
            if (BK) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float x[8];
                    rows_load8(bw[j] + k, x);
                    rows_split8<NSPLIT>(x, fb[kk][j], lb[kk][j]);
                }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                fa[kk][i] = *reinterpret_cast<const bf16x8*>(ah[i] + k);
                if (NSPLIT == 3) la[kk][i] = *reinterpret_cast<const bf16x8*>(al[i] + k);
            }
        }
        // (requested behind the first tile's operands, in front of its MFMAs: vector-memory operations retire in order, so a slow
        //  residual line in FRONT of the operands would hold the first MFMA's counted wait -- 0.4 us on gemm.hip's 16-wide tile)
        if (t == wave) epilogue_loads();
        if (!BK) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's images have landed
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float x[8];
                    rows_img_col8(bimg + j * RT_FIMG, kk, lane, x);
                    rows_split8<NSPLIT>(x, fb[kk][j], lb[kk][j]);
                }
        }
#if SLNLP_PROBE_FENCES == 256
        if (t == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RTS_MARK(1); }
#endif
        f32x4 acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (NSPLIT == 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(la[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][i], lb[kk][j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk][i], fb[kk][j], acc[i][j], 0, 0, 0);
                }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4*>(part + (((t * MT + i) * NT + j) * 64 + lane) * 4) = acc[i][j];
    }
    if (wave >= ktiles) epilogue_loads();          // (a wave without a K tile still owes its epilogue requests)
    RTS_MARK(2);
    __syncthreads();
    if (wave >= MT * NT) return;
    // the K sum of this wave's MFMA tile, in tile order
    f32x4 acc = *reinterpret_cast<const f32x4*>(part + (wave * 64 + lane) * 4);
    for (int t = 1; t < ktiles; ++t) acc += *reinterpret_cast<const f32x4*>(part + ((t * MT * NT + wave) * 64 + lane) * 4);
    RTS_MARK(3);
    if (!live) return;

    // ---- epilogue in the accumulator layout
    unsigned lot[4] = {0u, 0u, 0u, 0u};
    if (g.drop_p > 0.f) {
        if (g.drop_head_dim == 0) {
            const uint4 bits = dropout_bits8(dkey, (unsigned)gm0 >> 2, drop_cc((unsigned)gn));
            const int h = drop_half((unsigned)gn);
#pragma unroll
            for (int r = 0; r < 4; ++r) lot[r] = pick_lot(bits, h, r);
        } else {                                  // one keep / drop decision per (row, head), see slnlp.h
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned rh = (unsigned)(gm0 + r) * (unsigned)(N / g.drop_head_dim) + (unsigned)(gn / g.drop_head_dim);
                lot[r] = pick_lot(dropout_bits8(dkey, rh >> 2, 0u), 0, (int)(rh & 3u));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int gm = gm0 + r;
        if (gm >= M) break;
        float v = acc[r] + bias;
        if (g.relu == 1) v = fmaxf(v, 0.f);
        else if (g.relu == 2) v = tanhf(v);
        if (g.gate) v = g.gate_mode == 1 ? v * (1.f - gt[r] * gt[r]) : (gt[r] > 0.f ? v * g.gate_scale : 0.f);
        if (g.drop_p > 0.f) v = (lot[r] >= p.drop_thr) ? v * p.drop_scale : 0.f;
        if (g.resid) v += rs[r];
        if (g.C) g.C[(long)gm * g.ldc + gn] = v;
        if (g.C_hi) {
            unsigned short h, l;
            split_bf16(v, h, l);
            g.C_hi[(long)gm * g.ldc_p + gn] = h;
            if (g.C_lo) g.C_lo[(long)gm * g.ldc_p + gn] = l;
        }
    }
}

// One weight-gradient unit: dW[16 WM m x 128 n] = dY^T x over the batch rows -- A = dY planes [k = row][m], B = x planes [k = row][n],
// both m-major; WAVE w computes the WM 16 x 16 tiles of columns n0 + 16 w.  The WM dY images are staged ONCE per workgroup (the waves
// share the DMA instructions; one barrier), each wave stages its own x image.  K = the batch rows (one 64-k tile at the headline
// batch of 50; more tiles: partials added in tile order, as everywhere in this file).  rowsum_a (the bias gradient: sum over the rows
// of dY[., m]) is the same dY fragments times an all-ones fragment, by the first column unit's wave 0.  WM = 1 for a solo fit's launch (128 units of
// 36 KB for a 512 x 512 gradient), 4 for merged lockstep launches (a quarter of the x-image reads).
template <int NSPLIT, int WM>
__device__ __forceinline__ void rows_wgrad_unit(const slnlp_gemm_args& g, int unit, unsigned short* smem) {
    constexpr int NP = NSPLIT == 3 ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = g.M, N = g.N, ktiles = (g.K + 63) >> 6;
    const int units_n = (N + 127) / 128, um = unit / units_n, un = unit - um * units_n;
    const int m0 = um * 16 * WM, n0 = un * 128 + wave * 16, m_last = ((M + 15) / 16 - 1) * 16;
    const bool active = n0 < N;                          // (a wave past the last column tile only keeps the barriers company)
    unsigned short* aimg = smem;                          // [WM][hi, lo][64 k][16 m]: the workgroup's
    unsigned short* bimg = smem + WM * 2 * RT_IMG + wave * 2 * RT_IMG;   // [hi, lo][64 k][16 n]: this wave's
    f32x4 total[WM], rsum[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) { total[i] = f32x4{0.f, 0.f, 0.f, 0.f}; rsum[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // the bias gradient is one more product of the same dY fragments: dY^T 1 (an all-ones B fragment, exact in bf16), hi and lo -- two
    // MFMAs per fragment instead of a lane-serial walk over the image (16 lanes x 64 k x WM dependent LDS reads: 17 us of a 20 us unit
    // at batch 256, r05 timeline).  Every column of the result tile holds the row sums; column 0's lanes store them.
    const bool do_rs = g.rowsum_a != nullptr && un == 0 && wave == 0;      // (wave-uniform)
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#if SLNLP_PROBE_FENCES == 256
    // (probe build: entry, first tile's images landed, first tile's MFMAs issued, K loop done, end; bit 31 of the block word = a weight-gradient unit,
    //  bit 30 = one that also sums the rows of dY)
    unsigned long long rts[RTS_W] = {0, 0, 0, 0, 0, 0};
    struct RtsFlushW {
        unsigned long long* t;
        unsigned tag;
        __device__ ~RtsFlushW() {
            if (threadIdx.x == 0) {
                t[4] = __builtin_amdgcn_s_memrealtime();
                t[5] = ((unsigned long long)(gridDim.x * gridDim.y) << 32) | tag | (unsigned)(blockIdx.y * gridDim.x + blockIdx.x);
                const unsigned i = atomicAdd(&g_rts_n, 1u) & (unsigned)(RTS_MAX - 1);
                for (int k = 0; k < RTS_W; ++k) g_rts[i][k] = t[k];
            }
        }
    } rts_flush{rts, 0x80000000u | (g.rowsum_a != nullptr && un == 0 ? 0x40000000u : 0u)};
    RTS_MARK(0);
#endif
    for (int t = 0; t < ktiles; ++t) {
        if (t) {                                          // every wave is done with the previous tile's images
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
        for (int im = wave; im < WM * NP; im += RT_WAVES) {
            const int i = im / NP, pl = im - i * NP;
            rows_dma_image(pl ? g.A_lo : g.A_hi, g.lda_p, t * 64, min(m0 + 16 * i, m_last), aimg + (2 * i + pl) * RT_IMG, lane);
        }
        if (active) {
            rows_dma_image(g.B_hi, g.ldb_p, t * 64, n0, bimg, lane);
            if (NSPLIT == 3) rows_dma_image(g.B_lo, g.ldb_p, t * 64, n0, bimg + RT_IMG, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#if SLNLP_PROBE_FENCES == 256
        if (t == 0) RTS_MARK(1);
#endif
        if (active) {
            bf16x8 fb[2], lb[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                fb[kk] = rows_tr_frag(bimg, kk, lane);
                if (NSPLIT == 3) lb[kk] = rows_tr_frag(bimg + RT_IMG, kk, lane);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, rs = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 fa = rows_tr_frag(aimg + (2 * i) * RT_IMG, kk, lane);
                    if (NSPLIT == 3) {
                        const bf16x8 la = rows_tr_frag(aimg + (2 * i + 1) * RT_IMG, kk, lane);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(la, fb[kk], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, lb[kk], acc, 0, 0, 0);
                        if (do_rs) rs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(la, ones, rs, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[kk], acc, 0, 0, 0);
                    if (do_rs) rs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, ones, rs, 0, 0, 0);
                }
                total[i] = t == 0 ? acc : total[i] + acc;
                if (do_rs) rsum[i] = t == 0 ? rs : rsum[i] + rs;
            }
        }
#if SLNLP_PROBE_FENCES == 256
        if (t == 0) RTS_MARK(2);
#endif
    }
    RTS_MARK(3);
    if (!active) return;
    const int gn = n0 + (lane & 15);
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int gm0 = m0 + 16 * i + ((lane >> 4) << 2);
        if (gn < N) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (gm0 + r < M) g.C[(long)(gm0 + r) * g.ldc + gn] = total[i][r];
        }
        if (do_rs && (lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (gm0 + r < M) g.rowsum_a[gm0 + r] = rsum[i][r];
        }
    }
}

template <int NSPLIT, int MT, int NT>
__device__ __forceinline__ void gemm_rows_body(RowsParams p) {
    extern __shared__ __attribute__((aligned(16))) float part[];      // [ktiles][MT x NT tiles][64 lanes][4]
    rows_tile<NSPLIT, MT, NT, true>(p, blockIdx.x, blockIdx.y, part);
}

// The backward pair of one dY in ONE launch: blocks [0, nd) are the data gradient's tiles (dX = dY W, W m-major), the rest the weight
// gradient's units (dW = dY^T x, db = column sums of dY).  Both read the same dY planes.
struct RowsBwdParams {
    RowsParams d;
    slnlp_gemm_args w;
};
__device__ __forceinline__ RowsBwdParams as_global(RowsBwdParams p) {
    launder(p.d.a);
    launder(p.w);
    return p;
}
template <int NSPLIT, int MT, int NT>
__device__ __forceinline__ void gemm_rows_bwd_body(RowsBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float part[];
    const int gx = (p.d.a.N + 16 * NT - 1) / (16 * NT), gy = (p.d.a.M + 16 * MT - 1) / (16 * MT), nd = gx * gy;
    const int id = blockIdx.x;
    if (id < nd) rows_tile<NSPLIT, MT, NT, false>(p.d, id % gx, id / gx, part);
    else rows_wgrad_unit<NSPLIT, MT>(p.w, id - nd, reinterpret_cast<unsigned short*>(part));      // (16 MT gradient rows per unit)
}

// geometries: MFMA tiles per workgroup
struct RowsGeo { int mt, nt; };
constexpr int RT_NGEO = 3;
constexpr RowsGeo RT_GEO[RT_NGEO] = {{1, 1}, {4, 1}, {4, 2}};
#define SLNLP_ROWS_KERNEL(NS, G)                                                                                          \
    __device__ __forceinline__ void gemm_rows_body_##NS##_##G(RowsParams p) { gemm_rows_body<NS, RT_GEO[G].mt, RT_GEO[G].nt>(p); } \
    SLNLP_ZKERNEL(gemm_rows_kernel_##NS##_##G, RT_THREADS, gemm_rows_body_##NS##_##G)                                     \
    __device__ __forceinline__ void gemm_rows_bwd_body_##NS##_##G(RowsBwdParams p) { gemm_rows_bwd_body<NS, RT_GEO[G].mt, RT_GEO[G].nt>(p); } \
    SLNLP_ZKERNEL(gemm_rows_bwd_kernel_##NS##_##G, RT_THREADS, gemm_rows_bwd_body_##NS##_##G)
SLNLP_ROWS_KERNEL(3, 0)
SLNLP_ROWS_KERNEL(3, 1)
SLNLP_ROWS_KERNEL(3, 2)
SLNLP_ROWS_KERNEL(1, 0)
SLNLP_ROWS_KERNEL(1, 1)
SLNLP_ROWS_KERNEL(1, 2)

using RowsKernel = void (*)(Pack<RowsParams>, const Pack<RowsParams>*);
using RowsBwdKernel = void (*)(Pack<RowsBwdParams>, const Pack<RowsBwdParams>*);
static RowsKernel rows_kernel(int precision, int geo) {
    if (precision == 3) return geo == 0 ? gemm_rows_kernel_3_0 : geo == 1 ? gemm_rows_kernel_3_1 : gemm_rows_kernel_3_2;
    return geo == 0 ? gemm_rows_kernel_1_0 : geo == 1 ? gemm_rows_kernel_1_1 : gemm_rows_kernel_1_2;
}
static RowsBwdKernel rows_bwd_kernel(int precision, int geo) {
    if (precision == 3) return geo == 0 ? gemm_rows_bwd_kernel_3_0 : geo == 1 ? gemm_rows_bwd_kernel_3_1 : gemm_rows_bwd_kernel_3_2;
    return geo == 0 ? gemm_rows_bwd_kernel_1_0 : geo == 1 ? gemm_rows_bwd_kernel_1_1 : gemm_rows_bwd_kernel_1_2;
}
static size_t rows_lds(int geo, int K) { return (size_t)ceil_div(K, 64) * RT_GEO[geo].mt * RT_GEO[geo].nt * 64 * 4 * sizeof(float); }
// backward launches: the data gradient's partial tiles + its waves' W images, or the weight gradient's four images per wave
static size_t rows_bwd_lds(int geo, int K) {
    const size_t d = rows_lds(geo, K) + (size_t)RT_WAVES * RT_GEO[geo].nt * RT_FIMG * sizeof(float);
    const size_t w = (size_t)(RT_GEO[geo].mt + RT_WAVES) * 2 * RT_IMG * sizeof(unsigned short);      // the workgroup's dY images + the waves' x images
    return d > w ? d : w;
}
static dim3 rows_grid(int geo, int M, int N) { return dim3(ceil_div(N, 16 * RT_GEO[geo].nt), ceil_div(M, 16 * RT_GEO[geo].mt)); }
static int rows_wgrad_units(const slnlp_gemm_args& w, int geo) { return ceil_div(w.M, 16 * RT_GEO[geo].mt) * ceil_div(w.N, 128); }

// -1 = automatic; 0 .. RT_NGEO-1 forced (slnlp_set_rows_tile: tests, tuning)
static std::atomic<int> g_rows_geo{[] { const char* e = getenv("SLNLP_ROWS_TILE"); const int v = e ? atoi(e) : -1; return v >= 0 && v < RT_NGEO ? v : -1; }()};
// Which tile a launch of `fits` products [M x N x K] takes: rounds of workgroups over the 256 compute units x the panel bytes one
// workgroup loads (a compute unit's load rate is the limit, see gemm_rows_body) -- the small tile while the launch fits the chip
// about once (one fit: 128 workgroups of 64 KB), the wide ones when K fits in lockstep fill it many times over.
constexpr size_t RT_LDS_MAX = 160 * 1024;    // per workgroup (MI355X: 160 KiB per compute unit)
static int rows_geo(int M, int N, int K, int fits, bool bwd) {
    const int forced = g_rows_geo.load(std::memory_order_relaxed);
    auto fits_lds = [&](int geo) { return (bwd ? rows_bwd_lds(geo, K) : rows_lds(geo, K)) <= RT_LDS_MAX; };
    if (forced >= 0 && fits_lds(forced)) return forced;
    int best = 0;
    long best_cost = -1;
    for (int geo = 0; geo < RT_NGEO; ++geo) {
        if (!fits_lds(geo)) continue;
        const dim3 gr = rows_grid(geo, M, N);
        // (a backward launch also carries the weight gradient's units: dW [K x N] in pieces of 16 mt x 128)
        const long wunits = bwd ? (long)ceil_div(K, 16 * RT_GEO[geo].mt) * ceil_div(N, 128) : 0;
        const long units = ((long)gr.x * gr.y + wunits) * fits, rounds = (units + 255) / 256;
        const long cost = rounds * (16L * (RT_GEO[geo].mt + RT_GEO[geo].nt) * K * 4 + 16384);     // (+ a fixed cost per round)
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = geo; }
    }
    return best;
}

static int rows_init() {
    static DeviceOnce once;
    return once.run([]() -> int {
        bool ok = true;
        for (int prec = 1; prec <= 3; prec += 2)
            for (int geo = 0; geo < RT_NGEO; ++geo)
                ok = ok && hipFuncSetAttribute((const void*)rows_kernel(prec, geo), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)std::min(rows_lds(geo, 64 * RT_MAX_KTILES), RT_LDS_MAX)) == hipSuccess &&
                     hipFuncSetAttribute((const void*)rows_bwd_kernel(prec, geo), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)std::min(rows_bwd_lds(geo, 64 * RT_MAX_KTILES), RT_LDS_MAX)) == hipSuccess;
        if (!ok) {
            set_error("gemm_rows: cannot raise dynamic LDS limit: %s", hipGetErrorString(hipGetLastError()));
            return SLNLP_ERR_LAUNCH;
        }
        return 0;
    });
}

// A merged (lockstep) launch of K fits' recorded gemm_rows call sites picks its tile again for K products (lockstep.hip): returns the
// kernel to launch and its grid / LDS, or nullptr when `fn` is not a gemm_rows kernel.  Results do not depend on the tile.
const void* gemm_rows_for_fits(const void* fn, const void* recorded_args, int fits, dim3* grid, size_t* lds) {
    int prec = 0;
    bool bwd = false;
    for (int geo = 0; geo < RT_NGEO; ++geo) {
        if (fn == (const void*)rows_kernel(3, geo)) prec = 3;
        if (fn == (const void*)rows_kernel(1, geo)) prec = 1;
        if (fn == (const void*)rows_bwd_kernel(3, geo)) { prec = 3; bwd = true; }
        if (fn == (const void*)rows_bwd_kernel(1, geo)) { prec = 1; bwd = true; }
    }
    if (!prec) return nullptr;
    if (bwd) {
        const RowsBwdParams& P = *reinterpret_cast<const RowsBwdParams*>(recorded_args);
        const int geo = rows_geo(P.d.a.M, P.d.a.N, P.d.a.K, fits, true);
        const dim3 gr = rows_grid(geo, P.d.a.M, P.d.a.N);
        *grid = dim3(gr.x * gr.y + rows_wgrad_units(P.w, geo));
        *lds = rows_bwd_lds(geo, P.d.a.K);
        return (const void*)rows_bwd_kernel(prec, geo);
    }
    const slnlp_gemm_args& a = reinterpret_cast<const RowsParams*>(recorded_args)->a;
    const int geo = rows_geo(a.M, a.N, a.K, fits, false);
    *grid = rows_grid(geo, a.M, a.N);
    *lds = rows_lds(geo, a.K);
    return (const void*)rows_kernel(prec, geo);
}

static int check_rows_job(const slnlp_gemm_args& a, bool b_kmajor, const char* who) {
    SLNLP_CHECK_ARG(a.A_hi && a.B, "%s: A as bf16 planes (A_hi / A_lo / lda_p) and the weight B as fp32 (B / ldb) required", who);
    SLNLP_CHECK_ARG(a.a_kmajor && (a.b_kmajor != 0) == b_kmajor, "%s: operand layouts (A k-major; B k-major for a forward product, m-major for a data gradient)", who);
    SLNLP_CHECK_ARG(a.C || a.C_hi, "%s: no output", who);
    SLNLP_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "%s: bad shape M=%d N=%d K=%d", who, a.M, a.N, a.K);
    SLNLP_CHECK_ARG(a.precision == 1 || (a.precision == 3 && a.A_lo), "%s: precision 1, or 3 with A's lo plane", who);
    SLNLP_CHECK_ARG(a.lda_p % 64 == 0 && a.lda_p >= a.K, "%s: A's plane row stride must be a multiple of 64 and cover K (zero-padded)", who);
    SLNLP_CHECK_ARG(a.K % 64 == 0 && a.ldb % 4 == 0 && a.ldb >= (b_kmajor ? a.K : a.N) && (b_kmajor || a.N % 4 == 0),
                    "%s: K must be a multiple of 64 (the fp32 weight has no padding), ldb a multiple of 4 that covers a row", who);
    SLNLP_CHECK_ARG((((uintptr_t)a.A_hi | (uintptr_t)a.A_lo | (uintptr_t)a.B) & 15) == 0, "%s: operands must be 16-byte aligned", who);
    SLNLP_CHECK_ARG(!a.C || a.ldc >= a.N, "%s: ldc < N", who);
    SLNLP_CHECK_ARG(!a.C_hi || a.ldc_p >= a.N, "%s: ldc_p < N", who);
    SLNLP_CHECK_ARG(a.drop_p >= 0.f && a.drop_p < 1.f && (a.drop_p == 0.f || a.rng), "%s: bad dropout args", who);
    SLNLP_CHECK_ARG(!a.gate || a.ldg >= a.N, "%s: ldg too small", who);
    SLNLP_CHECK_ARG(!a.resid || a.ldr >= a.N, "%s: ldr too small", who);
    SLNLP_CHECK_ARG(!a.rowsum_a && a.batch <= 1, "%s: no row sums, no batched jobs", who);
    SLNLP_CHECK_ARG(a.drop_head_dim >= 0 && (a.drop_head_dim == 0 || a.N % a.drop_head_dim == 0), "%s: drop_head_dim %d does not divide N %d", who,
                    a.drop_head_dim, a.N);
    SLNLP_CHECK_ARG(ceil_div(a.K, 64) <= RT_MAX_KTILES, "%s: K = %d > %d (the workgroup's partial tiles live in LDS)", who, a.K, 64 * RT_MAX_KTILES);
    return 0;
}

// dX = dY W (+ epilogue) and dW = dY^T x, db = colsum(dY) in one launch (gemm_rows_bwd_body).  dgrad: A = dY planes k-major, B = W planes
// m-major; wgrad: A = dY planes, B = x planes, both m-major, K = the batch rows (planes zero beyond them), C = dW fp32, rowsum_a = db or null.
int gemm_rows_bwd(const slnlp_gemm_args& dgrad, const slnlp_gemm_args& wgrad, hipStream_t st) {
    SLNLP_TRY(check_rows_job(dgrad, false, "gemm_rows_bwd (dgrad)"));
    const slnlp_gemm_args& w = wgrad;
    SLNLP_CHECK_ARG(w.A_hi && w.B_hi && w.C && !w.a_kmajor && !w.b_kmajor, "gemm_rows_bwd (wgrad): m-major operand planes and an fp32 output required");
    SLNLP_CHECK_ARG(w.precision == dgrad.precision && (w.precision == 1 || (w.A_lo && w.B_lo)), "gemm_rows_bwd (wgrad): same precision as the dgrad, lo planes for 3");
    SLNLP_CHECK_ARG(w.M > 0 && w.N > 0 && w.K > 0 && w.lda_p % 64 == 0 && w.ldb_p % 64 == 0 && w.lda_p >= w.M && w.ldb_p >= w.N && w.ldc >= w.N,
                    "gemm_rows_bwd (wgrad): bad shape / strides");
    SLNLP_CHECK_ARG((((uintptr_t)w.A_hi | (uintptr_t)w.B_hi | (uintptr_t)w.A_lo | (uintptr_t)w.B_lo) & 15) == 0, "gemm_rows_bwd (wgrad): planes must be 16-byte aligned");
    SLNLP_CHECK_ARG(!w.bias && !w.gate && !w.resid && w.drop_p == 0.f && !w.C_hi && w.relu == 0 && w.batch <= 1, "gemm_rows_bwd (wgrad): no epilogue");
    RowsBwdParams p;
    p.d.a = dgrad;
    p.d.drop_thr = dropout_threshold(dgrad.drop_p);
    p.d.drop_scale = 1.f / (1.f - dgrad.drop_p);
    p.w = wgrad;
    SLNLP_TRY(rows_init());
    const int geo = rows_geo(dgrad.M, dgrad.N, dgrad.K, 1, true);
    const dim3 gr = rows_grid(geo, dgrad.M, dgrad.N);
    return zlaunch(rows_bwd_kernel(dgrad.precision, geo), dim3(gr.x * gr.y + rows_wgrad_units(wgrad, geo)), RT_THREADS, rows_bwd_lds(geo, dgrad.K), st,
                   "gemm_rows_bwd", p);
}

int gemm_rows(const slnlp_gemm_args& a, hipStream_t st) {
    SLNLP_TRY(check_rows_job(a, true, "gemm_rows"));
    RowsParams p;
    p.a = a;
    p.drop_thr = dropout_threshold(a.drop_p);
    p.drop_scale = 1.f / (1.f - a.drop_p);
    SLNLP_TRY(rows_init());
    const int geo = rows_geo(a.M, a.N, a.K, 1, false);
    return zlaunch(rows_kernel(a.precision, geo), rows_grid(geo, a.M, a.N), RT_THREADS, rows_lds(geo, a.K), st, "gemm_rows", p);
}

}  // namespace slnlp

#if SLNLP_PROBE_FENCES == 256
// probe build only: copy the recorded workgroup timelines to the host and reset the recorder; returns the number recorded
extern "C" int slnlp_probe_rows_ts(unsigned long long* dst, int max_entries) {
    unsigned n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(slnlp::g_rts_n), sizeof(n)) != hipSuccess) return -1;
    if (n > (unsigned)slnlp::RTS_MAX) n = slnlp::RTS_MAX;
    if ((int)n > max_entries) n = max_entries;
    if (n && hipMemcpyFromSymbol(dst, HIP_SYMBOL(slnlp::g_rts), (size_t)n * slnlp::RTS_W * sizeof(unsigned long long)) != hipSuccess) return -1;
    const unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(slnlp::g_rts_n), &zero, sizeof(zero)) != hipSuccess) return -1;
    return (int)n;
}
#endif

extern "C" int slnlp_gemm_rows_bwd(const slnlp_gemm_args* dgrad, const slnlp_gemm_args* wgrad, void* stream) {
    if (!dgrad || !wgrad) {
        slnlp::set_error("gemm_rows_bwd: null args");
        return SLNLP_ERR_INVALID_ARG;
    }
    return slnlp::gemm_rows_bwd(*dgrad, *wgrad, (hipStream_t)stream);
}

extern "C" int slnlp_set_rows_tile(int tile) {
    if (tile < -1 || tile >= slnlp::RT_NGEO) {
        slnlp::set_error("set_rows_tile: %d (-1 = automatic, 0 = 16 x 16, 1 = 64 x 16, 2 = 64 x 32 outputs per workgroup)", tile);
        return SLNLP_ERR_INVALID_ARG;
    }
    slnlp::g_rows_geo.store(tile, std::memory_order_relaxed);
    return 0;
}

extern "C" int slnlp_gemm_rows(const slnlp_gemm_args* args, void* stream) {
    if (!args) {
        slnlp::set_error("gemm_rows: null args");
        return SLNLP_ERR_INVALID_ARG;
    }
    return slnlp::gemm_rows(*args, (hipStream_t)stream);
}
