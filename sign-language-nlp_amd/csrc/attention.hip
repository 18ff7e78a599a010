// attention.hip -- attention cores for the short (S <= 64) ASL-Phono sequences.
//
// Self-attention (encoder; /root/reference/model/transformer.py:68-73,82-87 ->
// nn.MultiheadAttention slow path): one workgroup per (batch, head); the four
// contractions of a head (Q K^T, P V and their gradients) run on MFMA in
// split-bf16 out of bf16 LDS images, the head dim (16 .. 256 in the reference
// grid, config-transformer.yaml:49,53) in chunks of 64; softmax, masks and
// dropout live in the accumulator layout.
//
// Cross-attention (decoder, tgt length 1, no masks): one workgroup per (batch, head),
// matrix-vector products out of LDS tiles.
#include "common.hpp"
#include "launch.hpp"

namespace slnlp {

constexpr int SMAX = 64;   // max sequence length held in one tile
constexpr int DCH = 64;    // head-dim chunk
constexpr int TLD = 68;    // LDS row stride (floats): 16-B aligned rows, conflict-free b128 column access

// ------------------------------------------------------- MFMA self-attention ---
// The 64x64x64 contractions of one (batch, head) run on v_mfma_f32_16x16x32_bf16 in split-bf16 (hi/lo, 3 passes,
// same arithmetic as the GEMMs); head dims above 64 are walked in 64-wide chunks.  All operand tiles live in LDS as bf16 [row][col] images with a 72-element row
// stride; an operand whose contraction index runs along the rows is read with ds_read_b64_tr_b16, so one image
// serves both orientations and nothing is transposed in HBM.  Softmax / softmax-backward / dropout happen in
// the MFMA accumulator layout (lane -> 4 consecutive rows x 1 column; a row's 64 columns sit in 16 lanes).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int ALD = 72;            // bf16 row stride: 144 B -> the 16 rows of a b128 fragment read hit 16 distinct 16-B slots
constexpr int ATILE = SMAX * ALD;  // one plane of one tile
constexpr int MFMA_DH = 64;        // head-dim chunk of the MFMA path

// issue the loads of a [S x dh] slice (token-major rows m = s*B + b); no use of the values here (see gemm.hip)
__device__ __forceinline__ void fetch_tile(const float* __restrict__ src, long ld, int B, int b, int S, int col0,
                                           int dc, int tid, float4 (&r)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + 256 * u, s = idx >> 4, c = (idx & 15) << 2;
        const bool ok = s < S && c < dc;
        r[u] = *reinterpret_cast<const float4*>(src + ((long)(ok ? s : 0) * B + b) * ld + col0 + (ok ? c : 0));
    }
}
// fp32 -> bf16 hi / lo planes (T, T + ATILE); rows >= S and cols >= dc are zero
__device__ __forceinline__ void stash_tile(unsigned short* __restrict__ T, int S, int dc, int tid, const float4 (&r)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + 256 * u, s = idx >> 4, c = (idx & 15) << 2;
        const bool ok = s < S && c < dc;
        const float x[4] = {ok ? r[u].x : 0.f, ok ? r[u].y : 0.f, ok ? r[u].z : 0.f, ok ? r[u].w : 0.f};
        unsigned short hh[4], ll[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split_bf16(x[e], hh[e], ll[e]);
        uint2 w;
        w.x = hh[0] | ((unsigned)hh[1] << 16); w.y = hh[2] | ((unsigned)hh[3] << 16);
        *reinterpret_cast<uint2*>(T + s * ALD + c) = w;
        w.x = ll[0] | ((unsigned)ll[1] << 16); w.y = ll[2] | ((unsigned)ll[3] << 16);
        *reinterpret_cast<uint2*>(T + ATILE + s * ALD + c) = w;
    }
}
// fragment whose contraction index k runs along the image columns: lane l <- (row r0 + (l&15), k = 32kk + 8(l>>4) + j)
__device__ __forceinline__ bf16x8 frag_rows(const unsigned short* __restrict__ T, int r0, int kk, int lane) {
    return *reinterpret_cast<const bf16x8*>(T + (r0 + (lane & 15)) * ALD + kk * 32 + ((lane >> 4) << 3));
}
// fragment whose contraction index runs along the image rows: lane l <- (k = 32kk + 8(l>>4) + j, col c0 + (l&15))
__device__ __forceinline__ bf16x8 frag_cols(const unsigned short* __restrict__ T, int c0, int kk, int lane) {
    const int i = lane & 15, kb = kk * 32 + ((lane >> 4) << 3) + (i >> 2);
    const unsigned short* p0 = T + kb * ALD + c0 + ((i & 3) << 2);
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0 + 4 * ALD));
    const s16x8 v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ f32x4 mfma3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
}
// reductions over the 16 lanes that hold one accumulator row: row16_max / row16_sum of common.hpp (DPP, no LDS crossbar)
// keep flags of rows row0..row0+3 at columns col and col + 16 (bit 4 of col clear) of a dropout site: the 4 x 2 block one Philox call
// serves when row0 % 4 == 0 (common.hpp); bits 0-3 = column col, bits 4-7 = column col + 16
__device__ __forceinline__ unsigned keep_mask8(const DropKey& key, unsigned row0, unsigned col, unsigned thr) {
    unsigned m = 0;
    if ((row0 & 3u) == 0u) {
        const uint4 bits = dropout_bits8(key, row0 >> 2, drop_cc(col));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) m |= (pick_lot(bits, h, r) >= thr ? 1u : 0u) << (4 * h + r);
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) m |= (dropout_keep(key, row0 + r, col + 16 * h, thr) ? 1u : 0u) << (4 * h + r);
    }
    return m;
}
__device__ __forceinline__ void put_planes(unsigned short* __restrict__ T, int off, float v) {
    unsigned short h, l;
    split_bf16(v, h, l);
    T[off] = h;
    T[ATILE + off] = l;
}

// This wave's 16 rows x dc columns of an output tile, as hi / lo planes, through ITS rows of a free LDS tile T: the accumulator
// layout holds a row's 64 columns in four pieces of 16 lanes, so storing from it takes 2 x 16 two-byte stores per lane in 32-byte
// segments (and a 64-bit address each); from the tile a lane copies 16 bytes of a row.  Wave-local: only this wave writes and
// reads these rows, LDS operations of a wave execute in order.  Same values as store_planes1 (split_bf16).
__device__ __forceinline__ void store_rows_via_lds(unsigned short* __restrict__ T, const f32x4 (&o)[4], int wave, int lane, int S, int dc,
                                                   const PlaneOut& po, long row_stride, long col0, int B, int b) {
    const int jc = lane & 15, i0 = wave * 16 + ((lane >> 4) << 2);
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) put_planes(T, (i0 + r) * ALD + 16 * n + jc, o[n][r]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int row = wave * 16 + pass * 8 + (lane >> 3), c8 = (lane & 7) << 3;
        if (row < S && c8 < dc) {
            const uint4 vh = *reinterpret_cast<const uint4*>(T + row * ALD + c8), vl = *reinterpret_cast<const uint4*>(T + ATILE + row * ALD + c8);
            const long at = ((long)row * B + b) * row_stride + col0 + c8;
            *reinterpret_cast<uint4*>(po.hi + at) = vh;
            *reinterpret_cast<uint4*>(po.lo + at) = vl;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();                         // (the rows may be reused by this wave's next output)
}
// planes only (no fp32 copy, no fp8 plane), 16-byte aligned row pieces
__device__ __forceinline__ bool planes_rows_ok(const PlaneOut& po, const void* fp32_out, int dh, long row_stride) {
    return po.hi && po.lo && !po.q8 && !fp32_out && (dh & 7) == 0 && (row_stride & 7) == 0 &&
           (((unsigned long long)po.hi | (unsigned long long)po.lo) & 15ull) == 0;
}

__device__ __forceinline__ void attn_self_fwd_mfma_body(
    const float* __restrict__ qkv, const long* __restrict__ ids, long ld_ids, long pad_idx, int causal, int B,
    int S, int H, int dh, float* __restrict__ ctx, float* __restrict__ probs, float drop_p, unsigned drop_thr,
    int drop_site, const unsigned long long* __restrict__ rng, PlaneOut po) {
    __shared__ __attribute__((aligned(16))) unsigned short sm[6 * ATILE];   // Q (then P) | K | V, hi+lo each: 55 296 B
    unsigned short* TQ = sm;
    unsigned short* TK = sm + 2 * ATILE;
    unsigned short* TV = sm + 4 * ATILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H, E = H * dh;
    const long ld = 3L * E;
    const int jc = lane & 15, i0 = wave * 16 + ((lane >> 4) << 2);
    const int nch = (dh + MFMA_DH - 1) / MFMA_DH;                           // head dim in chunks of 64

    long idv[4] = {0, 0, 0, 0};
    if (ids) {
#pragma unroll
        for (int n = 0; n < 4; ++n) { const int j = 16 * n + jc; idv[n] = ids[(long)b * ld_ids + (j < S ? j : 0)]; }
    }
    // scores: rows 16w..16w+15 of Q K^T, accumulated over the head-dim chunks
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < nch; ++ch) {
        const int d0 = ch * MFMA_DH, dc = dh - d0 < MFMA_DH ? dh - d0 : MFMA_DH;
        float4 rq[4], rk[4], rv[4];
        fetch_tile(qkv, ld, B, b, S, h * dh + d0, dc, tid, rq);
        fetch_tile(qkv, ld, B, b, S, E + h * dh + d0, dc, tid, rk);
        if (nch == 1) fetch_tile(qkv, ld, B, b, S, 2 * E + h * dh, dc, tid, rv);
        if (ch > 0) __syncthreads();                                        // every wave is done with the previous chunk
        stash_tile(TQ, S, dc, tid, rq);
        stash_tile(TK, S, dc, tid, rk);
        if (nch == 1) stash_tile(TV, S, dc, tid, rv);
        __syncthreads();
        const int nk = (dc + 31) >> 5;
        for (int kk = 0; kk < nk; ++kk) {
            const bf16x8 ah = frag_rows(TQ, wave * 16, kk, lane), al = frag_rows(TQ + ATILE, wave * 16, kk, lane);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n] = mfma3(ah, al, frag_rows(TK, 16 * n, kk, lane), frag_rows(TK + ATILE, 16 * n, kk, lane), acc[n]);
        }
    }
    // softmax + dropout in the accumulator layout; P (dropped) replaces this wave's own Q rows
    const float scale = rsqrtf((float)dh), inv_keep = 1.f / (1.f - drop_p);
    const long prow0 = ((long)b * H + h) * S + i0;
    unsigned keep[4] = {15u, 15u, 15u, 15u};
    bool kblock[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int j = 16 * n + jc;
        kblock[n] = j >= S || (ids && idv[n] == pad_idx);
    }
    if (drop_p > 0.f && i0 < S) {
        const DropKey dkey = dropout_key(rng, drop_site);
#pragma unroll
        for (int np = 0; np < 2; ++np) {
            if (32 * np + jc >= S) continue;     // (the pair's second column may lie past S: its flags are never used)
            const unsigned m = keep_mask8(dkey, (unsigned)prow0, (unsigned)(32 * np + jc), drop_thr);
            keep[2 * np] = m & 15u;
            keep[2 * np + 1] = m >> 4;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + r;
        float v[4], m = -INFINITY;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const bool blocked = kblock[n] || (causal && 16 * n + jc > i);
            v[n] = blocked ? -INFINITY : acc[n][r] * scale;
            m = fmaxf(m, v[n]);
        }
        m = row16_max(m);
        float sum = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) { v[n] = expf(v[n] - m); sum += v[n]; }   // all-masked row: NaN, as torch
        sum = row16_sum(sum);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int j = 16 * n + jc;
            float p = v[n] / sum;
            if (i < S && j < S) probs[(prow0 + r) * S + j] = p;
            else p = 0.f;
            if (drop_p > 0.f) p = ((keep[n] >> r) & 1u) ? p * inv_keep : 0.f;
            put_planes(TQ, i * ALD + j, p);
        }
    }
    // ctx = P V, one head-dim chunk at a time
    const int nkk = (S + 31) >> 5;
    for (int ch = 0; ch < nch; ++ch) {
        const int d0 = ch * MFMA_DH, dc = dh - d0 < MFMA_DH ? dh - d0 : MFMA_DH, nd = (dc + 15) >> 4;
        if (nch > 1) {
            float4 rv[4];
            fetch_tile(qkv, ld, B, b, S, 2 * E + h * dh + d0, dc, tid, rv);
            __syncthreads();                                                // previous V chunk consumed
            stash_tile(TV, S, dc, tid, rv);
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < nkk; ++kk) {
            const bf16x8 ah = frag_rows(TQ, wave * 16, kk, lane), al = frag_rows(TQ + ATILE, wave * 16, kk, lane);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n < nd) acc[n] = mfma3(ah, al, frag_cols(TV, 16 * n, kk, lane), frag_cols(TV + ATILE, 16 * n, kk, lane), acc[n]);
        }
        if (planes_rows_ok(po, ctx, dh, E)) {               // (K's tile is free since the scores; this wave's rows of it)
            store_rows_via_lds(TK, acc, wave, lane, S, dc, po, E, (long)h * dh + d0, B, b);
            continue;
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int d = 16 * n + jc;
            if (d >= dc) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + r;
                if (i >= S) continue;
                const long at = ((long)i * B + b) * E + h * dh + d0 + d;
                if (ctx) ctx[at] = acc[n][r];               // (a caller whose consumers read the planes passes ctx = nullptr)
                store_planes1(po, at, acc[n][r]);
            }
        }
    }
}
SLNLP_ZKERNEL(attn_self_fwd_mfma_kernel, 256, attn_self_fwd_mfma_body)

// Backward.  Tiles (hi+lo each): TQ | TK | TO (dO) | TV (V, later dropped P) | TS (dS; only when the head dim needs
// more than one 64-wide chunk, else dS reuses TO): 73 728 B or 92 160 B of dynamic LDS.
__device__ __forceinline__ void attn_self_bwd_mfma_body(
    const float* __restrict__ qkv, const float* __restrict__ probs, const float* __restrict__ dctx, int B, int S,
    int H, int dh, float* __restrict__ dqkv, float drop_p, unsigned drop_thr, int drop_site,
    const unsigned long long* __restrict__ rng, PlaneOut po) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* TQ = reinterpret_cast<unsigned short*>(smem);
    unsigned short* TK = TQ + 2 * ATILE;
    unsigned short* TO = TQ + 4 * ATILE;
    unsigned short* TV = TQ + 6 * ATILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / H, h = blockIdx.x % H, E = H * dh;
    const long ld = 3L * E;
    const int jc = lane & 15, i0 = wave * 16 + ((lane >> 4) << 2);
    const long prow0 = ((long)b * H + h) * S + i0;
    const int nch = (dh + MFMA_DH - 1) / MFMA_DH;
    unsigned short* TS = nch > 1 ? TQ + 8 * ATILE : TO;

    float pr[4][4];   // [n][r]: probs of (row i0 + r, key 16n + jc); clamped address, masked after the loads
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * n + jc;
            const bool ok = (i0 + r) < S && j < S;
            pr[n][r] = probs[ok ? (prow0 + r) * S + j : 0];
        }
    // dP = dO V^T  (rows 16w..), accumulated over the head-dim chunks
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 rq[4], rk[4];
    for (int ch = 0; ch < nch; ++ch) {
        const int d0 = ch * MFMA_DH, dc = dh - d0 < MFMA_DH ? dh - d0 : MFMA_DH;
        float4 ro[4], rv[4];
        fetch_tile(dctx, E, B, b, S, h * dh + d0, dc, tid, ro);
        fetch_tile(qkv, ld, B, b, S, 2 * E + h * dh + d0, dc, tid, rv);
        if (nch == 1) {
            fetch_tile(qkv, ld, B, b, S, h * dh, dc, tid, rq);
            fetch_tile(qkv, ld, B, b, S, E + h * dh, dc, tid, rk);
        }
        if (ch > 0) __syncthreads();
        stash_tile(TO, S, dc, tid, ro);
        stash_tile(TV, S, dc, tid, rv);
        __syncthreads();
        const int nk = (dc + 31) >> 5;
        for (int kk = 0; kk < nk; ++kk) {
            const bf16x8 ah = frag_rows(TO, wave * 16, kk, lane), al = frag_rows(TO + ATILE, wave * 16, kk, lane);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n] = mfma3(ah, al, frag_rows(TV, 16 * n, kk, lane), frag_rows(TV + ATILE, 16 * n, kk, lane), acc[n]);
        }
    }
    if (nch == 1) {
        stash_tile(TQ, S, dh, tid, rq);
        stash_tile(TK, S, dh, tid, rk);
    }
    // softmax backward in registers: ds = p (dp - sum_j dp p) scale;  pd = dropped p
    const float scale = rsqrtf((float)dh), inv_keep = 1.f / (1.f - drop_p);
    float ds[4][4], pd[4][4];
    {
        unsigned keep[4] = {15u, 15u, 15u, 15u};
        if (drop_p > 0.f) {
            const DropKey dkey = dropout_key(rng, drop_site);
#pragma unroll
            for (int np = 0; np < 2; ++np) {
                if (32 * np + jc >= S || i0 >= S) continue;
                const unsigned m = keep_mask8(dkey, (unsigned)prow0, (unsigned)(32 * np + jc), drop_thr);
                keep[2 * np] = m & 15u;
                keep[2 * np + 1] = m >> 4;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float dp[4], s = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const bool ok = (i0 + r) < S && (16 * n + jc) < S;
                const float p = ok ? pr[n][r] : 0.f;
                const bool k = (keep[n] >> r) & 1u;
                pr[n][r] = p;
                dp[n] = drop_p > 0.f ? (k ? acc[n][r] * inv_keep : 0.f) : acc[n][r];
                pd[n][r] = drop_p > 0.f ? (k ? p * inv_keep : 0.f) : p;
                s += dp[n] * p;
            }
            s = row16_sum(s);
#pragma unroll
            for (int n = 0; n < 4; ++n) ds[n][r] = pr[n][r] * (dp[n] - s) * scale;
        }
    }
    __syncthreads();   // every wave is done with V (and, single chunk, Q / K tiles are complete)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            put_planes(TV, (i0 + r) * ALD + 16 * n + jc, pd[n][r]);
            if (nch > 1) put_planes(TS, (i0 + r) * ALD + 16 * n + jc, ds[n][r]);
        }
    const int nkk = (S + 31) >> 5;
    for (int ch = 0; ch < nch; ++ch) {
        const int d0 = ch * MFMA_DH, dc = dh - d0 < MFMA_DH ? dh - d0 : MFMA_DH, nd = (dc + 15) >> 4;
        auto store = [&](const f32x4 (&o)[4], int part) {   // rows 16w.. of dQ (0) / dK (1) / dV (2), this chunk's columns
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int d = 16 * n + jc;
                if (d >= dc) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + r;
                    if (i >= S) continue;
                    const long at = ((long)i * B + b) * ld + (long)part * E + h * dh + d0 + d;
                    if (dqkv) dqkv[at] = o[n][r];           // (likewise: planes only when dqkv = nullptr)
                    store_planes1(po, at, o[n][r]);
                }
            }
        };
        if (nch > 1) {   // this chunk's dO, Q, K columns
            float4 ro[4];
            fetch_tile(dctx, E, B, b, S, h * dh + d0, dc, tid, ro);
            fetch_tile(qkv, ld, B, b, S, h * dh + d0, dc, tid, rq);
            fetch_tile(qkv, ld, B, b, S, E + h * dh + d0, dc, tid, rk);
            __syncthreads();                                // previous chunk's tiles consumed
            stash_tile(TO, S, dc, tid, ro);
            stash_tile(TQ, S, dc, tid, rq);
            stash_tile(TK, S, dc, tid, rk);
        }
        __syncthreads();
        // dV = Pd^T dO   (rows = keys 16w..; contraction over s)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < nkk; ++kk) {
            const bf16x8 ah = frag_cols(TV, wave * 16, kk, lane), al = frag_cols(TV + ATILE, wave * 16, kk, lane);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n < nd) acc[n] = mfma3(ah, al, frag_cols(TO, 16 * n, kk, lane), frag_cols(TO + ATILE, 16 * n, kk, lane), acc[n]);
        }
        // outputs as row pieces through this wave's rows of a tile nobody reads any more: Pd's (single chunk: dead after dV) or this
        // chunk's dO tile (several chunks: dead after dV too, one more barrier)
        const bool rows_out = planes_rows_ok(po, dqkv, dh, ld);
        unsigned short* TR = nch == 1 ? TV : TO;
        if (!rows_out) store(acc, 2);
        if (nch > 1 && rows_out) {
            __syncthreads();                                 // every wave is done with this chunk's dO
            store_rows_via_lds(TR, acc, wave, lane, S, dc, po, ld, 2L * E + h * dh + d0, B, b);
        }
        if (nch == 1) {   // dS takes dO's tile
            __syncthreads();
            if (rows_out) store_rows_via_lds(TR, acc, wave, lane, S, dc, po, ld, 2L * E + h * dh + d0, B, b);   // (every wave is done with Pd)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) put_planes(TS, (i0 + r) * ALD + 16 * n + jc, ds[n][r]);
            __syncthreads();
        }
        // dQ = dS K  (rows 16w.., contraction over keys);  dK = dS^T Q  (rows = keys 16w.., contraction over s)
        f32x4 acq[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) { acc[n] = f32x4{0.f, 0.f, 0.f, 0.f}; acq[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int kk = 0; kk < nkk; ++kk) {
            const bf16x8 qh = frag_rows(TS, wave * 16, kk, lane), ql = frag_rows(TS + ATILE, wave * 16, kk, lane);
            const bf16x8 kh = frag_cols(TS, wave * 16, kk, lane), kl = frag_cols(TS + ATILE, wave * 16, kk, lane);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n < nd) {
                    acq[n] = mfma3(qh, ql, frag_cols(TK, 16 * n, kk, lane), frag_cols(TK + ATILE, 16 * n, kk, lane), acq[n]);
                    acc[n] = mfma3(kh, kl, frag_cols(TQ, 16 * n, kk, lane), frag_cols(TQ + ATILE, 16 * n, kk, lane), acc[n]);
                }
        }
        if (rows_out) {
            store_rows_via_lds(TR, acq, wave, lane, S, dc, po, ld, (long)h * dh + d0, B, b);
            store_rows_via_lds(TR, acc, wave, lane, S, dc, po, ld, (long)E + h * dh + d0, B, b);
        } else {
            store(acq, 0);
            store(acc, 1);
        }
    }
}
SLNLP_ZKERNEL(attn_self_bwd_mfma_kernel, 256, attn_self_bwd_mfma_body)

// ------------------------------------------------------------------ cross ---
constexpr int XDH = 256;  // max head dim

// One workgroup per (batch, head).  The work is a handful of 48 x 64 matrix-vector products, so the kernel is
// pure memory latency: every global load of a phase is issued at once (K and V tiles go to LDS with 16-B loads,
// three per thread), and the arithmetic then runs out of LDS.  Head dims above 64 loop over 64-wide chunks.
__device__ __forceinline__ void xload_tile(float* __restrict__ T, const float* __restrict__ src, long ld, int B, int b,
                                           int S, int col0, int dc, int tid) {
    float4 r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + 256 * u, s = idx >> 4, c = (idx & 15) << 2;
        const bool ok = s < S && c < dc;
        r[u] = *reinterpret_cast<const float4*>(src + ((long)(ok ? s : 0) * B + b) * ld + col0 + (ok ? c : 0));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = tid + 256 * u, s = idx >> 4, c = (idx & 15) << 2;
        const bool ok = s < S && c < dc;
        *reinterpret_cast<float4*>(T + s * TLD + c) = ok ? r[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// dot of tile row j = tid >> 2 with vec over the 16 columns owned by part = tid & 3, reduced over the 4 parts
__device__ __forceinline__ float xrow_dot(const float* __restrict__ T, const float* __restrict__ vec, int tid) {
    const int j = tid >> 2, c0 = (tid & 3) << 4;
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float4 t = *reinterpret_cast<const float4*>(T + j * TLD + c0 + 4 * u);
        const float4 v = *reinterpret_cast<const float4*>(vec + c0 + 4 * u);
        a += t.x * v.x + t.y * v.y + t.z * v.z + t.w * v.w;
    }
    a += dpp_mov<DPP_XOR1>(a);
    a += dpp_mov<DPP_XOR2>(a);
    return a;
}
// sum_j w[j] * T[j][d] for d = tid & 63: rows split over the 4 waves, combined through red[4][64]
__device__ __forceinline__ float xcol_sum(const float* __restrict__ T, const float* __restrict__ wj, float* __restrict__ red,
                                          int tid) {
    const int d = tid & 63, g = tid >> 6;
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) a += wj[16 * g + u] * T[(16 * g + u) * TLD + d];
    red[g * 64 + d] = a;
    __syncthreads();
    return red[d] + red[64 + d] + red[128 + d] + red[192 + d];
}

__device__ __forceinline__ void attn_cross_fwd_body(
    const float* __restrict__ q, const float* __restrict__ kv, long ld_kv, int B, int S, int H, int dh,
    float* __restrict__ ctx, float* __restrict__ probs, float drop_p, unsigned drop_thr, int drop_site,
    const unsigned long long* __restrict__ rng) {
    __shared__ __attribute__((aligned(16))) float Ks[SMAX * TLD];
    __shared__ __attribute__((aligned(16))) float Vs[SMAX * TLD];
    __shared__ __attribute__((aligned(16))) float qs[DCH];
    __shared__ float sc[SMAX], pd[SMAX], red[4 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int bh = blockIdx.x, b = bh / H, h = bh % H, E = H * dh;
    const int nch = (dh + DCH - 1) / DCH;
    float acc = 0.f;
    for (int ch = 0; ch < nch; ++ch) {
        const int d0 = ch * DCH, dc = dh - d0 < DCH ? dh - d0 : DCH;
        if (ch > 0) __syncthreads();
        if (tid < DCH) qs[tid] = tid < dc ? q[(long)b * E + h * dh + d0 + tid] : 0.f;
        xload_tile(Ks, kv, ld_kv, B, b, S, h * dh + d0, dc, tid);
        if (nch == 1) xload_tile(Vs, kv, ld_kv, B, b, S, E + h * dh, dc, tid);
        __syncthreads();
        acc += xrow_dot(Ks, qs, tid);
    }
    if ((tid & 3) == 0) sc[tid >> 2] = acc * rsqrtf((float)dh);
    __syncthreads();
    {   // softmax over the S keys: every wave computes it (no masks here), wave 0 publishes
        const float v = lane < S ? sc[lane] : -INFINITY;
        const float m = wave_max(v);
        const float e = expf(v - m);
        const float p = e / wave_sum(e);
        if (tid < 64) {
            float d = 0.f;
            if (lane < S) {
                probs[(long)bh * S + lane] = p;
                d = p;
                if (drop_p > 0.f) d = dropout_keep(rng, drop_site, (unsigned)bh, (unsigned)lane, drop_thr) ? p / (1.f - drop_p) : 0.f;
            }
            pd[lane] = d;
        }
    }
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int d0 = ch * DCH, dc = dh - d0 < DCH ? dh - d0 : DCH;
        if (nch > 1) {
            __syncthreads();
            xload_tile(Vs, kv, ld_kv, B, b, S, E + h * dh + d0, dc, tid);
            __syncthreads();
        }
        const float o = xcol_sum(Vs, pd, red, tid);
        if (tid < dc) ctx[(long)b * E + h * dh + d0 + tid] = o;
    }
}
SLNLP_ZKERNEL(attn_cross_fwd_kernel, 256, attn_cross_fwd_body)

__device__ __forceinline__ void attn_cross_bwd_body(
    const float* __restrict__ q, const float* __restrict__ kv, long ld_kv, const float* __restrict__ probs,
    const float* __restrict__ dctx, int B, int S, int H, int dh, float* __restrict__ dq, float* __restrict__ dkv,
    long ld_dkv, float drop_p, unsigned drop_thr, int drop_site, const unsigned long long* __restrict__ rng,
    PlaneOut po) {
    __shared__ __attribute__((aligned(16))) float Ks[SMAX * TLD];
    __shared__ __attribute__((aligned(16))) float Vs[SMAX * TLD];
    __shared__ __attribute__((aligned(16))) float qs[DCH];
    __shared__ __attribute__((aligned(16))) float gs[DCH];
    __shared__ float dpv[SMAX], dss[SMAX], pds[SMAX], red[4 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int bh = blockIdx.x, b = bh / H, h = bh % H, E = H * dh;
    const int nch = (dh + DCH - 1) / DCH;
    const float pj = lane < S ? probs[(long)bh * S + lane] : 0.f;
    // dP[j] = dctx . V[j]
    float acc = 0.f;
    for (int ch = 0; ch < nch; ++ch) {
        const int d0 = ch * DCH, dc = dh - d0 < DCH ? dh - d0 : DCH;
        if (ch > 0) __syncthreads();
        if (tid < DCH) gs[tid] = tid < dc ? dctx[(long)b * E + h * dh + d0 + tid] : 0.f;
        else if (tid < 2 * DCH && nch == 1) qs[tid - DCH] = tid - DCH < dc ? q[(long)b * E + h * dh + tid - DCH] : 0.f;
        xload_tile(Vs, kv, ld_kv, B, b, S, E + h * dh + d0, dc, tid);
        if (nch == 1) xload_tile(Ks, kv, ld_kv, B, b, S, h * dh, dc, tid);
        __syncthreads();
        acc += xrow_dot(Vs, gs, tid);
    }
    if ((tid & 3) == 0) dpv[tid >> 2] = acc;
    __syncthreads();
    {   // softmax backward (every wave; wave 0 publishes): ds = p (dp - sum dp p) / sqrt(dh), pd = dropped p
        float dp = lane < S ? dpv[lane] : 0.f, d = pj;
        if (drop_p > 0.f && lane < S) {
            const bool keep = dropout_keep(rng, drop_site, (unsigned)bh, (unsigned)lane, drop_thr);
            const float ik = 1.f / (1.f - drop_p);
            d = keep ? pj * ik : 0.f;
            dp = keep ? dp * ik : 0.f;
        }
        const float ssum = wave_sum(dp * pj);
        if (tid < 64) {
            dss[lane] = pj * (dp - ssum) * rsqrtf((float)dh);
            pds[lane] = d;
        }
    }
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int d0 = ch * DCH, dc = dh - d0 < DCH ? dh - d0 : DCH;
        if (nch > 1) {
            __syncthreads();
            if (tid < DCH) gs[tid] = tid < dc ? dctx[(long)b * E + h * dh + d0 + tid] : 0.f;
            else if (tid < 2 * DCH) qs[tid - DCH] = tid - DCH < dc ? q[(long)b * E + h * dh + d0 + tid - DCH] : 0.f;
            xload_tile(Ks, kv, ld_kv, B, b, S, h * dh + d0, dc, tid);
            __syncthreads();
        }
        const float o = xcol_sum(Ks, dss, red, tid);                 // dq = sum_j ds[j] K[j]
        if (tid < dc) dq[(long)b * E + h * dh + d0 + tid] = o;
        // dK[j] = ds[j] q,  dV[j] = pd[j] dctx : 16 rows x 64 columns per pass, 16-B stores
        const int c = (tid & 15) << 2;
        if (c < dc) {
            const float4 q4 = *reinterpret_cast<const float4*>(qs + c), g4 = *reinterpret_cast<const float4*>(gs + c);
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int j = (tid >> 4) + 16 * ps;
                if (j >= S) continue;
                const float dsj = dss[j], pdj = pds[j];
                const long at = ((long)j * B + b) * ld_dkv + h * dh + d0 + c;
                const float4 dk = make_float4(dsj * q4.x, dsj * q4.y, dsj * q4.z, dsj * q4.w);
                const float4 dv = make_float4(pdj * g4.x, pdj * g4.y, pdj * g4.z, pdj * g4.w);
                *reinterpret_cast<float4*>(dkv + at) = dk;
                *reinterpret_cast<float4*>(dkv + at + E) = dv;
                store_planes4(po, at, dk);
                store_planes4(po, at + E, dv);
            }
        }
    }
}
SLNLP_ZKERNEL(attn_cross_bwd_kernel, 256, attn_cross_bwd_body)

constexpr size_t ATTN_BWD_MFMA_LDS = 10 * ATILE * sizeof(unsigned short);  // up to 92 160 B (head dim > 64)

// one-time opt-in to > 64 KiB dynamic LDS; called from plan creation so it never lands inside a graph capture
int attn_init() {
    static DeviceOnce once;
    return once.run([]() -> int {
        if (hipFuncSetAttribute((const void*)attn_self_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)ATTN_BWD_MFMA_LDS) != hipSuccess) {
            set_error("attn_init: cannot raise dynamic LDS limit: %s", hipGetErrorString(hipGetLastError()));
            return SLNLP_ERR_LAUNCH;
        }
        return 0;
    });
}

// S > 64: the wave-per-row kernels of attention_long.hip
int attn_self_fwd_long(const float* qkv, const int64_t* ids, int64_t ld_ids, int64_t pad_idx, int causal, int B, int S, int H,
                       int dh, float* ctx, float* probs, float drop_p, int drop_site, const unsigned long long* rng,
                       hipStream_t st, PlaneOut po);
int attn_self_bwd_long(const float* qkv, const float* probs, const float* dctx, int B, int S, int H, int dh, float* dqkv,
                       float* scratch, float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st, PlaneOut po);
int attn_cross_fwd_long(const float* q, const float* kv, int64_t ld_kv, int B, int S, int H, int dh, float* ctx, float* probs,
                        float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st);
int attn_cross_bwd_long(const float* q, const float* kv, int64_t ld_kv, const float* probs, const float* dctx, int B, int S, int H,
                        int dh, float* dq, float* dkv, int64_t ld_dkv, float drop_p, int drop_site, const unsigned long long* rng,
                        hipStream_t st, PlaneOut po);

static int check_attn(const char* who, int B, int S, int H, int dh) {
    SLNLP_CHECK_ARG(B > 0 && H > 0, "%s: bad B=%d H=%d", who, B, H);
    SLNLP_CHECK_ARG(S > 0 && S <= SMAX, "%s: S=%d outside 1..%d (single-tile kernel)", who, S, SMAX);
    SLNLP_CHECK_ARG(dh > 0 && dh % 4 == 0 && dh <= XDH && (dh <= DCH || dh % DCH == 0),
                    "%s: head_dim %d unsupported (need multiple of 4, <= %d, and a multiple of %d above it)", who,
                    dh, XDH, DCH);
    return 0;
}

int attn_self_fwd(const float* qkv, const int64_t* ids, int64_t ld_ids, int64_t pad_idx, int causal, int B, int S,
                  int H, int dh, float* ctx, float* probs, float drop_p, int drop_site,
                  const unsigned long long* rng, hipStream_t st, PlaneOut po) {
    SLNLP_CHECK_ARG(qkv && probs && (ctx || (po.hi && S <= SMAX)), "attn_self_fwd: null pointer (ctx may be NULL only with output planes, S <= %d)", SMAX);
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "attn_self_fwd: bad dropout args");
    if (S > SMAX) return attn_self_fwd_long(qkv, ids, ld_ids, pad_idx, causal, B, S, H, dh, ctx, probs, drop_p, drop_site, rng, st, po);
    SLNLP_TRY(check_attn("attn_self_fwd", B, S, H, dh));
    SLNLP_TRY(zlaunch(attn_self_fwd_mfma_kernel, dim3(B * H), 256, 0, st, "attn_self_fwd",
                      qkv, (const long*)ids, (long)ld_ids, (long)pad_idx, causal, B, S, H, dh, ctx, probs, drop_p, dropout_threshold(drop_p), drop_site, rng, po));
    return 0;
}

int attn_self_bwd(const float* qkv, const float* probs, const float* dctx, int B, int S, int H, int dh, float* dqkv,
                  float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st, PlaneOut po, float* long_scratch) {
    SLNLP_CHECK_ARG(qkv && probs && dctx && (dqkv || (po.hi && S <= SMAX)), "attn_self_bwd: null pointer (dqkv may be NULL only with output planes, S <= %d)", SMAX);
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "attn_self_bwd: bad dropout args");
    if (S > SMAX) return attn_self_bwd_long(qkv, probs, dctx, B, S, H, dh, dqkv, long_scratch, drop_p, drop_site, rng, st, po);
    SLNLP_TRY(check_attn("attn_self_bwd", B, S, H, dh));
    SLNLP_TRY(attn_init());
    SLNLP_TRY(zlaunch(attn_self_bwd_mfma_kernel, dim3(B * H), 256, (dh <= MFMA_DH ? 8 : 10) * ATILE * sizeof(unsigned short), st, "attn_self_bwd",
                      qkv, probs, dctx, B, S, H, dh, dqkv, drop_p, dropout_threshold(drop_p), drop_site, rng, po));
    return 0;
}

int attn_cross_fwd(const float* q, const float* kv, int64_t ld_kv, int B, int S, int H, int dh, float* ctx,
                   float* probs, float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st) {
    SLNLP_CHECK_ARG(q && kv && ctx && probs, "attn_cross_fwd: null pointer");
    SLNLP_CHECK_ARG(ld_kv % 4 == 0 && ld_kv >= 2L * H * dh, "attn_cross_fwd: ld_kv=%ld", (long)ld_kv);
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "attn_cross_fwd: bad dropout args");
    if (S > SMAX) return attn_cross_fwd_long(q, kv, ld_kv, B, S, H, dh, ctx, probs, drop_p, drop_site, rng, st);
    SLNLP_TRY(check_attn("attn_cross_fwd", B, S, H, dh));
    SLNLP_TRY(zlaunch(attn_cross_fwd_kernel, dim3(B * H), 256, 0, st, "attn_cross_fwd",
                      q, kv, (long)ld_kv, B, S, H, dh, ctx, probs, drop_p, dropout_threshold(drop_p), drop_site, rng));
    return 0;
}

int attn_cross_bwd(const float* q, const float* kv, int64_t ld_kv, const float* probs, const float* dctx, int B,
                   int S, int H, int dh, float* dq, float* dkv, int64_t ld_dkv, float drop_p, int drop_site,
                   const unsigned long long* rng, hipStream_t st, PlaneOut po) {
    SLNLP_CHECK_ARG(q && kv && probs && dctx && dq && dkv, "attn_cross_bwd: null pointer");
    SLNLP_CHECK_ARG(ld_kv % 4 == 0 && ld_kv >= 2L * H * dh && ld_dkv >= 2L * H * dh && ld_dkv % 4 == 0, "attn_cross_bwd: bad ld");
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "attn_cross_bwd: bad dropout args");
    if (S > SMAX) return attn_cross_bwd_long(q, kv, ld_kv, probs, dctx, B, S, H, dh, dq, dkv, ld_dkv, drop_p, drop_site, rng, st, po);
    SLNLP_TRY(check_attn("attn_cross_bwd", B, S, H, dh));
    SLNLP_TRY(zlaunch(attn_cross_bwd_kernel, dim3(B * H), 256, 0, st, "attn_cross_bwd",
                      q, kv, (long)ld_kv, probs, dctx, B, S, H, dh, dq, dkv, (long)ld_dkv, drop_p, dropout_threshold(drop_p), drop_site, rng, po));
    return 0;
}

}  // namespace slnlp

extern "C" {
int slnlp_attn_self_fwd(const float* qkv, const int64_t* ids, int64_t ld_ids, int64_t pad_idx, int causal, int B,
                        int S, int H, int dh, float* ctx, float* probs, float drop_p, int drop_site,
                        const unsigned long long* rng, void* stream) {
    return slnlp::attn_self_fwd(qkv, ids, ld_ids, pad_idx, causal, B, S, H, dh, ctx, probs, drop_p, drop_site, rng,
                                (hipStream_t)stream, slnlp::PlaneOut{});
}
int slnlp_attn_self_bwd(const float* qkv, const float* probs, const float* dctx, int B, int S, int H, int dh,
                        float* dqkv, float drop_p, int drop_site, const unsigned long long* rng, void* stream) {
    return slnlp::attn_self_bwd(qkv, probs, dctx, B, S, H, dh, dqkv, drop_p, drop_site, rng, (hipStream_t)stream);
}
int64_t slnlp_attn_long_scratch_bytes(int B, int S, int H) {
    return (B > 0 && S > 64 && H > 0) ? (int64_t)slnlp::attn_long_scratch_bytes(B, S, H) : 0;
}
int slnlp_attn_self_bwd_long(const float* qkv, const float* probs, const float* dctx, int B, int S, int H, int dh,
                             float* dqkv, float* scratch, float drop_p, int drop_site, const unsigned long long* rng, void* stream) {
    return slnlp::attn_self_bwd(qkv, probs, dctx, B, S, H, dh, dqkv, drop_p, drop_site, rng, (hipStream_t)stream, slnlp::PlaneOut{}, scratch);
}
int slnlp_attn_cross_fwd(const float* q, const float* kv, int64_t ld_kv, int B, int S, int H, int dh, float* ctx,
                         float* probs, float drop_p, int drop_site, const unsigned long long* rng, void* stream) {
    return slnlp::attn_cross_fwd(q, kv, ld_kv, B, S, H, dh, ctx, probs, drop_p, drop_site, rng, (hipStream_t)stream);
}
int slnlp_attn_cross_bwd(const float* q, const float* kv, int64_t ld_kv, const float* probs, const float* dctx,
                         int B, int S, int H, int dh, float* dq, float* dkv, int64_t ld_dkv, float drop_p,
                         int drop_site, const unsigned long long* rng, void* stream) {
    return slnlp::attn_cross_bwd(q, kv, ld_kv, probs, dctx, B, S, H, dh, dq, dkv, ld_dkv, drop_p, drop_site, rng,
                                 (hipStream_t)stream);
}
}
