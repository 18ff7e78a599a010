// attention_mem.hip -- the decoder's cross-attention for a target of length 1, WITHOUT projecting the memory.
//
// Reference arithmetic (/root/reference/model/transformer.py:82-87 -> nn.MultiheadAttention, tgt length 1, no masks):
//     k_s = Wk mem_s + bk,  v_s = Wv mem_s + bv,   score_s = q_h . k_{s,h} / sqrt(dh),   ctx_h = sum_s p_s v_{s,h}
// i.e. two [S*B, E] x [E, E] GEMMs per decoder layer (24 % of the step's GEMM FLOPs with their gradients) whose 2 x S
// projected rows per sequence are each used ONCE, by a single query.  With one query per sequence the products
// re-associate exactly:
//     score_s = (Wk_h^T q_h) . mem_s / sqrt(dh)  + q_h . bk_h / sqrt(dh)     -- the second term is the same for every s:
//                                                                              softmax does not see it
//     ctx_h   = Wv_h (sum_s p_s mem_s) + bv_h sum_s p_s                       -- (sum_s p_s != 1 under dropout)
// so the layer needs  qk = Wk_h^T q_h  and  Wv_h mbar  -- B-row products -- and never forms K or V.  The backward
// re-associates the same way: d mbar = Wv_h^T d ctx_h, d p_s = d mbar . mem_s, d qk = sum_s d s_s mem_s,
// d q_h = Wk_h d qk, d mem_s += sum_h (p_s d mbar_h + d s_s qk_h); the weight gradients are per-head B-row products
// (d Wk_h = q_h^T (x) d qk, d Wv_h = d ctx_h^T (x) mbar) that run as grouped GEMM jobs, d bk is exactly zero (the
// reference computes rounding noise of the order 1e-9 there) and d bv = sum_b (sum_s p_s) d ctx.
//
// One workgroup per (sequence, head), fp32 FMA arithmetic; every global read is a coalesced row (weights [row][E],
// memory rows [E]).  The dropout element of p_s is (row b*H + h, column s), as in attention.hip's cross kernels.
#include "common.hpp"
#include "launch.hpp"

namespace slnlp {

// The four per-head weight products (qk = Wk_h^T q_h, ctx_h = Wv_h mbar, d mbar = Wv_h^T d ctx_h, d q_h = Wk_h d qk) and the
// two weight gradients are B-row GEMMs: they run as BATCHED jobs (batch = H) of the grouped fp32-operand GEMM kernel
// (gemm.hip), issued by the plan.  (A first version did them as matrix-vector loops inside the attention workgroups:
// 450 KB of weight rows per workgroup through dependent loads -- the cfg2 step went from 3.21 to 4.07 ms.)  What stays
// here touches only the memory rows: scores, softmax, dropout, the weighted sums over s, and their gradients.

// dots of an LDS vector v[E] with memory rows s = wave, wave + 4, ... of sequence b: FOUR rows per trip, so their loads are
// in flight together (one row per trip left each wave with a dependent load -> reduce chain per row: 19 us per kernel)
// workgroups go round-robin over the 8 XCDs, each with its own L2: with bh = blockIdx.x the H heads of a sequence sit on H
// different XCDs and every XCD streams ALL of mem (4.9 MB at cfg2, more than its 4 MB L2) -- 19 us per kernel.  Give XCD x the
// contiguous range [x nb/8, (x+1) nb/8) instead: each L2 then holds only the ~B/8 sequences its workgroups share.
__device__ __forceinline__ int xcd_local(int i, int nb) { return (nb & 7) == 0 ? (i & 7) * (nb >> 3) + (i >> 3) : i; }

template <class F>
__device__ __forceinline__ void row_dots(const float* __restrict__ v, const float* __restrict__ mem, int B, int b, int S, int E, int lane,
                                         int wave, F&& put) {
    for (int s0 = wave; s0 < S; s0 += 16) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + 4 * u;
            if (s < S) {
                const float* mr = mem + ((long)s * B + b) * E;
                for (int e = lane * 4; e < E; e += 256) {
                    const float4 m4 = *reinterpret_cast<const float4*>(mr + e), q4 = *reinterpret_cast<const float4*>(v + e);
                    a[u] += q4.x * m4.x + q4.y * m4.y + q4.z * m4.z + q4.w * m4.w;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float r = wave_sum(a[u]);
            if (lane == 0 && s0 + 4 * u < S) put(s0 + 4 * u, r);
        }
    }
}

// ---------------------------------------------------------------------------------------------------- forward ----
// qk [B*H, E] (from the batched GEMM); mem rows m = s*B + b, [S*B, E]; outputs: mbar [B*H, E], psum [B*H],
// probs [B*H, S] (pre-dropout), ctx0 [B, E] = bv * sum_s p_s (the batched GEMM Wv_h mbar adds onto it)
__device__ __forceinline__ void xmem_fwd_body(const float* __restrict__ qk, const float* __restrict__ mem, const float* __restrict__ bv,
                                              int B, int S, int H, int dh, float* __restrict__ mbar_out, float* __restrict__ psum_out,
                                              float* __restrict__ probs, float* __restrict__ ctx0, float drop_p, unsigned drop_thr,
                                              int drop_site, const unsigned long long* __restrict__ rng) {
    extern __shared__ __attribute__((aligned(16))) float xm_lds[];   // qk[E] | sc[S] | psum
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = xcd_local(blockIdx.x, gridDim.x), b = bh / H, h = bh % H, E = H * dh;
    float* qv = xm_lds;
    float* sc = qv + E;
    float* ps = sc + S;
    for (int e = tid * 4; e < E; e += 1024) *reinterpret_cast<float4*>(qv + e) = *reinterpret_cast<const float4*>(qk + (long)bh * E + e);
    __syncthreads();
    const float scale = rsqrtf((float)dh), inv_keep = 1.f / (1.f - drop_p);
    row_dots(qv, mem, B, b, S, E, lane, wave, [&](int s_, float r) { sc[s_] = r * scale; });   // scores: a wave-wide dot product per memory row
    __syncthreads();
    if (wave == 0) {
        float m = -INFINITY;
        for (int s = lane; s < S; s += 64) m = fmaxf(m, sc[s]);
        m = wave_max(m);
        float sum = 0.f;
        for (int s = lane; s < S; s += 64) sum += expf(sc[s] - m);
        sum = wave_sum(sum);
        float tot = 0.f;
        for (int s = lane; s < S; s += 64) {
            float p = expf(sc[s] - m) / sum;
            probs[(long)bh * S + s] = p;
            if (drop_p > 0.f) p = dropout_keep(rng, drop_site, (unsigned)bh, (unsigned)s, drop_thr) ? p * inv_keep : 0.f;
            sc[s] = p;
            tot += p;
        }
        tot = wave_sum(tot);
        if (lane == 0) {
            ps[0] = tot;
            psum_out[bh] = tot;
        }
    }
    __syncthreads();
    for (int e = tid * 4; e < E; e += 1024) {            // mbar = sum_s p_s mem_s
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 16
        for (int s = 0; s < S; ++s) {
            const float p = sc[s];
            const float4 m4 = *reinterpret_cast<const float4*>(mem + ((long)s * B + b) * E + e);
            a.x += p * m4.x; a.y += p * m4.y; a.z += p * m4.z; a.w += p * m4.w;
        }
        *reinterpret_cast<float4*>(mbar_out + (long)bh * E + e) = a;
    }
    for (int j = tid; j < dh; j += 256) ctx0[(long)b * E + h * dh + j] = bv[h * dh + j] * ps[0];
}
SLNLP_ZKERNEL(xmem_fwd_kernel, 256, xmem_fwd_body)

// --------------------------------------------------------------------------------------- backward, per (b, h) ----
// in: dmbar [B*H, E] (= Wv_h^T d ctx_h, from the batched GEMM), d ctx [B,E]; out: dsc [B*H, S] (= d score_s, scale folded
// in), dqk [B*H, E], dcp [B,E] = d ctx * sum_s p_s (its column sums are d bv).
// d p_s = d mbar . mem_s + d ctx_h . bv_h  -- the second term (from ctx_h's  bv_h sum_s p_s) is the same for every s and
// cancels in the softmax backward ONLY when no dropout mask sits between p and the sum.
__device__ __forceinline__ void xmem_bwd_body(const float* __restrict__ mem, const float* __restrict__ bv, const float* __restrict__ probs,
                                              const float* __restrict__ psum, const float* __restrict__ dmbar,
                                              const float* __restrict__ dctx, int B, int S, int H, int dh, float* __restrict__ dsc_out,
                                              float* __restrict__ dqk_out, float* __restrict__ dcp, float drop_p, unsigned drop_thr,
                                              int drop_site, const unsigned long long* __restrict__ rng) {
    extern __shared__ __attribute__((aligned(16))) float xm_lds[];   // dmb[E] | t[S] | c
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = xcd_local(blockIdx.x, gridDim.x), b = bh / H, h = bh % H, E = H * dh;
    float* dmb = xm_lds;
    float* t = dmb + E;
    float* cc = t + S;
    const float ps = psum[bh];
    for (int e = tid * 4; e < E; e += 1024) *reinterpret_cast<float4*>(dmb + e) = *reinterpret_cast<const float4*>(dmbar + (long)bh * E + e);
    if (wave == 0) {
        float c = 0.f;
        for (int j = lane; j < dh; j += 64) {
            const float g = dctx[(long)b * E + h * dh + j];
            c += g * bv[h * dh + j];
            dcp[(long)b * E + h * dh + j] = g * ps;
        }
        c = wave_sum(c);
        if (lane == 0) cc[0] = c;
    }
    __syncthreads();
    row_dots(dmb, mem, B, b, S, E, lane, wave, [&](int s_, float r) { t[s_] = r + cc[0]; });   // d p_s (after dropout) = d mbar . mem_s + d ctx_h . bv_h
    __syncthreads();
    const float scale = rsqrtf((float)dh), inv_keep = 1.f / (1.f - drop_p);
    if (wave == 0) {                                     // through the dropout mask and the softmax
        float dot = 0.f;
        for (int s = lane; s < S; s += 64) {
            float dp = t[s];
            if (drop_p > 0.f) dp = dropout_keep(rng, drop_site, (unsigned)bh, (unsigned)s, drop_thr) ? dp * inv_keep : 0.f;
            t[s] = dp;
            dot += probs[(long)bh * S + s] * dp;
        }
        dot = wave_sum(dot);
        for (int s = lane; s < S; s += 64) {
            const float v = probs[(long)bh * S + s] * (t[s] - dot) * scale;
            t[s] = v;
            dsc_out[(long)bh * S + s] = v;
        }
    }
    __syncthreads();
    for (int e = tid * 4; e < E; e += 1024) {            // d qk = sum_s d score_s mem_s
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 16
        for (int s = 0; s < S; ++s) {
            const float w = t[s];
            const float4 m4 = *reinterpret_cast<const float4*>(mem + ((long)s * B + b) * E + e);
            a.x += w * m4.x; a.y += w * m4.y; a.z += w * m4.z; a.w += w * m4.w;
        }
        *reinterpret_cast<float4*>(dqk_out + (long)bh * E + e) = a;
    }
}
SLNLP_ZKERNEL(xmem_bwd_kernel, 256, xmem_bwd_body)

constexpr int XMEM_MAX_HEADS = 64;      // heads per sequence the d-memory kernel keeps in LDS (xmem_check rejects more)

// ------------------------------------------------------------------------------------ backward, per memory row ----
// d mem[s*B+b, :] (+)= sum_h ( p_s(b,h) d mbar(b,h,:) + d score_s(b,h) qk(b,h,:) ),  p after dropout; heads in fixed order.
// A workgroup takes DMEM_ROWS rows s of ONE sequence b: the 2 H head vectors d mbar(b,h,:), qk(b,h,:) -- 32 KB at cfg2 -- are what it
// streams, so one row per workgroup read them S times per sequence (77 MB through the L2 per launch, 1.2 GB for 15 fits in
// lockstep); eight rows share one pass.  A thread holds 4 columns of its rows; the sum over h runs in head order per row as before.
constexpr int DMEM_ROWS = 8;
__device__ __forceinline__ void xmem_dmem_body(const float* __restrict__ probs, const float* __restrict__ dsc, const float* __restrict__ dmbar,
                                               const float* __restrict__ qk, int B, int S, int H, int E, float* __restrict__ dmem,
                                               int accumulate, float drop_p, unsigned drop_thr, int drop_site,
                                               const unsigned long long* __restrict__ rng, const float* __restrict__ dcp,
                                               float* __restrict__ dbv) {
    __shared__ float ph[DMEM_ROWS][XMEM_MAX_HEADS], dh_[DMEM_ROWS][XMEM_MAX_HEADS];   // p_s (after dropout), d score_s per (row, head)
    const int chunks = (S + DMEM_ROWS - 1) / DMEM_ROWS, tid = threadIdx.x;
    if ((int)blockIdx.x >= chunks * B) {                 // the launch's last ceil(E / 256) workgroups: d bv = column sums of dcp, in row
        const int c = ((int)blockIdx.x - chunks * B) * 256 + tid;      // order (was a launch of its own: 5 us of dispatch for 100 KB)
        if (c < E) {
            float a = 0.f;
#pragma unroll 8
            for (int r = 0; r < B; ++r) a += dcp[(long)r * E + c];
            dbv[c] = a;
        }
        return;
    }
    const int b = blockIdx.x / chunks, s0 = (blockIdx.x % chunks) * DMEM_ROWS;
    for (int i = tid; i < DMEM_ROWS * H; i += 256) {
        const int r = i / H, h = i - r * H, s_ = s0 + r;
        const long bh = (long)b * H + h;
        float p = 0.f, d = 0.f;
        if (s_ < S) {
            p = probs[bh * S + s_];
            if (drop_p > 0.f) p = dropout_keep(rng, drop_site, (unsigned)bh, (unsigned)s_, drop_thr) ? p / (1.f - drop_p) : 0.f;
            d = dsc[bh * S + s_];
        }
        ph[r][h] = p;
        dh_[r][h] = d;
    }
    __syncthreads();
    // thread -> (row set, 4 columns): with E / 4 < 256 column chunks the spare threads take other rows of the workgroup
    const int per = E >> 2, sets = per >= 256 ? 1 : 256 / per;
    const int set = sets == 1 ? 0 : tid / per, c0 = sets == 1 ? tid : tid - set * per;
    if (set >= sets) return;
    for (int e = c0 * 4; e < E; e += (sets == 1 ? 1024 : E)) {
        float4 a[DMEM_ROWS];
#pragma unroll
        for (int i = 0; i < DMEM_ROWS; ++i) {
            const int r = set + i * sets, s_ = s0 + r;
            a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (accumulate && r < DMEM_ROWS && s_ < S) a[i] = *reinterpret_cast<const float4*>(dmem + ((long)s_ * B + b) * E + e);
        }
#pragma unroll 8                                         // (all eight heads' vectors requested together: 16 loads in flight, not 4 x 4 round trips)
        for (int h = 0; h < H; ++h) {
            const long bh = (long)b * H + h;
            const float4 g = *reinterpret_cast<const float4*>(dmbar + bh * E + e), k = *reinterpret_cast<const float4*>(qk + bh * E + e);
#pragma unroll
            for (int i = 0; i < DMEM_ROWS; ++i) {
                const int r = set + i * sets;
                if (r < DMEM_ROWS) {
                    const float p = ph[r][h], d = dh_[r][h];
                    a[i].x += p * g.x + d * k.x; a[i].y += p * g.y + d * k.y; a[i].z += p * g.z + d * k.z; a[i].w += p * g.w + d * k.w;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < DMEM_ROWS; ++i) {
            const int r = set + i * sets, s_ = s0 + r;
            if (r < DMEM_ROWS && s_ < S) *reinterpret_cast<float4*>(dmem + ((long)s_ * B + b) * E + e) = a[i];
        }
    }
}
SLNLP_ZKERNEL(xmem_dmem_kernel, 256, xmem_dmem_body)

// out[c] = sum_r in[r, c] in row order (d bv)
__device__ __forceinline__ void xmem_colsum_body(const float* __restrict__ in, int R, int C, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f;
#pragma unroll 8
    for (int r = 0; r < R; ++r) a += in[(long)r * C + c];
    out[c] = a;
}
SLNLP_ZKERNEL(xmem_colsum_kernel, 256, xmem_colsum_body)

// ------------------------------------------------------------------------------- HP heads per workgroup ----
// The kernels above stream the S memory rows of their sequence twice per (sequence, head) workgroup, and the H heads of a sequence
// stream the SAME rows: 2 x S x E x 4 bytes through the L2 per head -- 78 MB per launch at cfg2, 1.2 GB for 15 fits in lockstep,
// where that traffic is the kernels' time.  Here a workgroup serves HP heads of one sequence from ONE pass over the rows: a row
// chunk is loaded once and used for the HP dot products (and later the HP weighted sums).  Per output element the arithmetic
// and its order are those of the kernels above (same lane partition of every dot product, same sequential sums over s): the
// results are bit-identical, HP is a scheduling choice (xmem_heads_per_wg).
__device__ __forceinline__ float dot4(const float4& q4, const float4& m4) { return q4.x * m4.x + q4.y * m4.y + q4.z * m4.z + q4.w * m4.w; }

// dots of the HP LDS vectors v[hp][E] (stride vs) with memory rows s = wave, wave + 4, ... : RU rows per trip, each row chunk loaded
// once for all HP heads.  The column loop is the OUTER one and a trip's RU row chunks are requested together: with the rows outside
// (rounds 2-4) every chunk was one load followed by its own s_waitcnt -- 24 dependent round trips per wave for a 48-frame sequence,
// most of the kernel's 13.9 us (ISA of xmem_fwd_h2_kernel).  Per (row, head) the sum still runs over the columns in increasing order
// with the same dot4 expression: same bits.
template <int HP, class F>
__device__ __forceinline__ void row_dots_multi(const float* __restrict__ v, int vs, const float* __restrict__ mem, int B, int b, int S, int E,
                                               int lane, int wave, F&& put) {
    constexpr int RU = 6;
    for (int s0 = wave; s0 < S; s0 += 4 * RU) {
        float a[RU][HP];
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int hp = 0; hp < HP; ++hp) a[u][hp] = 0.f;
        for (int e = lane * 4; e < E; e += 256) {
            float4 m4[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int s = s0 + 4 * u < S ? s0 + 4 * u : s0;             // (a row past the end: a repeat of the first, never reported)
                m4[u] = *reinterpret_cast<const float4*>(mem + ((long)s * B + b) * E + e);
            }
#pragma unroll
            for (int hp = 0; hp < HP; ++hp) {
                const float4 q4 = *reinterpret_cast<const float4*>(v + (long)hp * vs + e);
#pragma unroll
                for (int u = 0; u < RU; ++u) a[u][hp] += dot4(q4, m4[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int hp = 0; hp < HP; ++hp) {
                const float r = wave_sum(a[u][hp]);
                if (lane == 0 && s0 + 4 * u < S) put(hp, s0 + 4 * u, r);
            }
    }
}

// LDS per head: v[E], sc[S padded to 4], 4 scalar slots
__device__ __forceinline__ int xm_head_floats(int S, int E) { return E + ((S + 3) & ~3) + 4; }

template <int HP>
__device__ __forceinline__ void xmem_fwd_multi_body(const float* __restrict__ qk, const float* __restrict__ mem, const float* __restrict__ bv,
                                                    int B, int S, int H, int dh, float* __restrict__ mbar_out, float* __restrict__ psum_out,
                                                    float* __restrict__ probs, float* __restrict__ ctx0, float drop_p, unsigned drop_thr,
                                                    int drop_site, const unsigned long long* __restrict__ rng) {
    extern __shared__ __attribute__((aligned(16))) float xm_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = H * dh, groups = H / HP;
    const int wg = xcd_local(blockIdx.x, gridDim.x), b = wg / groups, h0 = (wg % groups) * HP;
    const int hf = xm_head_floats(S, E), sp = (S + 3) & ~3;
    float* heads = xm_lds;
    for (int i = tid * 4; i < HP * E; i += 1024) {
        const int hp = i / E, e = i - hp * E;
        *reinterpret_cast<float4*>(heads + (long)hp * hf + e) = *reinterpret_cast<const float4*>(qk + ((long)b * H + h0 + hp) * E + e);
    }
    __syncthreads();
    const float scale = rsqrtf((float)dh), inv_keep = 1.f / (1.f - drop_p);
    row_dots_multi<HP>(heads, hf, mem, B, b, S, E, lane, wave, [&](int hp, int s_, float r) { heads[(long)hp * hf + E + s_] = r * scale; });
    __syncthreads();
    for (int hp = wave; hp < HP; hp += 4) {              // softmax + dropout of one head per wave (xmem_fwd_body's wave-0 block)
        float* sc = heads + (long)hp * hf + E;
        const long bh = (long)b * H + h0 + hp;
        float m = -INFINITY;
        for (int s_ = lane; s_ < S; s_ += 64) m = fmaxf(m, sc[s_]);
        m = wave_max(m);
        float sum = 0.f;
        for (int s_ = lane; s_ < S; s_ += 64) sum += expf(sc[s_] - m);
        sum = wave_sum(sum);
        float tot = 0.f;
        for (int s_ = lane; s_ < S; s_ += 64) {
            float p = expf(sc[s_] - m) / sum;
            probs[bh * S + s_] = p;
            if (drop_p > 0.f) p = dropout_keep(rng, drop_site, (unsigned)bh, (unsigned)s_, drop_thr) ? p * inv_keep : 0.f;
            sc[s_] = p;
            tot += p;
        }
        tot = wave_sum(tot);
        if (lane == 0) {
            sc[sp] = tot;
            psum_out[bh] = tot;
        }
    }
    __syncthreads();
    for (int e = tid * 4; e < E; e += 1024) {            // mbar = sum_s p_s mem_s, every row chunk loaded once for the HP heads
        float4 a[HP];
#pragma unroll
        for (int hp = 0; hp < HP; ++hp) a[hp] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int s_ = 0; s_ < S; ++s_) {
            const float4 m4 = *reinterpret_cast<const float4*>(mem + ((long)s_ * B + b) * E + e);
#pragma unroll
            for (int hp = 0; hp < HP; ++hp) {
                const float p = heads[(long)hp * hf + E + s_];
                a[hp].x += p * m4.x; a[hp].y += p * m4.y; a[hp].z += p * m4.z; a[hp].w += p * m4.w;
            }
        }
#pragma unroll
        for (int hp = 0; hp < HP; ++hp) *reinterpret_cast<float4*>(mbar_out + ((long)b * H + h0 + hp) * E + e) = a[hp];
    }
    for (int i = tid; i < HP * dh; i += 256) {
        const int hp = i / dh, j = i - hp * dh, h = h0 + hp;
        ctx0[(long)b * E + h * dh + j] = bv[h * dh + j] * heads[(long)hp * hf + E + sp];
    }
}
__device__ __forceinline__ void xmem_fwd_h2_body(const float* qk, const float* mem, const float* bv, int B, int S, int H, int dh, float* mbar_out,
                                                 float* psum_out, float* probs, float* ctx0, float drop_p, unsigned drop_thr, int drop_site,
                                                 const unsigned long long* rng) {
    xmem_fwd_multi_body<2>(qk, mem, bv, B, S, H, dh, mbar_out, psum_out, probs, ctx0, drop_p, drop_thr, drop_site, rng);
}
__device__ __forceinline__ void xmem_fwd_h4_body(const float* qk, const float* mem, const float* bv, int B, int S, int H, int dh, float* mbar_out,
                                                 float* psum_out, float* probs, float* ctx0, float drop_p, unsigned drop_thr, int drop_site,
                                                 const unsigned long long* rng) {
    xmem_fwd_multi_body<4>(qk, mem, bv, B, S, H, dh, mbar_out, psum_out, probs, ctx0, drop_p, drop_thr, drop_site, rng);
}
SLNLP_ZKERNEL(xmem_fwd_h2_kernel, 256, xmem_fwd_h2_body)
SLNLP_ZKERNEL(xmem_fwd_h4_kernel, 256, xmem_fwd_h4_body)

template <int HP>
__device__ __forceinline__ void xmem_bwd_multi_body(const float* __restrict__ mem, const float* __restrict__ bv, const float* __restrict__ probs,
                                                    const float* __restrict__ psum, const float* __restrict__ dmbar,
                                                    const float* __restrict__ dctx, int B, int S, int H, int dh, float* __restrict__ dsc_out,
                                                    float* __restrict__ dqk_out, float* __restrict__ dcp, float drop_p, unsigned drop_thr,
                                                    int drop_site, const unsigned long long* __restrict__ rng) {
    extern __shared__ __attribute__((aligned(16))) float xm_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = H * dh, groups = H / HP;
    const int wg = xcd_local(blockIdx.x, gridDim.x), b = wg / groups, h0 = (wg % groups) * HP;
    const int hf = xm_head_floats(S, E), sp = (S + 3) & ~3;
    float* heads = xm_lds;
    for (int i = tid * 4; i < HP * E; i += 1024) {
        const int hp = i / E, e = i - hp * E;
        *reinterpret_cast<float4*>(heads + (long)hp * hf + e) = *reinterpret_cast<const float4*>(dmbar + ((long)b * H + h0 + hp) * E + e);
    }
    for (int hp = wave; hp < HP; hp += 4) {              // c = d ctx_h . bv_h; d ctx * sum_s p_s (xmem_bwd_body's wave-0 block)
        const int h = h0 + hp;
        const float ps = psum[(long)b * H + h];
        float c = 0.f;
        for (int j = lane; j < dh; j += 64) {
            const float g = dctx[(long)b * E + h * dh + j];
            c += g * bv[h * dh + j];
            dcp[(long)b * E + h * dh + j] = g * ps;
        }
        c = wave_sum(c);
        if (lane == 0) heads[(long)hp * hf + E + sp] = c;
    }
    __syncthreads();
    row_dots_multi<HP>(heads, hf, mem, B, b, S, E, lane, wave,
                       [&](int hp, int s_, float r) { heads[(long)hp * hf + E + s_] = r + heads[(long)hp * hf + E + sp]; });
    __syncthreads();
    const float scale = rsqrtf((float)dh), inv_keep = 1.f / (1.f - drop_p);
    for (int hp = wave; hp < HP; hp += 4) {              // through the dropout mask and the softmax
        float* t = heads + (long)hp * hf + E;
        const long bh = (long)b * H + h0 + hp;
        float dot = 0.f;
        for (int s_ = lane; s_ < S; s_ += 64) {
            float dp = t[s_];
            if (drop_p > 0.f) dp = dropout_keep(rng, drop_site, (unsigned)bh, (unsigned)s_, drop_thr) ? dp * inv_keep : 0.f;
            t[s_] = dp;
            dot += probs[bh * S + s_] * dp;
        }
        dot = wave_sum(dot);
        for (int s_ = lane; s_ < S; s_ += 64) {
            const float v = probs[bh * S + s_] * (t[s_] - dot) * scale;
            t[s_] = v;
            dsc_out[bh * S + s_] = v;
        }
    }
    __syncthreads();
    for (int e = tid * 4; e < E; e += 1024) {            // d qk = sum_s d score_s mem_s
        float4 a[HP];
#pragma unroll
        for (int hp = 0; hp < HP; ++hp) a[hp] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int s_ = 0; s_ < S; ++s_) {
            const float4 m4 = *reinterpret_cast<const float4*>(mem + ((long)s_ * B + b) * E + e);
#pragma unroll
            for (int hp = 0; hp < HP; ++hp) {
                const float w = heads[(long)hp * hf + E + s_];
                a[hp].x += w * m4.x; a[hp].y += w * m4.y; a[hp].z += w * m4.z; a[hp].w += w * m4.w;
            }
        }
#pragma unroll
        for (int hp = 0; hp < HP; ++hp) *reinterpret_cast<float4*>(dqk_out + ((long)b * H + h0 + hp) * E + e) = a[hp];
    }
}
__device__ __forceinline__ void xmem_bwd_h2_body(const float* mem, const float* bv, const float* probs, const float* psum, const float* dmbar,
                                                 const float* dctx, int B, int S, int H, int dh, float* dsc_out, float* dqk_out, float* dcp,
                                                 float drop_p, unsigned drop_thr, int drop_site, const unsigned long long* rng) {
    xmem_bwd_multi_body<2>(mem, bv, probs, psum, dmbar, dctx, B, S, H, dh, dsc_out, dqk_out, dcp, drop_p, drop_thr, drop_site, rng);
}
__device__ __forceinline__ void xmem_bwd_h4_body(const float* mem, const float* bv, const float* probs, const float* psum, const float* dmbar,
                                                 const float* dctx, int B, int S, int H, int dh, float* dsc_out, float* dqk_out, float* dcp,
                                                 float drop_p, unsigned drop_thr, int drop_site, const unsigned long long* rng) {
    xmem_bwd_multi_body<4>(mem, bv, probs, psum, dmbar, dctx, B, S, H, dh, dsc_out, dqk_out, dcp, drop_p, drop_thr, drop_site, rng);
}
SLNLP_ZKERNEL(xmem_bwd_h2_kernel, 256, xmem_bwd_h2_body)
SLNLP_ZKERNEL(xmem_bwd_h4_kernel, 256, xmem_bwd_h4_body)

// ------------------------------------------------------------------------------------------------ launchers ----
static int xmem_init() {                  // dynamic LDS beyond 64 KiB is only needed for S in the thousands; raise once per device
    static DeviceOnce once;
    return once.run([]() -> int {
        const int bytes = (1024 + 5000 + 8) * (int)sizeof(float);
        if (hipFuncSetAttribute((const void*)xmem_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
            hipFuncSetAttribute((const void*)xmem_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
            set_error("attention_mem: cannot raise the dynamic LDS limit: %s", hipGetErrorString(hipGetLastError()));
            return SLNLP_ERR_LAUNCH;
        }
        return 0;
    });
}

// heads per workgroup (1: the kernels above).  env SLNLP_XMEM_HEADS = 1 / 2 / 4 forces it (A / B runs).
static int xmem_heads_per_wg(int B, int S, int H, int dh) {
    static const int forced = [] { const char* e = getenv("SLNLP_XMEM_HEADS"); return e ? atoi(e) : 0; }();
    const int want = forced ? forced : 2;
    for (int hp = want; hp >= 2; hp >>= 1)
        if (H % hp == 0 && (size_t)hp * (H * dh + ((S + 3) & ~3) + 4) * sizeof(float) <= 65536) return hp;
    return 1;
}

static int xmem_check(const char* who, int B, int S, int H, int dh) {
    SLNLP_CHECK_ARG(B > 0 && S > 0 && S <= 5000 && H > 0, "%s: bad B=%d S=%d H=%d", who, B, S, H);
    SLNLP_CHECK_ARG(dh > 0 && dh % 4 == 0 && dh <= 256 && H * dh <= 1024, "%s: head_dim %d / model dim %d unsupported", who, dh, H * dh);
    SLNLP_CHECK_ARG(H <= XMEM_MAX_HEADS, "%s: %d heads (the per-head LDS tables of the d-memory kernel hold %d)", who, H, XMEM_MAX_HEADS);
    return 0;
}

// scores / softmax / dropout / mbar of every (sequence, head) from qk = Wk_h^T q_h; ctx0 = bv * sum_s p_s
int xmem_fwd(const float* qk, const float* mem, const float* bv, int B, int S, int H, int dh, float* mbar, float* psum, float* probs,
             float* ctx0, float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st) {
    SLNLP_TRY(xmem_check("xmem_fwd", B, S, H, dh));
    SLNLP_CHECK_ARG(qk && mem && bv && mbar && psum && probs && ctx0, "xmem_fwd: null pointer");
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "xmem_fwd: bad dropout args");
    if (const int hp = xmem_heads_per_wg(B, S, H, dh); hp > 1) {
        const size_t l = (size_t)hp * (H * dh + ((S + 3) & ~3) + 4) * sizeof(float);
        if (hp == 4) return zlaunch(xmem_fwd_h4_kernel, dim3(B * (H / 4)), 256, l, st, "xmem_fwd", qk, mem, bv, B, S, H, dh, mbar, psum, probs, ctx0,
                                    drop_p, dropout_threshold(drop_p), drop_site, rng);
        return zlaunch(xmem_fwd_h2_kernel, dim3(B * (H / 2)), 256, l, st, "xmem_fwd", qk, mem, bv, B, S, H, dh, mbar, psum, probs, ctx0, drop_p,
                       dropout_threshold(drop_p), drop_site, rng);
    }
    const size_t lds = (size_t)(H * dh + S + 8) * sizeof(float);
    if (lds > 65536) SLNLP_TRY(xmem_init());
    return zlaunch(xmem_fwd_kernel, dim3(B * H), 256, lds, st, "xmem_fwd", qk, mem, bv, B, S, H, dh, mbar, psum, probs, ctx0, drop_p,
                   dropout_threshold(drop_p), drop_site, rng);
}

// from d mbar = Wv_h^T d ctx_h: d scores, d qk, d ctx * sum_s p_s; then d memory (+= over the decoder layers) and d bv
int xmem_bwd(const float* mem, const float* bv, const float* probs, const float* psum, const float* qk, const float* dmbar,
             const float* dctx, int B, int S, int H, int dh, float* dsc, float* dqk, float* dcp, float* dbv, float* dmem, int accumulate,
             float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st) {
    SLNLP_TRY(xmem_check("xmem_bwd", B, S, H, dh));
    SLNLP_CHECK_ARG(mem && bv && probs && psum && qk && dmbar && dctx && dsc && dqk && dcp && dbv && dmem, "xmem_bwd: null pointer");
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "xmem_bwd: bad dropout args");
    const int E = H * dh;
    if (const int hp = xmem_heads_per_wg(B, S, H, dh); hp > 1) {
        const size_t l = (size_t)hp * (E + ((S + 3) & ~3) + 4) * sizeof(float);
        if (hp == 4) SLNLP_TRY(zlaunch(xmem_bwd_h4_kernel, dim3(B * (H / 4)), 256, l, st, "xmem_bwd", mem, bv, probs, psum, dmbar, dctx, B, S, H, dh,
                                       dsc, dqk, dcp, drop_p, dropout_threshold(drop_p), drop_site, rng));
        else SLNLP_TRY(zlaunch(xmem_bwd_h2_kernel, dim3(B * (H / 2)), 256, l, st, "xmem_bwd", mem, bv, probs, psum, dmbar, dctx, B, S, H, dh, dsc,
                               dqk, dcp, drop_p, dropout_threshold(drop_p), drop_site, rng));
    } else {
        const size_t lds = (size_t)(E + S + 8) * sizeof(float);
        if (lds > 65536) SLNLP_TRY(xmem_init());
        SLNLP_TRY(zlaunch(xmem_bwd_kernel, dim3(B * H), 256, lds, st, "xmem_bwd", mem, bv, probs, psum, dmbar, dctx, B, S, H, dh, dsc, dqk, dcp,
                          drop_p, dropout_threshold(drop_p), drop_site, rng));
    }
    return zlaunch(xmem_dmem_kernel, dim3(ceil_div(S, DMEM_ROWS) * B + ceil_div(E, 256)), 256, 0, st, "xmem_dmem", probs, (const float*)dsc, dmbar, qk, B, S, H, E, dmem,
                   accumulate, drop_p, dropout_threshold(drop_p), drop_site, rng, (const float*)dcp, dbv);
}

}  // namespace slnlp
