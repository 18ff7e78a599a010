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
// d mem[s*B+b, :] (+)= sum_h ( p_s(b,h) d mbar(b,h,:) + d score_s(b,h) qk(b,h,:) ),  p after dropout; heads in fixed order
__device__ __forceinline__ void xmem_dmem_body(const float* __restrict__ probs, const float* __restrict__ dsc, const float* __restrict__ dmbar,
                                               const float* __restrict__ qk, int B, int S, int H, int E, float* __restrict__ dmem,
                                               int accumulate, float drop_p, unsigned drop_thr, int drop_site,
                                               const unsigned long long* __restrict__ rng, const float* __restrict__ dcp,
                                               float* __restrict__ dbv) {
    __shared__ float ph[XMEM_MAX_HEADS], dh_[XMEM_MAX_HEADS];   // this row's p_s (after dropout) and d score_s per head: computed once
    if ((int)blockIdx.x >= S * B) {                      // the launch's last ceil(E / 256) workgroups: d bv = column sums of dcp, in row
        const int c = ((int)blockIdx.x - S * B) * 256 + threadIdx.x;   // order (was a launch of its own: 5 us of dispatch for 100 KB)
        if (c < E) {
            float a = 0.f;
#pragma unroll 8
            for (int r = 0; r < B; ++r) a += dcp[(long)r * E + c];
            dbv[c] = a;
        }
        return;
    }
    const int m = blockIdx.x, s = m / B, b = m % B;
    if (threadIdx.x < H) {
        const long bh = (long)b * H + threadIdx.x;
        float p = probs[bh * S + s];
        if (drop_p > 0.f) p = dropout_keep(rng, drop_site, (unsigned)bh, (unsigned)s, drop_thr) ? p / (1.f - drop_p) : 0.f;
        ph[threadIdx.x] = p;
        dh_[threadIdx.x] = dsc[bh * S + s];
    }
    __syncthreads();
    for (int e = threadIdx.x * 4; e < E; e += 1024) {
        float4 a = accumulate ? *reinterpret_cast<const float4*>(dmem + (long)m * E + e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int h = 0; h < H; ++h) {
            const long bh = (long)b * H + h;
            const float p = ph[h], d = dh_[h];
            const float4 g = *reinterpret_cast<const float4*>(dmbar + bh * E + e), k = *reinterpret_cast<const float4*>(qk + bh * E + e);
            a.x += p * g.x + d * k.x; a.y += p * g.y + d * k.y; a.z += p * g.z + d * k.z; a.w += p * g.w + d * k.w;
        }
        *reinterpret_cast<float4*>(dmem + (long)m * E + e) = a;
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
    const size_t lds = (size_t)(E + S + 8) * sizeof(float);
    if (lds > 65536) SLNLP_TRY(xmem_init());
    SLNLP_TRY(zlaunch(xmem_bwd_kernel, dim3(B * H), 256, lds, st, "xmem_bwd", mem, bv, probs, psum, dmbar, dctx, B, S, H, dh, dsc, dqk, dcp,
                      drop_p, dropout_threshold(drop_p), drop_site, rng));
    return zlaunch(xmem_dmem_kernel, dim3(S * B + ceil_div(E, 256)), 256, 0, st, "xmem_dmem", probs, (const float*)dsc, dmbar, qk, B, S, H, E, dmem,
                   accumulate, drop_p, dropout_threshold(drop_p), drop_site, rng, (const float*)dcp, dbv);
}

}  // namespace slnlp
