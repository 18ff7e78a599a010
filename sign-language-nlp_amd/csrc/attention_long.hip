// attention_long.hip -- attention cores for sequences longer than one 64-key tile (S > 64).
//
// The reference has no length limit short of its 5000-row positional table
// (/root/reference/model/component/positional_encoding.py:23); ASL-Phono batches are padded to the corpus-wide
// maximum length (dataset/asl_dataset.py:157-169), so a corpus with one long clip moves EVERY batch past 64 frames.
// The MFMA kernels of attention.hip hold a whole head in one 64x64 tile; these kernels take over above that:
// same arguments, same probs / dropout-site conventions (probs [B,H,S,S] pre-dropout, dropout element
// (row (b*H+h)*S + i, column j)), same torch semantics (a fully masked row is NaN), plain fp32 FMA arithmetic.
//
// Shape: ONE WAVE PER ATTENTION ROW -- (b, h, i) for self-attention, (b, h) for the decoder's single-query
// cross-attention.  A score is a head_dim-long dot product reduced across the wave; the row's scores live in LDS
// (4 rows per workgroup x S floats, dynamic), the softmax is two strided passes over them, and the context is
// accumulated with the head dim across lanes so every global access of K / V / Q rows is coalesced.  The backward is
// split the usual way: a pass over rows (dP, softmax backward, dQ; dS kept in a scratch buffer the size of probs) and
// a pass over key columns (dK, dV).  No tiling of keys through LDS and no MFMA: at these lengths the step is
// dominated by the S*B-row GEMMs anyway, and this path is about being correct for any S, not about the roofline.
#include "common.hpp"
#include "launch.hpp"

namespace slnlp {

constexpr int LONG_DH_MAX = 256;   // head dim <= 256: 4 values per lane

__device__ __forceinline__ void load_head(const float* __restrict__ p, int dh, int lane, float (&v)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (lane + 64 * u < dh) ? p[lane + 64 * u] : 0.f;
}
__device__ __forceinline__ float dot_head(const float (&a)[4], const float* __restrict__ p, int dh, int lane) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < dh) s += a[u] * p[lane + 64 * u];
    return wave_sum(s);
}

// ------------------------------------------------------------------ self-attention, forward
__device__ __forceinline__ void attn_self_fwd_long_body(const float* __restrict__ qkv, const long* __restrict__ ids, long ld_ids,
                                                        long pad_idx, int causal, int B, int S, int H, int dh,
                                                        float* __restrict__ ctx, float* __restrict__ probs, float drop_p,
                                                        unsigned drop_thr, int drop_site,
                                                        const unsigned long long* __restrict__ rng, PlaneOut po) {
    extern __shared__ float lds_rows[];                       // [4 waves][S]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;             // (b*H + h)*S + i
    if (row >= (long)B * H * S) return;
    float* sc = lds_rows + (long)wave * S;
    const int i = (int)(row % S), bh = (int)(row / S), h = bh % H, b = bh / H, E = H * dh;
    const long ld = 3L * E;
    float q[4];
    load_head(qkv + ((long)i * B + b) * ld + h * dh, dh, lane, q);
    const float scale = rsqrtf((float)dh), inv_keep = 1.f / (1.f - drop_p);
    for (int j = 0; j < S; ++j) {                             // scores: one wave-wide dot product per key
        const float s = dot_head(q, qkv + ((long)j * B + b) * ld + E + h * dh, dh, lane) * scale;
        if (lane == 0) sc[j] = s;
    }
    __builtin_amdgcn_wave_barrier();
    float m = -INFINITY;
    for (int j = lane; j < S; j += 64) {
        const bool blocked = (causal && j > i) || (ids && ids[(long)b * ld_ids + j] == pad_idx);
        const float v = blocked ? -INFINITY : sc[j];
        sc[j] = v;
        m = fmaxf(m, v);
    }
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < S; j += 64) {
        const float e = expf(sc[j] - m);                      // all-masked row: exp(-inf - -inf) = NaN, as torch
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    for (int j = lane; j < S; j += 64) {
        float p = sc[j] / sum;
        probs[row * S + j] = p;
        if (drop_p > 0.f) p = dropout_keep(rng, drop_site, (unsigned)row, (unsigned)j, drop_thr) ? p * inv_keep : 0.f;
        sc[j] = p;
    }
    __builtin_amdgcn_wave_barrier();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int jend = causal ? i + 1 : S;                      // blocked keys have weight exactly 0 (or the row is NaN through `sum`)
    for (int j = 0; j < S; ++j) {
        const float p = sc[j];
        if (j >= jend && p == 0.f) continue;
        const float* v = qkv + ((long)j * B + b) * ld + 2 * E + h * dh;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lane + 64 * u < dh) acc[u] += p * v[lane + 64 * u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < dh) {
            const long at = ((long)i * B + b) * E + h * dh + lane + 64 * u;
            ctx[at] = acc[u];
            store_planes1(po, at, acc[u]);
        }
}
SLNLP_ZKERNEL(attn_self_fwd_long_kernel, 256, attn_self_fwd_long_body)

// ------------------------------------------------------------------ self-attention, backward over rows
// dP_ij = keep_ij/(1-p) * dO_i . V_j;  dS_ij = P_ij (dP_ij - sum_j' P_ij' dP_ij');  ds[row, j] = dS_ij / sqrt(dh);
// dQ_i = sum_j ds_ij K_j
__device__ __forceinline__ void attn_self_bwd_long_rows_body(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                             const float* __restrict__ dctx, int B, int S, int H, int dh,
                                                             float* __restrict__ dqkv, float* __restrict__ ds, float drop_p,
                                                             unsigned drop_thr, int drop_site,
                                                             const unsigned long long* __restrict__ rng, PlaneOut po) {
    extern __shared__ float lds_rows[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= (long)B * H * S) return;
    float* sc = lds_rows + (long)wave * S;
    const int i = (int)(row % S), bh = (int)(row / S), h = bh % H, b = bh / H, E = H * dh;
    const long ld = 3L * E;
    float go[4];
    load_head(dctx + ((long)i * B + b) * E + h * dh, dh, lane, go);
    const float scale = rsqrtf((float)dh), inv_keep = 1.f / (1.f - drop_p);
    for (int j = 0; j < S; ++j) {
        const float d = dot_head(go, qkv + ((long)j * B + b) * ld + 2 * E + h * dh, dh, lane);
        if (lane == 0) sc[j] = d;
    }
    __builtin_amdgcn_wave_barrier();
    float dot = 0.f;
    for (int j = lane; j < S; j += 64) {
        float dp = sc[j];
        if (drop_p > 0.f) dp = dropout_keep(rng, drop_site, (unsigned)row, (unsigned)j, drop_thr) ? dp * inv_keep : 0.f;
        sc[j] = dp;
        dot += probs[row * S + j] * dp;
    }
    dot = wave_sum(dot);
    for (int j = lane; j < S; j += 64) {
        const float v = probs[row * S + j] * (sc[j] - dot) * scale;
        sc[j] = v;
        ds[row * S + j] = v;
    }
    __builtin_amdgcn_wave_barrier();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < S; ++j) {
        const float w = sc[j];
        if (w == 0.f) continue;                               // blocked keys (probability exactly 0)
        const float* k = qkv + ((long)j * B + b) * ld + E + h * dh;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lane + 64 * u < dh) acc[u] += w * k[lane + 64 * u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < dh) {
            const long at = ((long)i * B + b) * ld + h * dh + lane + 64 * u;
            dqkv[at] = acc[u];
            store_planes1(po, at, acc[u]);
        }
}
SLNLP_ZKERNEL(attn_self_bwd_long_rows_kernel, 256, attn_self_bwd_long_rows_body)

// ------------------------------------------------------------------ self-attention, backward over key columns
// dK_j = sum_i ds_ij Q_i;  dV_j = sum_i (keep_ij/(1-p) P_ij) dO_i
__device__ __forceinline__ void attn_self_bwd_long_cols_body(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                             const float* __restrict__ dctx, const float* __restrict__ ds, int B,
                                                             int S, int H, int dh, float* __restrict__ dqkv, float drop_p,
                                                             unsigned drop_thr, int drop_site,
                                                             const unsigned long long* __restrict__ rng, PlaneOut po) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long col = (long)blockIdx.x * 4 + wave;             // (b*H + h)*S + j
    if (col >= (long)B * H * S) return;
    const int j = (int)(col % S), bh = (int)(col / S), h = bh % H, b = bh / H, E = H * dh;
    const long ld = 3L * E, prow0 = (long)bh * S;
    const float inv_keep = 1.f / (1.f - drop_p);
    float dk[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < S; ++i) {
        const float w = ds[(prow0 + i) * S + j];
        float p = probs[(prow0 + i) * S + j];
        if (drop_p > 0.f) p = dropout_keep(rng, drop_site, (unsigned)(prow0 + i), (unsigned)j, drop_thr) ? p * inv_keep : 0.f;
        if (w == 0.f && p == 0.f) continue;                   // wave-uniform: blocked pair
        const float* q = qkv + ((long)i * B + b) * ld + h * dh;
        const float* g = dctx + ((long)i * B + b) * E + h * dh;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lane + 64 * u < dh) {
                dk[u] += w * q[lane + 64 * u];
                dv[u] += p * g[lane + 64 * u];
            }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < dh) {
            const long at = ((long)j * B + b) * ld + h * dh + lane + 64 * u;
            dqkv[at + E] = dk[u];
            dqkv[at + 2 * E] = dv[u];
            store_planes1(po, at + E, dk[u]);
            store_planes1(po, at + 2 * E, dv[u]);
        }
}
SLNLP_ZKERNEL(attn_self_bwd_long_cols_kernel, 256, attn_self_bwd_long_cols_body)

// ------------------------------------------------------------------ cross-attention (one query per sequence, no masks)
// q [B, E]; kv [S*B rows, K | V] with row stride ld_kv; probs [B*H, S]; dropout element (row b*H + h, column s)
__device__ __forceinline__ void attn_cross_fwd_long_body(const float* __restrict__ q, const float* __restrict__ kv, long ld_kv, int B,
                                                         int S, int H, int dh, float* __restrict__ ctx, float* __restrict__ probs,
                                                         float drop_p, unsigned drop_thr, int drop_site,
                                                         const unsigned long long* __restrict__ rng) {
    extern __shared__ float lds_rows[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= B * H) return;
    float* sc = lds_rows + (long)wave * S;
    const int h = bh % H, b = bh / H, E = H * dh;
    float qv[4];
    load_head(q + (long)b * E + h * dh, dh, lane, qv);
    const float scale = rsqrtf((float)dh), inv_keep = 1.f / (1.f - drop_p);
    for (int s = 0; s < S; ++s) {
        const float v = dot_head(qv, kv + ((long)s * B + b) * ld_kv + h * dh, dh, lane) * scale;
        if (lane == 0) sc[s] = v;
    }
    __builtin_amdgcn_wave_barrier();
    float m = -INFINITY;
    for (int s = lane; s < S; s += 64) m = fmaxf(m, sc[s]);
    m = wave_max(m);
    float sum = 0.f;
    for (int s = lane; s < S; s += 64) {
        const float e = expf(sc[s] - m);
        sc[s] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    for (int s = lane; s < S; s += 64) {
        float p = sc[s] / sum;
        probs[(long)bh * S + s] = p;
        if (drop_p > 0.f) p = dropout_keep(rng, drop_site, (unsigned)bh, (unsigned)s, drop_thr) ? p * inv_keep : 0.f;
        sc[s] = p;
    }
    __builtin_amdgcn_wave_barrier();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        const float p = sc[s];
        const float* v = kv + ((long)s * B + b) * ld_kv + E + h * dh;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lane + 64 * u < dh) acc[u] += p * v[lane + 64 * u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < dh) ctx[(long)b * E + h * dh + lane + 64 * u] = acc[u];
}
SLNLP_ZKERNEL(attn_cross_fwd_long_kernel, 256, attn_cross_fwd_long_body)

__device__ __forceinline__ void attn_cross_bwd_long_body(const float* __restrict__ q, const float* __restrict__ kv, long ld_kv,
                                                         const float* __restrict__ probs, const float* __restrict__ dctx, int B,
                                                         int S, int H, int dh, float* __restrict__ dq, float* __restrict__ dkv,
                                                         long ld_dkv, float drop_p, unsigned drop_thr, int drop_site,
                                                         const unsigned long long* __restrict__ rng, PlaneOut po) {
    extern __shared__ float lds_rows[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= B * H) return;
    float* sc = lds_rows + (long)wave * S;
    const int h = bh % H, b = bh / H, E = H * dh;
    float qv[4], go[4];
    load_head(q + (long)b * E + h * dh, dh, lane, qv);
    load_head(dctx + (long)b * E + h * dh, dh, lane, go);
    const float scale = rsqrtf((float)dh), inv_keep = 1.f / (1.f - drop_p);
    for (int s = 0; s < S; ++s) {                             // dP_s = dO . V_s  (through the dropout mask)
        const float d = dot_head(go, kv + ((long)s * B + b) * ld_kv + E + h * dh, dh, lane);
        if (lane == 0) sc[s] = d;
    }
    __builtin_amdgcn_wave_barrier();
    float dot = 0.f;
    for (int s = lane; s < S; s += 64) {
        float dp = sc[s];
        if (drop_p > 0.f) dp = dropout_keep(rng, drop_site, (unsigned)bh, (unsigned)s, drop_thr) ? dp * inv_keep : 0.f;
        sc[s] = dp;
        dot += probs[(long)bh * S + s] * dp;
    }
    dot = wave_sum(dot);
    for (int s = lane; s < S; s += 64) sc[s] = probs[(long)bh * S + s] * (sc[s] - dot) * scale;     // dS_s / sqrt(dh)
    __builtin_amdgcn_wave_barrier();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
        const float w = sc[s];
        float p = probs[(long)bh * S + s];
        if (drop_p > 0.f) p = dropout_keep(rng, drop_site, (unsigned)bh, (unsigned)s, drop_thr) ? p * inv_keep : 0.f;
        const float* k = kv + ((long)s * B + b) * ld_kv + h * dh;
        const long at = ((long)s * B + b) * ld_dkv + h * dh;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lane + 64 * u < dh) {
                const int d = lane + 64 * u;
                acc[u] += w * k[d];
                const float gk = w * qv[u], gv = p * go[u];
                dkv[at + d] = gk;
                dkv[at + E + d] = gv;
                store_planes1(po, at + d, gk);
                store_planes1(po, at + E + d, gv);
            }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < dh) dq[(long)b * E + h * dh + lane + 64 * u] = acc[u];
}
SLNLP_ZKERNEL(attn_cross_bwd_long_kernel, 256, attn_cross_bwd_long_body)

// ------------------------------------------------------------------ launchers (called by attention.hip when S > 64)
// 4 rows x S floats of dynamic LDS: above 64 KiB (S > 4096) the kernels' limit must be raised, once per device
static int long_lds_init() {
    static DeviceOnce once;
    return once.run([]() -> int {
    const int bytes = 4 * 5000 * (int)sizeof(float);
    const bool ok = hipFuncSetAttribute((const void*)attn_self_fwd_long_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
                    hipFuncSetAttribute((const void*)attn_self_bwd_long_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
                    hipFuncSetAttribute((const void*)attn_cross_fwd_long_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
                    hipFuncSetAttribute((const void*)attn_cross_bwd_long_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    if (!ok) {
        set_error("attention_long: cannot raise the dynamic LDS limit: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    return 0;
    });
}

static int check_long(const char* who, int B, int S, int H, int dh) {
    SLNLP_CHECK_ARG(B > 0 && H > 0 && S > 0 && S <= 5000, "%s: S=%d outside 1..5000", who, S);
    SLNLP_CHECK_ARG(dh > 0 && dh <= LONG_DH_MAX, "%s: head_dim %d above %d", who, dh, LONG_DH_MAX);
    SLNLP_CHECK_ARG((long)B * H * S * S < (1L << 40), "%s: probs tensor too large", who);
    return S > 4096 ? long_lds_init() : 0;
}

size_t attn_long_scratch_bytes(int B, int S, int H) { return ((size_t)B * H * S * S * sizeof(float) + 255) & ~(size_t)255; }

int attn_self_fwd_long(const float* qkv, const int64_t* ids, int64_t ld_ids, int64_t pad_idx, int causal, int B, int S, int H,
                       int dh, float* ctx, float* probs, float drop_p, int drop_site, const unsigned long long* rng,
                       hipStream_t st, PlaneOut po) {
    SLNLP_TRY(check_long("attn_self_fwd", B, S, H, dh));
    const long rows = (long)B * H * S;
    return zlaunch(attn_self_fwd_long_kernel, dim3((unsigned)((rows + 3) / 4)), 256, 4 * (size_t)S * sizeof(float), st, "attn_self_fwd_long",
                   qkv, (const long*)ids, (long)ld_ids, (long)pad_idx, causal, B, S, H, dh, ctx, probs, drop_p,
                   dropout_threshold(drop_p), drop_site, rng, po);
}

int attn_self_bwd_long(const float* qkv, const float* probs, const float* dctx, int B, int S, int H, int dh, float* dqkv,
                       float* scratch, float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st, PlaneOut po) {
    SLNLP_TRY(check_long("attn_self_bwd", B, S, H, dh));
    SLNLP_CHECK_ARG(scratch, "attn_self_bwd: sequences longer than 64 need a scratch buffer of slnlp_attn_long_scratch_bytes(B, S, H)");
    const long rows = (long)B * H * S;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    SLNLP_TRY(zlaunch(attn_self_bwd_long_rows_kernel, dim3(grid), 256, 4 * (size_t)S * sizeof(float), st, "attn_self_bwd_long_rows",
                      qkv, probs, dctx, B, S, H, dh, dqkv, scratch, drop_p, dropout_threshold(drop_p), drop_site, rng, po));
    return zlaunch(attn_self_bwd_long_cols_kernel, dim3(grid), 256, 0, st, "attn_self_bwd_long_cols",
                   qkv, probs, dctx, (const float*)scratch, B, S, H, dh, dqkv, drop_p, dropout_threshold(drop_p), drop_site, rng, po);
}

int attn_cross_fwd_long(const float* q, const float* kv, int64_t ld_kv, int B, int S, int H, int dh, float* ctx, float* probs,
                        float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st) {
    SLNLP_TRY(check_long("attn_cross_fwd", B, S, H, dh));
    return zlaunch(attn_cross_fwd_long_kernel, dim3((unsigned)((B * H + 3) / 4)), 256, 4 * (size_t)S * sizeof(float), st, "attn_cross_fwd_long",
                   q, kv, (long)ld_kv, B, S, H, dh, ctx, probs, drop_p, dropout_threshold(drop_p), drop_site, rng);
}

int attn_cross_bwd_long(const float* q, const float* kv, int64_t ld_kv, const float* probs, const float* dctx, int B, int S, int H,
                        int dh, float* dq, float* dkv, int64_t ld_dkv, float drop_p, int drop_site, const unsigned long long* rng,
                        hipStream_t st, PlaneOut po) {
    SLNLP_TRY(check_long("attn_cross_bwd", B, S, H, dh));
    return zlaunch(attn_cross_bwd_long_kernel, dim3((unsigned)((B * H + 3) / 4)), 256, 4 * (size_t)S * sizeof(float), st, "attn_cross_bwd_long",
                   q, kv, (long)ld_kv, probs, dctx, B, S, H, dh, dq, dkv, (long)ld_dkv, drop_p, dropout_threshold(drop_p), drop_site, rng, po);
}

}  // namespace slnlp
