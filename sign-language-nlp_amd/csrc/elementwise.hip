// elementwise.hip -- HBM-bound pieces of the step: embedding gather + positional
// add, LayerNorm, log-softmax / NLL criterion, grad-norm clip + SGD-momentum.
// All fp32, 16-B vector accesses, one wave per row for the row-wise reductions.
#include "common.hpp"
#include "launch.hpp"

namespace slnlp {

// ===================================================================== embed
// /root/reference/model/transformer.py:106-109 + component/positional_encoding.py:48-49
__device__ __forceinline__ void embed_fwd_body(const long* __restrict__ ids, long ld_ids, int B, int S,
                                                        int E, int V, const float* __restrict__ table,
                                                        const float* __restrict__ pe, float* __restrict__ out,
                                                        float scale, float drop_p, unsigned drop_thr, int drop_site,
                                                        const unsigned long long* __restrict__ rng, long nan_idx,
                                                        PlaneOut po, unsigned char* __restrict__ keep_mask) {
    const int e4 = E >> 2;
    const long total = (long)B * S * e4;
    const float ik = 1.f / (1.f - drop_p);
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int m = (int)(idx / e4), c = (int)(idx % e4) * 4;
        const int s = m / B, b = m % B;
        const long id = ids[(long)b * ld_ids + s];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0 && id < V) v = *reinterpret_cast<const float4*>(table + id * E + c);
        const float4 p = pe ? *reinterpret_cast<const float4*>(pe + (long)s * E + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = v.x * scale + p.x; v.y = v.y * scale + p.y; v.z = v.z * scale + p.z; v.w = v.w * scale + p.w;
        if (drop_p > 0.f) {
            const bool k0 = dropout_keep(rng, drop_site, m, c + 0, drop_thr), k1 = dropout_keep(rng, drop_site, m, c + 1, drop_thr);
            const bool k2 = dropout_keep(rng, drop_site, m, c + 2, drop_thr), k3 = dropout_keep(rng, drop_site, m, c + 3, drop_thr);
            v.x = k0 ? v.x * ik : 0.f; v.y = k1 ? v.y * ik : 0.f; v.z = k2 ? v.z * ik : 0.f; v.w = k3 ? v.w * ik : 0.f;
            // optional 4-bit keep record per float4: the backward of a 2400 x 512 embedding spent 54 us regenerating
            // these Philox words (one call per element there: its rows are gathered by token id)
            if (keep_mask) keep_mask[idx] = (unsigned char)((k0 ? 1 : 0) | (k1 ? 2 : 0) | (k2 ? 4 : 0) | (k3 ? 8 : 0));
        }
        // decoder input token == <pad>: its single self-attention key is masked
        // (transformer.py:72-73) -> softmax over an empty set -> NaN row in torch.
        if (id == nan_idx) v = make_float4(NAN, NAN, NAN, NAN);
        *reinterpret_cast<float4*>(out + (long)m * E + c) = v;
        store_planes4(po, (long)m * E + c, v);
    }
}
SLNLP_ZKERNEL(embed_fwd_kernel, 256, embed_fwd_body)

// Embedding backward = segmented sum of dx rows by token id, deterministic (fixed order, no float
// atomics), in two levels so that a hot id (the <pad> row takes ~40 % of all tokens) is never one
// long serial chain of dependent loads:
//  level 1: tokens are cut into chunks of 64 consecutive positions m = s*B+b; inside a chunk the first
//           occurrence of an id sums the chunk's occurrences in increasing m -> partial[m], pid[m] = id;
//  level 2: the first chunk-partial of an id sums the (<= #chunks) partials of that id in increasing m
//           and writes the table row.  Rows of ids that do not occur were zeroed by a memset.
__device__ __forceinline__ float4 load_dx_row(const float* __restrict__ dx, int t, int E, int c, float drop_p,
                                              float ik, unsigned thr, int site,
                                              const unsigned long long* __restrict__ rng,
                                              const unsigned char* __restrict__ keep_mask) {
    float4 g = *reinterpret_cast<const float4*>(dx + (long)t * E + c);
    if (drop_p > 0.f && keep_mask) {          // keep bits recorded by embed_fwd
        const unsigned k = keep_mask[((long)t * E + c) >> 2];
        g.x = (k & 1u) ? g.x * ik : 0.f; g.y = (k & 2u) ? g.y * ik : 0.f;
        g.z = (k & 4u) ? g.z * ik : 0.f; g.w = (k & 8u) ? g.w * ik : 0.f;
    } else if (drop_p > 0.f) {
        g.x = dropout_keep(rng, site, t, c + 0, thr) ? g.x * ik : 0.f;
        g.y = dropout_keep(rng, site, t, c + 1, thr) ? g.y * ik : 0.f;
        g.z = dropout_keep(rng, site, t, c + 2, thr) ? g.z * ik : 0.f;
        g.w = dropout_keep(rng, site, t, c + 3, thr) ? g.w * ik : 0.f;
    }
    return g;
}

constexpr int EMB_CHUNK = 64;

__device__ __forceinline__ void embed_bwd_chunk_body(const long* __restrict__ ids, long ld_ids, int B, int S,
                                                              int E, int V, const float* __restrict__ dx,
                                                              float* __restrict__ partial, int* __restrict__ pid,
                                                              float drop_p, unsigned drop_thr, int drop_site,
                                                              const unsigned long long* __restrict__ rng,
                                                              const unsigned char* __restrict__ keep_mask) {
    // one workgroup per token: almost all exit at once (not the first of their id in the 64-token chunk); the
    // chunk-first of an id sums the chunk's rows of that id in increasing-m order, all 256 threads across columns
    const int M = B * S, lane = threadIdx.x & 63;
    const int m = blockIdx.x, m0 = (m / EMB_CHUNK) * EMB_CHUNK, t = m - m0, mine = m0 + lane;
    int my_id = -1;                        // lane l holds the id of token m0 + l (or -1)
    if (mine < M) {
        const long v = ids[(long)(mine % B) * ld_ids + (mine / B)];
        my_id = (v < 0 || v >= V) ? -1 : (int)v;
    }
    const float ik = 1.f / (1.f - drop_p);
    const int id = __builtin_amdgcn_readlane(my_id, __builtin_amdgcn_readfirstlane(t));   // t is block-uniform
    const unsigned long long mask = __ballot(my_id == id && id >= 0);
    const bool first = id >= 0 && (mask & ((1ull << t) - 1ull)) == 0ull;
    if (threadIdx.x == 0) pid[m] = first ? id : -1;
    if (!first) return;
    // The matching rows are summed as a FIXED tree: wave w adds the rows at chunk positions [16w, 16w+16) in
    // increasing order, then the four wave sums are added in wave order -- bit-reproducible, and the serial
    // load -> add chain is a quarter as long as one running sum (the <pad> id matches ~half of every chunk).
    __shared__ __attribute__((aligned(16))) float red[4][256];
    const int wave = threadIdx.x >> 6;
    const unsigned long long wmask = mask & (0xFFFFull << (16 * wave));
    for (int c0 = 0; c0 < E; c0 += 256) {
        const int c = c0 + lane * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < E) {
            unsigned long long bits = wmask;
            while (bits) {                 // 4 independent loads in flight
                int k[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    k[u] = bits ? __ffsll((long long)bits) - 1 : -1;
                    bits &= bits - 1;      // 0 & anything stays 0
                }
                float4 g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    g[u] = k[u] >= 0 ? load_dx_row(dx, m0 + k[u], E, c, drop_p, ik, drop_thr, drop_site, rng, keep_mask)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w; }
            }
        }
        if (c0 > 0) __syncthreads();
        *reinterpret_cast<float4*>(&red[wave][lane * 4]) = acc;
        __syncthreads();
        if (wave == 0 && c < E) {
            float4 r = *reinterpret_cast<const float4*>(&red[0][lane * 4]);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 q = *reinterpret_cast<const float4*>(&red[w][lane * 4]);
                r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
            }
            *reinterpret_cast<float4*>(partial + (long)m * E + c) = r;
        }
    }
}
SLNLP_ZKERNEL(embed_bwd_chunk_kernel, 256, embed_bwd_chunk_body)

__device__ __forceinline__ void embed_bwd_combine_body(const int* __restrict__ pid, int M, int E,
                                                                const float* __restrict__ partial,
                                                                float* __restrict__ dtable, float scale) {
    __shared__ int list[1024];             // slots holding a partial of this id, increasing (<= #chunks)
    __shared__ int nlist;
    const int m = blockIdx.x, tid = threadIdx.x;
    const int id = pid[m];
    if (id < 0) return;
    int dup = 0;
    for (int mm = tid; mm < m; mm += 256) dup |= (pid[mm] == id);
    if (__syncthreads_or(dup)) return;     // an earlier chunk owns this id
    if (tid == 0) nlist = 0;
    __syncthreads();
    // later chunk-firsts of the same id: at most one per chunk -> probe only slots >= m, chunk by chunk
    for (int base = m; base < M; base += 256) {
        const int mm = base + tid;
        const bool hit = mm < M && pid[mm] == id;
        const unsigned long long bal = __ballot(hit);
        __shared__ unsigned long long match[4];
        __syncthreads();
        if ((tid & 63) == 0) match[tid >> 6] = bal;
        __syncthreads();
        if (tid == 0) {
            int n = nlist;
            for (int w = 0; w < 4; ++w) {
                unsigned long long bits = match[w];
                while (bits && n < 1024) {
                    list[n++] = base + w * 64 + __ffsll((long long)bits) - 1;
                    bits &= bits - 1;
                }
            }
            nlist = n;
        }
        __syncthreads();
    }
    const int n = nlist;
    // same fixed tree: wave w adds list entries [w*q, (w+1)*q), then the wave sums are added in wave order
    __shared__ __attribute__((aligned(16))) float red[4][256];
    const int wave = tid >> 6, lane = tid & 63, per = (n + 3) / 4;
    const int i0 = wave * per, i1 = (i0 + per < n) ? i0 + per : n;
    for (int c0 = 0; c0 < E; c0 += 256) {
        const int c = c0 + lane * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < E) {
            for (int i = i0; i < i1; i += 4) {
                float4 g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    g[u] = (i + u < i1) ? *reinterpret_cast<const float4*>(partial + (long)list[i + u] * E + c)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w; }
            }
        }
        if (c0 > 0) __syncthreads();
        *reinterpret_cast<float4*>(&red[wave][lane * 4]) = acc;
        __syncthreads();
        if (wave == 0 && c < E) {
            float4 r = *reinterpret_cast<const float4*>(&red[0][lane * 4]);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 q = *reinterpret_cast<const float4*>(&red[w][lane * 4]);
                r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
            }
            r.x *= scale; r.y *= scale; r.z *= scale; r.w *= scale;
            *reinterpret_cast<float4*>(dtable + (long)id * E + c) = r;
        }
    }
}
SLNLP_ZKERNEL(embed_bwd_combine_kernel, 256, embed_bwd_combine_body)

// The whole scatter-add in ONE launch when the batch has at most EMB_CHUNK tokens (the target side: one token per sequence):
// grid max(M, V).  Workgroup i < V zero-fills table row i if no token carries id i; workgroup m < M, if token m is the first of
// its id, sums that id's rows with the same fixed tree as embed_bwd_chunk (wave w adds chunk positions [16w, 16w + 16) in order,
// then the wave sums in wave order) and writes sum * scale -- the bits of chunk + combine with a single chunk, without the
// fill_zero and combine launches.
__device__ __forceinline__ void embed_bwd_small_body(const long* __restrict__ ids, long ld_ids, int B, int S, int E, int V,
                                                     const float* __restrict__ dx, float* __restrict__ dtable, float scale,
                                                     float drop_p, unsigned drop_thr, int drop_site,
                                                     const unsigned long long* __restrict__ rng) {
    const int M = B * S, lane = threadIdx.x & 63, m = blockIdx.x;
    int my_id = -1;                        // lane l holds the id of token l (or -1)
    if (lane < M) {
        const long v = ids[(long)(lane % B) * ld_ids + (lane / B)];
        my_id = (v < 0 || v >= V) ? -1 : (int)v;
    }
    if (m < V && __ballot(my_id == m) == 0ull) {          // nobody carries id m: the row's gradient is zero
        for (int c = threadIdx.x * 4; c < E; c += 1024) *reinterpret_cast<float4*>(dtable + (long)m * E + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (m >= M) return;
    const float ik = 1.f / (1.f - drop_p);
    const int id = __builtin_amdgcn_readlane(my_id, __builtin_amdgcn_readfirstlane(m));   // m is block-uniform
    const unsigned long long mask = __ballot(my_id == id && id >= 0);
    const bool first = id >= 0 && (mask & ((1ull << m) - 1ull)) == 0ull;
    if (!first) return;
    __shared__ __attribute__((aligned(16))) float red[4][256];
    const int wave = threadIdx.x >> 6;
    const unsigned long long wmask = mask & (0xFFFFull << (16 * wave));
    for (int c0 = 0; c0 < E; c0 += 256) {
        const int c = c0 + lane * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < E) {
            unsigned long long bits = wmask;
            while (bits) {
                int k[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    k[u] = bits ? __ffsll((long long)bits) - 1 : -1;
                    bits &= bits - 1;
                }
                float4 g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    g[u] = k[u] >= 0 ? load_dx_row(dx, k[u], E, c, drop_p, ik, drop_thr, drop_site, rng, nullptr)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w; }
            }
        }
        if (c0 > 0) __syncthreads();
        *reinterpret_cast<float4*>(&red[wave][lane * 4]) = acc;
        __syncthreads();
        if (wave == 0 && c < E) {
            float4 r = *reinterpret_cast<const float4*>(&red[0][lane * 4]);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 q = *reinterpret_cast<const float4*>(&red[w][lane * 4]);
                r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
            }
            r.x *= scale; r.y *= scale; r.z *= scale; r.w *= scale;
            *reinterpret_cast<float4*>(dtable + (long)id * E + c) = r;
        }
    }
}
SLNLP_ZKERNEL(embed_bwd_small_kernel, 256, embed_bwd_small_body)

int embed_fwd(const int64_t* ids, int64_t ld_ids, int B, int S, int E, int V, const float* table, const float* pe,
              float* out, float scale, float drop_p, int drop_site, const unsigned long long* rng, int64_t nan_idx,
              hipStream_t st, PlaneOut po, unsigned char* keep_mask) {
    SLNLP_CHECK_ARG(ids && table && out, "embed_fwd: null pointer");
    SLNLP_CHECK_ARG(B > 0 && S > 0 && V > 0 && E > 0 && E % 4 == 0, "embed_fwd: bad shape B=%d S=%d E=%d V=%d", B, S, E, V);
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "embed_fwd: bad dropout args");
    const long total = (long)B * S * (E / 4);
    int grid = ceil_div(total, 256);
    if (grid > 2048) grid = 2048;
    SLNLP_TRY(zlaunch(embed_fwd_kernel, dim3(grid), 256, 0, st, "embed_fwd",
                      (const long*)ids, (long)ld_ids, B, S, E, V, table, pe, out, scale, drop_p, dropout_threshold(drop_p), drop_site, rng, (long)nan_idx, po, keep_mask));
    return 0;
}

size_t embed_bwd_scratch_bytes(int B, int S, int E) {
    return ((size_t)B * S * E * sizeof(float) + (size_t)B * S * sizeof(int) + 255) & ~(size_t)255;
}

int embed_bwd(const int64_t* ids, int64_t ld_ids, int B, int S, int E, int V, const float* dx, float* dtable,
              float scale, int64_t zero_row, float drop_p, int drop_site, const unsigned long long* rng, void* scratch,
              hipStream_t st, const unsigned char* keep_mask) {
    SLNLP_CHECK_ARG(ids && dx && dtable && scratch, "embed_bwd: null pointer");
    SLNLP_CHECK_ARG(B > 0 && S > 0 && V > 0 && E > 0 && E % 4 == 0, "embed_bwd: bad shape");
    SLNLP_CHECK_ARG((long)B * S <= 65536, "embed_bwd: more than 65536 tokens per batch");
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "embed_bwd: bad dropout args");
    const int M = B * S;
    if (M <= EMB_CHUNK && !keep_mask) {      // one chunk: zero-fill, sum and scale in one launch
        SLNLP_TRY(zlaunch(embed_bwd_small_kernel, dim3(M > V ? M : V), 256, 0, st, "embed_bwd_small",
                          (const long*)ids, (long)ld_ids, B, S, E, V, dx, dtable, scale, drop_p, dropout_threshold(drop_p), drop_site, rng));
        if (zero_row >= 0 && zero_row < V) SLNLP_TRY(fill_zero(dtable + zero_row * E, (size_t)E * sizeof(float), st));
        return 0;
    }
    SLNLP_TRY(fill_zero(dtable, (size_t)V * E * sizeof(float), st));    // rows of ids that do not occur (our kernel: recordable)
    float* partial = (float*)scratch;
    int* pid = (int*)(partial + (size_t)M * E);
    SLNLP_TRY(zlaunch(embed_bwd_chunk_kernel, dim3(M), 256, 0, st, "embed_bwd_chunk",
                      (const long*)ids, (long)ld_ids, B, S, E, V, dx, partial, pid, drop_p, dropout_threshold(drop_p), drop_site, rng, keep_mask));
    SLNLP_TRY(zlaunch(embed_bwd_combine_kernel, dim3(M), 256, 0, st, "embed_bwd_combine",
                      pid, M, E, partial, dtable, scale));
    if (zero_row >= 0 && zero_row < V) SLNLP_TRY(fill_zero(dtable + zero_row * E, (size_t)E * sizeof(float), st));
    return 0;
}

// ================================================================= layernorm
constexpr int LN_MAXU = 4;  // float4 per lane -> E <= 1024 in the backward (register-resident columns)

__device__ __forceinline__ void layernorm_fwd_body(const float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int rows, int E,
                                                            float eps, float* __restrict__ y,
                                                            float* __restrict__ stats, PlaneOut po) {
    // one wave per row; the row is read ONCE and kept in registers (E <= LN_MAXU * 256) for the two-pass statistics
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const float* xr = x + (long)row * E;
    float4 v[LN_MAXU], g[LN_MAXU], bt[LN_MAXU];
#pragma unroll
    for (int u = 0; u < LN_MAXU; ++u) {
        const int c = lane * 4 + u * 256;
        const bool in = c < E;
        v[u] = in ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        g[u] = in ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        bt[u] = in ? *reinterpret_cast<const float4*>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < LN_MAXU; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;      // out-of-range slots hold zeros
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < LN_MAXU; ++u) {
        if (lane * 4 + u * 256 < E) {
            const float a = v[u].x - mean, b = v[u].y - mean, cc = v[u].z - mean, d = v[u].w - mean;
            q += a * a + b * b + cc * cc + d * d;
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)E + eps);
#pragma unroll
    for (int u = 0; u < LN_MAXU; ++u) {
        const int c = lane * 4 + u * 256;
        if (c < E) {
            float4 o;
            o.x = (v[u].x - mean) * rstd * g[u].x + bt[u].x; o.y = (v[u].y - mean) * rstd * g[u].y + bt[u].y;
            o.z = (v[u].z - mean) * rstd * g[u].z + bt[u].z; o.w = (v[u].w - mean) * rstd * g[u].w + bt[u].w;
            *reinterpret_cast<float4*>(y + (long)row * E + c) = o;
            store_planes4(po, (long)row * E + c, o);
        }
    }
    if (lane == 0 && stats) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}
SLNLP_ZKERNEL(layernorm_fwd_kernel, 256, layernorm_fwd_body)

// LayerNorm backward is TWO kernels since round 3:
//   layernorm_bwd_rows   dx (and its dropout-masked copy, and their bf16 planes): purely row-wise, one wave per GS consecutive rows;
//   ln_param_partial     the (dgamma, dbeta) column sums of the same (dy, x, stats), per chunk of rows, for EVERY LayerNorm of the
//                        step in one table-driven launch at the end of backward; ln_param_reduce adds the chunks in order.
// History: the round-2 kernel also kept the (dgamma, dbeta) accumulators and combined them through LDS at its end, and was the
// kernel in which the multi-queue nondeterminism of DESIGN.md section 6 first showed (40-85 % of its runs differed beside two
// other fits).  The cause turned out to be the packed fp32 instructions its arithmetic compiled to (the library is built without
// them now, see the Makefile), not its structure; the split stayed because it is faster (15.0 -> 12.8 us per launch at cfg2, one
// 32 us column-sum launch per step) and keeps every instance under 256 registers.
//
// U = float4 slots per lane (E <= 256 * U), GS = rows per wave (4: many rows; 1: the decoder's B rows, spread over as many waves as
// possible), RB = rows whose loads are in flight together (U * RB <= 8: every instance stays under 256 registers).
//
// FUSE (round 5): a workgroup of four waves takes a CHUNK of `chunk_rows` rows (16 for batches of >= 1024 rows, ln_partial_chunk),
// a group of GS rows per wave as before (more groups: wave w takes groups w, w + 4, ...), every wave keeps the (dgamma, dbeta) column
// sums of its rows in registers (rows ascending), and the four waves' sums meet in LDS and are added in wave order:
// partial[chunk][0][c] = sum dy xhat, partial[chunk][1][c] = sum dy -- instead of ln_param_partial reading dy and x a second time at
// the end of backward (15 fits in lockstep: 2.5 GB, 380 us of a 14.7 ms step; configs[4]: 350 us of 13.7 ms), for a twelfth of those
// bytes in partial sums written and read back.  (One wave walking the chunk's four groups one after the other was 2 x slower per
// launch at every size -- a wave has one group's loads in flight: configs[1] +6 %.)
template <int U, int GS, int RB, bool FUSE>
__device__ __forceinline__ void layernorm_bwd_rows(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ stats, int rows, int E, const float* __restrict__ add_to_dx, float* __restrict__ dx,
    float* __restrict__ dx_drop, float drop_p, unsigned drop_thr, int drop_site,
    const unsigned long long* __restrict__ rng, PlaneOut po_dx, PlaneOut po_drop, float* __restrict__ partial, int chunk_rows, int nchunks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float invE = 1.f / (float)E, ik = 1.f / (1.f - drop_p);
    // the dropout-masked copy goes out as fp32 (dx_drop), as bf16 planes (po_drop), or both: a caller whose consumers read planes
    // only (the plane-GEMM path) passes dx_drop = nullptr and saves the fp32 store
    const bool want_drop = dx_drop != nullptr || po_drop.hi != nullptr;
    const bool drop = want_drop && drop_p > 0.f;
    const int unit = FUSE ? (int)blockIdx.x : blockIdx.x * (int)(blockDim.x >> 6) + wave;   // a chunk per workgroup / a group per wave
    float4 ag[U], ab[U];
#pragma unroll
    for (int u = 0; u < U; ++u) ag[u] = ab[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int row_begin = FUSE ? unit * chunk_rows + wave * GS : unit * GS;
    const int row_end = FUSE ? ((unit + 1) * chunk_rows < rows ? (unit + 1) * chunk_rows : rows) : row_begin + GS;
    const int row_step = FUSE ? 4 * GS : GS;
    if (!FUSE && row_begin >= rows) return;
    DropKey dkey = {};
    if (drop) dkey = dropout_key(rng, drop_site);
    float4 g[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (lane * 4 + u * 256 < E) g[u] = *reinterpret_cast<const float4*>(gamma + lane * 4 + u * 256);
#pragma unroll 1
    for (int row0 = row_begin; row0 < row_end; row0 += row_step)      // (a chunk past the batch's last row: no trip, zero sums -- the reduce table is the full batch's)
#pragma unroll
    for (int i0 = 0; i0 < GS; i0 += RB) {
        float4 d[RB][U], v[RB][U];
        float mean[RB], rstd[RB], s1[RB], s2[RB];
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = row0 + i0 + i < rows ? row0 + i0 + i : rows - 1;   // clamped: the tail rows are masked below
            mean[i] = stats[2 * row];
            rstd[i] = stats[2 * row + 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = lane * 4 + u * 256;
                if (c < E) {
                    d[i][u] = *reinterpret_cast<const float4*>(dy + (long)row * E + c);
                    v[i][u] = *reinterpret_cast<const float4*>(x + (long)row * E + c);
                }
            }
        }
        // phase 1, per row: the two row sums
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (lane * 4 + u * 256 < E) {
                    const float4 dd = d[i][u], vv = v[i][u];
                    const float4 xh = make_float4((vv.x - mean[i]) * rstd[i], (vv.y - mean[i]) * rstd[i], (vv.z - mean[i]) * rstd[i], (vv.w - mean[i]) * rstd[i]);
                    const float4 gv = make_float4(dd.x * g[u].x, dd.y * g[u].y, dd.z * g[u].z, dd.w * g[u].w);
                    a1 += gv.x + gv.y + gv.z + gv.w;
                    a2 += gv.x * xh.x + gv.y * xh.y + gv.z * xh.z + gv.w * xh.w;
                }
            }
            s1[i] = wave_sum(a1) * invE;
            s2[i] = wave_sum(a2) * invE;
        }
        // phase 2, per column slot: the dropout lots of the slot's 4 columns (one Philox call per column serves the group's
        // four rows -- and the same rows 16 columns on, which belong to lane ^ 4: this HBM-bound kernel does not share them,
        // common.hpp) live only here, then every row's outputs for the slot
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = lane * 4 + u * 256;
            if (c >= E) continue;
            uint4 kb[4];
            const int half = drop_half((unsigned)c);                           // (c % 4 == 0: the slot's four columns share bit 4)
            if (drop) {
#pragma unroll
                for (int e = 0; e < 4; ++e) kb[e] = dropout_bits8(dkey, (unsigned)(row0 + i0) >> 2, drop_cc((unsigned)(c + e)));
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int row = row0 + i0 + i;
                if (row >= rows) continue;
                const int wsel = row & 3;                                     // which Philox word is this row's
                const float4 dd = d[i][u], vv = v[i][u];
                const float4 xh = make_float4((vv.x - mean[i]) * rstd[i], (vv.y - mean[i]) * rstd[i], (vv.z - mean[i]) * rstd[i], (vv.w - mean[i]) * rstd[i]);
                const float4 gv = make_float4(dd.x * g[u].x, dd.y * g[u].y, dd.z * g[u].z, dd.w * g[u].w);
                if (FUSE) {
                    ag[u].x += dd.x * xh.x; ag[u].y += dd.y * xh.y; ag[u].z += dd.z * xh.z; ag[u].w += dd.w * xh.w;
                    ab[u].x += dd.x; ab[u].y += dd.y; ab[u].z += dd.z; ab[u].w += dd.w;
                }
                float4 o;
                o.x = rstd[i] * (gv.x - s1[i] - xh.x * s2[i]); o.y = rstd[i] * (gv.y - s1[i] - xh.y * s2[i]);
                o.z = rstd[i] * (gv.z - s1[i] - xh.z * s2[i]); o.w = rstd[i] * (gv.w - s1[i] - xh.w * s2[i]);
                if (add_to_dx) {
                    const float4 a = *reinterpret_cast<const float4*>(add_to_dx + (long)row * E + c);
                    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
                }
                *reinterpret_cast<float4*>(dx + (long)row * E + c) = o;
                store_planes4(po_dx, (long)row * E + c, o);
                if (want_drop) {
                    if (drop) {
                        o.x = pick_lot(kb[0], half, wsel) >= drop_thr ? o.x * ik : 0.f;
                        o.y = pick_lot(kb[1], half, wsel) >= drop_thr ? o.y * ik : 0.f;
                        o.z = pick_lot(kb[2], half, wsel) >= drop_thr ? o.z * ik : 0.f;
                        o.w = pick_lot(kb[3], half, wsel) >= drop_thr ? o.w * ik : 0.f;
                    }
                    if (dx_drop) *reinterpret_cast<float4*>(dx_drop + (long)row * E + c) = o;
                    store_planes4(po_drop, (long)row * E + c, o);
                }
            }
        }
    }
    if (FUSE) {
        __shared__ float4 red[4][2][U * 64];             // [wave][dgamma, dbeta][column slot]
#pragma unroll
        for (int u = 0; u < U; ++u) {
            red[wave][0][u * 64 + lane] = ag[u];
            red[wave][1][u * 64 + lane] = ab[u];
        }
        __syncthreads();
        // thread t adds the four waves' sums of one (which, slot) pair, in wave order
        for (int q = threadIdx.x; q < 2 * U * 64; q += 256) {
            const int which = q / (U * 64), slot = q - which * (U * 64), c = (slot & 63) * 4 + (slot >> 6) * 256;
            if (c >= E) continue;
            float4 t = red[0][which][slot];
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const float4 o = red[k][which][slot];
                t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
            }
            *reinterpret_cast<float4*>(partial + ((long)unit * 2 + which) * E + c) = t;
        }
    }
}

// one kernel per (row-length class, rows per wave): a kernel's register allocation is the maximum over its branches
#define SLNLP_LN_BWD_BODY(name, U, GS, RB, FUSE)                                                                                    \
    __device__ __forceinline__ void name(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma, \
                                         const float* __restrict__ stats, int rows, int E, const float* __restrict__ add_to_dx,     \
                                         float* __restrict__ dx, float* __restrict__ dx_drop, float drop_p, unsigned drop_thr,       \
                                         int drop_site, const unsigned long long* __restrict__ rng, PlaneOut po_dx,                  \
                                         PlaneOut po_drop, float* __restrict__ partial, int chunk_rows, int nchunks) {               \
        layernorm_bwd_rows<U, GS, RB, FUSE>(dy, x, gamma, stats, rows, E, add_to_dx, dx, dx_drop, drop_p, drop_thr, drop_site, rng, \
                                            po_dx, po_drop, partial, chunk_rows, nchunks);                                          \
    }
SLNLP_LN_BWD_BODY(layernorm_bwd_body_u1g4, 1, 4, 4, false)
SLNLP_LN_BWD_BODY(layernorm_bwd_body_u2g4, 2, 4, 4, false)
SLNLP_LN_BWD_BODY(layernorm_bwd_body_u4g4, 4, 4, 2, false)
SLNLP_LN_BWD_BODY(layernorm_bwd_body_u1g1, 1, 1, 1, false)
SLNLP_LN_BWD_BODY(layernorm_bwd_body_u2g1, 2, 1, 1, false)
SLNLP_LN_BWD_BODY(layernorm_bwd_body_u4g1, 4, 1, 1, false)
SLNLP_LN_BWD_BODY(layernorm_bwd_body_u1g4f, 1, 4, 4, true)
SLNLP_LN_BWD_BODY(layernorm_bwd_body_u2g4f, 2, 4, 4, true)
SLNLP_LN_BWD_BODY(layernorm_bwd_body_u4g4f, 4, 4, 2, true)
SLNLP_ZKERNEL(layernorm_bwd_kernel_u1g4f, 256, layernorm_bwd_body_u1g4f)
SLNLP_ZKERNEL(layernorm_bwd_kernel_u2g4f, 256, layernorm_bwd_body_u2g4f)
SLNLP_ZKERNEL(layernorm_bwd_kernel_u4g4f, 256, layernorm_bwd_body_u4g4f)
SLNLP_ZKERNEL(layernorm_bwd_kernel_u1g4, 256, layernorm_bwd_body_u1g4)
SLNLP_ZKERNEL(layernorm_bwd_kernel_u2g4, 256, layernorm_bwd_body_u2g4)
SLNLP_ZKERNEL(layernorm_bwd_kernel_u4g4, 256, layernorm_bwd_body_u4g4)
SLNLP_ZKERNEL(layernorm_bwd_kernel_u1g1, 256, layernorm_bwd_body_u1g1)
SLNLP_ZKERNEL(layernorm_bwd_kernel_u2g1, 256, layernorm_bwd_body_u2g1)
SLNLP_ZKERNEL(layernorm_bwd_kernel_u4g1, 256, layernorm_bwd_body_u4g1)

// (dgamma, dbeta) partial sums of one LayerNorm per chunk of rows: partial[chunk][0][c] = sum_r dy[r, c] xhat[r, c],
// partial[chunk][1][c] = sum_r dy[r, c] over the chunk's rows in increasing order.  grid (chunks, entries); a thread owns 4
// columns (blockDim = E / 4 rounded up to a wave) and walks its chunk's rows: coalesced 16-byte loads, no LDS, no barrier.  A chunk
// past the batch's last row writes zeros, so the reduce table can stay the full batch's (ln_param_reduce adds `nblk` chunks).
__device__ __forceinline__ void ln_param_partial_body(const LnPartialEntry* __restrict__ table, LnPartialEntry single, int E, int rows_enc,
                                                      int rows_dec, int chunk_enc, int chunk_dec, int nchunk_enc, int nchunk_dec) {
    LnPartialEntry e = table ? table[blockIdx.y] : single;
    const float* __restrict__ dy = as_global(e.dy);
    const float* __restrict__ x = as_global(e.x);
    const float* __restrict__ stats = as_global(e.stats);
    float* __restrict__ partial = as_global(e.partial);
    const int rows = e.dec ? rows_dec : rows_enc, chunk = e.dec ? chunk_dec : chunk_enc;
    const int c = threadIdx.x * 4;
    if (c >= E || (int)blockIdx.x >= (e.dec ? nchunk_dec : nchunk_enc)) return;   // the grid is the larger class's chunk count
    const int r0 = blockIdx.x * chunk;
    const int r1 = r0 + chunk < rows ? r0 + chunk : rows;
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](const float4& dd, const float4& vv, float mean, float rstd) {
        ag.x += dd.x * ((vv.x - mean) * rstd); ag.y += dd.y * ((vv.y - mean) * rstd);
        ag.z += dd.z * ((vv.z - mean) * rstd); ag.w += dd.w * ((vv.w - mean) * rstd);
        ab.x += dd.x; ab.y += dd.y; ab.z += dd.z; ab.w += dd.w;
    };
    int r = r0;
    for (; r + 8 <= r1; r += 8) {                    // 8 rows' loads in flight, added in row order
        float4 dd[8], vv[8];
        float2 ms[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dd[i] = *reinterpret_cast<const float4*>(dy + (long)(r + i) * E + c);
            vv[i] = *reinterpret_cast<const float4*>(x + (long)(r + i) * E + c);
            ms[i] = *reinterpret_cast<const float2*>(stats + 2 * (r + i));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) add(dd[i], vv[i], ms[i].x, ms[i].y);
    }
    for (; r < r1; ++r)
        add(*reinterpret_cast<const float4*>(dy + (long)r * E + c), *reinterpret_cast<const float4*>(x + (long)r * E + c), stats[2 * r], stats[2 * r + 1]);
    *reinterpret_cast<float4*>(partial + ((long)blockIdx.x * 2 + 0) * E + c) = ag;
    *reinterpret_cast<float4*>(partial + ((long)blockIdx.x * 2 + 1) * E + c) = ab;
}
SLNLP_ZKERNEL(ln_param_partial_kernel, 256, ln_param_partial_body)

// grid (entry, ceil(E/64)); block = 64 columns x 4 partial-groups, combined through LDS in fixed order.
__device__ __forceinline__ void ln_param_reduce_body(const slnlp_ln_reduce_entry* __restrict__ table) {
    __shared__ float red[2][4][64];
    const slnlp_ln_reduce_entry e = table[blockIdx.x];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6, c = blockIdx.y * 64 + col;
    float g = 0.f, b = 0.f;
    if (c < e.E) {
        // up to 64 partials per thread: 8 independent load pairs in flight per trip (the adds stay in increasing-k order)
        int k = grp;
        for (; k + 28 < e.nblk; k += 32) {
            float pg[8], pb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                pg[u] = e.partial[((long)(k + 4 * u) * 2 + 0) * e.E + c];
                pb[u] = e.partial[((long)(k + 4 * u) * 2 + 1) * e.E + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { g += pg[u]; b += pb[u]; }
        }
        for (; k < e.nblk; k += 4) {
            g += e.partial[((long)k * 2 + 0) * e.E + c];
            b += e.partial[((long)k * 2 + 1) * e.E + c];
        }
    }
    red[0][grp][col] = g;
    red[1][grp][col] = b;
    __syncthreads();
    if (grp == 0 && c < e.E) {
        e.dgamma[c] = (red[0][0][col] + red[0][1][col]) + (red[0][2][col] + red[0][3][col]);
        e.dbeta[c] = (red[1][0][col] + red[1][1][col]) + (red[1][2][col] + red[1][3][col]);
    }
}
SLNLP_ZKERNEL(ln_param_reduce_kernel, 256, ln_param_reduce_body)

int layernorm_fwd(const float* x, const float* gamma, const float* beta, int rows, int E, float eps, float* y,
                  float* stats, hipStream_t st, PlaneOut po) {
    SLNLP_CHECK_ARG(x && gamma && beta && y, "layernorm_fwd: null pointer");
    SLNLP_CHECK_ARG(rows > 0 && E > 0 && E % 4 == 0 && E <= LN_MAXU * 256, "layernorm_fwd: need E %% 4 == 0 and E <= %d, got rows=%d E=%d", LN_MAXU * 256, rows, E);
    SLNLP_TRY(zlaunch(layernorm_fwd_kernel, dim3(ceil_div(rows, 4)), 256, 0, st, "layernorm_fwd",
                      x, gamma, beta, rows, E, eps, y, stats, po));
    return 0;
}

static int ln_bwd_group(int rows) { return rows >= 1024 ? 4 : 1; }   // rows per wave

// (dgamma, dbeta) sums inside the row kernel (FUSE) for batches of many rows; the few rows of a decoder batch keep one wave per row
// and the separate column-sum launch
bool ln_bwd_fused(int full_rows) { return ln_bwd_group(full_rows) == 4; }
// rows per chunk of the (dgamma, dbeta) partial sums: 16 -- the fused class's workgroup (four waves of four rows), and four times the
// workgroups of a 64-row chunk for the column-sum launch of the decoder's few rows (11.6 -> 6 us at 50 rows) -- more when that would
// exceed SLNLP_LN_MAX_PARTIALS chunks
int ln_partial_chunk(int rows) {
    int c = 16;
    while (ceil_div(rows, c) > SLNLP_LN_MAX_PARTIALS) c *= 2;
    return c;
}
int ln_bwd_blocks(int rows) { return ceil_div(rows, ln_partial_chunk(rows)); }   // chunks = entries ln_param_reduce adds

// the (dgamma, dbeta) chunk sums of `n` LayerNorms (device table) or of one (table == nullptr: `single`), one launch
int ln_param_partial(const LnPartialEntry* table_dev, const LnPartialEntry* single_host, int n, int E, int rows_enc, int rows_dec, int full_rows_enc,
                     int full_rows_dec, hipStream_t st) {
    SLNLP_CHECK_ARG(n > 0 && E > 0 && E % 4 == 0 && E <= LN_MAXU * 256, "ln_param_partial: bad args");
    LnPartialEntry single;
    memset(&single, 0, sizeof(single));
    if (single_host) single = *single_host;
    // chunk sizes (and so the chunk count every entry writes) follow the FULL batch: smaller batches leave zero chunks.  Encoder
    // entries of a fused-class batch were summed by their row kernels (layernorm_bwd with a partial pointer): no chunk of theirs here
    const int ce = ln_partial_chunk(full_rows_enc > 0 ? full_rows_enc : 1), cd = ln_partial_chunk(full_rows_dec > 0 ? full_rows_dec : 1);
    const int ne = full_rows_enc > 0 && !(table_dev && ln_bwd_fused(full_rows_enc)) ? ceil_div(full_rows_enc, ce) : 0;
    const int nd = full_rows_dec > 0 ? ceil_div(full_rows_dec, cd) : 0;
    const int nb = std::max(ne, nd);
    if (nb == 0) return 0;
    const int threads = (ceil_div(E, 4) + 63) / 64 * 64;
    return zlaunch(ln_param_partial_kernel, dim3(nb, n), threads, 0, st, "ln_param_partial", table_dev, single, E,
                   rows_enc, rows_dec, ce, cd, ne, nd);
}

int layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* stats, int rows, int E,
                  const float* add_to_dx, float* dx, float* dx_drop, float drop_p, int drop_site,
                  const unsigned long long* rng, float* partial, int* nblk_out, int full_rows, hipStream_t st,
                  PlaneOut po_dx, PlaneOut po_drop) {
    SLNLP_CHECK_ARG(dy && x && gamma && stats && dx, "layernorm_bwd: null pointer");
    SLNLP_CHECK_ARG(rows > 0 && E > 0 && E % 4 == 0 && E <= LN_MAXU * 256, "layernorm_bwd: need E %% 4 == 0 and E <= %d, got %d", LN_MAXU * 256, E);
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "layernorm_bwd: bad dropout args");
    // full_rows: the rows of the caller's FULL batch (0: this call's) -- the chunk geometry of the (dgamma, dbeta) sums follows it, so
    // that a shorter last batch writes the same number of chunks (the trailing ones zero) and the reduce table never changes
    if (full_rows < rows) full_rows = rows;
    const int u = E <= 256 ? 1 : E <= 512 ? 2 : 4;
    if (partial && ln_bwd_fused(full_rows)) {     // one wave per chunk: dx and the chunk's column sums in one pass over dy and x
        const int cr = ln_partial_chunk(full_rows), nchunks = ceil_div(full_rows, cr);
        auto kern = u == 1 ? layernorm_bwd_kernel_u1g4f : u == 2 ? layernorm_bwd_kernel_u2g4f : layernorm_bwd_kernel_u4g4f;
        SLNLP_TRY(zlaunch(kern, dim3(nchunks), 256, 0, st, "layernorm_bwd (fused)", dy, x, gamma, stats, rows, E, add_to_dx, dx, dx_drop, drop_p,
                          dropout_threshold(drop_p), drop_site, rng, po_dx, po_drop, partial, cr, nchunks));
        if (nblk_out) *nblk_out = nchunks;
        return 0;
    }
    const int gs = ln_bwd_group(rows);
    auto kern = gs == 4 ? (u == 1 ? layernorm_bwd_kernel_u1g4 : u == 2 ? layernorm_bwd_kernel_u2g4 : layernorm_bwd_kernel_u4g4)
                        : (u == 1 ? layernorm_bwd_kernel_u1g1 : u == 2 ? layernorm_bwd_kernel_u2g1 : layernorm_bwd_kernel_u4g1);
    // 4-row groups: ONE wave per workgroup -- 2400 rows are 600 waves, which 150 workgroups of four would park on 150 of the 256
    // CUs; single-wave workgroups spread over all of them (the kernel is a stream of 16-byte loads and stores per wave)
    const int wpb = gs == 4 ? 1 : 4;
    SLNLP_TRY(zlaunch(kern, dim3(ceil_div(rows, wpb * gs)), 64 * wpb, 0, st, "layernorm_bwd",
                      dy, x, gamma, stats, rows, E, add_to_dx, dx, dx_drop, drop_p, dropout_threshold(drop_p), drop_site, rng, po_dx, po_drop,
                      (float*)nullptr, 0, 0));
    // partial != nullptr: also this LayerNorm's (dgamma, dbeta) chunk sums, for callers that reduce one LayerNorm at a time
    // (the C API, tests); a plan passes nullptr and runs ONE table-driven ln_param_partial launch for all its LayerNorms
    if (nblk_out) *nblk_out = ln_bwd_blocks(full_rows);
    if (!partial) return 0;
    LnPartialEntry e;
    e.dy = dy; e.x = x; e.stats = stats; e.partial = partial; e.dec = 0; e.pad = 0;
    return ln_param_partial(nullptr, &e, 1, E, rows, 0, full_rows, 0, st);
}

int ln_param_reduce(const slnlp_ln_reduce_entry* table_dev, int n, int max_E, hipStream_t st) {
    SLNLP_CHECK_ARG(table_dev && n > 0 && max_E > 0, "ln_param_reduce: bad args");
    SLNLP_TRY(zlaunch(ln_param_reduce_kernel, dim3(n, ceil_div(max_E, 64)), 256, 0, st, "ln_param_reduce",
                      table_dev));
    return 0;
}

// ====================================================================== loss
// log_softmax (transformer.py:88-89) + CrossEntropyLoss(ignore_index) applied to
// the log-probs (helper.py:61-70).  One workgroup; one wave per row.
constexpr int LOSS_MAXB = 1024;

// grid = B rows, one wave each.  n_valid (targets != ignore_index) is recomputed by every row (B <= 1024
// compares) so d loss/d logits needs no second pass; the scalar loss is summed by lsm_loss_reduce_kernel.
__device__ __forceinline__ void lsm_nll_body(const float* __restrict__ logits, long ld, const long* __restrict__ y,
                                                     int B, int V, long ignore, float* __restrict__ logp,
                                                     float* __restrict__ row_nll, float* __restrict__ dlogits, long ldd,
                                                     float* __restrict__ logp2, const int* __restrict__ logp2_row) {
    // logp2 (optional): a second copy of the log-probs for the caller, at row offset *logp2_row (a device scalar, so a
    // recorded launch can walk an epoch's output buffer) -- replaces a device-to-device copy per step
    const int lane = threadIdx.x, b = blockIdx.x;
    if (logp2) logp2 += (long)(logp2_row ? *logp2_row : 0) * V;
    int cnt = 0;
    for (int i = lane; i < B; i += 64) {
        const long t = y[i];
        cnt += (t != ignore && t >= 0 && t < V) ? 1 : 0;
    }
    const float nvalid = wave_sum((float)cnt);
    const float* xr = logits + (long)b * ld;
    float m = -INFINITY;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, xr[v]);
    m = wave_max(m);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(xr[v] - m);
    const float lse = m + logf(wave_sum(s));
    // second log_softmax (the criterion's), over the log-probs
    float m2 = -INFINITY;
    for (int v = lane; v < V; v += 64) {
        const float lp = xr[v] - lse;
        logp[(long)b * V + v] = lp;
        if (logp2) logp2[(long)b * V + v] = lp;
        m2 = fmaxf(m2, lp);
    }
    m2 = wave_max(m2);
    float s2 = 0.f;
    for (int v = lane; v < V; v += 64) s2 += expf((xr[v] - lse) - m2);
    const float lse2 = m2 + logf(wave_sum(s2));
    const long t = y[b];
    const bool valid = (t != ignore) && t >= 0 && t < V;
    if (lane == 0) row_nll[b] = valid ? -((xr[t] - lse) - lse2) : NAN;   // NaN marks "ignored"
    if (!dlogits) return;
    // d loss / d logp = w * (softmax(logp) - onehot); then back through the model's log_softmax
    const float w = valid ? 1.f / nvalid : 0.f;
    float sum = 0.f;
    for (int v = lane; v < V; v += 64) sum += w * (expf((xr[v] - lse) - lse2) - (v == t ? 1.f : 0.f));
    sum = wave_sum(sum);
    for (int v = lane; v < V; v += 64) {
        const float lp = xr[v] - lse;
        const float dlp = w * (expf(lp - lse2) - (v == t ? 1.f : 0.f));
        dlogits[(long)b * ldd + v] = dlp - expf(lp) * sum;
    }
}
SLNLP_ZKERNEL(lsm_nll_kernel, 64, lsm_nll_body)

__device__ __forceinline__ void lsm_loss_reduce_body(const float* __restrict__ row_nll, int B, float* __restrict__ loss,
                                                     float* __restrict__ loss_hist, const int* __restrict__ hist_idx) {
    float tot = 0.f, n = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) {
        const float v = row_nll[b];
        if (v == v) { tot += v; n += 1.f; }
    }
    tot = wave_sum(tot);
    n = wave_sum(n);
    if (threadIdx.x == 0) {
        loss[0] = tot / n;   // 0/0 = NaN when every target is ignored, as torch
        if (loss_hist) loss_hist[hist_idx ? *hist_idx : 0] = tot / n;    // per-batch losses of an epoch, no host round trip
    }
}
SLNLP_ZKERNEL(lsm_loss_reduce_kernel, 64, lsm_loss_reduce_body)

__device__ __forceinline__ void lsm_bwd_body(const float* __restrict__ logp, const float* __restrict__ dlogp, int B,
                                                      int V, float* __restrict__ dlogits, long ldd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    float sum = 0.f;
    for (int v = lane; v < V; v += 64) sum += dlogp[(long)b * V + v];
    sum = wave_sum(sum);
    for (int v = lane; v < V; v += 64)
        dlogits[(long)b * ldd + v] = dlogp[(long)b * V + v] - expf(logp[(long)b * V + v]) * sum;
}
SLNLP_ZKERNEL(lsm_bwd_kernel, 256, lsm_bwd_body)

int lsm_nll(const float* logits, int64_t ld_logits, const int64_t* y, int B, int V, int64_t ignore_index, float* logp,
            float* loss, float* dlogits, int64_t ld_dlogits, float* row_scratch, hipStream_t st, hipStream_t loss_st,
            float* logp2, const int* logp2_row, float* loss_hist, const int* hist_idx) {
    SLNLP_CHECK_ARG(logits && y && logp && loss && row_scratch, "lsm_nll: null pointer");
    SLNLP_CHECK_ARG(B > 0 && B <= LOSS_MAXB && V > 0 && ld_logits >= V, "lsm_nll: bad shape B=%d V=%d", B, V);
    SLNLP_CHECK_ARG(!dlogits || ld_dlogits >= V, "lsm_nll: ld_dlogits too small");
    SLNLP_TRY(zlaunch(lsm_nll_kernel, dim3(B), 64, 0, st, "lsm_nll",
                      logits, (long)ld_logits, (const long*)y, B, V, (long)ignore_index, logp, row_scratch, dlogits, (long)ld_dlogits,
                      logp2, logp2_row));
    SLNLP_TRY(zlaunch(lsm_loss_reduce_kernel, dim3(1), 64, 0, loss_st ? loss_st : st, "lsm_loss_reduce",
                      row_scratch, B, loss, loss_hist, hist_idx));
    return 0;
}

int lsm_bwd(const float* logp, const float* dlogp, int B, int V, float* dlogits, int64_t ld_dlogits, hipStream_t st) {
    SLNLP_CHECK_ARG(logp && dlogp && dlogits && B > 0 && V > 0 && ld_dlogits >= V, "lsm_bwd: bad args");
    SLNLP_TRY(zlaunch(lsm_bwd_kernel, dim3(ceil_div(B, 4)), 256, 0, st, "lsm_bwd",
                      logp, dlogp, B, V, dlogits, (long)ld_dlogits));
    return 0;
}

// ================================================================= optimizer
// clip_grad_norm_(max_norm) + torch.optim.SGD(momentum) over one flat arena.
constexpr int OPT_BLOCKS = 1024;

__device__ __forceinline__ void sumsq_body(const float* __restrict__ g, long n4, float* __restrict__ partials) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)OPT_BLOCKS * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
SLNLP_ZKERNEL(sumsq_kernel, 256, sumsq_body)

__device__ __forceinline__ void sgd_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  long n4, const float* __restrict__ lr_dev, float momentum,
                                                  float max_norm, const float* __restrict__ partials,
                                                  float* __restrict__ norm_out, unsigned long long* __restrict__ rng,
                                                  PlaneOut wp, long wp_begin4, long wp_end4) {
    // wp (optional): the updated weights also leave as bf16 hi / lo planes (same offsets as the arena) -- the operand
    // form the plane GEMMs of the NEXT step stage by LDS-DMA -- instead of a separate pass that re-reads the arena.
    // Only float4 indices in [wp_begin4, wp_end4) are written: the plan passes the range of the weights that FEED plane GEMMs
    // (the encoder layers: a third of a Transformer's parameters), the rest of the plane arena has no reader
    __shared__ float red[4];
    // every block re-derives the total in the same fixed order: deterministic, no third launch
    float s = 0.f;
    for (int i = threadIdx.x; i < OPT_BLOCKS; i += 256) s += partials[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
    const float lr = lr_dev[0];
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 b = reinterpret_cast<float4*>(buf)[i];
        float4 w = reinterpret_cast<float4*>(p)[i];
        b.x = momentum * b.x + gv.x * coef; b.y = momentum * b.y + gv.y * coef;
        b.z = momentum * b.z + gv.z * coef; b.w = momentum * b.w + gv.w * coef;
        w.x -= lr * b.x; w.y -= lr * b.y; w.z -= lr * b.z; w.w -= lr * b.w;
        reinterpret_cast<float4*>(buf)[i] = b;
        reinterpret_cast<float4*>(p)[i] = w;
        if (i >= wp_begin4 && i < wp_end4) store_planes4(wp, i * 4, w);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (norm_out) norm_out[0] = norm;
        if (rng) rng[1] += 1ull;
    }
}
SLNLP_ZKERNEL(sgd_kernel, 256, sgd_body)

int clip_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t n, const float* lr_dev,
                  float momentum, float max_norm, float* partials, float* norm_out, unsigned long long* rng,
                  hipStream_t st, PlaneOut wp, int64_t wp_begin, int64_t wp_end) {
    SLNLP_CHECK_ARG(params && grads && momentum_buf && lr_dev && partials, "clip_sgd_step: null pointer");
    SLNLP_CHECK_ARG(n > 0 && n % 4 == 0, "clip_sgd_step: n=%ld must be a positive multiple of 4", (long)n);
    SLNLP_CHECK_ARG(((uintptr_t)params & 15) == 0 && ((uintptr_t)grads & 15) == 0 && ((uintptr_t)momentum_buf & 15) == 0,
                    "clip_sgd_step: arenas must be 16-byte aligned");
    SLNLP_TRY(zlaunch(sumsq_kernel, dim3(OPT_BLOCKS), 256, 0, st, "sumsq",
                      grads, (long)(n / 4), partials));
    int grid = ceil_div(n / 4, 256);
    if (grid > 2048) grid = 2048;
    SLNLP_TRY(zlaunch(sgd_kernel, dim3(grid), 256, 0, st, "sgd",
                      params, grads, momentum_buf, (long)(n / 4), lr_dev, momentum, max_norm, partials, norm_out, rng, wp,
                      (long)(wp_begin / 4), (long)(wp_end < 0 ? n / 4 : (wp_end + 3) / 4)));
    return 0;
}

// torch.optim.Adam (amsgrad False, maximize False) fused with clip_grad_norm_, same two-launch shape as clip + SGD:
//   g' = g * clip_coef (+ weight_decay * p);  m += (1 - b1)(g' - m);  v = b2 v + (1 - b2) g'^2;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)            (torch/optim/adam.py _single_tensor_adam)
// The step count t lives in device memory (step_f[0], a float: exact to 2^24 steps) and is advanced here, so a captured
// or recorded step needs no host-side argument that changes per step.
__device__ __forceinline__ void adam_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                          float* __restrict__ v, long n4, const float* __restrict__ lr_dev, float beta1,
                                          float beta2, float eps, float weight_decay, float max_norm,
                                          const float* __restrict__ partials, float* __restrict__ norm_out,
                                          unsigned long long* __restrict__ rng, float* __restrict__ step_f, PlaneOut wp,
                                          long wp_begin4, long wp_end4) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < OPT_BLOCKS; i += 256) s += partials[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    float coef = 1.f;
    if (max_norm > 0.f) coef = fminf(max_norm / (norm + 1e-6f), 1.f);
    const float lr = lr_dev[0];
    const float t = step_f[0] + 1.f;                       // every block reads the OLD count (adam_count_kernel advances it afterwards)
    const float bc1 = 1.f - powf(beta1, t), bc2 = 1.f - powf(beta2, t);
    const float step_size = lr / bc1, rsq_bc2 = 1.f / sqrtf(bc2);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i], w = reinterpret_cast<float4*>(p)[i];
        float ge[4] = {gv.x * coef, gv.y * coef, gv.z * coef, gv.w * coef};
        float me[4] = {mm.x, mm.y, mm.z, mm.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w}, we[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (weight_decay != 0.f) ge[e] += weight_decay * we[e];
            me[e] += (1.f - beta1) * (ge[e] - me[e]);
            ve[e] = beta2 * ve[e] + (1.f - beta2) * ge[e] * ge[e];
            we[e] -= step_size * (me[e] / (sqrtf(ve[e]) * rsq_bc2 + eps));
        }
        reinterpret_cast<float4*>(m)[i] = make_float4(me[0], me[1], me[2], me[3]);
        reinterpret_cast<float4*>(v)[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
        const float4 wn = make_float4(we[0], we[1], we[2], we[3]);
        reinterpret_cast<float4*>(p)[i] = wn;
        if (i >= wp_begin4 && i < wp_end4) store_planes4(wp, i * 4, wn);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (norm_out) norm_out[0] = norm;
        if (rng) rng[1] += 1ull;
    }
}
SLNLP_ZKERNEL(adam_kernel, 256, adam_body)

// the count is advanced by its own one-thread launch AFTER the update (every block of adam_kernel must read the same old value)
__device__ __forceinline__ void adam_count_body(float* __restrict__ step_f) { if (threadIdx.x == 0 && blockIdx.x == 0) step_f[0] += 1.f; }
SLNLP_ZKERNEL(adam_count_kernel, 64, adam_count_body)

int clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr_dev,
                   float beta1, float beta2, float eps, float weight_decay, float max_norm, float* partials, float* norm_out,
                   unsigned long long* rng, float* step_f, hipStream_t st, PlaneOut wp, int64_t wp_begin, int64_t wp_end) {
    SLNLP_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && lr_dev && partials && step_f, "clip_adam_step: null pointer");
    SLNLP_CHECK_ARG(n > 0 && n % 4 == 0, "clip_adam_step: n=%ld must be a positive multiple of 4", (long)n);
    SLNLP_CHECK_ARG((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                    "clip_adam_step: arenas must be 16-byte aligned");
    SLNLP_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "clip_adam_step: bad betas / eps");
    SLNLP_TRY(zlaunch(sumsq_kernel, dim3(OPT_BLOCKS), 256, 0, st, "sumsq", grads, (long)(n / 4), partials));
    int grid = ceil_div(n / 4, 256);
    if (grid > 2048) grid = 2048;
    SLNLP_TRY(zlaunch(adam_kernel, dim3(grid), 256, 0, st, "adam", params, grads, exp_avg, exp_avg_sq, (long)(n / 4), lr_dev, beta1, beta2,
                      eps, weight_decay, max_norm, partials, norm_out, rng, step_f, wp, (long)(wp_begin / 4),
                      (long)(wp_end < 0 ? n / 4 : (wp_end + 3) / 4)));
    SLNLP_TRY(zlaunch(adam_count_kernel, dim3(1), 64, 0, st, "adam_count", step_f));
    return 0;
}

// ============================================================== dropout mask
__global__ void dropout_mask_kernel(float* out, int R, int C, unsigned thr, int site, const unsigned long long* rng) {
    const long total = (long)R * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const unsigned r = (unsigned)(i / C), c = (unsigned)(i % C);
        out[i] = dropout_keep(rng, site, r, c, thr) ? 1.f : 0.f;
    }
}

}  // namespace slnlp

extern "C" {
int slnlp_embed_fwd(const int64_t* ids, int64_t ld_ids, int B, int S, int E, int V, const float* table,
                    const float* pe, float* out, float scale, float drop_p, int drop_site,
                    const unsigned long long* rng, int64_t nan_idx, void* stream) {
    return slnlp::embed_fwd(ids, ld_ids, B, S, E, V, table, pe, out, scale, drop_p, drop_site, rng, nan_idx,
                            (hipStream_t)stream);
}
int64_t slnlp_embed_bwd_scratch_bytes(int B, int S, int E) { return (int64_t)slnlp::embed_bwd_scratch_bytes(B, S, E); }
int slnlp_embed_bwd(const int64_t* ids, int64_t ld_ids, int B, int S, int E, int V, const float* dx, float* dtable,
                    float scale, int64_t zero_row, float drop_p, int drop_site, const unsigned long long* rng,
                    void* scratch, void* stream) {
    return slnlp::embed_bwd(ids, ld_ids, B, S, E, V, dx, dtable, scale, zero_row, drop_p, drop_site, rng, scratch,
                            (hipStream_t)stream);
}
int slnlp_layernorm_fwd(const float* x, const float* gamma, const float* beta, int rows, int E, float eps, float* y,
                        float* stats, void* stream) {
    return slnlp::layernorm_fwd(x, gamma, beta, rows, E, eps, y, stats, (hipStream_t)stream);
}
int slnlp_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* stats, int rows, int E,
                        const float* add_to_dx, float* dx, float* dx_drop, float drop_p, int drop_site,
                        const unsigned long long* rng, float* partial, int* nblk_out, void* stream) {
    return slnlp::layernorm_bwd(dy, x, gamma, stats, rows, E, add_to_dx, dx, dx_drop, drop_p, drop_site, rng, partial,
                                nblk_out, 0, (hipStream_t)stream);
}
int slnlp_ln_param_reduce(const slnlp_ln_reduce_entry* table_dev, int n, int max_E, void* stream) {
    return slnlp::ln_param_reduce(table_dev, n, max_E, (hipStream_t)stream);
}
int slnlp_lsm_nll(const float* logits, int64_t ld_logits, const int64_t* y, int B, int V, int64_t ignore_index,
                  float* logp, float* loss, float* dlogits, int64_t ld_dlogits, float* row_scratch, void* stream) {
    return slnlp::lsm_nll(logits, ld_logits, y, B, V, ignore_index, logp, loss, dlogits, ld_dlogits, row_scratch,
                          (hipStream_t)stream, nullptr);
}
int slnlp_lsm_bwd(const float* logp, const float* dlogp, int B, int V, float* dlogits, int64_t ld_dlogits,
                  void* stream) {
    return slnlp::lsm_bwd(logp, dlogp, B, V, dlogits, ld_dlogits, (hipStream_t)stream);
}
int slnlp_clip_sgd_step(float* params, const float* grads, float* momentum_buf, int64_t n, const float* lr_dev,
                        float momentum, float max_norm, float* partials, float* norm_out, unsigned long long* rng,
                        void* stream) {
    return slnlp::clip_sgd_step(params, grads, momentum_buf, n, lr_dev, momentum, max_norm, partials, norm_out, rng,
                                (hipStream_t)stream);
}
int slnlp_clip_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* lr_dev,
                         float beta1, float beta2, float eps, float weight_decay, float max_norm, float* partials, float* norm_out,
                         float* step_count, void* stream) {
    return slnlp::clip_adam_step(params, grads, exp_avg, exp_avg_sq, n, lr_dev, beta1, beta2, eps, weight_decay, max_norm, partials,
                                 norm_out, nullptr, step_count, (hipStream_t)stream);
}
int slnlp_dropout_mask(float* out, int R, int C, float p, int site, const unsigned long long* rng, void* stream) {
    if (!out || !rng || R <= 0 || C <= 0 || p < 0.f || p >= 1.f) {
        slnlp::set_error("slnlp_dropout_mask: bad args");
        return SLNLP_ERR_INVALID_ARG;
    }
    long total = (long)R * C;
    int grid = slnlp::ceil_div(total, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(slnlp::dropout_mask_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, out, R, C,
                       slnlp::dropout_threshold(p), site, rng);
    return hipGetLastError() == hipSuccess ? 0 : SLNLP_ERR_LAUNCH;
}
}
