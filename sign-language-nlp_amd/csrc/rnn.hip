// rnn.hip -- point-wise LSTM / GRU cell (forward + backward, length-masked = packed-sequence
// semantics) and the single-step Bahdanau attention of the reference's
// EncoderDecoder{LSTM,GRU}Attn (/root/reference/model/base/encoder_decoder_attn_bkp.py).
//
// The matmuls around the cell (x W_ih^T for all timesteps at once, h_{t-1} W_hh^T per step, and
// every dgrad / wgrad) run on the MFMA GEMM (gemm.hip); these kernels are the fp32 glue between
// them.  Gate order follows torch.nn.LSTM (i,f,g,o) / torch.nn.GRU (r,z,n), which the reference
// instantiates at bkp.py:95-100,186-190.
#include "launch.hpp"

namespace slnlp {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------- forward
// One launch = one timestep of up to two directions (blockIdx.y).  Sequence b advances only while
// t < lengths[b] (pack_padded_sequence, bkp.py:110-114): otherwise state is carried and the layer
// output is `fill` (pad_packed_sequence(padding_value=pad_idx), bkp.py:120-123).
template <bool LSTM>
__device__ __forceinline__ void rnn_cell_fwd_body(slnlp_rnn_cell_dir d0, slnlp_rnn_cell_dir d1, int B, int Hd,
                                                  const long* __restrict__ lengths, float fill, long ld_out, float drop_p,
                                                  unsigned drop_thr, int drop_site, const unsigned long long* __restrict__ rng) {
    const slnlp_rnn_cell_dir d = blockIdx.y == 0 ? d0 : d1;
    const int G = LSTM ? 4 : 3;
    const long n = (long)B * Hd;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int b = (int)(idx / Hd), j = (int)(idx % Hd);
        const bool valid = lengths ? (d.t < lengths[b]) : true;
        const float* xp = d.xproj + (long)b * G * Hd;
        const float* hp = d.hproj + (long)b * G * Hd;
        const float hprev = d.h[idx];
        float hnew;
        if (LSTM) {
            const float cprev = d.c[idx];
            const float gi = sigmoidf_(xp[j] + hp[j]);
            const float gf = sigmoidf_(xp[Hd + j] + hp[Hd + j]);
            const float gg = tanhf(xp[2 * Hd + j] + hp[2 * Hd + j]);
            const float go = sigmoidf_(xp[3 * Hd + j] + hp[3 * Hd + j]);
            const float cnew = gf * cprev + gi * gg;
            hnew = go * tanhf(cnew);
            float* a = d.acts + (long)b * G * Hd;
            a[j] = gi; a[Hd + j] = gf; a[2 * Hd + j] = gg; a[3 * Hd + j] = go;
            d.cprev_save[idx] = cprev;
            d.c[idx] = valid ? cnew : cprev;
        } else {
            const float hn = hp[2 * Hd + j];
            const float r = sigmoidf_(xp[j] + hp[j]);
            const float z = sigmoidf_(xp[Hd + j] + hp[Hd + j]);
            const float nn = tanhf(xp[2 * Hd + j] + r * hn);
            hnew = (1.f - z) * nn + z * hprev;
            float* a = d.acts + (long)b * G * Hd;
            a[j] = r; a[Hd + j] = z; a[2 * Hd + j] = nn;
            d.hn_save[idx] = hn;
        }
        d.hprev_save[idx] = hprev;
        d.h[idx] = valid ? hnew : hprev;
        if (d.out) {
            float o = valid ? hnew : fill;
            if (drop_p > 0.f && valid)
                o = dropout_keep(rng, drop_site, (unsigned)(d.out_row0 + b), (unsigned)(d.out_col0 + j), drop_thr)
                        ? o / (1.f - drop_p) : 0.f;
            d.out[(long)b * ld_out + j] = o;
        }
    }
}

__device__ __forceinline__ void rnn_cell_fwd_lstm_body(slnlp_rnn_cell_dir d0, slnlp_rnn_cell_dir d1, int B, int Hd, const long* lengths,
                                                       float fill, long ld_out, float drop_p, unsigned drop_thr, int drop_site,
                                                       const unsigned long long* rng) {
    rnn_cell_fwd_body<true>(d0, d1, B, Hd, lengths, fill, ld_out, drop_p, drop_thr, drop_site, rng);
}
__device__ __forceinline__ void rnn_cell_fwd_gru_body(slnlp_rnn_cell_dir d0, slnlp_rnn_cell_dir d1, int B, int Hd, const long* lengths,
                                                      float fill, long ld_out, float drop_p, unsigned drop_thr, int drop_site,
                                                      const unsigned long long* rng) {
    rnn_cell_fwd_body<false>(d0, d1, B, Hd, lengths, fill, ld_out, drop_p, drop_thr, drop_site, rng);
}
SLNLP_ZKERNEL(rnn_cell_fwd_lstm_kernel, 256, rnn_cell_fwd_lstm_body)
SLNLP_ZKERNEL(rnn_cell_fwd_gru_kernel, 256, rnn_cell_fwd_gru_body)

// ------------------------------------------------------------------ backward
// dh_state / dc_state are the running gradients w.r.t. the state AFTER step t; the kernel emits the
// gate gradients of step t and the part of dh that bypasses the recurrent matmul (`carry`); the
// caller then forms dh_state(t-1) = dgh_t W_hh + carry with one GEMM.
template <bool LSTM>
__device__ __forceinline__ void rnn_cell_bwd_body(slnlp_rnn_cell_bwd_dir d0, slnlp_rnn_cell_bwd_dir d1, int B, int Hd,
                                                  const long* __restrict__ lengths, long ld_dout, float drop_p, unsigned drop_thr,
                                                  int drop_site, const unsigned long long* __restrict__ rng) {
    const slnlp_rnn_cell_bwd_dir d = blockIdx.y == 0 ? d0 : d1;
    const int G = LSTM ? 4 : 3;
    const long n = (long)B * Hd;
    for (long idx = blockIdx.x * 256L + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const int b = (int)(idx / Hd), j = (int)(idx % Hd);
        // ---- every operand is requested before the first one is used: ONE memory round trip per element.  (Rounds 1-4 read the length,
        // branched on it, then read the gate activations, then the saved states: three to four dependent round trips in a kernel that
        // runs 192 times per configs[2] step on the backward chain.)  All of them exist for masked timesteps too.
        const long len = lengths ? lengths[b] : 0;
        float dh = d.dh_state[idx];
        float x[2][8];
#pragma unroll
        for (int e0 = 0; e0 < 2; ++e0)
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e0][e] = 8 * e0 + e < d.n_extra ? d.dh_extra[(8 * e0 + e) * d.extra_stride + idx] : 0.f;
        float g_out = d.dout ? d.dout[(long)b * ld_dout + j] : 0.f;
        const float* a = d.acts + (long)b * G * Hd;
        float a0 = a[j], a1 = a[Hd + j], a2 = a[2 * Hd + j], a3 = LSTM ? a[3 * Hd + j] : 0.f;
        float s0 = LSTM ? d.cprev_save[idx] : d.hprev_save[idx], s1 = LSTM ? d.dc_state[idx] : d.hn_save[idx];
        asm volatile("" : "+v"(dh), "+v"(g_out), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(s0), "+v"(s1));    // (keeps the loads above the branches)
        const bool valid = lengths ? (d.t < len) : true;
        for (int e0 = 0; e0 < 2 && 8 * e0 < d.n_extra; ++e0) {        // fixed order (the K-slices of the recurrent data gradient: 3, or up to 15)
#pragma unroll
            for (int e = 0; e < 8; ++e) dh += x[e0][e];
        }
        for (int e = 16; e < d.n_extra; ++e) dh += d.dh_extra[e * d.extra_stride + idx];
        float* gx = d.dgx + (long)b * G * Hd;
        float* gh = LSTM ? gx : d.dgh + (long)b * G * Hd;
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
            float g = g_out;
            if (drop_p > 0.f)
                g = dropout_keep(rng, drop_site, (unsigned)(d.out_row0 + b), (unsigned)(d.out_col0 + j), drop_thr)
                        ? g / (1.f - drop_p) : 0.f;
            dh += g;
        }
        if (LSTM) {
            const float gi = a0, gf = a1, gg = a2, go = a3;
            const float cprev = s0;
            const float tc = tanhf(gf * cprev + gi * gg);
            const float dc = s1 + dh * go * (1.f - tc * tc);
            gx[j] = dc * gg * gi * (1.f - gi);
            gx[Hd + j] = dc * cprev * gf * (1.f - gf);
            gx[2 * Hd + j] = dc * gi * (1.f - gg * gg);
            gx[3 * Hd + j] = dh * tc * go * (1.f - go);
            d.dc_state[idx] = dc * gf;
            d.carry[idx] = 0.f;
        } else {
            const float r = a0, z = a1, nn = a2;
            const float hprev = s0, hn = s1;
            const float dn_pre = dh * (1.f - z) * (1.f - nn * nn);
            const float dr_pre = dn_pre * hn * r * (1.f - r);
            const float dz_pre = dh * (hprev - nn) * z * (1.f - z);
            gx[j] = dr_pre; gx[Hd + j] = dz_pre; gx[2 * Hd + j] = dn_pre;
            gh[j] = dr_pre; gh[Hd + j] = dz_pre; gh[2 * Hd + j] = dn_pre * r;
            d.carry[idx] = dh * z;
        }
    }
}

__device__ __forceinline__ void rnn_cell_bwd_lstm_body(slnlp_rnn_cell_bwd_dir d0, slnlp_rnn_cell_bwd_dir d1, int B, int Hd,
                                                       const long* lengths, long ld_dout, float drop_p, unsigned drop_thr, int drop_site,
                                                       const unsigned long long* rng) {
    rnn_cell_bwd_body<true>(d0, d1, B, Hd, lengths, ld_dout, drop_p, drop_thr, drop_site, rng);
}
__device__ __forceinline__ void rnn_cell_bwd_gru_body(slnlp_rnn_cell_bwd_dir d0, slnlp_rnn_cell_bwd_dir d1, int B, int Hd,
                                                      const long* lengths, long ld_dout, float drop_p, unsigned drop_thr, int drop_site,
                                                      const unsigned long long* rng) {
    rnn_cell_bwd_body<false>(d0, d1, B, Hd, lengths, ld_dout, drop_p, drop_thr, drop_site, rng);
}
SLNLP_ZKERNEL(rnn_cell_bwd_lstm_kernel, 256, rnn_cell_bwd_lstm_body)
SLNLP_ZKERNEL(rnn_cell_bwd_gru_kernel, 256, rnn_cell_bwd_gru_body)

// ------------------------------------------------------------------ Bahdanau
// bkp.py:304-327 with one query per sequence: scores[s] = w_e . tanh(q + proj_key[s]); positions
// where the source token is <pad> are masked (bkp.py:404-406); softmax; context = alphas . value.
// One workgroup per sequence b; rows of proj_key / value are time-major (m = s*B + b).
constexpr int BAH_MAXS = 2048;   // source positions per sequence held in LDS (the softmax walks them 64 at a time)

__device__ __forceinline__ void bahdanau_fwd_body(const float* __restrict__ q, const float* __restrict__ pk,
                                                  const float* __restrict__ val, const float* __restrict__ we,
                                                  const long* __restrict__ ids, long ld_ids, long pad, int B, int S, int Hd,
                                                  float* __restrict__ alphas, float* __restrict__ ctx) {
    __shared__ float sc[BAH_MAXS];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s = wave; s < S; s += 4) {
        const float* pr = pk + ((long)s * B + b) * Hd;
        float a = 0.f;
        for (int j = lane; j < Hd; j += 64) a += we[j] * tanhf(q[(long)b * Hd + j] + pr[j]);
        a = wave_sum(a);
        if (lane == 0) sc[s] = (ids[(long)b * ld_ids + s] == pad) ? -INFINITY : a;
    }
    __syncthreads();
    if (wave == 0) {                     // softmax over the S scores, 64 at a time (S <= 64: one trip, one value per lane)
        float m = -INFINITY;
        for (int s = lane; s < S; s += 64) m = fmaxf(m, sc[s]);
        m = wave_max(m);
        float e = 0.f;
        for (int s = lane; s < S; s += 64) e += expf(sc[s] - m);
        const float tot = wave_sum(e);
        for (int s = lane; s < S; s += 64) {
            const float p = expf(sc[s] - m) / tot;
            sc[s] = p;
            alphas[(long)b * S + s] = p;
        }
    }
    __syncthreads();
    const int V2 = 2 * Hd;
    for (int c = threadIdx.x; c < V2; c += 256) {
        float a = 0.f;
#pragma unroll 8
        for (int s = 0; s < S; ++s) a += sc[s] * val[((long)s * B + b) * V2 + c];   // 8 independent loads in flight per trip
        ctx[(long)b * V2 + c] = a;
    }
}

SLNLP_ZKERNEL(bahdanau_fwd_kernel, 256, bahdanau_fwd_body)

// dctx [B,2Hd] -> dq [B,Hd], dpk [S*B,Hd], dval [S*B,2Hd] (written, not accumulated), dwe_part [B,Hd]
__device__ __forceinline__ void bahdanau_bwd_body(const float* __restrict__ q, const float* __restrict__ pk,
                                                  const float* __restrict__ val, const float* __restrict__ we,
                                                  const float* __restrict__ alphas, const float* __restrict__ dctx, int B, int S,
                                                  int Hd, float* __restrict__ dq, float* __restrict__ dpk,
                                                  float* __restrict__ dval, float* __restrict__ dwe_part) {
    __shared__ float al[BAH_MAXS], dsc[BAH_MAXS];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, V2 = 2 * Hd;
    for (int s = threadIdx.x; s < S; s += 256) al[s] = alphas[(long)b * S + s];
    __syncthreads();
    // dalpha[s] = dctx . value[s];  dvalue[s] = alpha[s] * dctx
    for (int s = wave; s < S; s += 4) {
        const long row = ((long)s * B + b) * V2;
        float a = 0.f;
        for (int c = lane; c < V2; c += 64) {
            const float g = dctx[(long)b * V2 + c];
            a += g * val[row + c];
            dval[row + c] = al[s] * g;
        }
        a = wave_sum(a);
        if (lane == 0) dsc[s] = a;
    }
    __syncthreads();
    if (wave == 0) {  // softmax backward
        float part = 0.f;
        for (int s = lane; s < S; s += 64) part += al[s] * dsc[s];
        const float dot = wave_sum(part);
        for (int s = lane; s < S; s += 64) dsc[s] = al[s] * (dsc[s] - dot);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < Hd; j += 256) {
        const float qj = q[(long)b * Hd + j], wj = we[j];
        float accq = 0.f, accw = 0.f;
#pragma unroll 8
        for (int s = 0; s < S; ++s) {
            const long row = ((long)s * B + b) * Hd;
            const float u = tanhf(qj + pk[row + j]);
            const float dpre = dsc[s] * wj * (1.f - u * u);
            dpk[row + j] = dpre;
            accq += dpre;
            accw += dsc[s] * u;
        }
        dq[(long)b * Hd + j] = accq;
        dwe_part[(long)b * Hd + j] = accw;
    }
}

SLNLP_ZKERNEL(bahdanau_bwd_kernel, 256, bahdanau_bwd_body)

// out[c] = sum_r in[r, c]   (fixed order; tiny)
__device__ __forceinline__ void colsum_body(const float* __restrict__ in, int R, int C, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f;
    for (int r = 0; r < R; ++r) a += in[(long)r * C + c];
    out[c] = a;
}

SLNLP_ZKERNEL(colsum_kernel, 256, colsum_body)

// out[r, :] (+)= in[r, :] for strided row blocks (final-state gather, gradient adds)
__device__ __forceinline__ void add_rows_body(const float* __restrict__ in, long ld_in, float* __restrict__ out, long ld_out, int R,
                                              int C, int accumulate) {
    const long n = (long)R * C;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int r = (int)(i / C), c = (int)(i % C);
        const float v = in[(long)r * ld_in + c];
        float* o = out + (long)r * ld_out + c;
        *o = accumulate ? *o + v : v;
    }
}

SLNLP_ZKERNEL(add_rows_kernel, 256, add_rows_body)

// out = dy * (1 - y^2)   (backward of y = tanh(z); the bridge, bkp.py:268-280)
__device__ __forceinline__ void tanh_bwd_body(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = dy[i] * (1.f - y[i] * y[i]);
}
SLNLP_ZKERNEL(tanh_bwd_kernel, 256, tanh_bwd_body)

int tanh_bwd(const float* dy, const float* y, float* out, int64_t n, hipStream_t st) {
    SLNLP_CHECK_ARG(dy && y && out && n > 0, "tanh_bwd: bad args");
    int gx = ceil_div(n, 256);
    if (gx > 1024) gx = 1024;
    return zlaunch(tanh_bwd_kernel, dim3(gx), 256, 0, st, "tanh_bwd", dy, y, out, (long)n);
}

int rnn_cell_fwd(int lstm, const slnlp_rnn_cell_dir* dirs, int ndir, int B, int Hd, const int64_t* lengths, float fill,
                 int64_t ld_out, float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st) {
    SLNLP_CHECK_ARG(dirs && (ndir == 1 || ndir == 2) && B > 0 && Hd > 0, "rnn_cell_fwd: bad args");
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "rnn_cell_fwd: bad dropout args");
    for (int k = 0; k < ndir; ++k)
        SLNLP_CHECK_ARG(dirs[k].xproj && dirs[k].hproj && dirs[k].h && dirs[k].hprev_save && dirs[k].acts &&
                            (lstm ? (dirs[k].c && dirs[k].cprev_save) : (dirs[k].hn_save != nullptr)),
                        "rnn_cell_fwd: null pointer in direction %d", k);
    int gx = ceil_div((long)B * Hd, 256);
    if (gx > 1024) gx = 1024;
    const slnlp_rnn_cell_dir d1 = dirs[ndir - 1];
    return zlaunch(lstm ? rnn_cell_fwd_lstm_kernel : rnn_cell_fwd_gru_kernel, dim3(gx, ndir), 256, 0, st, "rnn_cell_fwd", dirs[0], d1, B, Hd,
                   (const long*)lengths, fill, (long)ld_out, drop_p, dropout_threshold(drop_p), drop_site, rng);
}

int rnn_cell_bwd(int lstm, const slnlp_rnn_cell_bwd_dir* dirs, int ndir, int B, int Hd, const int64_t* lengths,
                 int64_t ld_dout, float drop_p, int drop_site, const unsigned long long* rng, hipStream_t st) {
    SLNLP_CHECK_ARG(dirs && (ndir == 1 || ndir == 2) && B > 0 && Hd > 0, "rnn_cell_bwd: bad args");
    SLNLP_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), "rnn_cell_bwd: bad dropout args");
    for (int k = 0; k < ndir; ++k)
        SLNLP_CHECK_ARG(dirs[k].dh_state && dirs[k].carry && dirs[k].dgx && dirs[k].acts &&
                            (lstm ? (dirs[k].dc_state && dirs[k].cprev_save)
                                  : (dirs[k].dgh && dirs[k].hprev_save && dirs[k].hn_save)),
                        "rnn_cell_bwd: null pointer in direction %d", k);
    int gx = ceil_div((long)B * Hd, 256);
    if (gx > 1024) gx = 1024;
    const slnlp_rnn_cell_bwd_dir d1 = dirs[ndir - 1];
    return zlaunch(lstm ? rnn_cell_bwd_lstm_kernel : rnn_cell_bwd_gru_kernel, dim3(gx, ndir), 256, 0, st, "rnn_cell_bwd", dirs[0], d1, B, Hd,
                   (const long*)lengths, (long)ld_dout, drop_p, dropout_threshold(drop_p), drop_site, rng);
}

int bahdanau_fwd(const float* q, const float* pk, const float* val, const float* we, const int64_t* ids,
                 int64_t ld_ids, int64_t pad, int B, int S, int Hd, float* alphas, float* ctx, hipStream_t st) {
    SLNLP_CHECK_ARG(q && pk && val && we && ids && alphas && ctx, "bahdanau_fwd: null pointer");
    SLNLP_CHECK_ARG(B > 0 && S > 0 && S <= BAH_MAXS && Hd > 0, "bahdanau_fwd: S=%d outside 1..%d", S, BAH_MAXS);
    return zlaunch(bahdanau_fwd_kernel, dim3(B), 256, 0, st, "bahdanau_fwd", q, pk, val, we, (const long*)ids, (long)ld_ids, (long)pad, B, S,
                   Hd, alphas, ctx);
}

int bahdanau_bwd(const float* q, const float* pk, const float* val, const float* we, const float* alphas,
                 const float* dctx, int B, int S, int Hd, float* dq, float* dpk, float* dval, float* dwe_part,
                 float* dwe, hipStream_t st) {
    SLNLP_CHECK_ARG(q && pk && val && we && alphas && dctx && dq && dpk && dval && dwe_part && dwe,
                    "bahdanau_bwd: null pointer");
    SLNLP_CHECK_ARG(B > 0 && S > 0 && S <= BAH_MAXS && Hd > 0, "bahdanau_bwd: S=%d outside 1..%d", S, BAH_MAXS);
    SLNLP_TRY(zlaunch(bahdanau_bwd_kernel, dim3(B), 256, 0, st, "bahdanau_bwd", q, pk, val, we, alphas, dctx, B, S, Hd, dq, dpk, dval, dwe_part));
    return zlaunch(colsum_kernel, dim3(ceil_div(Hd, 256)), 256, 0, st, "colsum", (const float*)dwe_part, B, Hd, dwe);
}

int add_rows(const float* in, int64_t ld_in, float* out, int64_t ld_out, int R, int C, int accumulate, hipStream_t st) {
    SLNLP_CHECK_ARG(in && out && R > 0 && C > 0, "add_rows: bad args");
    int gx = ceil_div((long)R * C, 256);
    if (gx > 1024) gx = 1024;
    return zlaunch(add_rows_kernel, dim3(gx), 256, 0, st, "add_rows", in, (long)ld_in, out, (long)ld_out, R, C, accumulate);
}

}  // namespace slnlp

extern "C" {
int slnlp_rnn_cell_fwd(int lstm, const slnlp_rnn_cell_dir* dirs, int ndir, int B, int Hd, const int64_t* lengths,
                       float fill, int64_t ld_out, float drop_p, int drop_site, const unsigned long long* rng,
                       void* stream) {
    return slnlp::rnn_cell_fwd(lstm, dirs, ndir, B, Hd, lengths, fill, ld_out, drop_p, drop_site, rng,
                               (hipStream_t)stream);
}
int slnlp_rnn_cell_bwd(int lstm, const slnlp_rnn_cell_bwd_dir* dirs, int ndir, int B, int Hd, const int64_t* lengths,
                       int64_t ld_dout, float drop_p, int drop_site, const unsigned long long* rng, void* stream) {
    return slnlp::rnn_cell_bwd(lstm, dirs, ndir, B, Hd, lengths, ld_dout, drop_p, drop_site, rng, (hipStream_t)stream);
}
int slnlp_bahdanau_fwd(const float* q, const float* pk, const float* val, const float* we, const int64_t* ids,
                       int64_t ld_ids, int64_t pad, int B, int S, int Hd, float* alphas, float* ctx, void* stream) {
    return slnlp::bahdanau_fwd(q, pk, val, we, ids, ld_ids, pad, B, S, Hd, alphas, ctx, (hipStream_t)stream);
}
int slnlp_bahdanau_bwd(const float* q, const float* pk, const float* val, const float* we, const float* alphas,
                       const float* dctx, int B, int S, int Hd, float* dq, float* dpk, float* dval, float* dwe_part,
                       float* dwe, void* stream) {
    return slnlp::bahdanau_bwd(q, pk, val, we, alphas, dctx, B, S, Hd, dq, dpk, dval, dwe_part, dwe,
                               (hipStream_t)stream);
}
}
