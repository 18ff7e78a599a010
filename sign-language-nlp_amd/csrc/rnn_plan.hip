// rnn_plan.hip -- host-side plan for model.EncoderDecoder{LSTM,GRU}Attn.
//
// Drop-in target: /root/reference/model/base/encoder_decoder_attn_bkp.py
//   EncoderDecoderAttnBaseBkp.forward :388-402, Encoder :102-132 (packed bidirectional RNN, padded
//   outputs = float(pad_idx)), Decoder.forward/forward_step/init_hidden :202-280 with exactly ONE
//   step (MAX_OUTPUT_LEN = 1, :332), BahdanauAttention :304-327, Generator :69-76 -- and the skorch
//   step around it.  The generator consumes the decoder STATE (:40-46), so
//   pre_output_layer is a dead weight: it lives in the arena (state_dict parity) and never moves.
//
// Layout of time: every activation is time-major, row m = t*B + b, so one timestep of the
// recurrence is a contiguous [B, *] block.  x W_ih^T runs as ONE GEMM over all timesteps; the
// recurrence is per step {h W_hh^T (MFMA GEMM) ; point-wise cell for both directions}.
#include <map>
#include <string>
#include <vector>

#include "common.hpp"
#include "launch.hpp"
#include <cstring>

namespace slnlp {

static inline long align_up_r(long v, long a) { return (v + a - 1) / a * a; }

// K-slices of the recurrent data gradient (see slnlp_rnn_plan::kslices).  SLNLP_RNN_KSLICE=gate: one per gate (A / B measurements)
static int rnn_kslices(const slnlp_rnn_config& c) {
    const int G = c.lstm ? 4 : 3;
    if (!(c.Hd >= 256 && c.Hd % 4 == 0)) return 1;
    static const int len = [] { const char* e = getenv("SLNLP_RNN_KSLICE"); return e && strcmp(e, "gate") ? atoi(e) : 0; }();   // (0: one per gate)
    return (len >= 64 && c.Hd % len == 0) ? G * c.Hd / len : G;
}
static bool rnn_kslice_wide() { static const bool w = [] { const char* e = getenv("SLNLP_RNN_KSLICE"); return e && strchr(e, 'w'); }(); return w; }

struct RParam {
    std::string name;
    int64_t shape[2];
    int ndim;
    int64_t off, numel;
};
struct RnnW { long w_ih, w_hh, b_ih, b_hh; int in; };
struct RLayout {
    std::vector<RParam> ents;
    std::vector<RnnW> enc[2];   // [direction][layer]
    std::vector<RnnW> dec;
    long key_w, query_w, energy_w, bridge_w, bridge_b, pre_out, src_emb, trg_emb, gen_w, total;
};

// reference state_dict order (bkp.py:358-381: encoder, decoder(attention, rnn, bridge, pre_output_layer),
// src_embed, trg_embed, generator); every tensor on a 16-byte boundary.
static RLayout build_rlayout(const slnlp_rnn_config& c) {
    RLayout L;
    long cur = 0;
    auto add = [&](const std::string& n, long d0, long d1) -> long {
        RParam e;
        e.name = n; e.shape[0] = d0; e.shape[1] = d1; e.ndim = d1 > 0 ? 2 : 1;
        e.numel = d1 > 0 ? d0 * d1 : d0;
        e.off = cur;
        cur = align_up_r(cur + e.numel, 4);
        L.ents.push_back(e);
        return e.off;
    };
    const long G = c.lstm ? 4 : 3, Hd = c.Hd, E = c.E;
    for (int l = 0; l < c.N; ++l) {
        const int in = l == 0 ? E : 2 * Hd;
        for (int d = 0; d < 2; ++d) {
            const std::string s = "_l" + std::to_string(l) + (d ? "_reverse" : "");
            RnnW w;
            w.in = in;
            w.w_ih = add("model.encoder.rnn.weight_ih" + s, G * Hd, in);
            w.w_hh = add("model.encoder.rnn.weight_hh" + s, G * Hd, Hd);
            w.b_ih = add("model.encoder.rnn.bias_ih" + s, G * Hd, 0);
            w.b_hh = add("model.encoder.rnn.bias_hh" + s, G * Hd, 0);
            L.enc[d].push_back(w);
        }
    }
    L.key_w = add("model.decoder.attention.key_layer.weight", Hd, 2 * Hd);
    L.query_w = add("model.decoder.attention.query_layer.weight", Hd, Hd);
    L.energy_w = add("model.decoder.attention.energy_layer.weight", 1, Hd);
    for (int l = 0; l < c.N; ++l) {
        const std::string s = "_l" + std::to_string(l);
        RnnW w;
        w.in = l == 0 ? E + 2 * Hd : Hd;
        w.w_ih = add("model.decoder.rnn.weight_ih" + s, G * Hd, w.in);
        w.w_hh = add("model.decoder.rnn.weight_hh" + s, G * Hd, Hd);
        w.b_ih = add("model.decoder.rnn.bias_ih" + s, G * Hd, 0);
        w.b_hh = add("model.decoder.rnn.bias_hh" + s, G * Hd, 0);
        L.dec.push_back(w);
    }
    L.bridge_w = add("model.decoder.bridge.weight", Hd, 2 * Hd);
    L.bridge_b = add("model.decoder.bridge.bias", Hd, 0);
    L.pre_out = add("model.decoder.pre_output_layer.weight", Hd, 3 * Hd + E);
    L.src_emb = add("model.src_embed.weight", c.Vs, E);
    L.trg_emb = add("model.trg_embed.weight", c.Vt, E);
    L.gen_w = add("model.generator.proj.weight", c.Vt, Hd);
    L.total = cur;
    return L;
}

static int check_rcfg(const slnlp_rnn_config* c) {
    SLNLP_CHECK_ARG(c, "rnn: null config");
    SLNLP_CHECK_ARG(c->lstm == 0 || c->lstm == 1, "rnn: Invalid `rnn_type`.");   // bkp.py:347
    SLNLP_CHECK_ARG(c->E > 0 && c->E % 4 == 0, "rnn: embedding_size %d must be a positive multiple of 4", c->E);
    SLNLP_CHECK_ARG(c->Hd > 0 && c->Hd % 4 == 0, "rnn: hidden_size %d must be a positive multiple of 4", c->Hd);
    SLNLP_CHECK_ARG(c->N > 0 && c->Vs > 1 && c->Vt > 1, "rnn: bad num_layers / vocab");
    SLNLP_CHECK_ARG(c->B > 0 && c->B <= 1024, "rnn: batch %d outside 1..1024", c->B);
    SLNLP_CHECK_ARG(c->S > 0 && c->S <= 2048, "rnn: seq_len %d outside 1..2048", c->S);
    SLNLP_CHECK_ARG((long)c->B * c->S <= 65536, "rnn: batch %d x seq_len %d exceeds 65536 tokens per step", c->B, c->S);
    SLNLP_CHECK_ARG(c->bos_idx >= 0 && c->bos_idx < c->Vt, "rnn: bos_idx %d outside the target vocabulary", c->bos_idx);
    SLNLP_CHECK_ARG(c->dropout >= 0.f && c->dropout < 1.f, "rnn: dropout %f", c->dropout);
    SLNLP_CHECK_ARG(c->precision == 1 || c->precision == 3, "rnn: precision %d", c->precision);
    return 0;
}

struct RBump {
    char* base;
    size_t cur = 0;
    explicit RBump(void* b) : base((char*)b) {}
    template <typename T>
    T* take(size_t n) {
        cur = (cur + 255) & ~(size_t)255;
        T* p = (T*)(base + cur);
        cur += n * sizeof(T);
        return p;
    }
};

struct EncDirA {   // per (layer, direction)
    float *xproj, *acts, *hprev, *cprev, *hn, *h, *c;
    float *dgx, *dgh, *dh, *dc, *carry, *dhx;   // dhx: partial products of the K-sliced recurrent dgrad
};
struct RPP {               // bf16 hi / lo planes of a GEMM operand (gemm_planes.hip)
    unsigned short *hi = nullptr, *lo = nullptr;
    PlaneOut out() const { PlaneOut o; o.hi = hi; o.lo = lo; return o; }
};
struct EncLayerA { EncDirA d[2]; float *out, *dout; RPP xinp; };    // out [M,2Hd]; dout = grad w.r.t. out; xinp = planes of the layer input
struct DecLayerA {
    float *xproj, *hproj, *acts, *hprev, *cprev, *hn, *h, *c, *out;
    float *dgx, *dgh, *dh, *dc, *carry, *dout;
};
struct RWs {
    float *zero_fwd, *zero_bwd;
    float *emb, *demb, *enc_final, *denc_final, *h0, *dh0, *dz, *pk, *dpk, *q, *dq, *alphas, *ctx, *dctx, *emb_bos,
        *demb_bos, *dwe_part, *logits, *dlogits, *logp, *row_nll, *opt_partials;
    int64_t* bos_ids;
    unsigned* sync;          // {barrier count, generation, error flag} of the persistent layer kernel
    RPP wp;                  // planes of the encoder's RNN weights (arena prefix [0, key_w)), split once per forward
    RPP dgxp[2], dghp[2], hprevp[2];   // per direction, reused by every layer's backward
    char *planes_begin, *planes_end;   // activation planes: zero padding rows, re-zeroed when the batch size changes
    void *emb_scratch_src, *emb_scratch_tgt;
    std::vector<EncLayerA> enc;
    std::vector<DecLayerA> dec;
    size_t bytes;
};

static RWs rcarve(const slnlp_rnn_config& c, void* base) {
    RWs w;
    RBump b(base);
    const size_t B = c.B, S = c.S, E = c.E, Hd = c.Hd, M = B * S, G = c.lstm ? 4 : 3, Vp = align_up_r(c.Vt, 4);
    w.emb = b.take<float>(M * E);
    w.demb = b.take<float>(M * E);
    // states that start every step at zero sit together: ONE memset per pass instead of one per (layer, direction)
    w.zero_fwd = b.take<float>(2 * c.N * B * Hd);                 // encoder c
    w.zero_bwd = b.take<float>((2 * c.N + 2 * c.N) * B * Hd);     // encoder dc, decoder dh, decoder dc
    for (int l = 0; l < c.N; ++l) {
        EncLayerA a;
        for (int d = 0; d < 2; ++d) {
            EncDirA& e = a.d[d];
            e.xproj = b.take<float>(M * G * Hd);
            e.acts = b.take<float>(M * G * Hd);
            e.hprev = b.take<float>(M * Hd);
            e.cprev = b.take<float>(M * Hd);
            e.hn = b.take<float>(M * Hd);
            e.h = b.take<float>(B * Hd);
            e.c = w.zero_fwd + (size_t)(2 * l + d) * B * Hd;
            e.dgx = b.take<float>(M * G * Hd);
            e.dgh = c.lstm ? e.dgx : b.take<float>(M * G * Hd);
            e.dh = b.take<float>(B * Hd);
            e.dc = w.zero_bwd + (size_t)(2 * l + d) * B * Hd;
            e.carry = b.take<float>(B * Hd);
            e.dhx = b.take<float>((size_t)(rnn_kslices(c) > 1 ? rnn_kslices(c) - 1 : 1) * B * Hd);
        }
        a.out = b.take<float>(M * 2 * Hd);
        a.dout = b.take<float>(M * 2 * Hd);
        w.enc.push_back(a);
    }
    w.enc_final = b.take<float>(c.N * B * 2 * Hd);
    w.denc_final = b.take<float>(c.N * B * 2 * Hd);
    w.h0 = b.take<float>(c.N * B * Hd);
    w.dh0 = b.take<float>(c.N * B * Hd);
    w.dz = b.take<float>(c.N * B * Hd);
    w.pk = b.take<float>(M * Hd);
    w.dpk = b.take<float>(M * Hd);
    w.q = b.take<float>(B * Hd);
    w.dq = b.take<float>(B * Hd);
    w.alphas = b.take<float>(B * S);
    w.ctx = b.take<float>(B * 2 * Hd);
    w.dctx = b.take<float>(B * 2 * Hd);
    w.emb_bos = b.take<float>(B * E);
    w.demb_bos = b.take<float>(B * E);
    w.dwe_part = b.take<float>(B * Hd);
    for (int l = 0; l < c.N; ++l) {
        DecLayerA a;
        a.xproj = b.take<float>(B * G * Hd);
        a.hproj = b.take<float>(B * G * Hd);
        a.acts = b.take<float>(B * G * Hd);
        a.hprev = b.take<float>(B * Hd);
        a.cprev = b.take<float>(B * Hd);
        a.hn = b.take<float>(B * Hd);
        a.h = b.take<float>(B * Hd);
        a.c = b.take<float>(B * Hd);
        a.out = b.take<float>(B * Hd);
        a.dgx = b.take<float>(B * G * Hd);
        a.dgh = c.lstm ? a.dgx : b.take<float>(B * G * Hd);
        a.dh = w.zero_bwd + (size_t)(2 * c.N + l) * B * Hd;
        a.dc = w.zero_bwd + (size_t)(3 * c.N + l) * B * Hd;
        a.carry = b.take<float>(B * Hd);
        a.dout = b.take<float>(B * Hd);
        w.dec.push_back(a);
    }
    w.logits = b.take<float>(B * Vp);
    w.dlogits = b.take<float>(B * Vp);
    w.logp = b.take<float>(B * c.Vt);
    w.row_nll = b.take<float>(B);
    w.opt_partials = b.take<float>(1024);
    w.bos_ids = b.take<int64_t>(B);
    w.sync = b.take<unsigned>(64);
    w.emb_scratch_src = b.take<char>(embed_bwd_scratch_bytes(c.B, c.S, c.E));
    w.emb_scratch_tgt = b.take<char>(embed_bwd_scratch_bytes(c.B, 1, c.E));
    {   // ---- bf16 operand planes of the M = S*B GEMMs (used when E and Hd are multiples of 64)
        const size_t Mp = (M + 63) / 64 * 64, GH = G * Hd;
        const size_t wlen = (size_t)build_rlayout(c).key_w + 64 * (2 * Hd > E ? 2 * Hd : E);   // tail pad: tiles may over-read rows
        w.wp.hi = b.take<unsigned short>(wlen);
        w.wp.lo = b.take<unsigned short>(wlen);
        b.cur = (b.cur + 255) & ~(size_t)255;
        w.planes_begin = b.base + b.cur;
        auto pp = [&](size_t cols) { RPP q; q.hi = b.take<unsigned short>(Mp * cols); q.lo = b.take<unsigned short>(Mp * cols); return q; };
        for (int l = 0; l < c.N; ++l) w.enc[l].xinp = pp(l == 0 ? E : 2 * Hd);
        for (int d = 0; d < 2; ++d) {
            w.dgxp[d] = pp(GH);
            w.dghp[d] = c.lstm ? w.dgxp[d] : pp(GH);
            w.hprevp[d] = pp(Hd);
        }
        b.cur = (b.cur + 255) & ~(size_t)255;
        w.planes_end = b.base + b.cur;
    }
    w.bytes = (b.cur + 255) & ~(size_t)255;
    return w;
}

}  // namespace slnlp

using namespace slnlp;

enum { RSITE_ENC0 = 32, RSITE_DEC0 = 64 };

struct slnlp_rnn_plan {
    slnlp_rnn_config cfg;
    slnlp_tf_buffers buf;
    RLayout L;
    RWs w;
    int last_B = 0;
    float last_p = 0.f;
    const int64_t *last_X = nullptr, *last_y = nullptr, *last_len = nullptr;
    std::map<int, hipGraphExec_t> graphs;
    bool use_planes = false;  // E, Hd multiples of 64: the M = S*B GEMMs run on pre-split bf16 planes (gemm_planes.hip)
    int planes_B = -1;        // batch size the activation planes' zero padding is valid for
    int destroy_sync = 1;     // slnlp_rnn_set_destroy_sync: wait for the device before the plan goes away (launch.hpp)
    int wgrad_p = 2, dgrad_p = 2;   // split-bf16 passes of the plane gradient products: the process default AT CREATION, fixed for the plan's life
    bool persistent = false;  // opt-in: all timesteps of an encoder layer in one launch (not yet faster; needs one fit per GPU)
    // backward through time: the cell kernel + K-sliced grouped GEMM pair per timestep (default), or ONE launch per timestep
    // (gemm.hip rnn_step_bwd_kernel; slnlp_rnn_set_fused_backward(plan, 1), env SLNLP_RNN_FUSED_BWD=1).  Round 4 built the
    // fused kernel to halve the 384 dependent launches of a cfg3 backward and measured it SLOWER solo (LSTM 9.11 vs 8.09 ms, GRU
    // 7.73 vs 7.27): its G K-slices share one CU's LDS-write and conversion bandwidth where the K-sliced launch spreads them over
    // 256 CUs, and a timestep is a latency chain either way.  16 GRU fits in lockstep gain 4 % from it.
    bool unfused_bwd = [] { const char* e = getenv("SLNLP_RNN_FUSED_BWD"); return !(e && atoi(e) != 0); }();
    // lockstep (lockstep.hip): where lsm_nll also puts the batch's log-probs / loss (device row and batch index in ls_dyn)
    float* ls_logp = nullptr;
    float* ls_loss = nullptr;
    const int* ls_dyn = nullptr;

    float* P(long off) const { return buf.params + off; }
    float* Gd(long off) const { return buf.grads + off; }

    // y[M,N] = x[M,K](lda) W[N,K](ldb)^T + bias, act (0 none / 2 tanh), + resid
    slnlp_gemm_args lin_args(const float* x, long lda, int M, int K, const float* W, long ldb, int N, const float* bias,
                             float* y, long ldy, int act, const float* resid) const {
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A = x; a.lda = lda; a.a_kmajor = 1;
        a.B = W; a.ldb = ldb; a.b_kmajor = 1;
        a.C = y; a.ldc = ldy; a.M = M; a.N = N; a.K = K;
        a.bias = bias; a.relu = act; a.resid = resid; a.ldr = ldy;
        a.precision = cfg.precision;
        return a;
    }
    int lin(const float* x, long lda, int M, int K, const float* W, long ldb, int N, const float* bias, float* y, long ldy,
            int act, const float* resid, hipStream_t st) const {
        return gemm(lin_args(x, lda, M, K, W, ldb, N, bias, y, ldy, act, resid), st);
    }
    // dx[M,Kin](ldx) = dy[M,Nout](ldy) W[Nout,Kin](ldw)  (+resid, same ld as dx)
    slnlp_gemm_args dgr_args(const float* dy, long ldy, int M, int Nout, const float* W, long ldw, int Kin, float* dx,
                             long ldx, const float* resid) const {
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A = dy; a.lda = ldy; a.a_kmajor = 1;
        a.B = W; a.ldb = ldw; a.b_kmajor = 0;
        a.C = dx; a.ldc = ldx; a.M = M; a.N = Kin; a.K = Nout;
        a.resid = resid; a.ldr = ldx;
        a.precision = cfg.precision;
        return a;
    }
    int dgr(const float* dy, long ldy, int M, int Nout, const float* W, long ldw, int Kin, float* dx, long ldx,
            const float* resid, hipStream_t st) const {
        return gemm(dgr_args(dy, ldy, M, Nout, W, ldw, Kin, dx, ldx, resid), st);
    }
    // zero padding of the activation planes is per batch size: re-zero when it changes (outside any capture)
    int prepare_planes(int B, hipStream_t st) {
        if (!use_planes || B == planes_B) return 0;
        if (hipMemsetAsync(w.planes_begin, 0, (size_t)(w.planes_end - w.planes_begin), st) != hipSuccess) {
            set_error("rnn: zeroing operand planes failed");
            return SLNLP_ERR_LAUNCH;
        }
        planes_B = B;
        return 0;
    }
    // plane GEMM argument builders (weights: planes of the arena at offset woff)
    slnlp_gemm_args lin_p(const RPP& x, int M, int K, long woff, int N, const float* bias, float* y) const {
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A_hi = x.hi; a.A_lo = x.lo; a.lda_p = K; a.a_kmajor = 1;
        a.B_hi = w.wp.hi + woff; a.B_lo = w.wp.lo + woff; a.ldb_p = K; a.b_kmajor = 1;
        a.C = y; a.ldc = N; a.M = M; a.N = N; a.K = K; a.bias = bias;
        a.precision = cfg.precision;
        return a;
    }
    slnlp_gemm_args dgr_p(const RPP& dy, int M, int Nout, long woff, int Kin, float* dx, const float* resid) const {
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A_hi = dy.hi; a.A_lo = dy.lo; a.lda_p = Nout; a.a_kmajor = 1;
        a.B_hi = w.wp.hi + woff; a.B_lo = w.wp.lo + woff; a.ldb_p = Kin; a.b_kmajor = 0;
        a.C = dx; a.ldc = Kin; a.M = M; a.N = Kin; a.K = Nout;
        a.resid = resid; a.ldr = Kin;
        a.precision = cfg.precision == 3 ? dgrad_p : cfg.precision;     // (slnlp_set_backward_passes: dY's bf16 head only)
        return a;
    }
    slnlp_gemm_args wgr_p(const RPP& dy, int T, int Nout, const RPP& x, int Kin, float* dW, float* db) const {
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A_hi = dy.hi; a.A_lo = dy.lo; a.lda_p = Nout; a.a_kmajor = 0;
        a.B_hi = x.hi; a.B_lo = x.lo; a.ldb_p = Kin; a.b_kmajor = 0;
        a.C = dW; a.ldc = Kin; a.M = Nout; a.N = Kin; a.K = T;
        a.rowsum_a = db;
        a.precision = cfg.precision == 3 ? wgrad_p : cfg.precision;
        return a;
    }
    // the recurrent dgrad dh(t-1) = dgh W_hh + carry contracts over the G*Hd gate columns, cut into K-slices (each its own GEMM job or
    // one GEMM of a batched job; partial products summed by the next cell kernel).  Rounds 1-4: one slice per gate on 64 x 16 tiles --
    // 256 workgroups of 160 KB (64 rows x 512 k of dgh, re-read by each of the 32 column tiles, + 512 k x 16 columns of W_hh).  Round 5:
    // slices of 128 on 64 x 64 tiles -- the same 256 workgroups, 64 KB each (32 + 32): a launch lasts as long as one workgroup loads.
    int kslices() const { return rnn_kslices(cfg); }
    // dW[Nout,Kin](ldw) = dy[T,Nout](ldy)^T x[T,Kin](ldx);  db = colsum(dy)
    int wgr(const float* dy, long ldy, int T, int Nout, const float* x, long ldx, int Kin, float* dW, long ldw, float* db,
            hipStream_t st) const {
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A = dy; a.lda = ldy; a.a_kmajor = 0;
        a.B = x; a.ldb = ldx; a.b_kmajor = 0;
        a.C = dW; a.ldc = ldw; a.M = Nout; a.N = Kin; a.K = T;
        a.rowsum_a = db;
        a.precision = cfg.precision;
        return gemm(a, st);
    }
};


extern "C" {

int slnlp_rnn_num_params(const slnlp_rnn_config* cfg) {
    if (check_rcfg(cfg)) return -1;
    return (int)build_rlayout(*cfg).ents.size();
}
int slnlp_rnn_param_info(const slnlp_rnn_config* cfg, int i, char* name, int64_t shape[2], int* ndim, int64_t* offset) {
    SLNLP_TRY(check_rcfg(cfg));
    RLayout L = build_rlayout(*cfg);
    SLNLP_CHECK_ARG(i >= 0 && i < (int)L.ents.size(), "rnn_param_info: index %d out of range", i);
    const RParam& e = L.ents[i];
    if (name) { strncpy(name, e.name.c_str(), 127); name[127] = 0; }
    if (shape) { shape[0] = e.shape[0]; shape[1] = e.shape[1]; }
    if (ndim) *ndim = e.ndim;
    if (offset) *offset = e.off;
    return 0;
}
int64_t slnlp_rnn_arena_floats(const slnlp_rnn_config* cfg) {
    if (check_rcfg(cfg)) return -1;
    return build_rlayout(*cfg).total;
}
int64_t slnlp_rnn_workspace_bytes(const slnlp_rnn_config* cfg) {
    if (check_rcfg(cfg)) return -1;
    return (int64_t)rcarve(*cfg, nullptr).bytes;
}

void slnlp_rnn_destroy(slnlp_rnn_plan* plan) {
    if (!plan) return;
    if (!plan->graphs.empty()) (void)hipDeviceSynchronize();   // graph execs are torn down below
    else destroy_sync(plan->destroy_sync);
    for (auto& kv : plan->graphs) (void)hipGraphExecDestroy(kv.second);
    delete plan;
}

int slnlp_rnn_create(const slnlp_rnn_config* cfg, const slnlp_tf_buffers* buf, slnlp_rnn_plan** out) {
    SLNLP_TRY(check_rcfg(cfg));
    SLNLP_CHECK_ARG(buf && out, "rnn_create: null argument");
    SLNLP_CHECK_ARG(buf->params && buf->grads && buf->momentum && buf->workspace && buf->rng && buf->lr && buf->scalars,
                    "rnn_create: every buffer pointer (except pe) is required");
    SLNLP_CHECK_ARG((((uintptr_t)buf->params | (uintptr_t)buf->grads | (uintptr_t)buf->momentum |
                      (uintptr_t)buf->workspace) & 255) == 0,
                    "rnn_create: arenas / workspace must be 256-byte aligned");
    slnlp_rnn_plan* p = new slnlp_rnn_plan();
    p->cfg = *cfg;
    p->wgrad_p = wgrad_passes();
    p->dgrad_p = dgrad_passes();
    p->buf = *buf;
    p->L = build_rlayout(*cfg);
    p->w = rcarve(*cfg, buf->workspace);
    p->use_planes = (cfg->E % 64 == 0) && (cfg->Hd % 64 == 0);
    std::vector<int64_t> bos(cfg->B, (int64_t)cfg->bos_idx);
    if (hipMemcpy(p->w.bos_ids, bos.data(), bos.size() * sizeof(int64_t), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(buf->grads, 0, p->L.total * sizeof(float)) != hipSuccess ||   // pre_output_layer + pads stay 0
        hipMemset(p->w.sync, 0, 64 * sizeof(unsigned)) != hipSuccess ||
        hipStreamSynchronize(nullptr) != hipSuccess ||      // null-stream memsets must not land inside the caller's first step
        rnn_layer_init() != 0 || rnn_step_bwd_init() != 0 || gemm_planes_init() != 0) {
        set_error("rnn_create: device initialisation failed: %s", hipGetErrorString(hipGetLastError()));
        delete p;
        return SLNLP_ERR_LAUNCH;
    }
    *out = p;
    return 0;
}

int slnlp_rnn_forward(slnlp_rnn_plan* pl, const int64_t* X, const int64_t* y, const int64_t* lengths, int B, int train,
                      float* logp_out, void* stream) {
    SLNLP_CHECK_ARG(pl && X && y && lengths, "rnn_forward: X, y and lengths are required");
    SLNLP_CHECK_ARG(B > 0 && B <= pl->cfg.B, "rnn_forward: batch %d outside 1..%d", B, pl->cfg.B);
    hipStream_t st = (hipStream_t)stream;
    StepScope scope(st);
    SLNLP_TRY(scope.rc);
    const slnlp_rnn_config& c = pl->cfg;
    const RWs& w = pl->w;
    const RLayout& L = pl->L;
    const int lstm = c.lstm, G = lstm ? 4 : 3, E = c.E, Hd = c.Hd, S = c.S, N = c.N, M = S * B, GH = G * Hd;
    const int Vp = (int)align_up_r(c.Vt, 4);
    const float p = train ? c.dropout : 0.f;
    const unsigned long long* rng = pl->buf.rng;
    pl->last_B = B; pl->last_p = p; pl->last_X = X; pl->last_y = y; pl->last_len = lengths;

    // src_embed (nn.Embedding(padding_idx), no scale, no positional term)  bkp.py:48-50
    const bool up = pl->use_planes;
    if (up) {   // the encoder's RNN weights as bf16 planes, once per forward (they changed in the optimizer step / load_state)
        if (!recording()) SLNLP_TRY(pl->prepare_planes(B, st));   // a memset: the lockstep driver runs it before it records
        SLNLP_TRY(split_planes(pl->buf.params, L.key_w, 1, (int)L.key_w, w.wp.hi, w.wp.lo, L.key_w, st));
    }
    SLNLP_TRY(embed_fwd(X, S, B, S, E, c.Vs, pl->P(L.src_emb), nullptr, w.emb, 1.f, 0.f, 0, rng, -1, st,
                        up ? w.enc[0].xinp.out() : PlaneOut{}));
    SLNLP_TRY(fill_zero(w.zero_fwd, (size_t)2 * N * c.B * Hd * sizeof(float), st));   // every c_0
    const float* x_in = w.emb;
    for (int l = 0; l < N; ++l) {
        const EncLayerA& a = w.enc[l];
        const int in = L.enc[0][l].in;
        if (up) {   // x W_ih^T + b_ih of both directions over all timesteps: one grouped plane-GEMM launch
            if (l > 0) SLNLP_TRY(split_planes(x_in, in, M, in, a.xinp.hi, a.xinp.lo, in, st));
            const slnlp_gemm_args xj[2] = {pl->lin_p(a.xinp, M, in, L.enc[0][l].w_ih, GH, pl->P(L.enc[0][l].b_ih), a.d[0].xproj),
                                           pl->lin_p(a.xinp, M, in, L.enc[1][l].w_ih, GH, pl->P(L.enc[1][l].b_ih), a.d[1].xproj)};
            SLNLP_TRY(gemm_planes_group(xj, nullptr, 2, nullptr, 0, st));
        }
        for (int d = 0; d < 2; ++d) {
            const RnnW& q = L.enc[d][l];
            if (!up) SLNLP_TRY(pl->lin(x_in, in, M, in, pl->P(q.w_ih), in, GH, pl->P(q.b_ih), a.d[d].xproj, GH, 0, nullptr, st));
            // the state chain lives in the per-timestep `hprev` slots: slot of the first processed timestep = h_0 = 0
            const int t0 = d == 0 ? 0 : S - 1;
            SLNLP_TRY(fill_zero(a.d[d].hprev + (long)t0 * B * Hd, (size_t)B * Hd * sizeof(float), st));
        }
        const bool last = l == N - 1;
        // inter-layer dropout (not after the last layer); padded outputs of the LAST layer = float(pad_idx)
        const float fill = last ? (float)c.pad_src : 0.f, pdrop = last ? 0.f : p;
        int launched = 0;
        if (pl->persistent) {   // all S timesteps of both directions in ONE launch (gemm.hip rnn_layer_fwd_kernel)
            slnlp_rnn_layer_dir ld[2];
            for (int d = 0; d < 2; ++d) {
                const RnnW& q = L.enc[d][l];
                const EncDirA& e = a.d[d];
                ld[d].hprev = e.hprev; ld[d].h_final = e.h; ld[d].w_hh = pl->P(q.w_hh); ld[d].b_hh = pl->P(q.b_hh);
                ld[d].xproj = e.xproj; ld[d].c = e.c; ld[d].cprev = e.cprev; ld[d].acts = e.acts; ld[d].hn = e.hn;
                ld[d].out = a.out + d * Hd; ld[d].out_col0 = d * Hd; ld[d].reverse = d;
            }
            SLNLP_TRY(rnn_layer_fwd(lstm, ld, 2, B, Hd, S, lengths, fill, 2 * Hd, pdrop, RSITE_ENC0 + l, rng, c.precision,
                                    w.sync, reinterpret_cast<int*>(w.sync + 2), &launched, st));
        }
        for (int step = 0; step < S && !launched; ++step) {
            // shapes the persistent kernel does not cover: one launch per timestep (recurrent GEMM of both directions + cell)
            slnlp_rnn_step_dir dirs[2];
            for (int d = 0; d < 2; ++d) {
                const int t = d == 0 ? step : S - 1 - step, tn = d == 0 ? t + 1 : t - 1;
                const RnnW& q = L.enc[d][l];
                const EncDirA& e = a.d[d];
                slnlp_rnn_step_dir& k = dirs[d];
                k.h_in = e.hprev + (long)t * B * Hd;
                k.h_out = step + 1 < S ? e.hprev + (long)tn * B * Hd : e.h;
                k.w_hh = pl->P(q.w_hh); k.b_hh = pl->P(q.b_hh);
                k.xproj = e.xproj + (long)t * B * GH;
                k.c = e.c;
                k.cprev_save = e.cprev + (long)t * B * Hd;
                k.acts = e.acts + (long)t * B * GH;
                k.hn_save = e.hn + (long)t * B * Hd;
                k.out = a.out + (long)t * B * 2 * Hd + d * Hd;
                k.t = t; k.out_row0 = t * B; k.out_col0 = d * Hd;
            }
            SLNLP_TRY(rnn_step_fwd(lstm, dirs, 2, B, Hd, lengths, fill, 2 * Hd, pdrop, RSITE_ENC0 + l, rng, c.precision, st));
        }
        // final states -> hidden[l] = fwd || bwd   (concatenate_directions, bkp.py:155-159)
        for (int d = 0; d < 2; ++d)
            SLNLP_TRY(add_rows(a.d[d].h, Hd, w.enc_final + (long)l * B * 2 * Hd + d * Hd, 2 * Hd, B, Hd, 0, st));
        x_in = a.out;
    }
    const float* enc_out = w.enc[N - 1].out;
    // Decoder.init_hidden :268-280: tanh(bridge(enc_final)); LSTM uses (h, h) as (h0, c0)
    SLNLP_TRY(pl->lin(w.enc_final, 2 * Hd, N * B, 2 * Hd, pl->P(L.bridge_w), 2 * Hd, Hd, pl->P(L.bridge_b), w.h0, Hd, 2,
                      nullptr, st));
    SLNLP_TRY(pl->lin(enc_out, 2 * Hd, M, 2 * Hd, pl->P(L.key_w), 2 * Hd, Hd, nullptr, w.pk, Hd, 0, nullptr, st));   // :246
    const float* h_top = w.h0 + (long)(N - 1) * B * Hd;                                                              // get_query
    SLNLP_TRY(pl->lin(h_top, Hd, B, Hd, pl->P(L.query_w), Hd, Hd, nullptr, w.q, Hd, 0, nullptr, st));
    SLNLP_TRY(bahdanau_fwd(w.q, w.pk, enc_out, pl->P(L.energy_w), X, S, c.pad_src, B, S, Hd, w.alphas, w.ctx, st));
    // prev_embed = trg_embed(<bos>)  :254 with max_len = 1
    SLNLP_TRY(embed_fwd(w.bos_ids, 1, B, 1, E, c.Vt, pl->P(L.trg_emb), nullptr, w.emb_bos, 1.f, 0.f, 0, rng, -1, st));
    const float* x_prev = nullptr;
    for (int l = 0; l < N; ++l) {
        const RnnW& q = L.dec[l];
        const DecLayerA& a = w.dec[l];
        if (l == 0) {   // rnn_input = cat[prev_embed, context]  -> two K-slices of W_ih, no concat buffer
            SLNLP_TRY(pl->lin(w.emb_bos, E, B, E, pl->P(q.w_ih), q.in, GH, pl->P(q.b_ih), a.xproj, GH, 0, nullptr, st));
            SLNLP_TRY(pl->lin(w.ctx, 2 * Hd, B, 2 * Hd, pl->P(q.w_ih) + E, q.in, GH, nullptr, a.xproj, GH, 0, a.xproj, st));
        } else {
            SLNLP_TRY(pl->lin(x_prev, Hd, B, Hd, pl->P(q.w_ih), Hd, GH, pl->P(q.b_ih), a.xproj, GH, 0, nullptr, st));
        }
        const float* h0l = w.h0 + (long)l * B * Hd;
        SLNLP_TRY(pl->lin(h0l, Hd, B, Hd, pl->P(q.w_hh), Hd, GH, pl->P(q.b_hh), a.hproj, GH, 0, nullptr, st));
        SLNLP_TRY(add_rows(h0l, Hd, a.h, Hd, B, Hd, 0, st));
        if (lstm) SLNLP_TRY(add_rows(h0l, Hd, a.c, Hd, B, Hd, 0, st));
        slnlp_rnn_cell_dir k;
        k.xproj = a.xproj; k.hproj = a.hproj; k.h = a.h; k.c = a.c;
        k.hprev_save = a.hprev; k.cprev_save = a.cprev; k.acts = a.acts; k.hn_save = a.hn;
        k.out = a.out; k.t = 0; k.out_row0 = 0; k.out_col0 = 0;
        SLNLP_TRY(rnn_cell_fwd(lstm, &k, 1, B, Hd, nullptr, 0.f, Hd, l == N - 1 ? 0.f : p, RSITE_DEC0 + l, rng, st));
        x_prev = a.out;
    }
    // generator on the decoder state (bkp.py:40-46,69-76) + criterion
    SLNLP_TRY(pl->lin(x_prev, Hd, B, Hd, pl->P(L.gen_w), Hd, c.Vt, nullptr, w.logits, Vp, 0, nullptr, st));
    SLNLP_TRY(lsm_nll(w.logits, Vp, y, B, c.Vt, c.pad_tgt, w.logp, pl->buf.scalars, train ? w.dlogits : nullptr, Vp,
                      w.row_nll, st, nullptr, logp_out ? nullptr : pl->ls_logp, logp_out ? nullptr : pl->ls_dyn,
                      logp_out ? nullptr : pl->ls_loss, (!logp_out && pl->ls_dyn) ? pl->ls_dyn + 1 : nullptr));
    if (logp_out &&
        hipMemcpyAsync(logp_out, w.logp, (size_t)B * c.Vt * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("rnn_forward: copy of log-probs failed");
        return SLNLP_ERR_LAUNCH;
    }
    return 0;
}

// on = 1: all timesteps of an encoder layer in ONE persistent launch (gemm.hip rnn_layer_fwd_kernel) instead of one
// launch per timestep.  Off by default: measured no faster yet, and its workgroups must all be resident at once, so it
// must not be used when several fits share the GPU.
int slnlp_rnn_set_fused_backward(slnlp_rnn_plan* pl, int on) {
    SLNLP_CHECK_ARG(pl, "rnn_set_fused_backward: null plan");
    pl->unfused_bwd = on == 0;
    return 0;
}
int slnlp_rnn_set_persistent(slnlp_rnn_plan* pl, int on) {
    SLNLP_CHECK_ARG(pl, "rnn_set_persistent: null plan");
    pl->persistent = on != 0;
    return 0;
}

// *status = 0: every device-wide barrier of the persistent layer kernels completed; 1: a workgroup timed out (bounded
// spin) and that step's results are invalid.  Synchronises the device.
int slnlp_rnn_health(slnlp_rnn_plan* pl, int* status) {
    SLNLP_CHECK_ARG(pl && status, "rnn_health: null argument");
    unsigned flag = 0;
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(&flag, pl->w.sync + 2, sizeof(flag), hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("rnn_health: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    *status = (int)flag;
    return 0;
}

int slnlp_rnn_seed_dlogp(slnlp_rnn_plan* pl, const float* dlogp, void* stream) {
    SLNLP_CHECK_ARG(pl && dlogp && pl->last_B > 0, "rnn_seed_dlogp: needs a prior forward");
    return lsm_bwd(pl->w.logp, dlogp, pl->last_B, pl->cfg.Vt, pl->w.dlogits, align_up_r(pl->cfg.Vt, 4), (hipStream_t)stream);
}

int slnlp_rnn_backward(slnlp_rnn_plan* pl, void* stream) {
    SLNLP_CHECK_ARG(pl && pl->last_B > 0, "rnn_backward: needs a prior forward(train)");
    hipStream_t st = (hipStream_t)stream;
    StepScope scope(st);
    SLNLP_TRY(scope.rc);
    const slnlp_rnn_config& c = pl->cfg;
    const RWs& w = pl->w;
    const RLayout& L = pl->L;
    const int B = pl->last_B, lstm = c.lstm, G = lstm ? 4 : 3, E = c.E, Hd = c.Hd, S = c.S, N = c.N, M = S * B;
    const int GH = G * Hd, Vp = (int)align_up_r(c.Vt, 4);
    const float p = pl->last_p;
    const unsigned long long* rng = pl->buf.rng;
    const int64_t *X = pl->last_X, *lengths = pl->last_len;
    const float* enc_out = w.enc[N - 1].out;

    SLNLP_TRY(fill_zero(w.zero_bwd, (size_t)4 * N * c.B * Hd * sizeof(float), st));   // dc / dh seeds
    // generator (no bias)
    const float* dec_out = w.dec[N - 1].out;
    SLNLP_TRY(pl->wgr(w.dlogits, Vp, B, c.Vt, dec_out, Hd, Hd, pl->Gd(L.gen_w), Hd, nullptr, st));
    SLNLP_TRY(pl->dgr(w.dlogits, Vp, B, c.Vt, pl->P(L.gen_w), Hd, Hd, w.dec[N - 1].dout, Hd, nullptr, st));
    // decoder RNN, one step, top layer first
    for (int l = N - 1; l >= 0; --l) {
        const RnnW& q = L.dec[l];
        const DecLayerA& a = w.dec[l];
        slnlp_rnn_cell_bwd_dir k = {};
        k.dh_state = a.dh; k.dc_state = a.dc; k.dout = a.dout; k.acts = a.acts; k.cprev_save = a.cprev;
        k.hprev_save = a.hprev; k.hn_save = a.hn; k.dgx = a.dgx; k.dgh = a.dgh; k.carry = a.carry;
        k.t = 0; k.out_row0 = 0; k.out_col0 = 0;
        SLNLP_TRY(rnn_cell_bwd(lstm, &k, 1, B, Hd, nullptr, Hd, l == N - 1 ? 0.f : p, RSITE_DEC0 + l, rng, st));
        float* dh0l = w.dh0 + (long)l * B * Hd;
        SLNLP_TRY(pl->dgr(a.dgh, GH, B, GH, pl->P(q.w_hh), Hd, Hd, dh0l, Hd, a.carry, st));        // d h0[l]
        if (lstm) SLNLP_TRY(add_rows(a.dc, Hd, dh0l, Hd, B, Hd, 1, st));                             // c0 = h0 too
        SLNLP_TRY(pl->wgr(a.dgh, GH, B, GH, a.hprev, Hd, Hd, pl->Gd(q.w_hh), Hd, pl->Gd(q.b_hh), st));
        if (l > 0) {
            SLNLP_TRY(pl->wgr(a.dgx, GH, B, GH, w.dec[l - 1].out, Hd, Hd, pl->Gd(q.w_ih), Hd, pl->Gd(q.b_ih), st));
            SLNLP_TRY(pl->dgr(a.dgx, GH, B, GH, pl->P(q.w_ih), Hd, Hd, w.dec[l - 1].dout, Hd, nullptr, st));
        } else {
            SLNLP_TRY(pl->wgr(a.dgx, GH, B, GH, w.emb_bos, E, E, pl->Gd(q.w_ih), q.in, pl->Gd(q.b_ih), st));
            SLNLP_TRY(pl->wgr(a.dgx, GH, B, GH, w.ctx, 2 * Hd, 2 * Hd, pl->Gd(q.w_ih) + E, q.in, nullptr, st));
            SLNLP_TRY(pl->dgr(a.dgx, GH, B, GH, pl->P(q.w_ih) + E, q.in, 2 * Hd, w.dctx, 2 * Hd, nullptr, st));
            SLNLP_TRY(pl->dgr(a.dgx, GH, B, GH, pl->P(q.w_ih), q.in, E, w.demb_bos, E, nullptr, st));
        }
    }
    // trg_embed: only the <bos> row sees gradient; padding_idx row gets none (bkp.py:377-379)
    SLNLP_TRY(embed_bwd(w.bos_ids, 1, B, 1, E, c.Vt, w.demb_bos, pl->Gd(L.trg_emb), 1.f, c.pad_tgt, 0.f, 0, rng,
                        w.emb_scratch_tgt, st));
    // attention
    float* denc_out = w.enc[N - 1].dout;
    SLNLP_TRY(bahdanau_bwd(w.q, w.pk, enc_out, pl->P(L.energy_w), w.alphas, w.dctx, B, S, Hd, w.dq, w.dpk, denc_out,
                           w.dwe_part, pl->Gd(L.energy_w), st));
    const float* h_top = w.h0 + (long)(N - 1) * B * Hd;
    float* dh_top = w.dh0 + (long)(N - 1) * B * Hd;
    SLNLP_TRY(pl->wgr(w.dq, Hd, B, Hd, h_top, Hd, Hd, pl->Gd(L.query_w), Hd, nullptr, st));
    SLNLP_TRY(pl->dgr(w.dq, Hd, B, Hd, pl->P(L.query_w), Hd, Hd, dh_top, Hd, dh_top, st));
    SLNLP_TRY(pl->wgr(w.dpk, Hd, M, Hd, enc_out, 2 * Hd, 2 * Hd, pl->Gd(L.key_w), 2 * Hd, nullptr, st));
    SLNLP_TRY(pl->dgr(w.dpk, Hd, M, Hd, pl->P(L.key_w), 2 * Hd, 2 * Hd, denc_out, 2 * Hd, denc_out, st));
    // bridge: h0 = tanh(z)
    SLNLP_TRY(tanh_bwd(w.dh0, w.h0, w.dz, (int64_t)N * B * Hd, st));
    SLNLP_TRY(pl->wgr(w.dz, Hd, N * B, Hd, w.enc_final, 2 * Hd, 2 * Hd, pl->Gd(L.bridge_w), 2 * Hd, pl->Gd(L.bridge_b), st));
    SLNLP_TRY(pl->dgr(w.dz, Hd, N * B, Hd, pl->P(L.bridge_w), 2 * Hd, 2 * Hd, w.denc_final, 2 * Hd, nullptr, st));

    // encoder, backward through time, top layer first
    for (int l = N - 1; l >= 0; --l) {
        const EncLayerA& a = w.enc[l];
        const bool last = l == N - 1;
        const int in = L.enc[0][l].in;
        const float* x_in = l > 0 ? w.enc[l - 1].out : w.emb;
        for (int d = 0; d < 2; ++d) {
            SLNLP_TRY(add_rows(w.denc_final + (long)l * B * 2 * Hd + d * Hd, 2 * Hd, a.d[d].dh, Hd, B, Hd, 0, st));
        }
        const int nsl = pl->kslices(), Ks = GH / nsl;
        // one launch per timestep (gemm.hip rnn_step_bwd_kernel: the recurrent dgrad of the step before + this step's cell
        // backward); shapes it does not cover -- and SLNLP_RNN_UNFUSED_BWD=1, the comparison path of the tests -- take the
        // cell kernel + K-sliced grouped GEMM pair
        const bool fused_bwd = rnn_step_bwd_covers(B, Hd) && !pl->unfused_bwd;
        for (int step = S - 1; step >= 0; --step) {
            slnlp_rnn_cell_bwd_dir dirs[2] = {};
            for (int d = 0; d < 2; ++d) {
                const int t = d == 0 ? step : S - 1 - step;
                const EncDirA& e = a.d[d];
                slnlp_rnn_cell_bwd_dir& k = dirs[d];
                k.dh_extra = e.dhx; k.extra_stride = (int64_t)B * Hd;
                k.n_extra = step == S - 1 ? 0 : nsl - 1;   // the first step starts from the final-state gradient only
                k.dh_state = e.dh; k.dc_state = e.dc;
                k.dout = a.dout + (long)t * B * 2 * Hd + d * Hd;
                k.acts = e.acts + (long)t * B * GH;
                k.cprev_save = e.cprev + (long)t * B * Hd;
                k.hprev_save = e.hprev + (long)t * B * Hd;
                k.hn_save = e.hn + (long)t * B * Hd;
                k.dgx = e.dgx + (long)t * B * GH;
                k.dgh = e.dgh + (long)t * B * GH;
                k.carry = e.carry;
                k.t = t; k.out_row0 = t * B; k.out_col0 = d * Hd;
            }
            if (fused_bwd) {
                slnlp_rnn_step_bwd_dir sd[2] = {};
                for (int d = 0; d < 2; ++d) {
                    sd[d].cell = dirs[d];
                    sd[d].cell.dh_extra = nullptr; sd[d].cell.n_extra = 0;
                    if (step < S - 1) {                    // the step processed just before: time t + 1 (forward dir) / t - 1 (backward dir)
                        const int tn = d == 0 ? step + 1 : S - 2 - step;
                        sd[d].dgh_next = a.d[d].dgh + (long)tn * B * GH;
                        sd[d].w_hh = pl->P(L.enc[d][l].w_hh);
                    }
                }
                SLNLP_TRY(rnn_step_bwd(lstm, sd, 2, B, Hd, lengths, 2 * Hd, last ? 0.f : p, RSITE_ENC0 + l, rng, c.precision, st));
                continue;
            }
            SLNLP_TRY(rnn_cell_bwd(lstm, dirs, 2, B, Hd, lengths, 2 * Hd, last ? 0.f : p, RSITE_ENC0 + l, rng, st));
            slnlp_gemm_args rec[8];
            int nj = 0;
            unsigned wide = 0;
            for (int d = 0; d < 2; ++d) {
                const RnnW& q = L.enc[d][l];
                if (nsl <= 4 && !rnn_kslice_wide()) {
                    for (int sl = 0; sl < nsl; ++sl)
                        rec[nj++] = pl->dgr_args(dirs[d].dgh + sl * Ks, GH, B, Ks, pl->P(q.w_hh) + (long)sl * Ks * Hd, Hd, Hd,
                                                 sl == 0 ? a.d[d].dh : a.d[d].dhx + (long)(sl - 1) * B * Hd, Hd,
                                                 sl == 0 ? a.d[d].carry : nullptr);
                    continue;
                }
                // slice 0 (+ carry) -> dh; slices 1 .. nsl-1 as ONE batched job -> dhx[0 .. nsl-2]; 64-column tiles
                rec[nj] = pl->dgr_args(dirs[d].dgh, GH, B, Ks, pl->P(q.w_hh), Hd, Hd, a.d[d].dh, Hd, a.d[d].carry);
                if (rnn_kslice_wide()) wide |= 1u << nj;
                ++nj;
                rec[nj] = pl->dgr_args(dirs[d].dgh + Ks, GH, B, Ks, pl->P(q.w_hh) + (long)Ks * Hd, Hd, Hd, a.d[d].dhx, Hd, nullptr);
                rec[nj].batch = nsl - 1;
                rec[nj].batch_stride_a = Ks; rec[nj].batch_stride_b = (long)Ks * Hd; rec[nj].batch_stride_c = (long)B * Hd;
                if (rnn_kslice_wide()) wide |= 1u << nj;
                ++nj;
            }
            SLNLP_TRY(gemm_group(rec, nj, st, wide));
        }
        float* dx = l > 0 ? w.enc[l - 1].dout : w.demb;
        for (int d = 0; d < 2; ++d) {
            const RnnW& q = L.enc[d][l];
            const EncDirA& e = a.d[d];
            if (pl->use_planes) {
                // both weight gradients and the input gradient of this direction: operands split to planes once, then ONE
                // grouped plane-GEMM launch (every job has >= 256 tiles, so no split-K is needed)
                SLNLP_TRY(split_planes(e.dgx, GH, M, GH, w.dgxp[d].hi, w.dgxp[d].lo, GH, st));
                if (!lstm) SLNLP_TRY(split_planes(e.dgh, GH, M, GH, w.dghp[d].hi, w.dghp[d].lo, GH, st));
                SLNLP_TRY(split_planes(e.hprev, Hd, M, Hd, w.hprevp[d].hi, w.hprevp[d].lo, Hd, st));
                const slnlp_gemm_args jobs[3] = {
                    pl->wgr_p(w.dgxp[d], M, GH, a.xinp, in, pl->Gd(q.w_ih), pl->Gd(q.b_ih)),
                    pl->wgr_p(w.dghp[d], M, GH, w.hprevp[d], Hd, pl->Gd(q.w_hh), pl->Gd(q.b_hh)),
                    pl->dgr_p(w.dgxp[d], M, GH, q.w_ih, in, dx, d == 0 ? nullptr : dx)};
                SLNLP_TRY(gemm_planes_group(jobs, nullptr, 3, nullptr, 0, st));
                continue;
            }
            SLNLP_TRY(pl->wgr(e.dgx, GH, M, GH, x_in, in, in, pl->Gd(q.w_ih), in, pl->Gd(q.b_ih), st));
            SLNLP_TRY(pl->wgr(e.dgh, GH, M, GH, e.hprev, Hd, Hd, pl->Gd(q.w_hh), Hd, pl->Gd(q.b_hh), st));
            SLNLP_TRY(pl->dgr(e.dgx, GH, M, GH, pl->P(q.w_ih), in, in, dx, in, d == 0 ? nullptr : dx, st));
        }
    }
    // src_embed: padding_idx row gets no gradient (bkp.py:374-376)
    SLNLP_TRY(embed_bwd(X, S, B, S, E, c.Vs, w.demb, pl->Gd(L.src_emb), 1.f, c.pad_src, 0.f, 0, rng, w.emb_scratch_src, st));
    return 0;
}

int slnlp_rnn_optim(slnlp_rnn_plan* pl, float momentum, float max_norm, void* stream) {
    SLNLP_CHECK_ARG(pl, "rnn_optim: null plan");
    StepScope scope((hipStream_t)stream);
    SLNLP_TRY(scope.rc);
    return clip_sgd_step(pl->buf.params, pl->buf.grads, pl->buf.momentum, pl->L.total, pl->buf.lr, momentum, max_norm,
                         pl->w.opt_partials, pl->buf.scalars + 1, pl->buf.rng, (hipStream_t)stream);
}

// clip_grad_norm_ + torch.optim.Adam on the arena (any torch optimizer is reachable in the reference through
// pydoc.locate, /root/reference/helper.py:91-104): exp_avg = buf.momentum, exp_avg_sq = the caller's arena-shaped buffer,
// step count = scalars[2] (advanced on the device) -- the same fused kernel as slnlp_tf_optim_adam.
int slnlp_rnn_optim_adam(slnlp_rnn_plan* pl, float* exp_avg_sq, float beta1, float beta2, float eps, float weight_decay,
                         float max_norm, void* stream) {
    SLNLP_CHECK_ARG(pl && exp_avg_sq, "rnn_optim_adam: null argument");
    StepScope scope((hipStream_t)stream);
    SLNLP_TRY(scope.rc);
    return clip_adam_step(pl->buf.params, pl->buf.grads, pl->buf.momentum, exp_avg_sq, pl->L.total, pl->buf.lr, beta1, beta2, eps,
                          weight_decay, max_norm, pl->w.opt_partials, pl->buf.scalars + 1, pl->buf.rng, pl->buf.scalars + 2,
                          (hipStream_t)stream);
}

int slnlp_rnn_set_destroy_sync(slnlp_rnn_plan* pl, int on) {
    SLNLP_CHECK_ARG(pl, "rnn_set_destroy_sync: null plan");
    pl->destroy_sync = on ? 1 : 0;
    return 0;
}

int slnlp_rnn_train_step(slnlp_rnn_plan* pl, const int64_t* X, const int64_t* y, const int64_t* lengths, int B,
                         float momentum, float max_norm, float* logp, void* stream) {
    StepScope scope((hipStream_t)stream);        // one scope for the whole step (the nested entry points re-enter it)
    SLNLP_TRY(scope.rc);
    SLNLP_TRY(slnlp_rnn_forward(pl, X, y, lengths, B, 1, logp, stream));
    SLNLP_TRY(slnlp_rnn_backward(pl, stream));
    return slnlp_rnn_optim(pl, momentum, max_norm, stream);
}

int slnlp_rnn_graph_capture_train(slnlp_rnn_plan* pl, const int64_t* X, const int64_t* y, const int64_t* lengths, int B,
                                  float momentum, float max_norm, float* logp, void* stream) {
    SLNLP_CHECK_ARG(pl && stream, "rnn_graph_capture_train: needs a plan and a non-default stream");
    hipStream_t st = (hipStream_t)stream;
    SLNLP_TRY(pl->prepare_planes(B, (hipStream_t)stream));   // must not be captured: it runs once per batch-size change
    auto old = pl->graphs.find(B);
    if (old != pl->graphs.end()) {
        (void)hipStreamSynchronize(st);
        (void)hipGraphExecDestroy(old->second);
        pl->graphs.erase(old);
    }
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        set_error("rnn_graph_capture_train: begin capture failed: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    int rc = slnlp_rnn_train_step(pl, X, y, lengths, B, momentum, max_norm, logp, stream);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != 0) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess || !g) {
        set_error("rnn_graph_capture_train: end capture failed: %s", hipGetErrorString(e));
        return SLNLP_ERR_LAUNCH;
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        set_error("rnn_graph_capture_train: instantiate failed: %s", hipGetErrorString(e));
        return SLNLP_ERR_LAUNCH;
    }
    pl->graphs[B] = exec;
    return 0;
}

int slnlp_rnn_graph_launch(slnlp_rnn_plan* pl, int B, void* stream) {
    SLNLP_CHECK_ARG(pl, "rnn_graph_launch: null plan");
    auto it = pl->graphs.find(B);
    SLNLP_CHECK_ARG(it != pl->graphs.end(), "rnn_graph_launch: no captured graph for batch %d", B);
    StepScope scope((hipStream_t)stream);
    SLNLP_TRY(scope.rc);
    SLNLP_TRY(pl->prepare_planes(B, (hipStream_t)stream));
    if (hipGraphLaunch(it->second, (hipStream_t)stream) != hipSuccess) {
        set_error("rnn_graph_launch: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    return 0;
}

int slnlp_rnn_tap(slnlp_rnn_plan* pl, const char* name, float* out, int64_t max_floats, int64_t* n_out, void* stream) {
    SLNLP_CHECK_ARG(pl && name && out && pl->last_B > 0, "rnn_tap: bad args / no forward yet");
    const slnlp_rnn_config& c = pl->cfg;
    const int B = pl->last_B, M = B * c.S, Hd = c.Hd, Vp = (int)align_up_r(c.Vt, 4);
    const std::string n(name);
    const float* src = nullptr;
    int64_t rows = 0, cols = 0, ld = 0;
    if (n == "enc_out") { src = pl->w.enc[c.N - 1].out; rows = M; cols = ld = 2 * Hd; }
    else if (n == "enc_final") { src = pl->w.enc_final; rows = (int64_t)c.N * B; cols = ld = 2 * Hd; }
    else if (n == "alphas") { src = pl->w.alphas; rows = B; cols = ld = c.S; }
    else if (n == "context") { src = pl->w.ctx; rows = B; cols = ld = 2 * Hd; }
    else if (n == "dec_out") { src = pl->w.dec[c.N - 1].out; rows = B; cols = ld = Hd; }
    else if (n == "logits") { src = pl->w.logits; rows = B; cols = c.Vt; ld = Vp; }
    SLNLP_CHECK_ARG(src, "rnn_tap: unknown tap '%s'", name);
    SLNLP_CHECK_ARG(rows * cols <= max_floats, "rnn_tap: buffer too small (%ld needed)", (long)(rows * cols));
    if (hipMemcpy2DAsync(out, cols * sizeof(float), src, ld * sizeof(float), cols * sizeof(float), rows,
                         hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
        set_error("rnn_tap: copy failed");
        return SLNLP_ERR_LAUNCH;
    }
    if (n_out) *n_out = rows * cols;
    return 0;
}

}  // extern "C"

// ---- hooks of the lockstep driver (lockstep.hip): the plan struct stays private to this file
namespace slnlp {
int rnn_ls_prepare(slnlp_rnn_plan* pl, int B, hipStream_t st) { return pl->prepare_planes(B, st); }
int rnn_ls_record(slnlp_rnn_plan* pl, const int64_t* X, const int64_t* y, const int64_t* len, int B, int train, float momentum,
                  float max_norm, const LsAdam* adam, float* exp_avg_sq, hipStream_t st) {
    SLNLP_CHECK_ARG(!pl->persistent, "lockstep: the persistent RNN layer kernel needs the GPU to itself -- switch it off");
    SLNLP_TRY(slnlp_rnn_forward(pl, X, y, len, B, train, nullptr, st));
    if (!train) return 0;
    SLNLP_TRY(slnlp_rnn_backward(pl, st));
    if (adam) return slnlp_rnn_optim_adam(pl, exp_avg_sq, adam->beta1, adam->beta2, adam->eps, adam->weight_decay, max_norm, st);
    return slnlp_rnn_optim(pl, momentum, max_norm, st);
}
void rnn_ls_outputs(slnlp_rnn_plan* pl, float* logp, float* loss, const int* dyn) {
    pl->ls_logp = logp; pl->ls_loss = loss; pl->ls_dyn = dyn;
}
void rnn_ls_replayed(slnlp_rnn_plan* pl, int B, int train) {
    pl->last_B = B;
    pl->last_p = train ? pl->cfg.dropout : 0.f;
}
const slnlp_rnn_config* rnn_ls_cfg(slnlp_rnn_plan* pl) { return &pl->cfg; }
}  // namespace slnlp
