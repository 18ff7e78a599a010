// tf_plan.hpp -- the Transformer plan object: arena layout, workspace carving and the per-launch helpers.  Shared by
// tf_plan.hip (the single-fit entry points) and lockstep.hip (K plans advancing through one launch sequence).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.hpp"
#include "launch.hpp"

namespace slnlp {

static inline long align_up(long v, long a) { return (v + a - 1) / a * a; }

struct ParamEnt {
    std::string name;
    int64_t shape[2];
    int ndim;
    int64_t off, numel;
};
struct EncP { long in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b; };
struct DecP {
    long sin_w, sin_b, sout_w, sout_b, cin_w, cin_b, cout_w, cout_b, l1_w, l1_b, l2_w, l2_b;
    long n1_w, n1_b, n2_w, n2_b, n3_w, n3_b;
};
struct Layout {
    std::vector<ParamEnt> ents;
    long src_emb, tgt_emb, encn_w, encn_b, decn_w, decn_b, lin_w, lin_b, total;
    std::vector<EncP> enc;
    std::vector<DecP> dec;
};

// Reference state_dict order (transformer.py:32-47 construction order; the
// *_pos_encoding.pe buffers are not parameters and live outside the arena).
// Every tensor starts on a 16-byte boundary so float4 access is always legal.
static Layout build_layout(const slnlp_tf_config& c) {
    Layout L;
    long cur = 0;
    auto add = [&](const std::string& n, long d0, long d1) -> long {
        ParamEnt e;
        e.name = n;
        e.shape[0] = d0;
        e.shape[1] = d1;
        e.ndim = d1 > 0 ? 2 : 1;
        e.numel = d1 > 0 ? d0 * d1 : d0;
        e.off = cur;
        cur = align_up(cur + e.numel, 4);
        L.ents.push_back(e);
        return e.off;
    };
    const long E = c.E, F = c.F;
    L.src_emb = add("src_embedding.weight", c.Vs, E);
    L.tgt_emb = add("tgt_embedding.weight", c.Vt, E);
    for (int i = 0; i < c.N; ++i) {
        const std::string p = "transformer.encoder.layers." + std::to_string(i) + ".";
        EncP e;
        e.in_w = add(p + "self_attn.in_proj_weight", 3 * E, E);
        e.in_b = add(p + "self_attn.in_proj_bias", 3 * E, 0);
        e.out_w = add(p + "self_attn.out_proj.weight", E, E);
        e.out_b = add(p + "self_attn.out_proj.bias", E, 0);
        e.l1_w = add(p + "linear1.weight", F, E);
        e.l1_b = add(p + "linear1.bias", F, 0);
        e.l2_w = add(p + "linear2.weight", E, F);
        e.l2_b = add(p + "linear2.bias", E, 0);
        e.n1_w = add(p + "norm1.weight", E, 0);
        e.n1_b = add(p + "norm1.bias", E, 0);
        e.n2_w = add(p + "norm2.weight", E, 0);
        e.n2_b = add(p + "norm2.bias", E, 0);
        L.enc.push_back(e);
    }
    L.encn_w = add("transformer.encoder.norm.weight", E, 0);
    L.encn_b = add("transformer.encoder.norm.bias", E, 0);
    for (int i = 0; i < c.N; ++i) {
        const std::string p = "transformer.decoder.layers." + std::to_string(i) + ".";
        DecP d;
        d.sin_w = add(p + "self_attn.in_proj_weight", 3 * E, E);
        d.sin_b = add(p + "self_attn.in_proj_bias", 3 * E, 0);
        d.sout_w = add(p + "self_attn.out_proj.weight", E, E);
        d.sout_b = add(p + "self_attn.out_proj.bias", E, 0);
        d.cin_w = add(p + "multihead_attn.in_proj_weight", 3 * E, E);
        d.cin_b = add(p + "multihead_attn.in_proj_bias", 3 * E, 0);
        d.cout_w = add(p + "multihead_attn.out_proj.weight", E, E);
        d.cout_b = add(p + "multihead_attn.out_proj.bias", E, 0);
        d.l1_w = add(p + "linear1.weight", F, E);
        d.l1_b = add(p + "linear1.bias", F, 0);
        d.l2_w = add(p + "linear2.weight", E, F);
        d.l2_b = add(p + "linear2.bias", E, 0);
        d.n1_w = add(p + "norm1.weight", E, 0);
        d.n1_b = add(p + "norm1.bias", E, 0);
        d.n2_w = add(p + "norm2.weight", E, 0);
        d.n2_b = add(p + "norm2.bias", E, 0);
        d.n3_w = add(p + "norm3.weight", E, 0);
        d.n3_b = add(p + "norm3.bias", E, 0);
        L.dec.push_back(d);
    }
    L.decn_w = add("transformer.decoder.norm.weight", E, 0);
    L.decn_b = add("transformer.decoder.norm.bias", E, 0);
    L.lin_w = add("linear.weight", c.Vt, E);
    L.lin_b = add("linear.bias", c.Vt, 0);
    L.total = cur;
    return L;
}

static int check_cfg(const slnlp_tf_config* c) {
    SLNLP_CHECK_ARG(c, "tf: null config");
    SLNLP_CHECK_ARG(c->E > 0 && c->H > 0 && c->E % c->H == 0, "tf: E=%d not divisible by H=%d", c->E, c->H);
    const int dh = c->E / c->H;
    SLNLP_CHECK_ARG(c->E % 4 == 0 && c->E <= 1024, "tf: E=%d must be a multiple of 4 and <= 1024", c->E);
    SLNLP_CHECK_ARG(dh % 4 == 0 && dh <= 256 && (dh <= 64 || dh % 64 == 0), "tf: head_dim %d unsupported", dh);
    SLNLP_CHECK_ARG(c->H <= 64, "tf: num_heads %d > 64 (per-head LDS tables of the cross-attention backward)", c->H);
    SLNLP_CHECK_ARG(c->F > 0 && c->F % 4 == 0, "tf: hidden_size %d must be a multiple of 4", c->F);
    SLNLP_CHECK_ARG(c->N > 0 && c->Vs > 1 && c->Vt > 1, "tf: bad N/vocab");
    SLNLP_CHECK_ARG(c->B > 0 && c->B <= 1024, "tf: batch %d outside 1..1024", c->B);
    // S <= 64: one-tile MFMA attention; longer sequences take the wave-per-row kernels (attention_long.hip).  5000 = rows of
    // the reference's positional table (positional_encoding.py:23); B * S <= 65536: the embedding backward's chunk table
    SLNLP_CHECK_ARG(c->S > 0 && c->S <= 5000, "tf: seq_len %d outside 1..5000", c->S);
    SLNLP_CHECK_ARG((long)c->B * c->S <= 65536, "tf: batch %d x seq_len %d exceeds 65536 tokens per step", c->B, c->S);
    SLNLP_CHECK_ARG(c->dropout >= 0.f && c->dropout < 1.f, "tf: dropout %f", c->dropout);
    SLNLP_CHECK_ARG(c->precision == 1 || c->precision == 3 || c->precision == 8, "tf: precision %d (1, 3 or 8)", c->precision);
    SLNLP_CHECK_ARG(c->precision != 8 || (c->E % 128 == 0 && c->F % 128 == 0),
                    "tf: precision 8 (fp8 forward products) needs embedding_size and hidden_size to be multiples of 128");
    return 0;
}

// ------------------------------------------------------------- workspace ----
struct Bump {
    char* base;
    size_t cur = 0;
    explicit Bump(void* b) : base((char*)b) {}
    template <typename T>
    T* take(size_t n) {
        cur = (cur + 255) & ~(size_t)255;
        T* p = (T*)(base + cur);
        cur += n * sizeof(T);
        return p;
    }
};

// forward activations kept for backward + this layer's gradient buffers.  Every gradient buffer is
// written exactly once per step: there is nothing to overwrite until the next step.
// bf16 hi/lo planes of a GEMM operand (same logical shape / row stride as its fp32 twin, rows
// zero-padded to a multiple of 64): written once by the producer, read by gemm_planes.hip
struct PP {
    unsigned short *hi = nullptr, *lo = nullptr;
    unsigned char* q8 = nullptr;        // precision 8: forward operands also as an e4m3 plane
    PlaneOut out() const { PlaneOut o; o.hi = hi; o.lo = lo; o.q8 = q8; return o; }
};
struct EncA {
    float *qkv, *probs, *ctx, *y1, *st1, *x1, *h, *y2, *st2, *x2, *lnp1, *lnp2;
    float *gA2, *gB2, *gh, *gx1, *gA1, *gB1, *gctx, *gqkv, *gx0;
    PP ctxp, x1p, hp, x2p, d2p, ghp, d1p, gqkvp;
};
struct DecA {
    float *v, *y1, *st1, *t1, *q, *xprobs, *xctx, *y2, *st2, *t2, *h, *y3, *st3, *t3, *lnp1, *lnp2, *lnp3;
    float *qk, *mbar, *psum;                 // cross-attention without K / V projections (attention_mem.hip): kept for the backward
    float *gA3, *gB3, *gh, *gt2, *gA2, *gB2, *gxctx, *gq, *gt1, *gA1, *gB1, *gv, *gt0;
    float *dmbar, *dsc, *dqk, *dcp;          // its gradients: d mbar, d scores, d qk [B*H, .], d ctx * sum_s p_s [B, E]
    PP vp, t1p, xctxp, t2p, hp, t3p;         // operands of the layer's B-row products as planes (gemm_rows.hip); rows padded to 64
    PP d3p, ghp, d2p, gqp, d1p, gvp;         // ... and the gradients its backward pairs contract (dY of linear2, linear1, cross out / q, self out / v)
};
struct Ws {
    float *x0, *t0, *mem, *st_mem, *lnp_mem, *tfin, *st_fin, *lnp_fin, *logits, *dlogits, *logp, *row_nll;
    std::vector<EncA> enc;
    std::vector<DecA> dec;
    float *gfin, *gtl, *gmem, *gxl;     // d tfin, d t_last, d memory, d x_last
    void *emb_scratch_src, *emb_scratch_tgt;
    unsigned char* emb_keep;   // 4 keep bits per float4 of the source embedding's dropout (read by its backward)
    PP x0p, memp, wp;                   // wp: planes of the whole parameter arena (same offsets)
    PP t0p, tfinp;                      // target embedding / final decoder LayerNorm as planes (operands of B-row products)
    unsigned char* wq;                  // precision 8: e4m3 plane of the arena (same offsets; only the GEMM weight rows are filled)
    float* wscale;                      // [qrows] per-row scales
    QuantRow* qrow_table;               // [qrows] device table for the quantiser
    long n_qrows;
    char *planes_begin, *planes_end;    // activation planes region (re-zeroed when the batch size changes)
    float* opt_partials;
    float* attn_scratch;    // S > 64: dS of one self-attention backward ([B,H,S,S], shared by all layers)
    char* gscr[1];          // split-K scratch of the grouped GEMM launches (one stream: one scratch)
    size_t gscr_bytes;
    slnlp_ln_reduce_entry* ln_table;
    LnPartialEntry* ln_ptable;   // the same LayerNorms' (dy, x, stats, partial) for the one ln_param_partial launch of a backward
    size_t bytes;
};

constexpr int MAX_SPLITK = WD_MAX_SPLITK;
// weight-gradient GEMMs contract over the T tokens: aim at ~10 K-tiles (of 64) per workgroup
static int splitk_for(int T) {
    const int n = ((T + 63) / 64 + 5) / 10;
    return n < 1 ? 1 : n > MAX_SPLITK ? MAX_SPLITK : n;
}

static Ws carve(const slnlp_tf_config& c, void* base) {
    Ws w;
    Bump b(base);
    const size_t B = c.B, S = c.S, E = c.E, F = c.F, H = c.H, M = B * S, Vp = align_up(c.Vt, 4);
    // (dgamma, dbeta) chunk sums per LayerNorm: [chunks of the full batch][2][E] (encoder: S B rows, decoder: B rows)
    const size_t lnpE = (size_t)ln_bwd_blocks((int)(B * S)) * 2 * E, lnpD = (size_t)ln_bwd_blocks((int)B) * 2 * E;
    // plane path with single-tile attention (E, F multiples of 64, S <= 64): the attention context, d qkv, the gated FFN gradient and the
    // dropout-masked LayerNorm gradients exist as bf16 planes only -- their fp32 twins have no writer and no reader and are not carved
    // (34 MB per layer at cfg2; the debug layout omits them)
    const bool lean = (E % 64 == 0) && (F % 64 == 0) && S <= 64;
    auto opt = [&](size_t n) -> float* { return lean ? nullptr : b.take<float>(n); };
    w.x0 = b.take<float>(M * E);
    w.t0 = b.take<float>(B * E);
    for (int i = 0; i < c.N; ++i) {
        EncA a;
        a.qkv = b.take<float>(M * 3 * E);
        a.probs = b.take<float>(B * H * S * S);
        a.ctx = opt(M * E);
        a.y1 = b.take<float>(M * E);
        a.st1 = b.take<float>(M * 2);
        a.x1 = b.take<float>(M * E);
        a.h = b.take<float>(M * F);
        a.y2 = b.take<float>(M * E);
        a.st2 = b.take<float>(M * 2);
        a.x2 = b.take<float>(M * E);
        a.lnp1 = b.take<float>(lnpE);
        a.lnp2 = b.take<float>(lnpE);
        a.gA2 = b.take<float>(M * E);
        a.gB2 = opt(M * E);
        a.gh = opt(M * F);
        a.gx1 = b.take<float>(M * E);
        a.gA1 = b.take<float>(M * E);
        a.gB1 = opt(M * E);
        a.gctx = b.take<float>(M * E);
        a.gqkv = opt(M * 3 * E);
        a.gx0 = b.take<float>(M * E);
        w.enc.push_back(a);
    }
    w.mem = b.take<float>(M * E);
    w.st_mem = b.take<float>(M * 2);
    w.lnp_mem = b.take<float>(lnpE);
    for (int i = 0; i < c.N; ++i) {
        DecA a;
        a.v = b.take<float>(B * E);
        a.y1 = b.take<float>(B * E);
        a.st1 = b.take<float>(B * 2);
        a.t1 = b.take<float>(B * E);
        a.q = b.take<float>(B * E);
        a.qk = b.take<float>(B * H * E);
        a.mbar = b.take<float>(B * H * E);
        a.psum = b.take<float>(B * H);
        a.xprobs = b.take<float>(B * H * S);
        a.xctx = b.take<float>(B * E);
        a.y2 = b.take<float>(B * E);
        a.st2 = b.take<float>(B * 2);
        a.t2 = b.take<float>(B * E);
        a.h = b.take<float>(B * F);
        a.y3 = b.take<float>(B * E);
        a.st3 = b.take<float>(B * 2);
        a.t3 = b.take<float>(B * E);
        a.lnp1 = b.take<float>(lnpD);
        a.lnp2 = b.take<float>(lnpD);
        a.lnp3 = b.take<float>(lnpD);
        a.gA3 = b.take<float>(B * E);
        a.gB3 = b.take<float>(B * E);
        a.gh = b.take<float>(B * F);
        a.gt2 = b.take<float>(B * E);
        a.gA2 = b.take<float>(B * E);
        a.gB2 = b.take<float>(B * E);
        a.gxctx = b.take<float>(B * E);
        a.gq = b.take<float>(B * E);
        a.dmbar = b.take<float>(B * H * E);
        a.dsc = b.take<float>(B * H * S);
        a.dqk = b.take<float>(B * H * E);
        a.dcp = b.take<float>(B * E);
        a.gt1 = b.take<float>(B * E);
        a.gA1 = b.take<float>(B * E);
        a.gB1 = b.take<float>(B * E);
        a.gv = b.take<float>(B * E);
        a.gt0 = b.take<float>(B * E);
        w.dec.push_back(a);
    }
    w.tfin = b.take<float>(B * E);
    w.st_fin = b.take<float>(B * 2);
    w.lnp_fin = b.take<float>(lnpD);
    w.logits = b.take<float>(B * Vp);
    w.dlogits = b.take<float>(B * Vp);
    w.logp = b.take<float>(B * c.Vt);
    w.row_nll = b.take<float>(B);
    w.gfin = b.take<float>(B * E);
    w.gtl = b.take<float>(B * E);
    w.gmem = b.take<float>(M * E);
    w.gxl = b.take<float>(M * E);
    w.emb_scratch_src = b.take<char>(embed_bwd_scratch_bytes(c.B, c.S, c.E));
    w.emb_scratch_tgt = b.take<char>(embed_bwd_scratch_bytes(c.B, 1, c.E));
    w.emb_keep = b.take<unsigned char>(M * E / 4);
    w.opt_partials = b.take<float>(1024);
    w.attn_scratch = c.S > 64 ? (float*)b.take<char>(attn_long_scratch_bytes(c.B, c.S, c.H)) : nullptr;
    w.ln_table = b.take<slnlp_ln_reduce_entry>(5 * c.N + 2);
    w.ln_ptable = b.take<LnPartialEntry>(5 * c.N + 2);
    // ---- bf16 operand planes (only used when E and F are multiples of 64)
    const size_t Mp = (M + 63) / 64 * 64;
    const bool q8 = c.precision == 8;
    auto pp = [&](size_t cols, bool fwd_operand = false) {
        PP q;
        q.hi = b.take<unsigned short>(Mp * cols);
        q.lo = b.take<unsigned short>(Mp * cols);
        if (q8 && fwd_operand) q.q8 = b.take<unsigned char>(Mp * cols);
        return q;
    };
    const size_t wtot = (size_t)build_layout(c).total + 64 * 3 * (E > F ? E : F);   // tail pad: tiles may over-read rows
    w.wp.hi = b.take<unsigned short>(wtot);
    w.wp.lo = b.take<unsigned short>(wtot);
    w.n_qrows = (long)c.N * (3 * E + E + F + E);                          // encoder in_proj, out_proj, linear1, linear2
    w.wq = q8 ? b.take<unsigned char>(wtot) : nullptr;
    w.wscale = q8 ? b.take<float>(w.n_qrows) : nullptr;
    w.qrow_table = q8 ? b.take<QuantRow>(w.n_qrows) : nullptr;
    b.cur = (b.cur + 255) & ~(size_t)255;
    w.planes_begin = b.base + b.cur;
    w.x0p = pp(E, true);
    w.memp = pp(E, true);
    for (int i = 0; i < c.N; ++i) {
        EncA& a = w.enc[i];
        a.ctxp = pp(E, true); a.x1p = pp(E, true); a.hp = pp(F, true); a.x2p = pp(E, true);
        a.d2p = pp(E); a.ghp = pp(F); a.d1p = pp(E); a.gqkvp = pp(3 * E);
    }
    {   // the decoder's B-row operands (rows padded to 64: the padding stays zero from the region's memset)
        const size_t Bp = (B + 63) / 64 * 64;
        auto ppd = [&](size_t cols) {
            PP q;
            q.hi = b.take<unsigned short>(Bp * cols);
            q.lo = b.take<unsigned short>(Bp * cols);
            return q;
        };
        w.t0p = ppd(E);
        w.tfinp = ppd(E);
        for (int i = 0; i < c.N; ++i) {
            DecA& a = w.dec[i];
            a.vp = ppd(E); a.t1p = ppd(E); a.xctxp = ppd(E); a.t2p = ppd(E); a.hp = ppd(F); a.t3p = ppd(E);
            a.d3p = ppd(E); a.ghp = ppd(F); a.d2p = ppd(E); a.gqp = ppd(E); a.d1p = ppd(E); a.gvp = ppd(E);
        }
    }
    {   // grouped-launch scratch lives in the zero-on-demand region: its arrival counters must start at zero
        // (the weight gradients' partial tiles: [3E or F] x [E or F] outputs rounded up to the widest (256 x 256) tile, x MAX_SPLITK)
        const size_t nx = ((E > F ? E : F) + 255) / 256 * 256, ny = ((3 * E > F ? 3 * E : F) + 255) / 256 * 256;
        w.gscr_bytes = 16384 + nx * ny * MAX_SPLITK * sizeof(float) + ny * MAX_SPLITK * sizeof(float);
        w.gscr_bytes = (w.gscr_bytes + 255) & ~(size_t)255;
        w.gscr[0] = b.take<char>(w.gscr_bytes);
    }
    b.cur = (b.cur + 255) & ~(size_t)255;
    w.planes_end = b.base + b.cur;
    w.bytes = (b.cur + 255) & ~(size_t)255;
    return w;
}

}  // namespace slnlp

using namespace slnlp;   // internal header, included only by the plan translation units

// dropout site ids
enum { SITE_SRC_EMB = 1, SITE_TGT_EMB = 2, SITE_LAYER0 = 16, SITE_PER_LAYER = 8 };

struct slnlp_tf_plan {
    slnlp_tf_config cfg;
    slnlp_tf_buffers buf;
    Layout L;
    Ws w;
    int last_B = 0;       // batch of the last forward
    float last_p = 0.f;   // dropout used by the last forward (0 in eval)
    const int64_t* last_X = nullptr;
    const int64_t* last_y = nullptr;
    std::map<int, hipGraphExec_t> graphs;   // one captured train step per batch size, kept until destroy
    int nbE = 0, nbD = 0;  // (dgamma, dbeta) chunk counts of the FULL batch (fixed: the reduce table is static)
    int destroy_sync = 1;  // slnlp_tf_set_destroy_sync: wait for the device before the plan goes away (launch.hpp)
    // Lockstep (lockstep.hip): where this fit's per-step outputs go while it advances as one of K fits -- an epoch-long
    // log-prob buffer and a per-batch loss history, indexed through two device scalars the driver updates per step
    float* ls_logp = nullptr;       // [rows of the epoch, Vt]
    float* ls_loss = nullptr;       // [batches of the epoch]
    const int* ls_dyn = nullptr;    // {first row of the batch, index of the batch}
    // precision 8: the forward products run on the fp8 MFMA (e4m3 activations, scale 1; e4m3 weights with one scale per
    // output row, re-quantised from the fp32 master weights whenever the arena has moved); the backward stays split-bf16
    int prec3() const { return cfg.precision == 8 ? 3 : cfg.precision; }
    unsigned long long wq_gen = 0;
    std::map<long, long> qrow0;          // arena offset of a quantised weight block -> its first row in wscale
    int ensure_wq(hipStream_t st) {
        if (cfg.precision != 8) return 0;
        unsigned long long g = params_generation(buf.params);
        if (g != 0 && g == wq_gen) return 0;
        SLNLP_TRY(quant_rows_fp8(buf.params, 0, (int)w.n_qrows, 0, w.wq, 0, w.wscale, w.qrow_table, st));
        if (g == 0) g = bump_params_generation(buf.params);
        wq_gen = g;
        return 0;
    }
    unsigned long long wplanes_gen = 0;   // generation of the parameter arena the weight planes were made from (0: never)
    // weights as bf16 planes: made by the optimizer kernel of the previous step, or here when the arena has changed since
    int ensure_wplanes(hipStream_t st) {
        if (!use_planes) return 0;
        unsigned long long g = params_generation(buf.params);
        if (g != 0 && g == wplanes_gen) return 0;
        SLNLP_TRY(split_planes(buf.params, L.total, 1, (int)L.total, w.wp.hi, w.wp.lo, L.total, st));
        if (g == 0) g = bump_params_generation(buf.params);
        wplanes_gen = g;
        return 0;
    }
    // the arena range whose planes have readers: the encoder layers' weights (plane GEMMs) and, for plans whose batches are too tall
    // for the B-row kernel (rows_for), the decoder's and the generator's.  Everything else -- the embedding tables, the decoder's weights
    // in a plan of B-row products (gemm_rows.hip splits them in registers) -- is read as fp32.  The optimizer kernels write planes for
    // this range only
    long wplane_begin() const { return L.enc.empty() ? 0 : L.enc[0].in_w; }
    long wplane_end() const { return L.enc.empty() ? 0 : (use_rows && cfg.B <= ROWS_MAX_B ? L.encn_w : L.total); }
    // The decoder's products of B rows: the B-row kernel up to one block of 64 rows (a launch lasts as long as one workgroup loads; 16 x 16
    // tiles), the plane GEMM above (configs[4], B = 256, E = 1024: 15 us per gradient pair against 22 -- 47 before the bias gradient
    // became an MFMA product -- and 11 against 13 forward; tools/bench_rows_shapes.py)
    static constexpr int ROWS_MAX_B = 64;
    bool rows_for(int B, int drop_head_dim) const { return B <= ROWS_MAX_B || drop_head_dim != 0; }   // (per-head dropout is not built into the plane GEMM)
    // the optimizer just rewrote the arena (and, with planes, the planes with it)
    void params_stepped() {
        const unsigned long long g = bump_params_generation(buf.params);
        if (use_planes) wplanes_gen = g;
    }
    bool use_planes = false;   // E, F multiples of 64: M = S*B GEMMs run on pre-split bf16 planes (gemm_planes.hip)
    bool use_rows = false;     // ... and the decoder's B-row products on planes, register-direct (gemm_rows.hip; K <= 1024)
    // split-bf16 passes of the plane GEMM's gradient products: the process default AT CREATION (slnlp_set_backward_passes), fixed for the
    // plan's life -- a captured graph, a recorded lockstep program and every host thread that steps this plan issue the same products
    int wgrad_np = 2, dgrad_np = 2;
    int planes_B = -1;         // batch size the activation planes' zero padding is valid for

    float* P(long off) const { return buf.params + off; }
    float* G(long off) const { return buf.grads + off; }
    int enc_site(int l, int k) const { return SITE_LAYER0 + l * SITE_PER_LAYER + k; }
    int dec_site(int l, int k) const { return SITE_LAYER0 + (cfg.N + l) * SITE_PER_LAYER + k; }

    int dec_self_block(int l, const float* t, const PP* tp, int B, float p, hipStream_t st) const;

    // y[M,N] = x[M,K] W[N,K]^T + b  (+relu) (+dropout) (+resid)
    int linear(const float* x, int M, int K, const float* W, int N, const float* bias, float* y, long ldy, int relu,
               float p, int site, const float* resid, hipStream_t st, int drop_head_dim = 0) const {
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A = x; a.lda = K; a.a_kmajor = 1;
        a.B = W; a.ldb = K; a.b_kmajor = 1;
        a.C = y; a.ldc = ldy; a.M = M; a.N = N; a.K = K;
        a.bias = bias; a.relu = relu;
        a.drop_p = p; a.drop_site = site; a.rng = buf.rng;
        a.resid = resid; a.ldr = ldy;
        a.precision = prec3();
        a.drop_head_dim = drop_head_dim;
        return gemm(a, st);
    }
    // dx[M,Kin] = dy[M,Nout] W[Nout,Kin]  (*gate) (+resid)
    slnlp_gemm_args dgrad_args(const float* dy, long ldy, int M, int Nout, const float* W, int Kin, float* dx,
                               const float* gate, float gate_scale, const float* resid) const {
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A = dy; a.lda = ldy; a.a_kmajor = 1;
        a.B = W; a.ldb = Kin; a.b_kmajor = 0;
        a.C = dx; a.ldc = Kin; a.M = M; a.N = Kin; a.K = Nout;
        a.gate = gate; a.ldg = Kin; a.gate_scale = gate_scale;
        a.resid = resid; a.ldr = Kin;
        a.precision = prec3();
        return a;
    }
    int dgrad(const float* dy, long ldy, int M, int Nout, const float* W, int Kin, float* dx, const float* gate,
              float gate_scale, const float* resid, hipStream_t st) const {
        return gemm(dgrad_args(dy, ldy, M, Nout, W, Kin, dx, gate, gate_scale, resid), st);
    }
    // dW[Nout,Kin] = dy[T,Nout]^T x[T,Kin];  db[Nout] = colsum(dy)
    slnlp_gemm_args wgrad_args(const float* dy, long ldy, int T, int Nout, const float* x, int Kin, float* dW, float* db) const {
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A = dy; a.lda = ldy; a.a_kmajor = 0;
        a.B = x; a.ldb = Kin; a.b_kmajor = 0;
        a.C = dW; a.ldc = Kin; a.M = Nout; a.N = Kin; a.K = T;
        a.rowsum_a = db;
        a.precision = prec3();
        return a;
    }
    int wgrad(const float* dy, long ldy, int T, int Nout, const float* x, int Kin, float* dW, float* db,
              hipStream_t st) const {
        return gemm(wgrad_args(dy, ldy, T, Nout, x, Kin, dW, db), st);
    }
    // `H` GEMMs of one shape in one job (gemm.hip, batched jobs): GEMM h reads A + h*sa, B + h*sb and writes C + h*sc
    static slnlp_gemm_args batched(slnlp_gemm_args a, int H, long sa, long sb, long sc) {
        a.batch = H; a.batch_stride_a = sa; a.batch_stride_b = sb; a.batch_stride_c = sc;
        return a;
    }
    // per-head products of the decoder's cross-attention (attention_mem.hip); W = rows h*dh.. of a [E, E] block of in_proj
    // x[B, H*dh] (columns h*dh..) -> out[B, H, E]:  out_h = x_h W_h      (qk = Wk_h^T q_h;  d mbar = Wv_h^T d ctx_h)
    slnlp_gemm_args head_expand(const float* x, const float* W, float* out, int B, int H, int dh) const {
        const int E = H * dh;
        slnlp_gemm_args a = dgrad_args(x, E, B, dh, W, E, out, nullptr, 0.f, nullptr);
        a.ldc = (long)H * E;
        return batched(a, H, dh, (long)dh * E, E);
    }
    // x[B, H, E] -> out[B, H*dh] (columns h*dh..):  out_h = x_h W_h^T (+ resid in place)      (ctx_h = Wv_h mbar;  d q_h = Wk_h d qk)
    slnlp_gemm_args head_reduce(const float* x, const float* W, float* out, const float* resid, int B, int H, int dh) const {
        const int E = H * dh;
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A = x; a.lda = (long)H * E; a.a_kmajor = 1;
        a.B = W; a.ldb = E; a.b_kmajor = 1;
        a.C = out; a.ldc = E; a.M = B; a.N = dh; a.K = E;
        a.resid = resid; a.ldr = E;
        a.precision = prec3();
        return batched(a, H, E, (long)dh * E, dh);
    }
    // dW_h[dh, E] = dy_h^T x_h:  dy[B, H*dh] (columns h*dh..), x[B, H, E], dW = rows h*dh.. of an [E, E] gradient block
    slnlp_gemm_args head_wgrad(const float* dy, const float* x, float* dW, int B, int H, int dh) const {
        const int E = H * dh;
        slnlp_gemm_args a = wgrad_args(dy, E, B, dh, x, E, dW, nullptr);
        a.ldb = (long)H * E;
        return batched(a, H, dh, E, (long)dh * E);
    }
    // weight- and data-gradient of one dY (fp32 operands) in one launch
    int wd_group_f(const slnlp_gemm_args& wg, const slnlp_gemm_args& dg, hipStream_t st) const {
        const slnlp_gemm_args jobs[2] = {wg, dg};
        return gemm_group(jobs, 2, st);
    }
    // ---- the same three GEMM roles over pre-split planes; weights: planes of the arena at offset woff
    int linear_p(const PP& x, int M, int K, long woff, int N, const float* bias, float* y, long ldy, int relu, float p,
                 int site, const float* resid, const PP* outp, hipStream_t st) const {
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        if (cfg.precision == 8) {       // fp8 forward product: e4m3 planes, per-row weight scales
            a.A_hi = reinterpret_cast<const uint16_t*>(x.q8); a.lda_p = K; a.a_kmajor = 1;
            a.B_hi = reinterpret_cast<const uint16_t*>(w.wq + woff); a.ldb_p = K; a.b_kmajor = 1;
            a.col_scale = w.wscale + qrow0.at(woff);
            a.C = y; a.ldc = ldy; a.M = M; a.N = N; a.K = K;
            a.bias = bias; a.relu = relu;
            a.drop_p = p; a.drop_site = site; a.rng = buf.rng;
            a.resid = resid; a.ldr = ldy;
            if (outp) { a.C_hi = outp->hi; a.C_lo = outp->lo; a.C_q8 = outp->q8; a.ldc_p = N; }
            a.precision = 8;
            return gemm(a, st);
        }
        a.A_hi = x.hi; a.A_lo = x.lo; a.lda_p = K; a.a_kmajor = 1;
        a.B_hi = w.wp.hi + woff; a.B_lo = w.wp.lo + woff; a.ldb_p = K; a.b_kmajor = 1;
        a.C = y; a.ldc = ldy; a.M = M; a.N = N; a.K = K;
        a.bias = bias; a.relu = relu;
        a.drop_p = p; a.drop_site = site; a.rng = buf.rng;
        a.resid = resid; a.ldr = ldy;
        if (outp) { a.C_hi = outp->hi; a.C_lo = outp->lo; a.ldc_p = N; }
        a.precision = prec3();
        return gemm(a, st);
    }
    // the same product for the decoder's B rows: x as planes, W as fp32 (split in registers), both register-direct (gemm_rows.hip)
    int linear_r(const PP& x, int M, int K, long woff, int N, const float* bias, float* y, long ldy, int relu, float p, int site,
                 const float* resid, const PP* outp, hipStream_t st, int drop_head_dim = 0) const {
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A_hi = x.hi; a.A_lo = x.lo; a.lda_p = K; a.a_kmajor = 1;
        const bool rows = rows_for(M, drop_head_dim);
        if (rows) { a.B = P(woff); a.ldb = K; }
        else { a.B_hi = w.wp.hi + woff; a.B_lo = w.wp.lo + woff; a.ldb_p = K; }
        a.b_kmajor = 1;
        a.C = y; a.ldc = ldy; a.M = M; a.N = N; a.K = K;
        a.bias = bias; a.relu = relu;
        a.drop_p = p; a.drop_site = site; a.rng = buf.rng;
        a.resid = resid; a.ldr = ldy;
        if (outp) { a.C_hi = outp->hi; a.C_lo = outp->lo; a.ldc_p = N; }
        a.precision = prec3();
        a.drop_head_dim = drop_head_dim;
        return rows ? gemm_rows(a, st) : gemm(a, st);
    }
    // the backward pair of a decoder Linear y[B, Nout] = x[B, Kin] W^T + b in ONE launch (gemm_rows.hip: gemm_rows_bwd): dX = dY W with
    // its epilogue (gate, per-head dropout, residual; fp32 and / or planes out) and dW = dY^T x, db = colsum(dY)
    int wd_rows(const PP& dy, int B, int Nout, long woff, int Kin, float* dx, const PP* dxp, const float* gate, float gate_scale,
                const float* resid, const PP& x, long gw, long gb, hipStream_t st, float drop_p = 0.f, int drop_site = 0, int drop_head_dim = 0) const {
        if (!rows_for(B, drop_head_dim))       // the plane GEMM's gradient pair, as in the encoder
            return wd_group(wgrad_p_args(dy, Nout, B, Nout, x, Kin, G(gw), G(gb)),
                            dgrad_p_args(dy, Nout, B, Nout, woff, Kin, dx, gate, gate_scale, resid, dxp), 0, st);
        slnlp_gemm_args d, g;
        memset(&d, 0, sizeof(d));
        memset(&g, 0, sizeof(g));
        d.A_hi = dy.hi; d.A_lo = dy.lo; d.lda_p = Nout; d.a_kmajor = 1;
        d.B = P(woff); d.ldb = Kin; d.b_kmajor = 0;
        d.C = dx; d.ldc = Kin; d.M = B; d.N = Kin; d.K = Nout;
        d.gate = gate; d.ldg = Kin; d.gate_scale = gate_scale;
        d.resid = resid; d.ldr = Kin;
        if (dxp) { d.C_hi = dxp->hi; d.C_lo = dxp->lo; d.ldc_p = Kin; }
        d.drop_p = drop_p; d.drop_site = drop_site; d.rng = buf.rng; d.drop_head_dim = drop_head_dim;
        d.precision = prec3();
        g.A_hi = dy.hi; g.A_lo = dy.lo; g.lda_p = Nout; g.a_kmajor = 0;
        g.B_hi = x.hi; g.B_lo = x.lo; g.ldb_p = Kin; g.b_kmajor = 0;
        g.C = G(gw); g.ldc = Kin; g.M = Nout; g.N = Kin; g.K = B;
        g.rowsum_a = G(gb);
        g.precision = prec3();
        return gemm_rows_bwd(d, g, st);
    }
    slnlp_gemm_args dgrad_p_args(const PP& dy, long ldy, int M, int Nout, long woff, int Kin, float* dx, const float* gate,
                                 float gate_scale, const float* resid, const PP* outp) const {
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A_hi = dy.hi; a.A_lo = dy.lo; a.lda_p = ldy; a.a_kmajor = 1;
        a.B_hi = w.wp.hi + woff; a.B_lo = w.wp.lo + woff; a.ldb_p = Kin; a.b_kmajor = 0;
        a.C = dx; a.ldc = Kin; a.M = M; a.N = Kin; a.K = Nout;
        a.gate = gate; a.ldg = Kin; a.gate_scale = gate_scale;
        a.resid = resid; a.ldr = Kin;
        if (outp) { a.C_hi = outp->hi; a.C_lo = outp->lo; a.ldc_p = Kin; }
        a.precision = prec3() == 3 ? dgrad_np : prec3();
        return a;
    }
    int dgrad_p(const PP& dy, long ldy, int M, int Nout, long woff, int Kin, float* dx, const float* gate,
                float gate_scale, const float* resid, const PP* outp, hipStream_t st) const {
        return gemm(dgrad_p_args(dy, ldy, M, Nout, woff, Kin, dx, gate, gate_scale, resid, outp), st);
    }
    slnlp_gemm_args wgrad_p_args(const PP& dy, long ldy, int T, int Nout, const PP& x, int Kin, float* dW, float* db) const {
        slnlp_gemm_args a;
        memset(&a, 0, sizeof(a));
        a.A_hi = dy.hi; a.A_lo = dy.lo; a.lda_p = ldy; a.a_kmajor = 0;
        a.B_hi = x.hi; a.B_lo = x.lo; a.ldb_p = Kin; a.b_kmajor = 0;
        a.C = dW; a.ldc = Kin; a.M = Nout; a.N = Kin; a.K = T;
        a.rowsum_a = db;
        a.precision = prec3() == 3 ? wgrad_np : prec3();
        return a;
    }
    int wgrad_p(const PP& dy, long ldy, int T, int Nout, const PP& x, int Kin, float* dW, float* db, hipStream_t st) const {
        return gemm(wgrad_p_args(dy, ldy, T, Nout, x, Kin, dW, db), st);
    }
    // weight gradient (split-K over the tokens) and data gradient of one dY in ONE launch: the wgrad's workgroups
    // fill the CUs the dgrad leaves idle, and there is no cross-queue edge to pay for (measured 4-10 us each)
    int wd_group(const slnlp_gemm_args& wg, const slnlp_gemm_args& dg, int which_scratch, hipStream_t st) const {
        return gemm_planes_wd(wg, dg, w.gscr[which_scratch], w.gscr_bytes, st);      // (split factor, one launch or two: gemm_planes.hip)
    }
    // zero padding of the activation planes is per batch size: re-zero when it changes (outside any capture)
    int prepare_planes(int B, hipStream_t st) {
        if (!use_planes || B == planes_B) return 0;
        if (hipMemsetAsync(w.planes_begin, 0, (size_t)(w.planes_end - w.planes_begin), st) != hipSuccess) {
            set_error("tf: zeroing operand planes failed");
            return SLNLP_ERR_LAUNCH;
        }
        planes_B = B;
        return 0;
    }
    int forward_impl(const int64_t* X, const int64_t* y, int B, int train, float* logp_out, hipStream_t st);
};

