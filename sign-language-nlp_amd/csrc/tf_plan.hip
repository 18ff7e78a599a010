// tf_plan.hip -- host-side plan for the whole model.Transformer step.
//
// Drop-in target: /root/reference/model/transformer.py:10-109 (constructor
// defines the parameter set, forward :60-90 defines the arithmetic) driven by
// the skorch step zero_grad -> forward -> CrossEntropyLoss(ignore_index) ->
// backward -> clip_grad_norm_ -> SGD.
//
// The plan owns (a) the layout of the flat fp32 parameter arena (names/shapes =
// reference state_dict), (b) the layout of the activation workspace and (c) the
// launch sequence.  Nothing here allocates device memory: the caller hands in
// arena / workspace pointers.  Every launch goes to the caller's stream, so the
// whole step can be captured into one hipGraph (slnlp_tf_graph_*).
#include <map>
#include <string>
#include <vector>

#include "common.hpp"

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
    SLNLP_CHECK_ARG(c->F > 0 && c->F % 4 == 0, "tf: hidden_size %d must be a multiple of 4", c->F);
    SLNLP_CHECK_ARG(c->N > 0 && c->Vs > 1 && c->Vt > 1, "tf: bad N/vocab");
    SLNLP_CHECK_ARG(c->B > 0 && c->B <= 1024, "tf: batch %d outside 1..1024", c->B);
    SLNLP_CHECK_ARG(c->S > 0 && c->S <= 64, "tf: seq_len %d outside 1..64 (single-tile attention)", c->S);
    SLNLP_CHECK_ARG(c->dropout >= 0.f && c->dropout < 1.f, "tf: dropout %f", c->dropout);
    SLNLP_CHECK_ARG(c->precision == 1 || c->precision == 3, "tf: precision %d", c->precision);
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
// written exactly once per step, so work forked to a side stream (wgrads) can keep reading it while
// the main stream moves on -- there is nothing to overwrite until the next step.
// bf16 hi/lo planes of a GEMM operand (same logical shape / row stride as its fp32 twin, rows
// zero-padded to a multiple of 64): written once by the producer, read by gemm_planes.hip
struct PP {
    unsigned short *hi = nullptr, *lo = nullptr;
    PlaneOut out() const { PlaneOut o; o.hi = hi; o.lo = lo; return o; }
};
struct EncA {
    float *qkv, *probs, *ctx, *y1, *st1, *x1, *h, *y2, *st2, *x2, *lnp1, *lnp2;
    float *gA2, *gB2, *gh, *gx1, *gA1, *gB1, *gctx, *gqkv, *gx0;
    PP ctxp, x1p, hp, x2p, d2p, ghp, d1p, gqkvp;
};
struct DecA {
    float *v, *y1, *st1, *t1, *q, *kv, *xprobs, *xctx, *y2, *st2, *t2, *h, *y3, *st3, *t3, *lnp1, *lnp2, *lnp3;
    float *gA3, *gB3, *gh, *gt2, *gA2, *gB2, *gxctx, *gq, *gkv, *gt1, *gA1, *gB1, *gv, *gt0;
    PP gkvp;
};
struct Ws {
    float *x0, *t0, *mem, *st_mem, *lnp_mem, *tfin, *st_fin, *lnp_fin, *logits, *dlogits, *logp, *row_nll;
    std::vector<EncA> enc;
    std::vector<DecA> dec;
    float *gfin, *gtl, *gmem, *gxl;     // d tfin, d t_last, d memory, d x_last
    void *emb_scratch_src, *emb_scratch_tgt;
    unsigned char* emb_keep;   // 4 keep bits per float4 of the source embedding's dropout (read by its backward)
    PP x0p, memp, wp;                   // wp: planes of the whole parameter arena (same offsets)
    char *planes_begin, *planes_end;    // activation planes region (re-zeroed when the batch size changes)
    float* opt_partials;
    char* gscr[2];          // split-K scratch of the grouped GEMM launches: [0] main stream, [1] side stream
    size_t gscr_bytes;
    slnlp_ln_reduce_entry* ln_table;
    size_t bytes;
};

constexpr int MAX_SPLITK = 8;
// weight-gradient GEMMs contract over the T tokens: aim at ~10 K-tiles (of 64) per workgroup
static int splitk_for(int T) {
    const int n = ((T + 63) / 64 + 5) / 10;
    return n < 1 ? 1 : n > MAX_SPLITK ? MAX_SPLITK : n;
}

static Ws carve(const slnlp_tf_config& c, void* base) {
    Ws w;
    Bump b(base);
    const size_t B = c.B, S = c.S, E = c.E, F = c.F, H = c.H, M = B * S, Vp = align_up(c.Vt, 4);
    const size_t lnp = (size_t)SLNLP_LN_MAX_PARTIALS * 2 * E;
    w.x0 = b.take<float>(M * E);
    w.t0 = b.take<float>(B * E);
    for (int i = 0; i < c.N; ++i) {
        EncA a;
        a.qkv = b.take<float>(M * 3 * E);
        a.probs = b.take<float>(B * H * S * S);
        a.ctx = b.take<float>(M * E);
        a.y1 = b.take<float>(M * E);
        a.st1 = b.take<float>(M * 2);
        a.x1 = b.take<float>(M * E);
        a.h = b.take<float>(M * F);
        a.y2 = b.take<float>(M * E);
        a.st2 = b.take<float>(M * 2);
        a.x2 = b.take<float>(M * E);
        a.lnp1 = b.take<float>(lnp);
        a.lnp2 = b.take<float>(lnp);
        a.gA2 = b.take<float>(M * E);
        a.gB2 = b.take<float>(M * E);
        a.gh = b.take<float>(M * F);
        a.gx1 = b.take<float>(M * E);
        a.gA1 = b.take<float>(M * E);
        a.gB1 = b.take<float>(M * E);
        a.gctx = b.take<float>(M * E);
        a.gqkv = b.take<float>(M * 3 * E);
        a.gx0 = b.take<float>(M * E);
        w.enc.push_back(a);
    }
    w.mem = b.take<float>(M * E);
    w.st_mem = b.take<float>(M * 2);
    w.lnp_mem = b.take<float>(lnp);
    for (int i = 0; i < c.N; ++i) {
        DecA a;
        a.v = b.take<float>(B * E);
        a.y1 = b.take<float>(B * E);
        a.st1 = b.take<float>(B * 2);
        a.t1 = b.take<float>(B * E);
        a.q = b.take<float>(B * E);
        a.kv = b.take<float>(M * 2 * E);
        a.xprobs = b.take<float>(B * H * S);
        a.xctx = b.take<float>(B * E);
        a.y2 = b.take<float>(B * E);
        a.st2 = b.take<float>(B * 2);
        a.t2 = b.take<float>(B * E);
        a.h = b.take<float>(B * F);
        a.y3 = b.take<float>(B * E);
        a.st3 = b.take<float>(B * 2);
        a.t3 = b.take<float>(B * E);
        a.lnp1 = b.take<float>(lnp);
        a.lnp2 = b.take<float>(lnp);
        a.lnp3 = b.take<float>(lnp);
        a.gA3 = b.take<float>(B * E);
        a.gB3 = b.take<float>(B * E);
        a.gh = b.take<float>(B * F);
        a.gt2 = b.take<float>(B * E);
        a.gA2 = b.take<float>(B * E);
        a.gB2 = b.take<float>(B * E);
        a.gxctx = b.take<float>(B * E);
        a.gq = b.take<float>(B * E);
        a.gkv = b.take<float>(M * 2 * E);
        a.gt1 = b.take<float>(B * E);
        a.gA1 = b.take<float>(B * E);
        a.gB1 = b.take<float>(B * E);
        a.gv = b.take<float>(B * E);
        a.gt0 = b.take<float>(B * E);
        w.dec.push_back(a);
    }
    w.tfin = b.take<float>(B * E);
    w.st_fin = b.take<float>(B * 2);
    w.lnp_fin = b.take<float>(lnp);
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
    w.ln_table = b.take<slnlp_ln_reduce_entry>(5 * c.N + 2);
    // ---- bf16 operand planes (only used when E and F are multiples of 64)
    const size_t Mp = (M + 63) / 64 * 64;
    auto pp = [&](size_t cols) { PP q; q.hi = b.take<unsigned short>(Mp * cols); q.lo = b.take<unsigned short>(Mp * cols); return q; };
    const size_t wtot = (size_t)build_layout(c).total + 64 * 3 * (E > F ? E : F);   // tail pad: tiles may over-read rows
    w.wp.hi = b.take<unsigned short>(wtot);
    w.wp.lo = b.take<unsigned short>(wtot);
    b.cur = (b.cur + 255) & ~(size_t)255;
    w.planes_begin = b.base + b.cur;
    w.x0p = pp(E);
    w.memp = pp(E);
    for (int i = 0; i < c.N; ++i) {
        EncA& a = w.enc[i];
        a.ctxp = pp(E); a.x1p = pp(E); a.hp = pp(F); a.x2p = pp(E);
        a.d2p = pp(E); a.ghp = pp(F); a.d1p = pp(E); a.gqkvp = pp(3 * E);
        w.dec[i].gkvp = pp(2 * E);
    }
    {   // grouped-launch scratch lives in the zero-on-demand region: its arrival counters must start at zero
        const size_t tx = ((E > F ? E : F) + 63) / 64, ty = ((3 * E > F ? 3 * E : F) + 63) / 64;
        w.gscr_bytes = 16384 + tx * ty * MAX_SPLITK * (512 * 8 * sizeof(float)) + ty * MAX_SPLITK * 64 * sizeof(float);
        w.gscr_bytes = (w.gscr_bytes + 255) & ~(size_t)255;
        for (int i = 0; i < 2; ++i) w.gscr[i] = b.take<char>(w.gscr_bytes);
    }
    b.cur = (b.cur + 255) & ~(size_t)255;
    w.planes_end = b.base + b.cur;
    w.bytes = (b.cur + 255) & ~(size_t)255;
    return w;
}

}  // namespace slnlp

using namespace slnlp;

// dropout site ids
enum { SITE_SRC_EMB = 1, SITE_TGT_EMB = 2, SITE_LAYER0 = 16, SITE_PER_LAYER = 8 };

constexpr int NSIDE = 2;

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
    int nbE = 0, nbD = 0;  // LN-backward block counts of the FULL batch (fixed: the reduce table is static)
    // Side streams: independent work (weight gradients, the memory K/V projections, embedding
    // gradients, the scalar loss) is forked off the dependent chain with events and joined before the
    // optimizer.  A single B=50 fit cannot fill 256 CUs with one kernel at a time; under stream
    // capture the forks become parallel branches of the hipGraph.
    hipStream_t side[NSIDE] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[NSIDE] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev_kv;
    bool side_dirty[NSIDE] = {false, false};
    bool use_planes = false;   // E, F multiples of 64: M = S*B GEMMs run on pre-split bf16 planes (gemm_planes.hip)
    int planes_B = -1;         // batch size the activation planes' zero padding is valid for

    float* P(long off) const { return buf.params + off; }
    float* G(long off) const { return buf.grads + off; }
    int enc_site(int l, int k) const { return SITE_LAYER0 + l * SITE_PER_LAYER + k; }
    int dec_site(int l, int k) const { return SITE_LAYER0 + (cfg.N + l) * SITE_PER_LAYER + k; }

    // side[k] may start once everything enqueued on `main` so far has finished
    int fork(hipStream_t main, int k) {
        if (hipEventRecord(ev_fork, main) != hipSuccess || hipStreamWaitEvent(side[k], ev_fork, 0) != hipSuccess) {
            set_error("tf: fork to side stream failed: %s", hipGetErrorString(hipGetLastError()));
            return SLNLP_ERR_LAUNCH;
        }
        side_dirty[k] = true;
        return 0;
    }
    // `main` waits for everything enqueued on side[k]
    int join(hipStream_t main, int k) {
        if (!side_dirty[k]) return 0;
        if (hipEventRecord(ev_join[k], side[k]) != hipSuccess || hipStreamWaitEvent(main, ev_join[k], 0) != hipSuccess) {
            set_error("tf: join of side stream failed: %s", hipGetErrorString(hipGetLastError()));
            return SLNLP_ERR_LAUNCH;
        }
        side_dirty[k] = false;
        return 0;
    }
    int join_all(hipStream_t main) {
        for (int k = 0; k < NSIDE; ++k) SLNLP_TRY(join(main, k));
        return 0;
    }

    int dec_self_block(int l, const float* t, int B, float p, hipStream_t st) const;

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
        a.precision = cfg.precision;
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
        a.precision = cfg.precision;
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
        a.precision = cfg.precision;
        return a;
    }
    int wgrad(const float* dy, long ldy, int T, int Nout, const float* x, int Kin, float* dW, float* db,
              hipStream_t st) const {
        return gemm(wgrad_args(dy, ldy, T, Nout, x, Kin, dW, db), st);
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
        a.A_hi = x.hi; a.A_lo = x.lo; a.lda_p = K; a.a_kmajor = 1;
        a.B_hi = w.wp.hi + woff; a.B_lo = w.wp.lo + woff; a.ldb_p = K; a.b_kmajor = 1;
        a.C = y; a.ldc = ldy; a.M = M; a.N = N; a.K = K;
        a.bias = bias; a.relu = relu;
        a.drop_p = p; a.drop_site = site; a.rng = buf.rng;
        a.resid = resid; a.ldr = ldy;
        if (outp) { a.C_hi = outp->hi; a.C_lo = outp->lo; a.ldc_p = N; }
        a.precision = cfg.precision;
        return gemm(a, st);
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
        a.precision = cfg.precision;
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
        a.precision = cfg.precision;
        return a;
    }
    int wgrad_p(const PP& dy, long ldy, int T, int Nout, const PP& x, int Kin, float* dW, float* db, hipStream_t st) const {
        return gemm(wgrad_p_args(dy, ldy, T, Nout, x, Kin, dW, db), st);
    }
    // weight gradient (split-K over the tokens) and data gradient of one dY in ONE launch: the wgrad's workgroups
    // fill the CUs the dgrad leaves idle, and there is no cross-queue edge to pay for (measured 4-10 us each)
    int wd_group(const slnlp_gemm_args& wg, const slnlp_gemm_args& dg, int which_scratch, hipStream_t st) const {
        const slnlp_gemm_args jobs[2] = {wg, dg};
        // Split factor of the weight gradient (K loop = tokens): the plane GEMM keeps 2 workgroups per CU resident
        // (512 slots) and a K-step costs about the same in every workgroup, so estimate
        //   time ~ rounds(total workgroups / 512) x longest K loop   (+1 step for the split-K meeting)
        // and take the best split; more workgroups than slots only adds a second, mostly empty round.
        auto cd = [](int a, int b) { return (a + b - 1) / b; };
        const int tw = cd(wg.M, 64) * cd(wg.N, 64), td = cd(dg.M, 64) * cd(dg.N, 64), kw = cd(wg.K, 64), kd = cd(dg.K, 64);
        int best = 1, best_cost = 1 << 30;
        for (int n = 1; n <= MAX_SPLITK && n <= kw; ++n) {
            const int len = cd(kw, n) + (n > 1 ? 1 : 0);
            const int cost = cd(tw * n + td, 512) * (len > kd ? len : kd);
            if (cost < best_cost) { best_cost = cost; best = n; }
        }
        const int split[2] = {best, 1};
        return gemm_planes_group(jobs, split, 2, w.gscr[which_scratch], w.gscr_bytes, st);
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
    int forward_impl(const int64_t* X, const int64_t* y, int B, int train, float* logp_out, hipStream_t st,
                     bool defer_join);
};

extern "C" {

int slnlp_tf_num_params(const slnlp_tf_config* cfg) {
    if (check_cfg(cfg)) return -1;
    return (int)build_layout(*cfg).ents.size();
}

int slnlp_tf_param_info(const slnlp_tf_config* cfg, int i, char* name, int64_t shape[2], int* ndim, int64_t* offset) {
    SLNLP_TRY(check_cfg(cfg));
    Layout L = build_layout(*cfg);
    SLNLP_CHECK_ARG(i >= 0 && i < (int)L.ents.size(), "tf_param_info: index %d out of range", i);
    const ParamEnt& e = L.ents[i];
    if (name) {
        strncpy(name, e.name.c_str(), 127);
        name[127] = 0;
    }
    if (shape) {
        shape[0] = e.shape[0];
        shape[1] = e.shape[1];
    }
    if (ndim) *ndim = e.ndim;
    if (offset) *offset = e.off;
    return 0;
}

int64_t slnlp_tf_arena_floats(const slnlp_tf_config* cfg) {
    if (check_cfg(cfg)) return -1;
    return build_layout(*cfg).total;
}

int64_t slnlp_tf_workspace_bytes(const slnlp_tf_config* cfg) {
    if (check_cfg(cfg)) return -1;
    return (int64_t)carve(*cfg, nullptr).bytes;
}

void slnlp_tf_destroy(slnlp_tf_plan* plan) {
    if (!plan) return;
    (void)hipDeviceSynchronize();             // nothing of this plan may still be in flight
    for (auto& kv : plan->graphs) (void)hipGraphExecDestroy(kv.second);
    for (int k = 0; k < NSIDE; ++k) {
        if (plan->side[k]) (void)hipStreamDestroy(plan->side[k]);
        if (plan->ev_join[k]) (void)hipEventDestroy(plan->ev_join[k]);
    }
    if (plan->ev_fork) (void)hipEventDestroy(plan->ev_fork);
    for (hipEvent_t e : plan->ev_kv) (void)hipEventDestroy(e);
    delete plan;
}

int slnlp_tf_create(const slnlp_tf_config* cfg, const slnlp_tf_buffers* buf, slnlp_tf_plan** out) {
    SLNLP_TRY(check_cfg(cfg));
    SLNLP_CHECK_ARG(buf && out, "tf_create: null argument");
    SLNLP_CHECK_ARG(buf->params && buf->grads && buf->momentum && buf->pe && buf->workspace && buf->rng && buf->lr &&
                        buf->scalars,
                    "tf_create: every buffer pointer is required");
    SLNLP_CHECK_ARG((((uintptr_t)buf->params | (uintptr_t)buf->grads | (uintptr_t)buf->momentum |
                      (uintptr_t)buf->workspace | (uintptr_t)buf->pe) & 255) == 0,
                    "tf_create: arenas / workspace / pe must be 256-byte aligned");
    slnlp_tf_plan* p = new slnlp_tf_plan();
    p->cfg = *cfg;
    p->buf = *buf;
    p->L = build_layout(*cfg);
    p->w = carve(*cfg, buf->workspace);
    bool ok = attn_init() == 0 && gemm_planes_init() == 0;
    p->use_planes = (cfg->E % 64 == 0) && (cfg->F % 64 == 0);
    for (int k = 0; ok && k < NSIDE; ++k)
        ok = hipStreamCreateWithFlags(&p->side[k], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&p->ev_join[k], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming) == hipSuccess;
    for (int l = 0; ok && l < cfg->N; ++l) {
        hipEvent_t e = nullptr;
        ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (ok) p->ev_kv.push_back(e);
    }
    // LN (dgamma, dbeta) reduction table: one entry per LayerNorm, uploaded once.  nblk is the FULL
    // batch's block count; smaller batches launch the same number of LN-backward blocks (nblk_force),
    // so every slot the reduce reads is rewritten each step.
    std::vector<slnlp_ln_reduce_entry> tab;
    const int nbE = p->nbE = ln_bwd_blocks(cfg->B * cfg->S), nbD = p->nbD = ln_bwd_blocks(cfg->B);
    auto ent = [&](const float* part, long gw, long gb, int nblk) {
        slnlp_ln_reduce_entry e;
        e.partial = part; e.dgamma = p->G(gw); e.dbeta = p->G(gb); e.nblk = nblk; e.E = cfg->E;
        tab.push_back(e);
    };
    for (int i = 0; i < cfg->N; ++i) {
        ent(p->w.enc[i].lnp1, p->L.enc[i].n1_w, p->L.enc[i].n1_b, nbE);
        ent(p->w.enc[i].lnp2, p->L.enc[i].n2_w, p->L.enc[i].n2_b, nbE);
    }
    ent(p->w.lnp_mem, p->L.encn_w, p->L.encn_b, nbE);
    for (int i = 0; i < cfg->N; ++i) {
        ent(p->w.dec[i].lnp1, p->L.dec[i].n1_w, p->L.dec[i].n1_b, nbD);
        ent(p->w.dec[i].lnp2, p->L.dec[i].n2_w, p->L.dec[i].n2_b, nbD);
        ent(p->w.dec[i].lnp3, p->L.dec[i].n3_w, p->L.dec[i].n3_b, nbD);
    }
    ent(p->w.lnp_fin, p->L.decn_w, p->L.decn_b, nbD);
    ok = ok && hipMemcpy(p->w.ln_table, tab.data(), tab.size() * sizeof(tab[0]), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemset(buf->grads, 0, p->L.total * sizeof(float)) == hipSuccess;
    if (!ok) {
        set_error("tf_create: device initialisation failed: %s", hipGetErrorString(hipGetLastError()));
        slnlp_tf_destroy(p);
        return SLNLP_ERR_LAUNCH;
    }
    *out = p;
    return 0;
}

}  // extern "C"

// Decoder layer l up to its cross-attention query: self-attention over ONE key (softmax == 1 -> out_proj(v_proj(t));
// the q/k rows of in_proj are dead; in train mode the weight-1 "attention" is still dropped per (row, head) -- fused into
// the V projection), residual + norm1, then q = in_proj_q(t1) (transformer.py:82-87).
int slnlp_tf_plan::dec_self_block(int l, const float* t, int B, float p, hipStream_t st) const {
    const slnlp_tf_plan* pl = this;
    const int E = cfg.E, dh = E / cfg.H;
    const unsigned long long* rng = buf.rng;
    (void)rng;
    const DecP& q = L.dec[l];
    const DecA& a = w.dec[l];
    SLNLP_TRY(pl->linear(t, B, E, pl->P(q.sin_w) + 2L * E * E, E, pl->P(q.sin_b) + 2 * E, a.v, E, 0, p, pl->dec_site(l, 0), nullptr, st, dh));
    SLNLP_TRY(pl->linear(a.v, B, E, pl->P(q.sout_w), E, pl->P(q.sout_b), a.y1, E, 0, p, pl->dec_site(l, 1), t, st));
    SLNLP_TRY(layernorm_fwd(a.y1, pl->P(q.n1_w), pl->P(q.n1_b), B, E, 1e-5f, a.t1, a.st1, st));
    SLNLP_TRY(pl->linear(a.t1, B, E, pl->P(q.cin_w), E, pl->P(q.cin_b), a.q, E, 0, 0.f, 0, nullptr, st));
    return 0;
}

int slnlp_tf_plan::forward_impl(const int64_t* X, const int64_t* y, int B, int train, float* logp_out, hipStream_t st,
                                bool defer_join) {
    slnlp_tf_plan* pl = this;
    const slnlp_tf_config& c = pl->cfg;
    const int E = c.E, F = c.F, H = c.H, S = c.S, dh = E / H, M = S * B, Vp = (int)align_up(c.Vt, 4);
    const float p = train ? c.dropout : 0.f;
    const unsigned long long* rng = pl->buf.rng;
    pl->last_B = B; pl->last_p = p; pl->last_X = X; pl->last_y = y;

    // The target side up to the first cross-attention (embedding, layer 0's single-key self-attention block and its
    // query projection: five B-row launches) depends on nothing the encoder computes: it runs on side[1] next to the
    // encoder and is joined right before layer 0's cross-attention.
    SLNLP_TRY(fork(st, 1));
    SLNLP_TRY(embed_fwd(y, 1, B, 1, E, c.Vt, pl->P(L.tgt_emb), pl->buf.pe, w.t0, sqrtf((float)E), p, SITE_TGT_EMB, rng, c.pad_tgt, side[1]));
    SLNLP_TRY(dec_self_block(0, w.t0, B, p, side[1]));
    const bool up = use_planes;
    if (up) {   // weights as bf16 planes, once per forward (they changed in the optimizer step / load_state_dict)
        SLNLP_TRY(prepare_planes(B, st));
        SLNLP_TRY(split_planes(pl->buf.params, L.total, 1, (int)L.total, w.wp.hi, w.wp.lo, L.total, st));
    }
    SLNLP_TRY(embed_fwd(X, S, B, S, E, c.Vs, pl->P(L.src_emb), pl->buf.pe, w.x0, sqrtf((float)E), p, SITE_SRC_EMB, rng, -1, st,
                        up ? w.x0p.out() : PlaneOut{}, w.emb_keep));

    const float* x = w.x0;
    const PP* xp = &w.x0p;
    for (int l = 0; l < c.N; ++l) {
        const EncP& q = L.enc[l];
        const EncA& a = w.enc[l];
        if (up) {
            SLNLP_TRY(pl->linear_p(*xp, M, E, q.in_w, 3 * E, pl->P(q.in_b), a.qkv, 3 * E, 0, 0.f, 0, nullptr, nullptr, st));
            SLNLP_TRY(attn_self_fwd(a.qkv, X, S, c.pad_src, 1, B, S, H, dh, a.ctx, a.probs, p, pl->enc_site(l, 0), rng, st, a.ctxp.out()));
            SLNLP_TRY(pl->linear_p(a.ctxp, M, E, q.out_w, E, pl->P(q.out_b), a.y1, E, 0, p, pl->enc_site(l, 1), x, nullptr, st));
            SLNLP_TRY(layernorm_fwd(a.y1, pl->P(q.n1_w), pl->P(q.n1_b), M, E, 1e-5f, a.x1, a.st1, st, a.x1p.out()));
            SLNLP_TRY(pl->linear_p(a.x1p, M, E, q.l1_w, F, pl->P(q.l1_b), a.h, F, 1, p, pl->enc_site(l, 2), nullptr, &a.hp, st));
            SLNLP_TRY(pl->linear_p(a.hp, M, F, q.l2_w, E, pl->P(q.l2_b), a.y2, E, 0, p, pl->enc_site(l, 3), a.x1, nullptr, st));
            SLNLP_TRY(layernorm_fwd(a.y2, pl->P(q.n2_w), pl->P(q.n2_b), M, E, 1e-5f, a.x2, a.st2, st, a.x2p.out()));
        } else {
            SLNLP_TRY(pl->linear(x, M, E, pl->P(q.in_w), 3 * E, pl->P(q.in_b), a.qkv, 3 * E, 0, 0.f, 0, nullptr, st));
            SLNLP_TRY(attn_self_fwd(a.qkv, X, S, c.pad_src, 1, B, S, H, dh, a.ctx, a.probs, p, pl->enc_site(l, 0), rng, st));
            SLNLP_TRY(pl->linear(a.ctx, M, E, pl->P(q.out_w), E, pl->P(q.out_b), a.y1, E, 0, p, pl->enc_site(l, 1), x, st));
            SLNLP_TRY(layernorm_fwd(a.y1, pl->P(q.n1_w), pl->P(q.n1_b), M, E, 1e-5f, a.x1, a.st1, st));
            SLNLP_TRY(pl->linear(a.x1, M, E, pl->P(q.l1_w), F, pl->P(q.l1_b), a.h, F, 1, p, pl->enc_site(l, 2), nullptr, st));
            SLNLP_TRY(pl->linear(a.h, M, F, pl->P(q.l2_w), E, pl->P(q.l2_b), a.y2, E, 0, p, pl->enc_site(l, 3), a.x1, st));
            SLNLP_TRY(layernorm_fwd(a.y2, pl->P(q.n2_w), pl->P(q.n2_b), M, E, 1e-5f, a.x2, a.st2, st));
        }
        x = a.x2;
        xp = &a.x2p;
    }
    SLNLP_TRY(layernorm_fwd(x, pl->P(L.encn_w), pl->P(L.encn_b), M, E, 1e-5f, w.mem, w.st_mem, st, up ? w.memp.out() : PlaneOut{}));

    // memory K|V projections of ALL decoder layers depend only on `mem`: run them on side[0]
    // while the main stream walks the decoder's chain of small (B-row) kernels.
    SLNLP_TRY(fork(st, 0));
    for (int l = 0; l < c.N; ++l) {
        const DecP& q = L.dec[l];
        if (up)
            SLNLP_TRY(pl->linear_p(w.memp, M, E, q.cin_w + (long)E * E, 2 * E, pl->P(q.cin_b) + E, w.dec[l].kv, 2 * E, 0, 0.f, 0,
                                   nullptr, nullptr, side[0]));
        else
            SLNLP_TRY(pl->linear(w.mem, M, E, pl->P(q.cin_w) + (long)E * E, 2 * E, pl->P(q.cin_b) + E, w.dec[l].kv, 2 * E, 0,
                                 0.f, 0, nullptr, side[0]));
        if (hipEventRecord(ev_kv[l], side[0]) != hipSuccess) {
            set_error("tf_forward: event record failed");
            return SLNLP_ERR_LAUNCH;
        }
    }

    const float* t = w.t0;
    for (int l = 0; l < c.N; ++l) {
        const DecP& q = L.dec[l];
        const DecA& a = w.dec[l];
        if (l == 0) SLNLP_TRY(join(st, 1));                     // layer 0's block ran on side[1] (above)
        else SLNLP_TRY(dec_self_block(l, t, B, p, st));
        if (hipStreamWaitEvent(st, ev_kv[l], 0) != hipSuccess) {
            set_error("tf_forward: wait for K/V projection failed");
            return SLNLP_ERR_LAUNCH;
        }
        SLNLP_TRY(attn_cross_fwd(a.q, a.kv, 2 * E, B, S, H, dh, a.xctx, a.xprobs, p, pl->dec_site(l, 2), rng, st));
        SLNLP_TRY(pl->linear(a.xctx, B, E, pl->P(q.cout_w), E, pl->P(q.cout_b), a.y2, E, 0, p, pl->dec_site(l, 3), a.t1, st));
        SLNLP_TRY(layernorm_fwd(a.y2, pl->P(q.n2_w), pl->P(q.n2_b), B, E, 1e-5f, a.t2, a.st2, st));
        SLNLP_TRY(pl->linear(a.t2, B, E, pl->P(q.l1_w), F, pl->P(q.l1_b), a.h, F, 1, p, pl->dec_site(l, 4), nullptr, st));
        SLNLP_TRY(pl->linear(a.h, B, F, pl->P(q.l2_w), E, pl->P(q.l2_b), a.y3, E, 0, p, pl->dec_site(l, 5), a.t2, st));
        SLNLP_TRY(layernorm_fwd(a.y3, pl->P(q.n3_w), pl->P(q.n3_b), B, E, 1e-5f, a.t3, a.st3, st));
        t = a.t3;
    }
    side_dirty[0] = false;  // every ev_kv has been waited for: side[0] is joined
    SLNLP_TRY(layernorm_fwd(t, pl->P(L.decn_w), pl->P(L.decn_b), B, E, 1e-5f, w.tfin, w.st_fin, st));
    SLNLP_TRY(pl->linear(w.tfin, B, E, pl->P(L.lin_w), c.Vt, pl->P(L.lin_b), w.logits, Vp, 0, 0.f, 0, nullptr, st));
    // log_softmax (transformer.py:88-89) + the criterion skorch applies to it (helper.py:61-70)
    SLNLP_TRY(lsm_nll(w.logits, Vp, y, B, c.Vt, c.pad_tgt, w.logp, pl->buf.scalars, train ? w.dlogits : nullptr, Vp,
                      w.row_nll, st, nullptr));
    if (logp_out &&
        hipMemcpyAsync(logp_out, w.logp, (size_t)B * c.Vt * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        set_error("tf_forward: copy of log-probs failed");
        return SLNLP_ERR_LAUNCH;
    }
    (void)defer_join;
    return 0;
}

extern "C" {

int slnlp_tf_forward(slnlp_tf_plan* pl, const int64_t* X, const int64_t* y, int B, int train, float* logp_out,
                     void* stream) {
    SLNLP_CHECK_ARG(pl && X && y, "tf_forward: `X` and `y` are required parameters");  // transformer.py:61-62
    SLNLP_CHECK_ARG(B > 0 && B <= pl->cfg.B, "tf_forward: batch %d outside 1..%d", B, pl->cfg.B);
    return pl->forward_impl(X, y, B, train, logp_out, (hipStream_t)stream, false);
}

int slnlp_tf_seed_dlogp(slnlp_tf_plan* pl, const float* dlogp, void* stream) {
    SLNLP_CHECK_ARG(pl && dlogp && pl->last_B > 0, "tf_seed_dlogp: needs a prior forward");
    return lsm_bwd(pl->w.logp, dlogp, pl->last_B, pl->cfg.Vt, pl->w.dlogits, align_up(pl->cfg.Vt, 4), (hipStream_t)stream);
}

int slnlp_tf_backward(slnlp_tf_plan* pl, void* stream) {
    SLNLP_CHECK_ARG(pl && pl->last_B > 0, "tf_backward: needs a prior forward(train)");
    hipStream_t st = (hipStream_t)stream;
    hipStream_t s0 = pl->side[0], s1 = pl->side[1];
    const slnlp_tf_config& c = pl->cfg;
    const Ws& w = pl->w;
    const Layout& L = pl->L;
    const int B = pl->last_B, E = c.E, F = c.F, H = c.H, S = c.S, dh = E / H, M = S * B, Vp = (int)align_up(c.Vt, 4);
    const float p = pl->last_p, ik = 1.f / (1.f - p);
    const unsigned long long* rng = pl->buf.rng;
    const int64_t *X = pl->last_X, *y = pl->last_y;
    const bool up = pl->use_planes;
    int nb;
    // Everything runs on the main stream; the weight gradient of each dY shares a launch with its data gradient.
    // Only the (large, independent) d memory / K|V weight-gradient groups go to side[0] and the target-embedding
    // gradient to side[1]: graph replay places parallel branches on its own queues and every cross-queue edge
    // costs 4-10 us, so fine-grained forks were measured slower than no forks.

    // generator: logits = tfin lin_w^T + lin_b
    SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(w.dlogits, Vp, B, c.Vt, w.tfin, E, pl->G(L.lin_w), pl->G(L.lin_b)),
                             pl->dgrad_args(w.dlogits, Vp, B, c.Vt, pl->P(L.lin_w), E, w.gfin, nullptr, 0.f, nullptr), st));
    SLNLP_TRY(layernorm_bwd(w.gfin, w.dec[c.N - 1].t3, pl->P(L.decn_w), w.st_fin, B, E, nullptr, w.gtl, nullptr, 0.f, 0,
                            rng, w.lnp_fin, &nb, pl->nbD, st));
    const float* dt = w.gtl;  // gradient w.r.t. the current decoder layer's output
    for (int l = c.N - 1; l >= 0; --l) {
        const DecP& q = L.dec[l];
        const DecA& a = w.dec[l];
        const float* t_in = l > 0 ? w.dec[l - 1].t3 : w.t0;
        // norm3 / FFN
        SLNLP_TRY(layernorm_bwd(dt, a.y3, pl->P(q.n3_w), a.st3, B, E, nullptr, a.gA3, p > 0.f ? a.gB3 : nullptr, p,
                                pl->dec_site(l, 5), rng, a.lnp3, &nb, pl->nbD, st));
        const float* d3 = p > 0.f ? a.gB3 : a.gA3;
        SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(d3, E, B, E, a.h, F, pl->G(q.l2_w), pl->G(q.l2_b)),
                                 pl->dgrad_args(d3, E, B, E, pl->P(q.l2_w), F, a.gh, a.h, ik, nullptr), st));
        SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(a.gh, F, B, F, a.t2, E, pl->G(q.l1_w), pl->G(q.l1_b)),
                                 pl->dgrad_args(a.gh, F, B, F, pl->P(q.l1_w), E, a.gt2, nullptr, 0.f, a.gA3), st));
        // norm2 / cross-attention
        SLNLP_TRY(layernorm_bwd(a.gt2, a.y2, pl->P(q.n2_w), a.st2, B, E, nullptr, a.gA2, p > 0.f ? a.gB2 : nullptr, p,
                                pl->dec_site(l, 3), rng, a.lnp2, &nb, pl->nbD, st));
        const float* d2 = p > 0.f ? a.gB2 : a.gA2;
        SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(d2, E, B, E, a.xctx, E, pl->G(q.cout_w), pl->G(q.cout_b)),
                                 pl->dgrad_args(d2, E, B, E, pl->P(q.cout_w), E, a.gxctx, nullptr, 0.f, nullptr), st));
        SLNLP_TRY(attn_cross_bwd(a.q, a.kv, 2 * E, a.xprobs, a.gxctx, B, S, H, dh, a.gq, a.gkv, 2 * E, p, pl->dec_site(l, 2), rng, st,
                                 up ? a.gkvp.out() : PlaneOut{}));
        // d memory accumulates over the decoder layers in a fixed order on side[0], next to its weight gradient
        SLNLP_TRY(pl->fork(st, 0));
        if (up) {
            SLNLP_TRY(pl->wd_group(pl->wgrad_p_args(a.gkvp, 2 * E, M, 2 * E, w.memp, E, pl->G(q.cin_w) + (long)E * E, pl->G(q.cin_b) + E),
                                   pl->dgrad_p_args(a.gkvp, 2 * E, M, 2 * E, q.cin_w + (long)E * E, E, w.gmem, nullptr, 0.f,
                                                    l == c.N - 1 ? nullptr : w.gmem, nullptr), 1, s0));
        } else {
            SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(a.gkv, 2 * E, M, 2 * E, w.mem, E, pl->G(q.cin_w) + (long)E * E, pl->G(q.cin_b) + E),
                                     pl->dgrad_args(a.gkv, 2 * E, M, 2 * E, pl->P(q.cin_w) + (long)E * E, E, w.gmem, nullptr, 0.f,
                                                    l == c.N - 1 ? nullptr : w.gmem), s0));
        }
        SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(a.gq, E, B, E, a.t1, E, pl->G(q.cin_w), pl->G(q.cin_b)),
                                 pl->dgrad_args(a.gq, E, B, E, pl->P(q.cin_w), E, a.gt1, nullptr, 0.f, a.gA2), st));
        // norm1 / self-attention (single key).  For layer 0 nothing downstream on the main stream needs these (they end
        // in weight gradients and the target-embedding gradient): they go to side[1] while the encoder backward starts.
        hipStream_t sb = st;
        if (l == 0) {
            SLNLP_TRY(pl->fork(st, 1));
            sb = s1;
        }
        SLNLP_TRY(layernorm_bwd(a.gt1, a.y1, pl->P(q.n1_w), a.st1, B, E, nullptr, a.gA1, p > 0.f ? a.gB1 : nullptr, p,
                                pl->dec_site(l, 1), rng, a.lnp1, &nb, pl->nbD, sb));
        const float* d1 = p > 0.f ? a.gB1 : a.gA1;
        {
            slnlp_gemm_args dg = pl->dgrad_args(d1, E, B, E, pl->P(q.sout_w), E, a.gv, nullptr, 0.f, nullptr);
            if (p > 0.f) { dg.drop_p = p; dg.drop_site = pl->dec_site(l, 0); dg.rng = rng; dg.drop_head_dim = dh; }   // same mask as forward
            SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(d1, E, B, E, a.v, E, pl->G(q.sout_w), pl->G(q.sout_b)), dg, sb));
        }
        // softmax over one element has zero gradient: the q/k rows of in_proj (weight and bias) get exactly 0.
        // Nothing ever writes them, and the gradient arena is zeroed at plan creation, so they stay zero.
        SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(a.gv, E, B, E, t_in, E, pl->G(q.sin_w) + 2L * E * E, pl->G(q.sin_b) + 2 * E),
                                 pl->dgrad_args(a.gv, E, B, E, pl->P(q.sin_w) + 2L * E * E, E, a.gt0, nullptr, 0.f, a.gA1), sb));
        dt = a.gt0;
    }
    SLNLP_TRY(embed_bwd(y, 1, B, 1, E, c.Vt, dt, pl->G(L.tgt_emb), sqrtf((float)E), -1, p, SITE_TGT_EMB, rng, w.emb_scratch_tgt, s1));

    // encoder: needs the complete d memory
    SLNLP_TRY(pl->join(st, 0));
    SLNLP_TRY(layernorm_bwd(w.gmem, w.enc[c.N - 1].x2, pl->P(L.encn_w), w.st_mem, M, E, nullptr, w.gxl, nullptr, 0.f, 0,
                            rng, w.lnp_mem, &nb, pl->nbE, st));
    const float* dx = w.gxl;
    for (int l = c.N - 1; l >= 0; --l) {
        const EncP& q = L.enc[l];
        const EncA& a = w.enc[l];
        const float* x_in = l > 0 ? w.enc[l - 1].x2 : w.x0;
        const PP& xp_in = l > 0 ? w.enc[l - 1].x2p : w.x0p;
        const PlaneOut none{};
        // LayerNorm backward also emits the bf16 planes of the gradient that feeds the sub-layer's GEMMs
        // (the dropout-masked copy when dropout is on, else dx itself)
        SLNLP_TRY(layernorm_bwd(dx, a.y2, pl->P(q.n2_w), a.st2, M, E, nullptr, a.gA2, p > 0.f ? a.gB2 : nullptr, p,
                                pl->enc_site(l, 3), rng, a.lnp2, &nb, pl->nbE, st, (up && p == 0.f) ? a.d2p.out() : none,
                                (up && p > 0.f) ? a.d2p.out() : none));
        const float* d2 = p > 0.f ? a.gB2 : a.gA2;
        if (up) {
            SLNLP_TRY(pl->wd_group(pl->wgrad_p_args(a.d2p, E, M, E, a.hp, F, pl->G(q.l2_w), pl->G(q.l2_b)),
                                   pl->dgrad_p_args(a.d2p, E, M, E, q.l2_w, F, a.gh, a.h, ik, nullptr, &a.ghp), 0, st));
            SLNLP_TRY(pl->wd_group(pl->wgrad_p_args(a.ghp, F, M, F, a.x1p, E, pl->G(q.l1_w), pl->G(q.l1_b)),
                                   pl->dgrad_p_args(a.ghp, F, M, F, q.l1_w, E, a.gx1, nullptr, 0.f, a.gA2, nullptr), 0, st));
        } else {
            SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(d2, E, M, E, a.h, F, pl->G(q.l2_w), pl->G(q.l2_b)),
                                     pl->dgrad_args(d2, E, M, E, pl->P(q.l2_w), F, a.gh, a.h, ik, nullptr), st));
            SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(a.gh, F, M, F, a.x1, E, pl->G(q.l1_w), pl->G(q.l1_b)),
                                     pl->dgrad_args(a.gh, F, M, F, pl->P(q.l1_w), E, a.gx1, nullptr, 0.f, a.gA2), st));
        }
        SLNLP_TRY(layernorm_bwd(a.gx1, a.y1, pl->P(q.n1_w), a.st1, M, E, nullptr, a.gA1, p > 0.f ? a.gB1 : nullptr, p,
                                pl->enc_site(l, 1), rng, a.lnp1, &nb, pl->nbE, st, (up && p == 0.f) ? a.d1p.out() : none,
                                (up && p > 0.f) ? a.d1p.out() : none));
        const float* d1 = p > 0.f ? a.gB1 : a.gA1;
        if (up) {
            SLNLP_TRY(pl->wd_group(pl->wgrad_p_args(a.d1p, E, M, E, a.ctxp, E, pl->G(q.out_w), pl->G(q.out_b)),
                                   pl->dgrad_p_args(a.d1p, E, M, E, q.out_w, E, a.gctx, nullptr, 0.f, nullptr, nullptr), 0, st));
            SLNLP_TRY(attn_self_bwd(a.qkv, a.probs, a.gctx, B, S, H, dh, a.gqkv, p, pl->enc_site(l, 0), rng, st, a.gqkvp.out()));
            SLNLP_TRY(pl->wd_group(pl->wgrad_p_args(a.gqkvp, 3 * E, M, 3 * E, xp_in, E, pl->G(q.in_w), pl->G(q.in_b)),
                                   pl->dgrad_p_args(a.gqkvp, 3 * E, M, 3 * E, q.in_w, E, a.gx0, nullptr, 0.f, a.gA1, nullptr), 0, st));
        } else {
            SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(d1, E, M, E, a.ctx, E, pl->G(q.out_w), pl->G(q.out_b)),
                                     pl->dgrad_args(d1, E, M, E, pl->P(q.out_w), E, a.gctx, nullptr, 0.f, nullptr), st));
            SLNLP_TRY(attn_self_bwd(a.qkv, a.probs, a.gctx, B, S, H, dh, a.gqkv, p, pl->enc_site(l, 0), rng, st));
            SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(a.gqkv, 3 * E, M, 3 * E, x_in, E, pl->G(q.in_w), pl->G(q.in_b)),
                                     pl->dgrad_args(a.gqkv, 3 * E, M, 3 * E, pl->P(q.in_w), E, a.gx0, nullptr, 0.f, a.gA1), st));
        }
        dx = a.gx0;
    }
    SLNLP_TRY(embed_bwd(X, S, B, S, E, c.Vs, dx, pl->G(L.src_emb), sqrtf((float)E), -1, p, SITE_SRC_EMB, rng, w.emb_scratch_src, st, w.emb_keep));
    SLNLP_TRY(pl->join_all(st));
    SLNLP_TRY(ln_param_reduce(w.ln_table, 5 * c.N + 2, E, st));
    return 0;
}

int slnlp_tf_optim(slnlp_tf_plan* pl, float momentum, float max_norm, void* stream) {
    SLNLP_CHECK_ARG(pl, "tf_optim: null plan");
    return clip_sgd_step(pl->buf.params, pl->buf.grads, pl->buf.momentum, pl->L.total, pl->buf.lr, momentum, max_norm,
                         pl->w.opt_partials, pl->buf.scalars + 1, pl->buf.rng, (hipStream_t)stream);
}

int slnlp_tf_train_step(slnlp_tf_plan* pl, const int64_t* X, const int64_t* y, int B, float momentum, float max_norm,
                        float* logp, void* stream) {
    SLNLP_TRY(slnlp_tf_forward(pl, X, y, B, 1, logp, stream));
    SLNLP_TRY(slnlp_tf_backward(pl, stream));
    return slnlp_tf_optim(pl, momentum, max_norm, stream);
}

// Capture one train step (fixed X / y / logp device buffers and batch size) into
// a hipGraph and keep the executable graph in the plan; replay with
// slnlp_tf_graph_launch.  lr, rng step and the data are read from device memory,
// so the same graph serves every step of a fit.
int slnlp_tf_graph_capture_train(slnlp_tf_plan* pl, const int64_t* X, const int64_t* y, int B, float momentum,
                                 float max_norm, float* logp, void* stream) {
    SLNLP_CHECK_ARG(pl && stream, "tf_graph_capture_train: needs a plan and a non-default stream");
    hipStream_t st = (hipStream_t)stream;
    SLNLP_TRY(pl->prepare_planes(B, st));   // must not be captured: it runs once per batch-size change
    auto old = pl->graphs.find(B);
    if (old != pl->graphs.end()) {          // re-capture for this batch size: the old exec may still be running
        (void)hipStreamSynchronize(st);
        (void)hipGraphExecDestroy(old->second);
        pl->graphs.erase(old);
    }
    if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        set_error("tf_graph_capture_train: begin capture failed: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    int rc = slnlp_tf_train_step(pl, X, y, B, momentum, max_norm, logp, stream);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != 0) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess || !g) {
        set_error("tf_graph_capture_train: end capture failed: %s", hipGetErrorString(e));
        return SLNLP_ERR_LAUNCH;
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        set_error("tf_graph_capture_train: instantiate failed: %s", hipGetErrorString(e));
        return SLNLP_ERR_LAUNCH;
    }
    pl->graphs[B] = exec;
    return 0;
}

int slnlp_tf_graph_launch(slnlp_tf_plan* pl, int B, void* stream) {
    SLNLP_CHECK_ARG(pl, "tf_graph_launch: null plan");
    auto it = pl->graphs.find(B);
    SLNLP_CHECK_ARG(it != pl->graphs.end(), "tf_graph_launch: no captured graph for batch %d", B);
    SLNLP_TRY(pl->prepare_planes(B, (hipStream_t)stream));
    if (hipGraphLaunch(it->second, (hipStream_t)stream) != hipSuccess) {
        set_error("tf_graph_launch: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    return 0;
}

int slnlp_tf_tap(slnlp_tf_plan* pl, const char* name, float* out, int64_t max_floats, int64_t* n_out, void* stream) {
    SLNLP_CHECK_ARG(pl && name && out && pl->last_B > 0, "tf_tap: bad args / no forward yet");
    const slnlp_tf_config& c = pl->cfg;
    const int B = pl->last_B, M = B * c.S, E = c.E, Vp = (int)align_up(c.Vt, 4);
    const std::string n(name);
    const float* src = nullptr;
    int64_t rows = 0, cols = E, ld = E;
    if (n == "src_embed") { src = pl->w.x0; rows = M; }
    else if (n == "tgt_embed") { src = pl->w.t0; rows = B; }
    else if (n == "memory") { src = pl->w.mem; rows = M; }
    else if (n == "logits") { src = pl->w.logits; rows = B; cols = c.Vt; ld = Vp; }
    else if (n == "dlogits") { src = pl->w.dlogits; rows = B; cols = c.Vt; ld = Vp; }
    else if (n.rfind("enc", 0) == 0) { int l = atoi(name + 3); SLNLP_CHECK_ARG(l >= 0 && l < c.N, "tf_tap: %s", name); src = pl->w.enc[l].x2; rows = M; }
    else if (n.rfind("dec", 0) == 0) { int l = atoi(name + 3); SLNLP_CHECK_ARG(l >= 0 && l < c.N, "tf_tap: %s", name); src = pl->w.dec[l].t3; rows = B; }
    SLNLP_CHECK_ARG(src, "tf_tap: unknown tap '%s'", name);
    SLNLP_CHECK_ARG(rows * cols <= max_floats, "tf_tap: buffer too small (%ld needed)", (long)(rows * cols));
    if (hipMemcpy2DAsync(out, cols * sizeof(float), src, ld * sizeof(float), cols * sizeof(float), rows,
                         hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
        set_error("tf_tap: copy failed");
        return SLNLP_ERR_LAUNCH;
    }
    if (n_out) *n_out = rows * cols;
    return 0;
}

}  // extern "C"
