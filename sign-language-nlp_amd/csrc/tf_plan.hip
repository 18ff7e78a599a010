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
#include "launch.hpp"
#include "tf_plan.hpp"

using namespace slnlp;

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
    if (!plan->graphs.empty()) (void)hipDeviceSynchronize();   // graph execs are torn down below
    else destroy_sync(plan->destroy_sync);    // nothing of this plan may still be in flight when its buffers go
    for (auto& kv : plan->graphs) (void)hipGraphExecDestroy(kv.second);
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
    p->wgrad_np = wgrad_passes();
    p->dgrad_np = dgrad_passes();
    {   // SLNLP_DEC_ROWS=0: the decoder's products on gemm.hip's fp32-operand kernel again (A / B measurements; another arithmetic:
        // that kernel splits its operands itself, with a truncated head)
        const char* e = getenv("SLNLP_DEC_ROWS");
        p->use_rows = p->use_planes && cfg->E <= 1024 && cfg->F <= 1024 && !(e && atoi(e) == 0);
    }
    if (ok && cfg->precision == 8) {     // rows the fp8 forward products read: one {offset, K} entry each, uploaded once
        std::vector<QuantRow> rows;
        auto block = [&](long off, int nrows, int K) {
            p->qrow0[off] = (long)rows.size();
            for (int r = 0; r < nrows; ++r) rows.push_back(QuantRow{off + (long)r * K, K, 0});
        };
        const int E = cfg->E, F = cfg->F;
        for (int i = 0; i < cfg->N; ++i) {
            const EncP& q = p->L.enc[i];
            block(q.in_w, 3 * E, E); block(q.out_w, E, E); block(q.l1_w, F, E); block(q.l2_w, E, F);
        }
        ok = (long)rows.size() == p->w.n_qrows &&
             hipMemcpy(p->w.qrow_table, rows.data(), rows.size() * sizeof(QuantRow), hipMemcpyHostToDevice) == hipSuccess;
    }
    // LN (dgamma, dbeta) tables, one entry per LayerNorm, uploaded once: what ln_param_partial reads (the dy / x / stats of the
    // LayerNorm's backward, all still intact at the end of backward: every gradient buffer is written once per step) and what
    // ln_param_reduce adds.  nblk is the FULL batch's chunk count; a smaller batch's trailing chunks are written as zeros.
    std::vector<slnlp_ln_reduce_entry> tab;
    std::vector<LnPartialEntry> ptab;
    const int nbE = p->nbE = ln_bwd_blocks(cfg->B * cfg->S), nbD = p->nbD = ln_bwd_blocks(cfg->B);
    auto ent = [&](float* part, long gw, long gb, int dec, const float* dy, const float* x, const float* stats) {
        slnlp_ln_reduce_entry e;
        e.partial = part; e.dgamma = p->G(gw); e.dbeta = p->G(gb); e.nblk = dec ? nbD : nbE; e.E = cfg->E;
        tab.push_back(e);
        LnPartialEntry q;
        q.dy = dy; q.x = x; q.stats = stats; q.partial = part; q.dec = dec; q.pad = 0;
        ptab.push_back(q);
    };
    {
        const Ws& w = p->w;
        const int N = cfg->N;
        for (int i = 0; i < N; ++i) {
            const EncA& a = w.enc[i];
            ent(a.lnp1, p->L.enc[i].n1_w, p->L.enc[i].n1_b, 0, a.gx1, a.y1, a.st1);
            ent(a.lnp2, p->L.enc[i].n2_w, p->L.enc[i].n2_b, 0, i + 1 < N ? w.enc[i + 1].gx0 : w.gxl, a.y2, a.st2);
        }
        ent(w.lnp_mem, p->L.encn_w, p->L.encn_b, 0, w.gmem, w.enc[N - 1].x2, w.st_mem);
        for (int i = 0; i < N; ++i) {
            const DecA& a = w.dec[i];
            ent(a.lnp1, p->L.dec[i].n1_w, p->L.dec[i].n1_b, 1, a.gt1, a.y1, a.st1);
            ent(a.lnp2, p->L.dec[i].n2_w, p->L.dec[i].n2_b, 1, a.gt2, a.y2, a.st2);
            ent(a.lnp3, p->L.dec[i].n3_w, p->L.dec[i].n3_b, 1, i + 1 < N ? w.dec[i + 1].gt0 : w.gtl, a.y3, a.st3);
        }
        ent(w.lnp_fin, p->L.decn_w, p->L.decn_b, 1, w.gfin, w.dec[N - 1].t3, w.st_fin);
    }
    ok = ok && hipMemcpy(p->w.ln_table, tab.data(), tab.size() * sizeof(tab[0]), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->w.ln_ptable, ptab.data(), ptab.size() * sizeof(ptab[0]), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemset(buf->grads, 0, p->L.total * sizeof(float)) == hipSuccess &&
         // the memset runs on the null stream, which does NOT order itself against the caller's non-blocking stream: without
         // this wait it can land after the first backward has written gradients (seen with several host threads, whose
         // initialisation kernels queue up on the null stream: grid scores differed from run to run)
         hipStreamSynchronize(nullptr) == hipSuccess;
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
int slnlp_tf_plan::dec_self_block(int l, const float* t, const PP* tp, int B, float p, hipStream_t st) const {
    const slnlp_tf_plan* pl = this;
    const int E = cfg.E, dh = E / cfg.H;
    const DecP& q = L.dec[l];
    const DecA& a = w.dec[l];
    if (use_rows) {        // B-row products on planes (gemm_rows.hip): every producer also emits its output as the next product's operand
        SLNLP_TRY(pl->linear_r(*tp, B, E, q.sin_w + 2L * E * E, E, pl->P(q.sin_b) + 2 * E, a.v, E, 0, p, pl->dec_site(l, 0), nullptr, &a.vp, st, dh));
        SLNLP_TRY(pl->linear_r(a.vp, B, E, q.sout_w, E, pl->P(q.sout_b), a.y1, E, 0, p, pl->dec_site(l, 1), t, nullptr, st));
        SLNLP_TRY(layernorm_fwd(a.y1, pl->P(q.n1_w), pl->P(q.n1_b), B, E, 1e-5f, a.t1, a.st1, st, a.t1p.out()));
        SLNLP_TRY(pl->linear_r(a.t1p, B, E, q.cin_w, E, pl->P(q.cin_b), a.q, E, 0, 0.f, 0, nullptr, nullptr, st));
        return 0;
    }
    SLNLP_TRY(pl->linear(t, B, E, pl->P(q.sin_w) + 2L * E * E, E, pl->P(q.sin_b) + 2 * E, a.v, E, 0, p, pl->dec_site(l, 0), nullptr, st, dh));
    SLNLP_TRY(pl->linear(a.v, B, E, pl->P(q.sout_w), E, pl->P(q.sout_b), a.y1, E, 0, p, pl->dec_site(l, 1), t, st));
    SLNLP_TRY(layernorm_fwd(a.y1, pl->P(q.n1_w), pl->P(q.n1_b), B, E, 1e-5f, a.t1, a.st1, st));
    SLNLP_TRY(pl->linear(a.t1, B, E, pl->P(q.cin_w), E, pl->P(q.cin_b), a.q, E, 0, 0.f, 0, nullptr, st));
    return 0;
}

int slnlp_tf_plan::forward_impl(const int64_t* X, const int64_t* y, int B, int train, float* logp_out, hipStream_t st) {
    slnlp_tf_plan* pl = this;
    const slnlp_tf_config& c = pl->cfg;
    const int E = c.E, F = c.F, H = c.H, S = c.S, dh = E / H, M = S * B, Vp = (int)align_up(c.Vt, 4);
    const float p = train ? c.dropout : 0.f;
    const unsigned long long* rng = pl->buf.rng;
    pl->last_B = B; pl->last_p = p; pl->last_X = X; pl->last_y = y;

    // The target side up to the first cross-attention (embedding, layer 0's single-key self-attention block and its
    // query projection: five B-row launches) depends on nothing the encoder computes; everything runs on the caller's
    // ONE stream in program order (round 1 forked it to a side stream: -3 % and a data race).
    const bool up = use_planes;
    if (up) {   // weights as bf16 planes: current unless the arena changed outside the fused optimizer step
        SLNLP_TRY(prepare_planes(B, st));
        SLNLP_TRY(ensure_wplanes(st));
        SLNLP_TRY(ensure_wq(st));
    }
    SLNLP_TRY(embed_fwd(y, 1, B, 1, E, c.Vt, pl->P(L.tgt_emb), pl->buf.pe, w.t0, sqrtf((float)E), p, SITE_TGT_EMB, rng, c.pad_tgt, st,
                        use_rows ? w.t0p.out() : PlaneOut{}));
    SLNLP_TRY(dec_self_block(0, w.t0, &w.t0p, B, p, st));
    SLNLP_TRY(embed_fwd(X, S, B, S, E, c.Vs, pl->P(L.src_emb), pl->buf.pe, w.x0, sqrtf((float)E), p, SITE_SRC_EMB, rng, -1, st,
                        up ? w.x0p.out() : PlaneOut{}, w.emb_keep));

    const float* x = w.x0;
    const PP* xp = &w.x0p;
    for (int l = 0; l < c.N; ++l) {
        const EncP& q = L.enc[l];
        const EncA& a = w.enc[l];
        if (up) {
            SLNLP_TRY(pl->linear_p(*xp, M, E, q.in_w, 3 * E, pl->P(q.in_b), a.qkv, 3 * E, 0, 0.f, 0, nullptr, nullptr, st));
            // (ctx and, in backward, d qkv leave the attention kernels as planes only when the sequence fits the single-tile kernels:
            //  their fp32 copies have no reader in the plane path)
            SLNLP_TRY(attn_self_fwd(a.qkv, X, S, c.pad_src, 1, B, S, H, dh, S <= 64 ? nullptr : a.ctx, a.probs, p, pl->enc_site(l, 0), rng, st, a.ctxp.out()));
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

    const float* t = w.t0;
    const PP* tp = &w.t0p;
    for (int l = 0; l < c.N; ++l) {
        const DecP& q = L.dec[l];
        const DecA& a = w.dec[l];
        if (l > 0) SLNLP_TRY(dec_self_block(l, t, tp, B, p, st));   // (layer 0's block ran ahead of the encoder, above)
        // cross-attention over the memory itself: with ONE query per sequence the K / V projections of the S memory rows
        // re-associate into B-row products (attention_mem.hip) -- no [S*B, 2E] projection, no K|V gradient GEMMs:
        // qk = Wk_h^T q_h (batched GEMM) -> scores / softmax / dropout / mbar (+ ctx0 = bv sum_s p_s) -> ctx = Wv_h mbar + ctx0
        {
            const float *Wk = pl->P(q.cin_w) + (long)E * E, *Wv = pl->P(q.cin_w) + 2L * E * E;
            const slnlp_gemm_args j1 = pl->head_expand(a.q, Wk, a.qk, B, H, dh);
            SLNLP_TRY(gemm_group(&j1, 1, st));
            SLNLP_TRY(xmem_fwd(a.qk, w.mem, pl->P(q.cin_b) + 2 * E, B, S, H, dh, a.mbar, a.psum, a.xprobs, a.xctx, p, pl->dec_site(l, 2), rng, st));
            slnlp_gemm_args j2 = pl->head_reduce(a.mbar, Wv, a.xctx, a.xctx, B, H, dh);
            if (use_rows) { j2.C_hi = a.xctxp.hi; j2.C_lo = a.xctxp.lo; j2.ldc_p = E; }      // ... and as planes, for the out projection
            SLNLP_TRY(gemm_group(&j2, 1, st));
        }
        if (use_rows) {
            SLNLP_TRY(pl->linear_r(a.xctxp, B, E, q.cout_w, E, pl->P(q.cout_b), a.y2, E, 0, p, pl->dec_site(l, 3), a.t1, nullptr, st));
            SLNLP_TRY(layernorm_fwd(a.y2, pl->P(q.n2_w), pl->P(q.n2_b), B, E, 1e-5f, a.t2, a.st2, st, a.t2p.out()));
            SLNLP_TRY(pl->linear_r(a.t2p, B, E, q.l1_w, F, pl->P(q.l1_b), a.h, F, 1, p, pl->dec_site(l, 4), nullptr, &a.hp, st));
            SLNLP_TRY(pl->linear_r(a.hp, B, F, q.l2_w, E, pl->P(q.l2_b), a.y3, E, 0, p, pl->dec_site(l, 5), a.t2, nullptr, st));
            SLNLP_TRY(layernorm_fwd(a.y3, pl->P(q.n3_w), pl->P(q.n3_b), B, E, 1e-5f, a.t3, a.st3, st, a.t3p.out()));
        } else {
            SLNLP_TRY(pl->linear(a.xctx, B, E, pl->P(q.cout_w), E, pl->P(q.cout_b), a.y2, E, 0, p, pl->dec_site(l, 3), a.t1, st));
            SLNLP_TRY(layernorm_fwd(a.y2, pl->P(q.n2_w), pl->P(q.n2_b), B, E, 1e-5f, a.t2, a.st2, st));
            SLNLP_TRY(pl->linear(a.t2, B, E, pl->P(q.l1_w), F, pl->P(q.l1_b), a.h, F, 1, p, pl->dec_site(l, 4), nullptr, st));
            SLNLP_TRY(pl->linear(a.h, B, F, pl->P(q.l2_w), E, pl->P(q.l2_b), a.y3, E, 0, p, pl->dec_site(l, 5), a.t2, st));
            SLNLP_TRY(layernorm_fwd(a.y3, pl->P(q.n3_w), pl->P(q.n3_b), B, E, 1e-5f, a.t3, a.st3, st));
        }
        t = a.t3;
        tp = &a.t3p;
    }
    if (use_rows) {
        SLNLP_TRY(layernorm_fwd(t, pl->P(L.decn_w), pl->P(L.decn_b), B, E, 1e-5f, w.tfin, w.st_fin, st, w.tfinp.out()));
        SLNLP_TRY(pl->linear_r(w.tfinp, B, E, L.lin_w, c.Vt, pl->P(L.lin_b), w.logits, Vp, 0, 0.f, 0, nullptr, nullptr, st));
    } else {
        SLNLP_TRY(layernorm_fwd(t, pl->P(L.decn_w), pl->P(L.decn_b), B, E, 1e-5f, w.tfin, w.st_fin, st));
        SLNLP_TRY(pl->linear(w.tfin, B, E, pl->P(L.lin_w), c.Vt, pl->P(L.lin_b), w.logits, Vp, 0, 0.f, 0, nullptr, st));
    }
    // log_softmax (transformer.py:88-89) + the criterion skorch applies to it (helper.py:61-70)
    // the caller's copy of the log-probs is written by the same kernel (no device-to-device copy); in lockstep it lands
    // in the epoch buffer at the batch's row offset and the loss in the epoch's loss history
    SLNLP_TRY(lsm_nll(w.logits, Vp, y, B, c.Vt, c.pad_tgt, w.logp, pl->buf.scalars, train ? w.dlogits : nullptr, Vp,
                      w.row_nll, st, nullptr, logp_out ? logp_out : ls_logp, logp_out ? nullptr : ls_dyn,
                      logp_out ? nullptr : ls_loss, (!logp_out && ls_dyn) ? ls_dyn + 1 : nullptr));
    return 0;
}

extern "C" {

int slnlp_tf_forward(slnlp_tf_plan* pl, const int64_t* X, const int64_t* y, int B, int train, float* logp_out,
                     void* stream) {
    SLNLP_CHECK_ARG(pl && X && y, "tf_forward: `X` and `y` are required parameters");  // transformer.py:61-62
    SLNLP_CHECK_ARG(B > 0 && B <= pl->cfg.B, "tf_forward: batch %d outside 1..%d", B, pl->cfg.B);
    StepScope scope((hipStream_t)stream);
    SLNLP_TRY(scope.rc);
    return pl->forward_impl(X, y, B, train, logp_out, (hipStream_t)stream);
}

int slnlp_tf_seed_dlogp(slnlp_tf_plan* pl, const float* dlogp, void* stream) {
    SLNLP_CHECK_ARG(pl && dlogp && pl->last_B > 0, "tf_seed_dlogp: needs a prior forward");
    return lsm_bwd(pl->w.logp, dlogp, pl->last_B, pl->cfg.Vt, pl->w.dlogits, align_up(pl->cfg.Vt, 4), (hipStream_t)stream);
}

int slnlp_tf_backward(slnlp_tf_plan* pl, void* stream) {
    SLNLP_CHECK_ARG(pl && pl->last_B > 0, "tf_backward: needs a prior forward(train)");
    hipStream_t st = (hipStream_t)stream;
    StepScope scope(st);
    SLNLP_TRY(scope.rc);
    const slnlp_tf_config& c = pl->cfg;
    const Ws& w = pl->w;
    const Layout& L = pl->L;
    const int B = pl->last_B, E = c.E, F = c.F, H = c.H, S = c.S, dh = E / H, M = S * B, Vp = (int)align_up(c.Vt, 4);
    const float p = pl->last_p, ik = 1.f / (1.f - p);
    const unsigned long long* rng = pl->buf.rng;
    const int64_t *X = pl->last_X, *y = pl->last_y;
    const bool up = pl->use_planes;
    // Everything runs on the caller's stream; the weight gradient of each dY shares a launch with its data gradient.

    // generator: logits = tfin lin_w^T + lin_b
    SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(w.dlogits, Vp, B, c.Vt, w.tfin, E, pl->G(L.lin_w), pl->G(L.lin_b)),
                             pl->dgrad_args(w.dlogits, Vp, B, c.Vt, pl->P(L.lin_w), E, w.gfin, nullptr, 0.f, nullptr), st));
    SLNLP_TRY(layernorm_bwd(w.gfin, w.dec[c.N - 1].t3, pl->P(L.decn_w), w.st_fin, B, E, nullptr, w.gtl, nullptr, 0.f, 0,
                            rng, nullptr, nullptr, 0, st));
    const float* dt = w.gtl;  // gradient w.r.t. the current decoder layer's output
    for (int l = c.N - 1; l >= 0; --l) {
        const DecP& q = L.dec[l];
        const DecA& a = w.dec[l];
        const float* t_in = l > 0 ? w.dec[l - 1].t3 : w.t0;
        if (pl->use_rows) {
            // the same chain on plane operands: LayerNorm backward emits the sub-layer's dY as planes (the dropout-masked copy when
            // dropout is on), every Linear's backward pair is ONE gemm_rows_bwd launch, whose data gradient emits the next dY
            const PlaneOut none{};
            const PP& t_inp = l > 0 ? w.dec[l - 1].t3p : w.t0p;
            SLNLP_TRY(layernorm_bwd(dt, a.y3, pl->P(q.n3_w), a.st3, B, E, nullptr, a.gA3, nullptr, p, pl->dec_site(l, 5), rng, nullptr, nullptr, 0, st,
                                    p == 0.f ? a.d3p.out() : none, p > 0.f ? a.d3p.out() : none));
            SLNLP_TRY(pl->wd_rows(a.d3p, B, E, q.l2_w, F, nullptr, &a.ghp, a.h, ik, nullptr, a.hp, q.l2_w, q.l2_b, st));
            SLNLP_TRY(pl->wd_rows(a.ghp, B, F, q.l1_w, E, a.gt2, nullptr, nullptr, 0.f, a.gA3, a.t2p, q.l1_w, q.l1_b, st));
            SLNLP_TRY(layernorm_bwd(a.gt2, a.y2, pl->P(q.n2_w), a.st2, B, E, nullptr, a.gA2, nullptr, p, pl->dec_site(l, 3), rng, nullptr, nullptr, 0, st,
                                    p == 0.f ? a.d2p.out() : none, p > 0.f ? a.d2p.out() : none));
            SLNLP_TRY(pl->wd_rows(a.d2p, B, E, q.cout_w, E, a.gxctx, nullptr, nullptr, 0.f, nullptr, a.xctxp, q.cout_w, q.cout_b, st));
            {
                const float *Wk = pl->P(q.cin_w) + (long)E * E, *Wv = pl->P(q.cin_w) + 2L * E * E;
                const slnlp_gemm_args j1 = pl->head_expand(a.gxctx, Wv, a.dmbar, B, H, dh);
                SLNLP_TRY(gemm_group(&j1, 1, st));
                SLNLP_TRY(xmem_bwd(w.mem, pl->P(q.cin_b) + 2 * E, a.xprobs, a.psum, a.qk, a.dmbar, a.gxctx, B, S, H, dh, a.dsc, a.dqk, a.dcp,
                                   pl->G(q.cin_b) + 2 * E, w.gmem, l == c.N - 1 ? 0 : 1, p, pl->dec_site(l, 2), rng, st));
                slnlp_gemm_args jobs[3] = {pl->head_reduce(a.dqk, Wk, a.gq, nullptr, B, H, dh),
                                           pl->head_wgrad(a.q, a.dqk, pl->G(q.cin_w) + (long)E * E, B, H, dh),
                                           pl->head_wgrad(a.gxctx, a.mbar, pl->G(q.cin_w) + 2L * E * E, B, H, dh)};
                jobs[0].C_hi = a.gqp.hi; jobs[0].C_lo = a.gqp.lo; jobs[0].ldc_p = E;
                SLNLP_TRY(gemm_group(jobs, 3, st));
            }
            SLNLP_TRY(pl->wd_rows(a.gqp, B, E, q.cin_w, E, a.gt1, nullptr, nullptr, 0.f, a.gA2, a.t1p, q.cin_w, q.cin_b, st));
            SLNLP_TRY(layernorm_bwd(a.gt1, a.y1, pl->P(q.n1_w), a.st1, B, E, nullptr, a.gA1, nullptr, p, pl->dec_site(l, 1), rng, nullptr, nullptr, 0, st,
                                    p == 0.f ? a.d1p.out() : none, p > 0.f ? a.d1p.out() : none));
            // (the single-key self-attention's per-(row, head) dropout: the same mask as the forward V projection)
            SLNLP_TRY(pl->wd_rows(a.d1p, B, E, q.sout_w, E, nullptr, &a.gvp, nullptr, 0.f, nullptr, a.vp, q.sout_w, q.sout_b, st, p, pl->dec_site(l, 0), p > 0.f ? dh : 0));
            // softmax over one element has zero gradient: the q/k rows of in_proj (weight and bias) get exactly 0 -- nothing writes them
            SLNLP_TRY(pl->wd_rows(a.gvp, B, E, q.sin_w + 2L * E * E, E, a.gt0, nullptr, nullptr, 0.f, a.gA1, t_inp, q.sin_w + 2L * E * E, q.sin_b + 2 * E, st));
            dt = a.gt0;
            continue;
        }
        // norm3 / FFN
        SLNLP_TRY(layernorm_bwd(dt, a.y3, pl->P(q.n3_w), a.st3, B, E, nullptr, a.gA3, p > 0.f ? a.gB3 : nullptr, p,
                                pl->dec_site(l, 5), rng, nullptr, nullptr, 0, st));
        const float* d3 = p > 0.f ? a.gB3 : a.gA3;
        SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(d3, E, B, E, a.h, F, pl->G(q.l2_w), pl->G(q.l2_b)),
                                 pl->dgrad_args(d3, E, B, E, pl->P(q.l2_w), F, a.gh, a.h, ik, nullptr), st));
        SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(a.gh, F, B, F, a.t2, E, pl->G(q.l1_w), pl->G(q.l1_b)),
                                 pl->dgrad_args(a.gh, F, B, F, pl->P(q.l1_w), E, a.gt2, nullptr, 0.f, a.gA3), st));
        // norm2 / cross-attention
        SLNLP_TRY(layernorm_bwd(a.gt2, a.y2, pl->P(q.n2_w), a.st2, B, E, nullptr, a.gA2, p > 0.f ? a.gB2 : nullptr, p,
                                pl->dec_site(l, 3), rng, nullptr, nullptr, 0, st));
        const float* d2 = p > 0.f ? a.gB2 : a.gA2;
        SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(d2, E, B, E, a.xctx, E, pl->G(q.cout_w), pl->G(q.cout_b)),
                                 pl->dgrad_args(d2, E, B, E, pl->P(q.cout_w), E, a.gxctx, nullptr, 0.f, nullptr), st));
        // d ctx -> d mbar = Wv_h^T d ctx_h (batched GEMM) -> d scores, d qk, d memory (accumulated over the decoder layers in
        // layer order), d bv -> ONE launch of three batched jobs: d q_h = Wk_h d qk, d Wk_h = q_h^T (x) d qk, d Wv_h = d ctx_h^T (x) mbar.
        // d bk is exactly zero (a shift of all scores): nothing writes it, the gradient arena was zeroed at plan creation.
        {
            const float *Wk = pl->P(q.cin_w) + (long)E * E, *Wv = pl->P(q.cin_w) + 2L * E * E;
            const slnlp_gemm_args j1 = pl->head_expand(a.gxctx, Wv, a.dmbar, B, H, dh);
            SLNLP_TRY(gemm_group(&j1, 1, st));
            SLNLP_TRY(xmem_bwd(w.mem, pl->P(q.cin_b) + 2 * E, a.xprobs, a.psum, a.qk, a.dmbar, a.gxctx, B, S, H, dh, a.dsc, a.dqk, a.dcp,
                               pl->G(q.cin_b) + 2 * E, w.gmem, l == c.N - 1 ? 0 : 1, p, pl->dec_site(l, 2), rng, st));
            const slnlp_gemm_args jobs[3] = {pl->head_reduce(a.dqk, Wk, a.gq, nullptr, B, H, dh),
                                             pl->head_wgrad(a.q, a.dqk, pl->G(q.cin_w) + (long)E * E, B, H, dh),
                                             pl->head_wgrad(a.gxctx, a.mbar, pl->G(q.cin_w) + 2L * E * E, B, H, dh)};
            SLNLP_TRY(gemm_group(jobs, 3, st));
        }
        SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(a.gq, E, B, E, a.t1, E, pl->G(q.cin_w), pl->G(q.cin_b)),
                                 pl->dgrad_args(a.gq, E, B, E, pl->P(q.cin_w), E, a.gt1, nullptr, 0.f, a.gA2), st));
        // norm1 / self-attention (single key)
        hipStream_t sb = st;
        SLNLP_TRY(layernorm_bwd(a.gt1, a.y1, pl->P(q.n1_w), a.st1, B, E, nullptr, a.gA1, p > 0.f ? a.gB1 : nullptr, p,
                                pl->dec_site(l, 1), rng, nullptr, nullptr, 0, sb));
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
    SLNLP_TRY(embed_bwd(y, 1, B, 1, E, c.Vt, dt, pl->G(L.tgt_emb), sqrtf((float)E), -1, p, SITE_TGT_EMB, rng, w.emb_scratch_tgt, st));

    // encoder: needs the complete d memory
    // (a tall batch: the encoder's LayerNorm backward kernels also write their (dgamma, dbeta) chunk sums, elementwise.hip)
    const int Mfull = c.B * c.S;
    const bool lnf = ln_bwd_fused(Mfull);
    SLNLP_TRY(layernorm_bwd(w.gmem, w.enc[c.N - 1].x2, pl->P(L.encn_w), w.st_mem, M, E, nullptr, w.gxl, nullptr, 0.f, 0,
                            rng, lnf ? w.lnp_mem : nullptr, nullptr, Mfull, st));
    const float* dx = w.gxl;
    for (int l = c.N - 1; l >= 0; --l) {
        const EncP& q = L.enc[l];
        const EncA& a = w.enc[l];
        const float* x_in = l > 0 ? w.enc[l - 1].x2 : w.x0;
        const PP& xp_in = l > 0 ? w.enc[l - 1].x2p : w.x0p;
        const PlaneOut none{};
        // LayerNorm backward also emits the bf16 planes of the gradient that feeds the sub-layer's GEMMs
        // (the dropout-masked copy when dropout is on, else dx itself)
        // (plane path: the sub-layer's GEMMs read the planes only, so neither the masked fp32 copy nor the fp32 ReLU-gated
        //  gradient of the FFN hidden layer is stored)
        SLNLP_TRY(layernorm_bwd(dx, a.y2, pl->P(q.n2_w), a.st2, M, E, nullptr, a.gA2, (p > 0.f && !up) ? a.gB2 : nullptr, p,
                                pl->enc_site(l, 3), rng, lnf ? a.lnp2 : nullptr, nullptr, Mfull, st, (up && p == 0.f) ? a.d2p.out() : none,
                                (up && p > 0.f) ? a.d2p.out() : none));
        const float* d2 = p > 0.f ? a.gB2 : a.gA2;
        if (up) {
            SLNLP_TRY(pl->wd_group(pl->wgrad_p_args(a.d2p, E, M, E, a.hp, F, pl->G(q.l2_w), pl->G(q.l2_b)),
                                   pl->dgrad_p_args(a.d2p, E, M, E, q.l2_w, F, nullptr, a.h, ik, nullptr, &a.ghp), 0, st));
            SLNLP_TRY(pl->wd_group(pl->wgrad_p_args(a.ghp, F, M, F, a.x1p, E, pl->G(q.l1_w), pl->G(q.l1_b)),
                                   pl->dgrad_p_args(a.ghp, F, M, F, q.l1_w, E, a.gx1, nullptr, 0.f, a.gA2, nullptr), 0, st));
        } else {
            SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(d2, E, M, E, a.h, F, pl->G(q.l2_w), pl->G(q.l2_b)),
                                     pl->dgrad_args(d2, E, M, E, pl->P(q.l2_w), F, a.gh, a.h, ik, nullptr), st));
            SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(a.gh, F, M, F, a.x1, E, pl->G(q.l1_w), pl->G(q.l1_b)),
                                     pl->dgrad_args(a.gh, F, M, F, pl->P(q.l1_w), E, a.gx1, nullptr, 0.f, a.gA2), st));
        }
        SLNLP_TRY(layernorm_bwd(a.gx1, a.y1, pl->P(q.n1_w), a.st1, M, E, nullptr, a.gA1, (p > 0.f && !up) ? a.gB1 : nullptr, p,
                                pl->enc_site(l, 1), rng, lnf ? a.lnp1 : nullptr, nullptr, Mfull, st, (up && p == 0.f) ? a.d1p.out() : none,
                                (up && p > 0.f) ? a.d1p.out() : none));
        const float* d1 = p > 0.f ? a.gB1 : a.gA1;
        if (up) {
            SLNLP_TRY(pl->wd_group(pl->wgrad_p_args(a.d1p, E, M, E, a.ctxp, E, pl->G(q.out_w), pl->G(q.out_b)),
                                   pl->dgrad_p_args(a.d1p, E, M, E, q.out_w, E, a.gctx, nullptr, 0.f, nullptr, nullptr), 0, st));
            SLNLP_TRY(attn_self_bwd(a.qkv, a.probs, a.gctx, B, S, H, dh, S <= 64 ? nullptr : a.gqkv, p, pl->enc_site(l, 0), rng, st, a.gqkvp.out(), w.attn_scratch));
            SLNLP_TRY(pl->wd_group(pl->wgrad_p_args(a.gqkvp, 3 * E, M, 3 * E, xp_in, E, pl->G(q.in_w), pl->G(q.in_b)),
                                   pl->dgrad_p_args(a.gqkvp, 3 * E, M, 3 * E, q.in_w, E, a.gx0, nullptr, 0.f, a.gA1, nullptr), 0, st));
        } else {
            SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(d1, E, M, E, a.ctx, E, pl->G(q.out_w), pl->G(q.out_b)),
                                     pl->dgrad_args(d1, E, M, E, pl->P(q.out_w), E, a.gctx, nullptr, 0.f, nullptr), st));
            SLNLP_TRY(attn_self_bwd(a.qkv, a.probs, a.gctx, B, S, H, dh, a.gqkv, p, pl->enc_site(l, 0), rng, st, PlaneOut{}, w.attn_scratch));
            SLNLP_TRY(pl->wd_group_f(pl->wgrad_args(a.gqkv, 3 * E, M, 3 * E, x_in, E, pl->G(q.in_w), pl->G(q.in_b)),
                                     pl->dgrad_args(a.gqkv, 3 * E, M, 3 * E, pl->P(q.in_w), E, a.gx0, nullptr, 0.f, a.gA1), st));
        }
        dx = a.gx0;
    }
    SLNLP_TRY(embed_bwd(X, S, B, S, E, c.Vs, dx, pl->G(L.src_emb), sqrtf((float)E), -1, p, SITE_SRC_EMB, rng, w.emb_scratch_src, st, w.emb_keep));
    // every LayerNorm's (dgamma, dbeta): chunk sums in one launch, then the chunks added in order
    SLNLP_TRY(ln_param_partial(w.ln_ptable, nullptr, 5 * c.N + 2, E, M, B, c.B * c.S, c.B, st));
    SLNLP_TRY(ln_param_reduce(w.ln_table, 5 * c.N + 2, E, st));
    return 0;
}

int slnlp_tf_optim(slnlp_tf_plan* pl, float momentum, float max_norm, void* stream) {
    SLNLP_CHECK_ARG(pl, "tf_optim: null plan");
    StepScope scope((hipStream_t)stream);
    SLNLP_TRY(scope.rc);
    SLNLP_TRY(clip_sgd_step(pl->buf.params, pl->buf.grads, pl->buf.momentum, pl->L.total, pl->buf.lr, momentum, max_norm,
                            pl->w.opt_partials, pl->buf.scalars + 1, pl->buf.rng, (hipStream_t)stream,
                            pl->use_planes ? pl->w.wp.out() : PlaneOut{}, pl->wplane_begin(), pl->wplane_end()));
    if (!recording()) pl->params_stepped();      // (a lockstep replay does this per step itself)
    return 0;
}

// clip_grad_norm_ + torch.optim.Adam on the arena: exp_avg = buf.momentum, exp_avg_sq = the caller's arena-shaped
// buffer, step count = scalars[2] (advanced on the device).
int slnlp_tf_optim_adam(slnlp_tf_plan* pl, float* exp_avg_sq, float beta1, float beta2, float eps, float weight_decay,
                        float max_norm, void* stream) {
    SLNLP_CHECK_ARG(pl && exp_avg_sq, "tf_optim_adam: null argument");
    StepScope scope((hipStream_t)stream);
    SLNLP_TRY(scope.rc);
    SLNLP_TRY(clip_adam_step(pl->buf.params, pl->buf.grads, pl->buf.momentum, exp_avg_sq, pl->L.total, pl->buf.lr, beta1, beta2, eps,
                             weight_decay, max_norm, pl->w.opt_partials, pl->buf.scalars + 1, pl->buf.rng, pl->buf.scalars + 2,
                             (hipStream_t)stream, pl->use_planes ? pl->w.wp.out() : PlaneOut{}, pl->wplane_begin(), pl->wplane_end()));
    if (!recording()) pl->params_stepped();
    return 0;
}

int slnlp_tf_set_destroy_sync(slnlp_tf_plan* pl, int on) {
    SLNLP_CHECK_ARG(pl, "tf_set_destroy_sync: null plan");
    pl->destroy_sync = on ? 1 : 0;
    return 0;
}

// The parameter arena was written from outside the library (load_state_dict, a torch optimizer, an in-place edit):
// derived data is stale.  The Python engines call this when the arena tensor's version counter has moved.
int slnlp_tf_params_changed(slnlp_tf_plan* pl) {
    SLNLP_CHECK_ARG(pl, "tf_params_changed: null plan");
    bump_params_generation(pl->buf.params);
    return 0;
}

int slnlp_tf_train_step(slnlp_tf_plan* pl, const int64_t* X, const int64_t* y, int B, float momentum, float max_norm,
                        float* logp, void* stream) {
    StepScope scope((hipStream_t)stream);        // one scope for the whole step (the nested entry points re-enter it)
    SLNLP_TRY(scope.rc);
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
    pl->wplanes_gen = 0;      // the captured step always re-splits the weights: a replay cannot check the arena's generation
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
    StepScope scope((hipStream_t)stream);
    SLNLP_TRY(scope.rc);
    SLNLP_TRY(pl->prepare_planes(B, (hipStream_t)stream));
    if (hipGraphLaunch(it->second, (hipStream_t)stream) != hipSuccess) {
        set_error("tf_graph_launch: %s", hipGetErrorString(hipGetLastError()));
        return SLNLP_ERR_LAUNCH;
    }
    pl->params_stepped();
    return 0;
}

// Debug helper: "name offset" lines (byte offsets into the workspace, in carve order) of every fp32 activation /
// gradient buffer -- lets a test locate a difference between two plans' workspaces.
int slnlp_tf_debug_layout(const slnlp_tf_config* cfg, char* out, int64_t out_bytes) {
    SLNLP_TRY(check_cfg(cfg));
    SLNLP_CHECK_ARG(out && out_bytes > 0, "tf_debug_layout: no output buffer");
    const Ws w = carve(*cfg, nullptr);
    std::string s;
    bool first = true;          // (only the very first buffer may sit at offset 0: any other null pointer is a buffer this configuration does not carve)
    auto add = [&](const std::string& n, const void* p) {
        if (p == nullptr && !first) return;
        first = false;
        s += n + " " + std::to_string((size_t)(const char*)p) + "\n";
    };
#define F(pre, st, f) add(pre + std::string(#f), st.f)
    add("x0", w.x0); add("t0", w.t0);
    for (int i = 0; i < cfg->N; ++i) {
        const EncA& a = w.enc[i];
        const std::string pre = "enc" + std::to_string(i) + ".";
        F(pre, a, qkv); F(pre, a, probs); F(pre, a, ctx); F(pre, a, y1); F(pre, a, st1); F(pre, a, x1); F(pre, a, h); F(pre, a, y2); F(pre, a, st2);
        F(pre, a, x2); F(pre, a, lnp1); F(pre, a, lnp2); F(pre, a, gA2); F(pre, a, gB2); F(pre, a, gh); F(pre, a, gx1); F(pre, a, gA1); F(pre, a, gB1);
        F(pre, a, gctx); F(pre, a, gqkv); F(pre, a, gx0);
    }
    add("mem", w.mem); add("st_mem", w.st_mem); add("lnp_mem", w.lnp_mem);
    for (int i = 0; i < cfg->N; ++i) {
        const DecA& a = w.dec[i];
        const std::string pre = "dec" + std::to_string(i) + ".";
        F(pre, a, v); F(pre, a, y1); F(pre, a, st1); F(pre, a, t1); F(pre, a, q); F(pre, a, qk); F(pre, a, mbar); F(pre, a, psum); F(pre, a, xprobs); F(pre, a, xctx); F(pre, a, y2);
        F(pre, a, st2); F(pre, a, t2); F(pre, a, h); F(pre, a, y3); F(pre, a, st3); F(pre, a, t3); F(pre, a, lnp1); F(pre, a, lnp2); F(pre, a, lnp3);
        F(pre, a, gA3); F(pre, a, gB3); F(pre, a, gh); F(pre, a, gt2); F(pre, a, gA2); F(pre, a, gB2); F(pre, a, gxctx); F(pre, a, gq); F(pre, a, dmbar); F(pre, a, dsc); F(pre, a, dqk); F(pre, a, dcp);
        F(pre, a, gt1); F(pre, a, gA1); F(pre, a, gB1); F(pre, a, gv); F(pre, a, gt0);
    }
#undef F
    add("tfin", w.tfin); add("st_fin", w.st_fin); add("lnp_fin", w.lnp_fin); add("logits", w.logits); add("dlogits", w.dlogits);
    add("logp", w.logp); add("row_nll", w.row_nll); add("gfin", w.gfin); add("gtl", w.gtl); add("gmem", w.gmem); add("gxl", w.gxl);
    add("emb_scratch_src", w.emb_scratch_src); add("emb_scratch_tgt", w.emb_scratch_tgt); add("emb_keep", w.emb_keep);
    add("opt_partials", w.opt_partials); add("ln_table", w.ln_table); add("wp.hi", w.wp.hi); add("wp.lo", w.wp.lo);
    add("planes_begin", w.planes_begin); add("gscr0", w.gscr[0]); add("planes_end", w.planes_end);
    add("end", (const char*)nullptr + w.bytes);
    SLNLP_CHECK_ARG((int64_t)s.size() + 1 <= out_bytes, "tf_debug_layout: needs %zu bytes", s.size() + 1);
    memcpy(out, s.c_str(), s.size() + 1);
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
