"""GPU parity of the EncoderDecoder{LSTM,GRU}Attn path (libslnlp slnlp_rnn_* through
slnlp.rnn_engine) against the golden vectors captured from the reference and the CPU oracle."""
import numpy as np
import pytest
import torch

import gold

pytestmark = pytest.mark.gpu

TOL = 1e-3
TOL_GRAD = 2e-2


def make_engine(c, sd, dropout=0.0, precision=3, seed=0):
    from slnlp import rnn_engine as re_
    cfg = re_.make_config(c["rnn_type"], c["E"], c["Hd"], c["N"], c["Vs"], c["Vt"], c["B"], c["S"], 1, 1, 0, dropout, precision)
    eng = re_.RnnEngine(cfg, seed=seed)
    eng.load_state(sd)
    return eng


CASES = [("lstm", "tiny"), ("lstm", "mid"), ("lstm", "cfg3"), ("gru", "tiny"), ("gru", "mid"), ("gru", "cfg3")]


def test_layout_is_reference_state_dict():
    from oracle import rnn_ref as rr
    from slnlp import rnn_engine as re_
    for rt in ("lstm", "gru"):
        cfg = re_.make_config(rt, 128, 256, 2, 3000, 202, 50, 48)
        ents, total = re_.layout(cfg)
        assert [(n, s) for n, s, _ in ents] == [(n, tuple(s)) for n, s in rr.param_shapes(rt, 128, 256, 2, 3000, 202)]
    n = sum(int(np.prod(s)) for _, s, _ in re_.layout(re_.make_config("lstm", 128, 256, 2, 3000, 200, 50, 48))[0])
    assert n == 4831744


@pytest.mark.parametrize("rnn_type,name", CASES)
def test_forward_vs_golden(rnn_type, name):
    g, c, sd, X, L, y = gold.rnn_case(rnn_type, name)
    eng = make_engine(c, sd)
    logp = eng.forward(X.cuda(), y.cuda(), L.cuda()).cpu()
    err = gold.rel_err(logp.numpy(), g["logp"])
    print(f"[{rnn_type} {name}] logp rel err {err:.2e}")
    assert err < TOL
    assert np.array_equal(logp.argmax(-1).numpy(), g["argmax"])
    B, S, Hd, N = c["B"], c["S"], c["Hd"], c["N"]
    al = eng.tap("alphas", B, S).cpu().numpy()
    assert gold.rel_err(al.reshape(g["tap_alphas"].shape), g["tap_alphas"]) < TOL
    fin = eng.tap("enc_final", N * B, 2 * Hd).cpu().view(N, B, 2 * Hd)
    if name == "tiny":
        assert gold.rel_err(fin.numpy(), g["tap_enc_final"]) < 2e-4
        eo = eng.tap("enc_out", S * B, 2 * Hd).cpu().view(S, B, 2 * Hd).transpose(0, 1)      # -> [B,S,2Hd]
        assert gold.rel_err(eo.numpy(), g["tap_enc_out"]) < 2e-4
        assert torch.all(eo[torch.arange(S)[None, :] >= L[:, None]] == 1.0)                   # pad rows = float(pad_idx)
        ctx = eng.tap("context", B, 2 * Hd).cpu().numpy()
        assert gold.rel_err(ctx.reshape(g["tap_context"].shape), g["tap_context"]) < 2e-4
        y2 = (y - 2 + 3) % (c["Vt"] - 2) + 2
        out2 = eng.forward(X.cuda(), y2.cuda(), L.cuda()).cpu()
        assert torch.equal(out2, logp)                                                        # output independent of y
    else:
        assert gold.rel_err(fin[:, :8].numpy(), g["tap_enc_final"]) < TOL


@pytest.mark.parametrize("rnn_type,name", CASES)
def test_train_steps_vs_golden(rnn_type, name):
    """Reference trajectories: 5 steps tiny, 2 steps mid and at the configs[2] shape (E512 Hd512 N4), LSTM and GRU."""
    g, c, sd, X, L, y = gold.rnn_case(rnn_type, name)
    assert len(g["losses"]) == (5 if name == "tiny" else 2)
    eng = make_engine(c, sd)
    eng.set_lr(0.01)
    Xc, yc, Lc = X.cuda(), y.cuda(), L.cuda()
    eng.forward(Xc, yc, Lc, train=True)
    eng.backward()
    gv = {k: v.cpu() for k, v in eng.views(eng.grads).items()}
    gold.check_summary(g, "grad0", gv, TOL_GRAD)
    assert float(gv["model.decoder.pre_output_layer.weight"].abs().max()) == 0.0      # dead weight
    assert float(gv["model.src_embed.weight"][1].abs().max()) == 0.0                  # padding_idx row
    assert float(gv["model.trg_embed.weight"][1:].abs().max()) == 0.0                 # only <bos> (=0) sees grad
    for s in range(len(g["losses"])):
        eng.train_step(Xc, yc, Lc, momentum=0.9, max_norm=0.5)
        torch.cuda.synchronize()
        print(f"[{rnn_type} {name}] step {s}: loss {eng.loss:.6f} (ref {g['losses'][s]:.6f}) norm {eng.grad_norm:.5f} (ref {g['grad_norms'][s]:.5f})")
        assert abs(eng.loss - g["losses"][s]) < TOL * g["losses"][s]
        assert abs(eng.grad_norm - g["grad_norms"][s]) < 2e-3 * g["grad_norms"][s]
    gold.check_summary(g, "wfinal", {k: v.cpu() for k, v in eng.views().items()}, TOL)


@pytest.mark.parametrize("rnn_type", ["lstm", "gru"])
def test_gradients_vs_oracle_and_graph(rnn_type):
    from oracle import rnn_ref, train_ref
    g, c, sd, X, L, y = gold.rnn_case(rnn_type, "tiny")
    fwd = lambda p, X, y, L: rnn_ref.forward(p, X, y, L, rnn_type=rnn_type, num_layers=c["N"])
    loss_o, _, grads_o = train_ref.Trainer(sd, fwd, frozen=("model.decoder.pre_output_layer.weight",)).loss_and_grads(X, y, L)
    eng = make_engine(c, sd)
    eng.forward(X.cuda(), y.cuda(), L.cuda(), train=True)
    eng.backward()
    torch.cuda.synchronize()
    assert abs(eng.loss - float(loss_o)) < TOL * float(loss_o)
    gv = {k: v.cpu() for k, v in eng.views(eng.grads).items()}
    for k, go in grads_o.items():
        if go is None:
            continue
        scale = float(go.abs().max())
        e = float((gv[k] - go).abs().max()) / max(scale, 1e-12) if scale > 0 else float(gv[k].abs().max())
        assert e < 2e-3, f"{k}: grad err {e:.2e}"
    # graph replay == eager, bitwise
    e1, e2 = make_engine(c, sd), make_engine(c, sd)
    e1.set_lr(0.01); e2.set_lr(0.01)
    with torch.cuda.stream(torch.cuda.Stream()):
        for _ in range(3):
            e1.train_step(X.cuda(), y.cuda(), L.cuda())
            e2.train_step_graph(X.cuda(), y.cuda(), L.cuda())
    torch.cuda.synchronize()
    assert torch.equal(e1.params, e2.params)


@pytest.mark.parametrize("rnn_type", ["lstm", "gru"])
def test_dropout_path_vs_oracle_with_same_masks(rnn_type):
    from oracle import rnn_ref, train_ref
    from slnlp import ops
    g, c, sd, X, L, y = gold.rnn_case(rnn_type, "tiny")
    p, B, S, Hd, N = 0.3, c["B"], c["S"], c["Hd"], c["N"]
    eng = make_engine(c, sd, dropout=p, seed=5)
    masks = {}
    for l in range(N - 1):
        m = ops.dropout_mask(S * B, 2 * Hd, p, 32 + l, eng.rng).cpu().view(S, B, 2 * Hd).transpose(0, 1)
        masks[f"model.encoder.rnn.dropout{l}"] = m
        masks[f"model.decoder.rnn.dropout{l}"] = ops.dropout_mask(B, Hd, p, 64 + l, eng.rng).cpu()
    logp = eng.forward(X.cuda(), y.cuda(), L.cuda(), train=True).cpu().clone()
    eng.backward()
    torch.cuda.synchronize()
    fwd = lambda pr, X, y, L: rnn_ref.forward(pr, X, y, L, rnn_type=rnn_type, num_layers=N, p_drop=p, masks=masks)
    loss_o, logp_o, grads_o = train_ref.Trainer(sd, fwd, frozen=("model.decoder.pre_output_layer.weight",)).loss_and_grads(X, y, L)
    assert gold.rel_err(logp.numpy(), logp_o.numpy()) < TOL
    gv = {k: v.cpu() for k, v in eng.views(eng.grads).items()}
    for k, go in grads_o.items():
        if go is None:
            continue
        scale = float(go.abs().max())
        e = float((gv[k] - go).abs().max()) / max(scale, 1e-12) if scale > 0 else float(gv[k].abs().max())
        assert e < 2e-3, f"{k}: grad err {e:.2e}"


@pytest.mark.parametrize("rnn_type,name", [("lstm", "cfg3"), ("gru", "cfg3"), ("lstm", "mid"), ("gru", "mid")])
def test_fused_backward_step_equals_cell_plus_ksliced_gemm(rnn_type, name):
    """One launch per backward timestep (slnlp_rnn_step_bwd: recurrent data gradient + cell backward, round 4) against the
    cell kernel + K-sliced grouped GEMM pair of rounds 1-3 (slnlp_rnn_set_fused_backward(plan, 0)) on the same plan: three
    train steps with dropout and ragged lengths -- the same partial products added in the same gate order, so every gradient
    and every weight agree to fp32 rounding (the stand-alone GEMM cuts its K loop over two wave groups: another order inside a
    gate's sum), and the loss trajectories are the same."""
    g, c, sd, X, L, y = gold.rnn_case(rnn_type, name)
    Xc, yc, Lc = X.cuda(), y.cuda(), L.cuda()
    res = []
    for fused in (True, False):
        eng = make_engine(c, sd, dropout=0.1)
        eng.set_fused_backward(fused)
        eng.set_lr(0.01)
        eng.forward(Xc, yc, Lc, train=True)
        eng.backward()
        torch.cuda.synchronize()
        grads = eng.grads.clone()
        losses = []
        for _ in range(3):
            eng.train_step(Xc, yc, Lc, momentum=0.9, max_norm=0.5)
            torch.cuda.synchronize()
            losses.append(eng.loss)
        res.append((grads, eng.params.clone(), losses))
    (g1, w1, l1), (g0, w0, l0) = res
    # fp32 rounding, amplified by the recurrence over S timesteps and N layers (measured: 1.3e-4 of the gradient scale for the
    # 12-step GRU, where the unfused path contracts all gates in ONE K loop and the fused one adds per-gate partial sums)
    scale = float(g0.abs().max())
    assert float((g1 - g0).abs().max()) < 5e-4 * scale, float((g1 - g0).abs().max()) / scale
    assert float((w1 - w0).abs().max()) < 1e-4 * float(w0.abs().max())
    assert all(abs(a - b) < 1e-5 * abs(b) for a, b in zip(l1, l0)), (l1, l0)


@pytest.mark.parametrize("lstm", [1, 0])
@pytest.mark.parametrize("B,Hd,ndir", [(50, 512, 2), (50, 512, 1), (33, 256, 2), (70, 64, 2), (20, 40, 1)])
def test_recurrent_step_tiles_return_the_same_bits(lstm, B, Hd, ndir):
    """The fused forward timestep on its two tiles -- 16 batch rows per workgroup with the gates on four waves (one fit's launch:
    every CU gets a workgroup) and 64 rows with the gates in one wave's accumulators (merged lockstep launches) -- must agree bit
    for bit (slnlp_set_rnn_step_tile): same K order, same halves, same cell arithmetic per element."""
    import ctypes as C
    from slnlp import ops
    from slnlp._lib import RnnStepDir, check, load, ptr, stream_ptr
    G = 4 if lstm else 3
    g = torch.Generator().manual_seed(B + Hd + lstm)
    rnd = lambda *s: torch.randn(*s, generator=g).cuda()
    h, c, W, bh, xp = rnd(ndir, B, Hd) * 0.5, rnd(ndir, B, Hd) * 0.5, rnd(ndir, G * Hd, Hd) * 0.05, rnd(ndir, G * Hd) * 0.1, rnd(ndir, B, G * Hd)
    lengths = torch.randint(1, 6, (B,), generator=g).cuda()
    rng = ops.make_rng(seed=3, step=1)
    t, p, site, fill, ld_out = 2, 0.2, 40, 1.0, 2 * Hd
    res = []
    try:
        for tile in (1, 0):
            load().slnlp_set_rnn_step_tile(tile)
            bufs = dict(acts=torch.zeros(ndir, B, G * Hd).cuda(), cprev=torch.zeros(ndir, B, Hd).cuda(), hn=torch.zeros(ndir, B, Hd).cuda(),
                        out=torch.zeros(B, ld_out).cuda(), c=c.clone(), h_out=torch.zeros(ndir, B, Hd).cuda())
            dirs = (RnnStepDir * ndir)(*[RnnStepDir(ptr(h[k]), ptr(bufs["h_out"][k]), ptr(W[k]), ptr(bh[k]), ptr(xp[k]), ptr(bufs["c"][k]), ptr(bufs["cprev"][k]),
                                                    ptr(bufs["acts"][k]), ptr(bufs["hn"][k]), bufs["out"].data_ptr() + 4 * k * Hd, t, t * B, k * Hd) for k in range(ndir)])
            check(load().slnlp_rnn_step_fwd(lstm, dirs, ndir, B, Hd, ptr(lengths), fill, ld_out, p, site, ptr(rng), 3, stream_ptr()), "step")
            torch.cuda.synchronize()
            res.append(bufs)
    finally:
        load().slnlp_set_rnn_step_tile(1)
    for k in ("h_out", "acts", "out", "c", "cprev", "hn"):
        assert torch.equal(res[0][k], res[1][k]), k
    assert float(res[0]["h_out"].abs().max()) > 0


@pytest.mark.parametrize("lstm", [1, 0])
@pytest.mark.parametrize("B,Hd", [(50, 512), (7, 40), (70, 64)])
def test_fused_step_equals_gemm_plus_cell(lstm, B, Hd):
    """slnlp_rnn_step_fwd (recurrent GEMM + cell in one launch) against slnlp_gemm + slnlp_rnn_cell_fwd on the same
    inputs: the same split-bf16 products -> the same state, gate activations and (dropped, length-masked) outputs.  Bit for
    bit where both run the K tiles in one sequence (Hd < 256: fewer than four K tiles); where the stand-alone GEMM cuts its K
    loop over two wave groups (gemm.hip, KS = 2) the two halves are added in another order: equal to fp32 rounding."""
    import ctypes as C
    from slnlp import ops
    from slnlp._lib import RnnCellDir, RnnStepDir, check, load, ptr, stream_ptr
    G = 4 if lstm else 3
    g = torch.Generator().manual_seed(B + Hd + lstm)
    rnd = lambda *s: torch.randn(*s, generator=g).cuda()
    h, c, W, bh, xp = rnd(B, Hd) * 0.5, rnd(B, Hd) * 0.5, rnd(G * Hd, Hd) * 0.05, rnd(G * Hd) * 0.1, rnd(B, G * Hd)
    lengths = torch.randint(1, 6, (B,), generator=g).cuda()
    rng = ops.make_rng(seed=3, step=1)
    t, p, site, fill, ld_out = 2, 0.2, 40, 1.0, 2 * Hd

    def buffers():
        return dict(acts=torch.zeros(B, G * Hd).cuda(), cprev=torch.zeros(B, Hd).cuda(), hn=torch.zeros(B, Hd).cuda(),
                    out=torch.zeros(B, ld_out).cuda(), c=c.clone())
    # reference: GEMM then cell
    r = buffers()
    hproj = ops.gemm(h, W, M=B, N=G * Hd, K=Hd, bias=bh)
    h_ref, hprev_ref = h.clone(), torch.zeros(B, Hd).cuda()
    d = RnnCellDir(ptr(xp), ptr(hproj), ptr(h_ref), ptr(r["c"]), ptr(hprev_ref), ptr(r["cprev"]), ptr(r["acts"]), ptr(r["hn"]),
                   ptr(r["out"]), t, t * B, Hd)
    check(load().slnlp_rnn_cell_fwd(lstm, C.byref(d), 1, B, Hd, ptr(lengths), fill, ld_out, p, site, ptr(rng), stream_ptr()), "cell")
    # fused
    f = buffers()
    h_out = torch.zeros(B, Hd).cuda()
    s = RnnStepDir(ptr(h), ptr(h_out), ptr(W), ptr(bh), ptr(xp), ptr(f["c"]), ptr(f["cprev"]), ptr(f["acts"]), ptr(f["hn"]),
                   ptr(f["out"]), t, t * B, Hd)
    check(load().slnlp_rnn_step_fwd(lstm, C.byref(s), 1, B, Hd, ptr(lengths), fill, ld_out, p, site, ptr(rng), 3, stream_ptr()), "step")
    torch.cuda.synchronize()
    same = torch.equal if Hd < 256 else (lambda a, b: torch.allclose(a, b, rtol=2e-6, atol=2e-6))
    assert same(h_out, h_ref) and torch.equal(hprev_ref, h)
    assert same(f["acts"], r["acts"]) and same(f["out"], r["out"])
    if lstm:
        assert same(f["c"], r["c"]) and torch.equal(f["cprev"], r["cprev"])
    else:
        assert same(f["hn"], r["hn"])
    assert (lengths <= t).any() and (lengths > t).any()          # both masked and live rows were exercised


@pytest.mark.parametrize("lstm", [1, 0])
@pytest.mark.parametrize("B,Hd,S", [(50, 512, 48), (9, 64, 5), (64, 128, 12)])
def test_persistent_layer_equals_stepwise(lstm, B, Hd, S):
    """slnlp_rnn_layer_fwd (all S timesteps of a bidirectional layer in one launch, W_hh slice resident in LDS,
    device-wide barrier between steps) against S calls of slnlp_rnn_step_fwd: bit-identical state chain, gate
    activations, outputs; barrier words back to rest, error flag clear."""
    import ctypes as C
    from slnlp import ops
    from slnlp._lib import RnnLayerDir, RnnStepDir, check, load, ptr, stream_ptr
    G = 4 if lstm else 3
    g = torch.Generator().manual_seed(B + Hd + S + lstm)
    rnd = lambda *s: torch.randn(*s, generator=g).cuda()
    W = [rnd(G * Hd, Hd) * 0.05 for _ in range(2)]
    bh = [rnd(G * Hd) * 0.1 for _ in range(2)]
    xp = [rnd(S, B, G * Hd) for _ in range(2)]
    lengths = torch.randint(1, S + 1, (B,), generator=g).cuda()
    rng = ops.make_rng(seed=3, step=1)
    p, site, fill, ld_out = 0.2, 33, 1.0, 2 * Hd

    def buffers():
        return [dict(hprev=torch.zeros(S, B, Hd).cuda(), h=torch.zeros(B, Hd).cuda(), c=torch.zeros(B, Hd).cuda(),
                     cprev=torch.zeros(S, B, Hd).cuda(), acts=torch.zeros(S, B, G * Hd).cuda(), hn=torch.zeros(S, B, Hd).cuda())
                for _ in range(2)]
    # reference: one launch per timestep
    r, out_r = buffers(), torch.zeros(S * B, ld_out).cuda()
    for step in range(S):
        dirs = (RnnStepDir * 2)()
        for d in range(2):
            t = step if d == 0 else S - 1 - step
            tn = t + 1 if d == 0 else t - 1
            e = r[d]
            h_out = e["hprev"][tn] if step + 1 < S else e["h"]
            dirs[d] = RnnStepDir(ptr(e["hprev"][t]), ptr(h_out), ptr(W[d]), ptr(bh[d]), ptr(xp[d][t]), ptr(e["c"]), ptr(e["cprev"][t]),
                                 ptr(e["acts"][t]), ptr(e["hn"][t]), out_r[t * B:].data_ptr() + 4 * d * Hd, t, t * B, d * Hd)
        check(load().slnlp_rnn_step_fwd(lstm, dirs, 2, B, Hd, ptr(lengths), fill, ld_out, p, site, ptr(rng), 3, stream_ptr()), "step")
    # persistent
    f, out_f = buffers(), torch.zeros(S * B, ld_out).cuda()
    sync = torch.zeros(4, dtype=torch.int32).cuda()
    dirs = (RnnLayerDir * 2)()
    for d in range(2):
        e = f[d]
        dirs[d] = RnnLayerDir(ptr(e["hprev"]), ptr(e["h"]), ptr(W[d]), ptr(bh[d]), ptr(xp[d]), ptr(e["c"]), ptr(e["cprev"]),
                              ptr(e["acts"]), ptr(e["hn"]), out_f.data_ptr() + 4 * d * Hd, d * Hd, d)
    launched = C.c_int32(0)
    check(load().slnlp_rnn_layer_fwd(lstm, dirs, 2, B, Hd, S, ptr(lengths), fill, ld_out, p, site, ptr(rng), 3, ptr(sync),
                                     C.byref(launched), stream_ptr()), "layer")
    torch.cuda.synchronize()
    assert launched.value == 1
    assert sync.tolist()[0] == 0 and sync.tolist()[2] == 0          # barrier at rest, no spin timeout
    for d in range(2):
        for k in ("hprev", "h", "acts") + (("c", "cprev") if lstm else ("hn",)):
            assert torch.equal(f[d][k], r[d][k]), (d, k)
    assert torch.equal(out_f, out_r)


def test_plan_with_persistent_layers_matches_stepwise_plan():
    """RnnEngine.set_persistent(True): the whole train step (forward through the persistent layer kernels, same
    backward) gives the same loss trajectory and weights as the default per-timestep plan; health() stays 0."""
    import bench
    from slnlp import rnn_engine as re_, synth
    c = dict(bench.WORKLOADS["cfg3"], precision=3, N=2)
    out = []
    for persistent in (False, True):
        cfg, sd = bench.build_sd(c, seed=1)
        eng = re_.RnnEngine(cfg, seed=1)
        eng.load_state(sd); eng.set_lr(0.01)
        eng.set_persistent(persistent)
        Xn, Ln, yn = synth.make_batch(3 * c["B"], c["S"], c["Vs"], c["Vt"], seed=1)
        X, L, y = torch.from_numpy(Xn).cuda(), torch.from_numpy(Ln).cuda(), torch.from_numpy(yn).cuda()
        B, losses = c["B"], []
        for i in range(3):
            eng.train_step(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B], L[i * B:(i + 1) * B], 0.9, 0.5)
            losses.append(eng.loss)
        assert eng.health() == 0
        out.append((losses, eng.params.clone()))
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])
