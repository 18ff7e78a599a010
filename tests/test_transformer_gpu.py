"""GPU parity of the whole Transformer path (libslnlp slnlp_tf_* through
slnlp.tf_engine) against (1) the golden vectors captured from the reference
and (2) the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): argmax class ids bit-exact; log-probs / loss
within 1e-3 relative.  precision=3 (split-bf16 MFMA) is the parity-grade mode;
precision=1 (single bf16 pass) is only required to stay within 3e-2."""
import numpy as np
import pytest
import torch

import gold
from slnlp import synth

pytestmark = pytest.mark.gpu

TOL = 1e-3          # north_star: logits / loss within 1e-3 rel
# Per-element gradients are ill-conditioned (a ReLU gate within rounding of 0 flips): emulating the
# GEMMs on the CPU with an exact-product 6-term bf16 split already differs from torch fp32 by 3e-3
# (cfg1) .. 4e-3 (cfg2) of the tensor max, the shipped 3-term split by 5e-3 .. 4e-2.
TOL_GRAD = 2e-2     # per-tensor gradient error relative to the tensor max (2-layer configs)
# pre-clip total gradient norm: 2e-3 on the first step from the reference's weights (measured 7.5e-4 at cfg2, <= 4e-4 at
# tiny / cfg1); later steps run from weights that already differ by the split-bf16 rounding of the previous updates, and the
# norm of this 12-layer post-LN net is the most sensitive scalar of the step (measured at cfg2: 2.4e-4, 8.9e-4, 3.1e-3 at
# steps 1-3 while the loss stays within 1.2e-5 of the reference) -> 5e-3 there through round 4.
# Round 5 (profiles/r05_grad_norm_variants.txt): four equally accurate arithmetics of the same step -- the decoder's products on
# plane operands or on fp32 operands, gradient products in two or three split-bf16 passes -- put the cfg2 norm of steps 1-4 at
# -2.9e-3 .. +1.8e-4, -1.7e-3 .. +1.0e-3, -3e-5 .. +4.1e-3 and -5.7e-3 .. -4.1e-3 of the reference's: the deviation has the sign and
# size of the STEP (the weights it starts from), 0.5e-3 .. 2e-3 of spread between variants, while every variant's loss stays within
# 1e-4 (5e-6 with plane operands).  The bar is that envelope with the spread once more on top: 8e-3.
TOL_NORM0, TOL_NORM = 2e-3, 8e-3


def make_engine(c, sd, dropout=0.0, precision=3, B=None, seed=0):
    from slnlp import tf_engine as te
    cfg = te.make_config(c["E"], c["H"], c["N"], c["F"], c["Vs"], c["Vt"], B or c["B"], c["S"], 1, 1, dropout, precision)
    eng = te.TransformerEngine(cfg, seed=seed)
    eng.load_state(sd)
    return eng


@pytest.mark.parametrize("name", ["tiny", "cfg1", "cfg2", "e1024", "cfg5"])
def test_forward_vs_golden(name):
    g, c, sd, X, L, y = gold.tf_case(name)
    eng = make_engine(c, sd)
    logp = eng.forward(X.cuda(), y.cuda()).cpu()
    err = gold.rel_err(logp.numpy(), g["logp"])
    print(f"[{name}] logp rel err {err:.2e}")
    assert err < TOL
    assert np.array_equal(logp.argmax(-1).numpy(), g["argmax"])          # bit-exact class ids
    assert np.allclose(np.exp(logp.double().numpy()).sum(-1), 1.0, atol=1e-5)
    if name == "tiny":
        M = c["B"] * c["S"]
        for k in [k for k in g if k.startswith("tap_")]:
            tap = k[4:]
            rows = M if tap in ("src_embed", "memory") or tap.startswith("enc") else c["B"]
            cols = c["Vt"] if tap == "logits" else c["E"]
            got = eng.tap(tap, rows, cols).cpu().numpy()
            e = gold.rel_err(got.reshape(g[k].shape), g[k])
            assert e < 2e-4, (tap, e)


def test_forward_precision1_is_bounded():
    g, c, sd, X, L, y = gold.tf_case("cfg1")
    eng = make_engine(c, sd, precision=1)
    logp = eng.forward(X.cuda(), y.cuda()).cpu()
    err = gold.rel_err(logp.numpy(), g["logp"])
    print(f"[cfg1 precision=1] logp rel err {err:.2e}")
    assert err < 3e-2


def test_behaviour_pins():
    g, c, sd, X, L, y = gold.tf_case("tiny")
    eng = make_engine(c, sd)
    base = eng.forward(X.cuda(), y.cuda()).cpu().clone()
    y2 = y.clone(); y2[1] = (y2[1] - 2 + 1) % (c["Vt"] - 2) + 2
    out = eng.forward(X.cuda(), y2.cuda()).cpu().clone()
    assert gold.rel_err(out.numpy(), g["pin_y_changed"]) < TOL
    ch = (out - base).abs().amax(1)
    assert ch[1] > 1e-3 and float(ch[[0, 2, 3]].max()) == 0.0            # output row i depends on y[i] only
    X2 = X.clone(); X2[:, 5] = (X2[:, 5] - 2 + 7) % (c["Vs"] - 2) + 2
    last = f"enc{c['N'] - 1}"
    M = c["B"] * c["S"]
    eng.forward(X.cuda(), y.cuda()); t1 = eng.tap(last, M, c["E"]).cpu().view(c["S"], c["B"], -1)
    eng.forward(X2.cuda(), y.cuda()); t2 = eng.tap(last, M, c["E"]).cpu().view(c["S"], c["B"], -1)
    assert gold.rel_err(t2.numpy(), g["pin_causal_enc_last"]) < 2e-4
    assert torch.equal(t1[:5], t2[:5]) and not torch.equal(t1[5:], t2[5:])   # encoder is causal
    # <pad> as decoder input -> NaN row (torch semantics), other rows untouched
    y3 = y.clone(); y3[2] = 1
    out = eng.forward(X.cuda(), y3.cuda()).cpu()
    assert torch.isnan(out[2]).all() and torch.equal(out[[0, 1, 3]], base[[0, 1, 3]])


def test_smaller_batch_matches_rows():
    g, c, sd, X, L, y = gold.tf_case("cfg1")
    eng = make_engine(c, sd)
    full = eng.forward(X.cuda(), y.cuda()).cpu().clone()
    part = eng.forward(X[:17].cuda(), y[:17].cuda()).cpu()
    assert gold.rel_err(part.numpy(), full[:17].numpy()) < 1e-5


def _oracle_trainer(c, sd, lr=0.01):
    from oracle import train_ref, transformer_ref as tr
    fwd = lambda p, X, y, L: tr.forward(p, X, y, num_heads=c["H"], num_layers=c["N"])
    return train_ref.Trainer(sd, fwd, pad_tgt=1, lr=lr, momentum=0.9, max_norm=0.5)


@pytest.mark.parametrize("name", ["tiny", "cfg1"])
def test_gradients_vs_oracle(name):
    g, c, sd, X, L, y = gold.tf_case(name)
    # (an ignored target == <pad> would also be the decoder INPUT and turn the row NaN in the
    #  reference -- transformer.py:65 -- so ignore_index is exercised in test_kernels_gpu::test_lsm_nll)
    loss_o, _, grads_o = _oracle_trainer(c, sd).loss_and_grads(X, y, L)
    eng = make_engine(c, sd)
    eng.forward(X.cuda(), y.cuda(), train=True)
    eng.backward()
    torch.cuda.synchronize()
    assert abs(eng.loss - float(loss_o)) < TOL * abs(float(loss_o))
    gv = {k: v.cpu() for k, v in eng.views(eng.grads).items()}
    worst = ("", 0.0)
    for k, go in grads_o.items():
        scale = float(go.abs().max())
        e = float((gv[k] - go).abs().max()) / max(scale, 1e-12) if scale > 0 else float(gv[k].abs().max())
        if e > worst[1]:
            worst = (k, e)
        assert e < TOL_GRAD, f"{k}: grad err {e:.2e} (scale {scale:.2e})"
    print(f"[{name}] worst grad err {worst[1]:.2e} at {worst[0]}")
    # decoder self-attention q/k projections are dead (softmax over one key): exactly zero
    E = c["E"]
    w = gv["transformer.decoder.layers.0.self_attn.in_proj_weight"]
    assert float(w[:2 * E].abs().max()) == 0.0 and float(w[2 * E:].abs().max()) > 0.0
    # embedding has no padding_idx: the <pad> row trains
    assert float(gv["src_embedding.weight"][1].abs().max()) > 0.0


@pytest.mark.parametrize("name", ["tiny", "cfg1", "cfg2", "e1024", "cfg5"])
def test_train_steps_vs_golden(name):
    """Reference training trajectories (tools/gen_golden.py::train_steps): 5 steps for tiny / cfg1 / cfg2 (the headline
    config), one step at the E1024 and configs[4] shapes -- per-tensor gradients of step 0, loss and pre-clip gradient
    norm of every step, every weight tensor after the last step."""
    g, c, sd, X, L, y = gold.tf_case(name)
    assert len(g["losses"]) == (5 if name in ("tiny", "cfg1", "cfg2") else 1)
    eng = make_engine(c, sd)
    eng.set_lr(0.01)
    Xc, yc = X.cuda(), y.cuda()
    eng.forward(Xc, yc, train=True)
    eng.backward()
    gold.check_summary(g, "grad0", {k: v.cpu() for k, v in eng.views(eng.grads).items()},
                       TOL_GRAD if c["N"] <= 2 else 5e-2)
    for s in range(len(g["losses"])):
        eng.train_step(Xc, yc, momentum=0.9, max_norm=0.5)
        torch.cuda.synchronize()
        print(f"[{name}] step {s}: loss {eng.loss:.6f} (ref {g['losses'][s]:.6f}) norm {eng.grad_norm:.5f} (ref {g['grad_norms'][s]:.5f})")
        assert abs(eng.loss - g["losses"][s]) < TOL * g["losses"][s]
        assert abs(eng.grad_norm - g["grad_norms"][s]) < (TOL_NORM0 if s == 0 else TOL_NORM) * g["grad_norms"][s]
    gold.check_summary(g, "wfinal", {k: v.cpu() for k, v in eng.views().items()}, TOL)


def test_a_plan_keeps_the_backward_passes_it_was_created_with():
    """slnlp_set_backward_passes is the default for plans created AFTERWARDS: a plan copies the counts at creation, so changing the
    default between (or during) its steps does not change its products -- eager steps, a captured graph and other threads' plans
    cannot end up with different arithmetic inside one fit."""
    import ctypes as C
    from slnlp._lib import load, check
    g, c, sd, X, L, y = gold.tf_case("cfg1")
    Xc, yc = X.cuda(), y.cuda()
    w, d = C.c_int32(0), C.c_int32(0)
    load().slnlp_get_backward_passes(C.byref(w), C.byref(d))
    saved = (int(w.value), int(d.value))
    try:
        check(load().slnlp_set_backward_passes(2, 2), "set_backward_passes")
        e1, e2 = make_engine(c, sd), make_engine(c, sd)
        e1.set_lr(0.01); e2.set_lr(0.01)
        e2.train_step(Xc, yc); e2.train_step(Xc, yc)
        check(load().slnlp_set_backward_passes(3, 3), "set_backward_passes")      # ... while e1 is between / inside its steps
        e3 = make_engine(c, sd)
        e3.set_lr(0.01)
        e1.train_step(Xc, yc); e1.train_step(Xc, yc)
        e3.train_step(Xc, yc); e3.train_step(Xc, yc)
        torch.cuda.synchronize()
        assert torch.equal(e1.params, e2.params)                   # e1 kept (2, 2)
        assert not torch.equal(e1.params, e3.params)               # a plan created under (3, 3) is another arithmetic
    finally:
        load().slnlp_set_backward_passes(*saved)


def test_graph_replay_equals_eager():
    g, c, sd, X, L, y = gold.tf_case("cfg1")
    Xc, yc = X.cuda(), y.cuda()
    e1, e2 = make_engine(c, sd), make_engine(c, sd)
    e1.set_lr(0.01); e2.set_lr(0.01)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(4):
            e1.train_step(Xc, yc)
            e2.train_step_graph(Xc, yc)
    torch.cuda.synchronize()
    assert torch.equal(e1.params, e2.params)                  # same kernels, same order -> bitwise equal
    assert e1.loss == e2.loss and int(e2.rng[1]) == 4


@pytest.mark.parametrize("name,reps", [("cfg1", 2), ("cfg2", 8)])
def test_train_is_deterministic(name, reps):
    """Identical fresh engines, identical steps -> bit-identical weights, every time.  (cfg2 x 8: in round 1 the backward
    forked work to side streams and 7 of 16 such steps differed at cfg2; the forks were deleted in round 3)"""
    g, c, sd, X, L, y = gold.tf_case(name)
    outs = []
    st = torch.cuda.Stream()
    for _ in range(reps):
        eng = make_engine(c, sd, dropout=0.1, seed=7)
        eng.set_lr(0.05)
        with torch.cuda.stream(st):
            for _ in range(3):
                eng.train_step(X.cuda(), y.cuda())
        torch.cuda.synchronize()
        outs.append(eng.params.clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])


def _dump_masks(eng, c, p):
    """Materialise every dropout mask the engine uses this step, keyed like the oracle."""
    from slnlp import ops
    B, S, E, F, H, N = c["B"], c["S"], c["E"], c["F"], c["H"], c["N"]
    M = B * S
    mk = lambda R, C_, site: ops.dropout_mask(R, C_, p, site, eng.rng).cpu()
    masks = {"src_pos_encoding.dropout": mk(M, E, 1).view(S, B, E),
             "tgt_pos_encoding.dropout": mk(B, E, 2).view(1, B, E)}
    for l in range(N):
        pre, base = f"transformer.encoder.layers.{l}.", 16 + 8 * l
        masks[pre + "self_attn.attn"] = mk(B * H * S, S, base).view(B, H, S, S)
        masks[pre + "dropout1"] = mk(M, E, base + 1).view(S, B, E)
        masks[pre + "dropout"] = mk(M, F, base + 2).view(S, B, F)
        masks[pre + "dropout2"] = mk(M, E, base + 3).view(S, B, E)
        pre, base = f"transformer.decoder.layers.{l}.", 16 + 8 * (N + l)
        masks[pre + "self_attn.attn"] = mk(B * H, 1, base).view(B, H, 1, 1)
        masks[pre + "dropout1"] = mk(B, E, base + 1).view(1, B, E)
        masks[pre + "multihead_attn.attn"] = mk(B * H, S, base + 2).view(B, H, 1, S)
        masks[pre + "dropout2"] = mk(B, E, base + 3).view(1, B, E)
        masks[pre + "dropout"] = mk(B, F, base + 4).view(1, B, F)
        masks[pre + "dropout3"] = mk(B, E, base + 5).view(1, B, E)
    return masks


@pytest.mark.parametrize("name", ["tiny", "cfg1"])
def test_dropout_path_vs_oracle_with_same_masks(name):
    """dropout > 0: the oracle is fed the very masks the GPU generates
    (slnlp_dropout_mask), so forward, loss and every gradient must agree."""
    from oracle import train_ref, transformer_ref as tr
    g, c, sd, X, L, y = gold.tf_case(name)
    p = 0.1 if name == "cfg1" else 0.3
    eng = make_engine(c, sd, dropout=p, seed=11)
    masks = _dump_masks(eng, c, p)
    logp = eng.forward(X.cuda(), y.cuda(), train=True).cpu().clone()
    eng.backward()
    torch.cuda.synchronize()
    fwd = lambda pr, X, y, L: tr.forward(pr, X, y, num_heads=c["H"], num_layers=c["N"], p_drop=p, masks=masks)
    loss_o, logp_o, grads_o = train_ref.Trainer(sd, fwd, pad_tgt=1).loss_and_grads(X, y, L)
    assert gold.rel_err(logp.numpy(), logp_o.numpy()) < TOL
    assert abs(eng.loss - float(loss_o)) < TOL * float(loss_o)
    gv = {k: v.cpu() for k, v in eng.views(eng.grads).items()}
    for k, go in grads_o.items():
        scale = float(go.abs().max())
        e = float((gv[k] - go).abs().max()) / max(scale, 1e-12) if scale > 0 else float(gv[k].abs().max())
        assert e < TOL_GRAD, f"{k}: grad err {e:.2e}"
    # eval mode ignores dropout
    ev = eng.forward(X.cuda(), y.cuda(), train=False).cpu()
    assert gold.rel_err(ev.numpy(), g["logp"]) < TOL


def test_precision8_fp8_forward_products():
    """precision=8 ("fp8 MFMA weights", BASELINE.json configs[4]): every S*B-row forward product on the fp8 MFMA (e4m3
    activations, e4m3 weights with per-row scales), backward in split-bf16 from the fp32 master weights.  Outside the 1e-3
    parity bar by construction -- the error is reported and bounded loosely; training still descends and stays deterministic."""
    from oracle import transformer_ref as tr
    from slnlp import tf_engine as te
    c = dict(E=128, H=4, N=2, F=256, Vs=300, Vt=40, B=20, S=24)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(tr.param_shapes(c["E"], c["H"], c["N"], c["F"], c["Vs"], c["Vt"]), seed=3).items()}
    X, L, y = [torch.from_numpy(a) for a in synth.make_batch(c["B"], c["S"], c["Vs"], c["Vt"], seed=5, min_len=5)]
    lo = tr.forward(sd, X, y, num_heads=c["H"], num_layers=c["N"])
    outs = {}
    for prec in (3, 8):
        eng = make_engine(c, sd, precision=prec)
        outs[prec] = eng.forward(X.cuda(), y.cuda()).cpu().clone()
    e3, e8 = gold.rel_err(outs[3].numpy(), lo.numpy()), gold.rel_err(outs[8].numpy(), lo.numpy())
    agree = float((outs[8].argmax(-1) == lo.argmax(-1)).float().mean())
    print(f"log-prob rel err: split-bf16 {e3:.2e}, fp8 forward {e8:.2e}; fp8 arg-max agreement {agree:.2f}")
    assert e3 < 1e-3 and 1e-3 < e8 < 0.15
    losses = []
    for rep in range(2):
        eng = make_engine(c, sd, dropout=0.1, precision=8, seed=4)
        eng.set_lr(0.05)
        ls = []
        for _ in range(6):
            eng.train_step(X.cuda(), y.cuda(), 0.9, 0.5)
            ls.append(eng.loss)
        losses.append(ls)
    assert losses[0] == losses[1] and losses[0][-1] < losses[0][0]
    with pytest.raises(RuntimeError, match="multiples of 128"):
        make_engine(dict(c, F=192), sd, precision=8)
