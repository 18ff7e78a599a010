"""CPU: the oracle (oracle/) against golden vectors captured from the reference
(tools/gen_golden.py).  This is the pin that makes the oracle trustworthy."""
import numpy as np
import pytest
import torch

import gold
from oracle import rnn_ref, train_ref, transformer_ref as tr

TOL_FWD = 2e-5   # fp32 restatement vs reference fp32 (different op order)
TOL_TRAIN = 2e-4


def test_masks_and_pe():
    g = gold.load("masks_pe")
    assert np.array_equal(tr.causal_mask(1).numpy(), g["mask1"])
    assert np.array_equal(tr.causal_mask(48).numpy(), g["mask48"])
    X, _, _ = __import__("slnlp.synth", fromlist=["x"]).make_batch(50, 48, 3000, 202, seed=1)
    src = torch.from_numpy(X).transpose(0, 1)
    assert np.array_equal(tr.padding_mask(src, 1).numpy(), g["padmask"])
    rows = g["pe_rows"]
    for E in (128, 512, 1024):
        pe = tr.positional_table(64, E).numpy()
        ref = g[f"pe{E}"]
        got = pe if E == 128 else pe[rows]
        assert np.array_equal(got, ref), f"pe{E} not bit-identical"


@pytest.mark.parametrize("name", ["tiny", "cfg1", "cfg2", "e1024", "cfg5"])
def test_transformer_forward(name):
    g, c, sd, X, L, y = gold.tf_case(name)
    taps = {}
    logp = tr.forward(sd, X, y, num_heads=c["H"], num_layers=c["N"], taps=taps)
    assert gold.rel_err(logp.numpy(), g["logp"]) < TOL_FWD
    assert np.array_equal(logp.argmax(-1).numpy(), g["argmax"])
    for k in [k for k in g if k.startswith("tap_")]:
        assert gold.rel_err(taps[k[4:]].numpy(), g[k]) < TOL_FWD, k


def test_transformer_behaviour_pins():
    g, c, sd, X, L, y = gold.tf_case("tiny")
    base = tr.forward(sd, X, y, num_heads=c["H"], num_layers=c["N"])
    y2 = y.clone(); y2[1] = (y2[1] - 2 + 1) % (c["Vt"] - 2) + 2
    out = tr.forward(sd, X, y2, num_heads=c["H"], num_layers=c["N"])
    assert gold.rel_err(out.numpy(), g["pin_y_changed"]) < TOL_FWD
    changed = (out - base).abs().amax(dim=1)
    assert changed[1] > 1e-3 and changed[[0, 2, 3]].max() == 0      # row-local y dependence
    X2 = X.clone(); X2[:, 5] = (X2[:, 5] - 2 + 7) % (c["Vs"] - 2) + 2
    t1, t2 = {}, {}
    tr.forward(sd, X, y, num_heads=c["H"], num_layers=c["N"], taps=t1)
    tr.forward(sd, X2, y, num_heads=c["H"], num_layers=c["N"], taps=t2)
    last = f"enc{c['N'] - 1}"
    assert gold.rel_err(t2[last].numpy(), g["pin_causal_enc_last"]) < TOL_FWD
    assert torch.equal(t1[last][:5], t2[last][:5])                   # encoder is causal
    assert not torch.equal(t1[last][5:], t2[last][5:])


def _tf_trainer(c, sd):
    fwd = lambda p, X, y, L: tr.forward(p, X, y, num_heads=c["H"], num_layers=c["N"])
    return train_ref.Trainer(sd, fwd, pad_tgt=1, lr=0.01, momentum=0.9, max_norm=0.5)


@pytest.mark.parametrize("name", ["tiny", "cfg1", "cfg2"])
def test_transformer_train(name):
    g, c, sd, X, L, y = gold.tf_case(name)
    trn = _tf_trainer(c, sd)
    _, _, grads = trn.loss_and_grads(X, y, L)
    gold.check_summary(g, "grad0", grads, TOL_TRAIN)
    for s in range(len(g["losses"])):
        loss, total, _ = trn.step(X, y, L)
        assert abs(float(loss) - g["losses"][s]) / g["losses"][s] < TOL_TRAIN, s
        assert abs(float(total) - g["grad_norms"][s]) / g["grad_norms"][s] < TOL_TRAIN, s
    gold.check_summary(g, "wfinal", trn.sd, TOL_TRAIN)


RNN = [("lstm", "tiny"), ("lstm", "mid"), ("lstm", "cfg3"), ("gru", "tiny"), ("gru", "mid")]


@pytest.mark.parametrize("rnn_type,name", RNN)
def test_rnn_forward(rnn_type, name):
    g, c, sd, X, L, y = gold.rnn_case(rnn_type, name)
    taps = {}
    logp = rnn_ref.forward(sd, X, y, L, rnn_type=rnn_type, num_layers=c["N"], taps=taps)
    assert gold.rel_err(logp.numpy(), g["logp"]) < TOL_FWD
    assert np.array_equal(logp.argmax(-1).numpy(), g["argmax"])
    if name == "tiny":
        for k in ("enc_out", "enc_final", "alphas", "context"):
            assert gold.rel_err(taps[k].numpy(), g["tap_" + k]) < TOL_FWD, k
        y2 = (y - 2 + 3) % (c["Vt"] - 2) + 2
        out = rnn_ref.forward(sd, X, y2, L, rnn_type=rnn_type, num_layers=c["N"])
        assert torch.equal(out, logp)                                # independent of y
        assert gold.rel_err(out.numpy(), g["pin_y_changed"]) < TOL_FWD
    else:
        assert gold.rel_err(taps["enc_final"][:, :8].numpy(), g["tap_enc_final"]) < TOL_FWD
        assert gold.rel_err(taps["alphas"].numpy(), g["tap_alphas"]) < TOL_FWD


@pytest.mark.parametrize("rnn_type,name", [r for r in RNN if r[1] != "cfg3"])
def test_rnn_train(rnn_type, name):
    g, c, sd, X, L, y = gold.rnn_case(rnn_type, name)
    fwd = lambda p, X, y, L: rnn_ref.forward(p, X, y, L, rnn_type=rnn_type, num_layers=c["N"])
    trn = train_ref.Trainer(sd, fwd, pad_tgt=1, lr=0.01, momentum=0.9, max_norm=0.5,
                            frozen=("model.decoder.pre_output_layer.weight",))
    _, _, grads = trn.loss_and_grads(X, y, L)
    gold.check_summary(g, "grad0", grads, TOL_TRAIN)
    # G7: src pad row gets no grad (padding_idx), trg_embed only the <bos>(=0) row
    assert float(grads["model.src_embed.weight"][1].abs().max()) == 0
    assert float(grads["model.trg_embed.weight"][1:].abs().max()) == 0
    for s in range(len(g["losses"])):
        loss, total, _ = trn.step(X, y, L)
        assert abs(float(loss) - g["losses"][s]) / g["losses"][s] < TOL_TRAIN, s
        assert abs(float(total) - g["grad_norms"][s]) / g["grad_norms"][s] < TOL_TRAIN, s
    gold.check_summary(g, "wfinal", trn.sd, TOL_TRAIN)
