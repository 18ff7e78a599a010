"""Edge shapes of the whole path against the CPU oracle on the same seeded inputs (maximum and
minimum sizes, ragged and uniform lengths, vocabularies that are not multiples of the vector width).  For each
case: eval forward (log-probs within 1e-3 rel, arg-max identical) and two fused train steps (loss and pre-clip
gradient norm within 1e-3 / 2e-3 rel) through slnlp_tf_* / slnlp_rnn_*."""
import numpy as np
import pytest
import torch

import gold
from slnlp import synth

pytestmark = pytest.mark.gpu

TF_CASES = {
    # name: (E, H, N, F, Vs, Vt, B, S, min_len)
    "max_len_64": (64, 4, 2, 128, 97, 23, 6, 64, 3),
    "len_1": (32, 2, 1, 64, 50, 11, 5, 1, 1),
    "batch_1": (32, 4, 2, 64, 64, 16, 1, 12, 3),
    "odd_vocab_head_dim_4": (32, 8, 1, 36, 101, 203, 7, 9, 2),          # Vt % 4 != 0, F % 64 != 0 (fp32-operand GEMMs), dh = 4
    "one_head_dim_64": (64, 1, 1, 64, 40, 9, 3, 17, 17),                # single head, every sequence at full length
    "head_dim_128": (256, 2, 1, 64, 40, 9, 4, 10, 2),
}


def _tf(case):
    E, H, N, F, Vs, Vt, B, S, min_len = TF_CASES[case]
    from oracle import train_ref, transformer_ref as tr
    from slnlp import tf_engine as te
    sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(tr.param_shapes(E, H, N, F, Vs, Vt), seed=3).items()}
    X, L, y = [torch.from_numpy(a) for a in synth.make_batch(B, S, Vs, Vt, seed=5, min_len=min_len)]
    eng = te.TransformerEngine(te.make_config(E, H, N, F, Vs, Vt, B, S))
    eng.load_state(sd)
    eng.set_lr(0.01)
    fwd = lambda p, X, y, L: tr.forward(p, X, y, num_heads=H, num_layers=N)
    return eng, train_ref.Trainer(sd, fwd, pad_tgt=1, lr=0.01, momentum=0.9, max_norm=0.5), fwd, sd, X, L, y


@pytest.mark.parametrize("case", sorted(TF_CASES))
def test_transformer_edge_shape(case):
    eng, trn, fwd, sd, X, L, y = _tf(case)
    lo = fwd(sd, X, y, L)
    lp = eng.forward(X.cuda(), y.cuda()).cpu()
    assert gold.rel_err(lp.numpy(), lo.numpy()) < 1e-3
    assert torch.equal(lp.argmax(-1), lo.argmax(-1))
    for step in range(2):
        eng.train_step(X.cuda(), y.cuda(), 0.9, 0.5)
        torch.cuda.synchronize()
        loss_o, norm_o, _ = trn.step(X, y, L)
        assert abs(eng.loss - float(loss_o)) < 1e-3 * abs(float(loss_o)), (case, step, eng.loss, float(loss_o))
        assert abs(eng.grad_norm - float(norm_o)) < 2e-3 * float(norm_o), (case, step, eng.grad_norm, float(norm_o))


RNN_CASES = {
    # name: (E, Hd, N, Vs, Vt, B, S, lengths)     lengths: "ragged" | "full" | "ones"
    "max_len_64": (32, 32, 2, 80, 21, 5, 64, "ragged"),
    "len_1": (16, 24, 1, 30, 9, 4, 1, "full"),
    "batch_1": (32, 64, 2, 64, 16, 1, 12, "ragged"),
    "all_lengths_one": (24, 40, 2, 50, 13, 6, 9, "ones"),                # every sequence is a single token: the backward
    "all_full_length": (64, 64, 1, 50, 203, 3, 11, "full"),             # direction's first processed step is its only one
}


@pytest.mark.parametrize("rnn_type", ["lstm", "gru"])
@pytest.mark.parametrize("case", sorted(RNN_CASES))
def test_rnn_edge_shape(rnn_type, case):
    E, Hd, N, Vs, Vt, B, S, how = RNN_CASES[case]
    from oracle import rnn_ref as rr, train_ref
    from slnlp import rnn_engine as re_
    sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(rr.param_shapes(rnn_type, E, Hd, N, Vs, Vt), seed=3).items()}
    X, L, y = [torch.from_numpy(a) for a in synth.make_batch(B, S, Vs, Vt, seed=5, min_len=1 if how != "full" else S)]
    if how == "ones":
        L = torch.ones_like(L)
        X[:, 1:] = 1                                                      # <pad> beyond the first token
    eng = re_.RnnEngine(re_.make_config(rnn_type, E, Hd, N, Vs, Vt, B, S, 1, 1, 0, 0.0, 3))
    eng.load_state(sd)
    eng.set_lr(0.01)
    fwd = lambda p, X, y, L: rr.forward(p, X, y, L, rnn_type=rnn_type, num_layers=N)
    trn = train_ref.Trainer(sd, fwd, pad_tgt=1, lr=0.01, momentum=0.9, max_norm=0.5, frozen=("model.decoder.pre_output_layer.weight",))
    lo = fwd(sd, X, y, L)
    lp = eng.forward(X.cuda(), y.cuda(), L.cuda()).cpu()
    assert gold.rel_err(lp.numpy(), lo.numpy()) < 1e-3
    assert torch.equal(lp.argmax(-1), lo.argmax(-1))
    for step in range(2):
        eng.train_step(X.cuda(), y.cuda(), L.cuda(), 0.9, 0.5)
        torch.cuda.synchronize()
        loss_o, norm_o, _ = trn.step(X, y, L)
        assert abs(eng.loss - float(loss_o)) < 1e-3 * abs(float(loss_o)), (case, step, eng.loss, float(loss_o))
        assert abs(eng.grad_norm - float(norm_o)) < 2e-3 * float(norm_o), (case, step, eng.grad_norm, float(norm_o))
