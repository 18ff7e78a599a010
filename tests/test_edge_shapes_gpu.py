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
    # beyond one 64-key tile: the wave-per-row attention kernels (attention_long.hip); the reference's only limit is its
    # 5000-row positional table (positional_encoding.py:23)
    "len_65": (64, 4, 2, 128, 97, 23, 5, 65, 3),
    "len_128_head_dim_128": (256, 2, 1, 128, 60, 13, 3, 128, 40),
    "len_200_planes": (128, 4, 1, 128, 97, 23, 2, 200, 150),           # E, F multiples of 64: the S*B-row GEMMs on bf16 planes
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
    "len_100": (32, 32, 1, 80, 21, 4, 100, "ragged"),                    # Bahdanau softmax over more than one wave of positions
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


@pytest.mark.parametrize("case,p", [("len_65", 0.25), ("len_128_head_dim_128", 0.1)])
def test_long_sequence_dropout_path_vs_oracle_with_same_masks(case, p):
    """S > 64 with dropout: the oracle is fed the very masks the GPU generates (slnlp_dropout_mask) -- the long kernels use the
    same (site, row, column) convention as the MFMA kernels -- so forward, loss and every gradient must agree."""
    from oracle import train_ref, transformer_ref as tr
    from slnlp import tf_engine as te
    from test_transformer_gpu import _dump_masks
    E, H, N, F, Vs, Vt, B, S, min_len = TF_CASES[case]
    c = dict(E=E, H=H, N=N, F=F, Vs=Vs, Vt=Vt, B=B, S=S)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(tr.param_shapes(E, H, N, F, Vs, Vt), seed=3).items()}
    X, L, y = [torch.from_numpy(a) for a in synth.make_batch(B, S, Vs, Vt, seed=5, min_len=min_len)]
    eng = te.TransformerEngine(te.make_config(E, H, N, F, Vs, Vt, B, S, 1, 1, p, 3), seed=11)
    eng.load_state(sd)
    masks = _dump_masks(eng, c, p)
    logp = eng.forward(X.cuda(), y.cuda(), train=True).cpu().clone()
    eng.backward()
    torch.cuda.synchronize()
    fwd = lambda pr, X, y, L: tr.forward(pr, X, y, num_heads=H, num_layers=N, p_drop=p, masks=masks)
    loss_o, logp_o, grads_o = train_ref.Trainer(sd, fwd, pad_tgt=1).loss_and_grads(X, y, L)
    assert gold.rel_err(logp.numpy(), logp_o.numpy()) < 1e-3
    assert abs(eng.loss - float(loss_o)) < 1e-3 * float(loss_o)
    gv = {k: v.cpu() for k, v in eng.views(eng.grads).items()}
    for k, go in grads_o.items():
        scale = float(go.abs().max())
        e = float((gv[k] - go).abs().max()) / max(scale, 1e-12) if scale > 0 else float(gv[k].abs().max())
        assert e < 2e-2, f"{k}: grad err {e:.2e}"


def test_long_sequence_through_the_estimator_and_lockstep():
    """A corpus padded to 80 frames: the drop-in module, the fused fit loop and a lockstep group all run (and agree)."""
    from slnlp.data import synthetic_dataset
    from slnlp.lockstep import fit_lockstep
    from test_lockstep_gpu import _net
    ds = synthetic_dataset(60, seq_len=80, src_vocab=64, n_labels=4, seed=5, min_len=30)
    nets = []
    for mode in ("solo", "lockstep"):
        torch.manual_seed(3)
        nets.append(_net(ds, max_epochs=2, early_stopping=None, lr_scheduler=None).initialize())
    nets[0].partial_fit(ds)
    fit_lockstep([nets[1]], [ds])
    strip = lambda h: [{k: v for k, v in row.items() if k != "dur"} for row in h]
    assert strip(nets[0].history) == strip(nets[1].history)
    assert np.isfinite(nets[0].history[-1]["valid_loss"]) and nets[0].predict(ds).shape == (60,)
