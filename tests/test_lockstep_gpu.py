"""GPU: K fits of one shape through ONE launch sequence (slnlp_tf_lockstep_*, slnlp/lockstep.py).

The contract is strict: block (x, y, z) of a merged launch does for fit z exactly what block (x, y) does in that fit's own
launch, so weights, losses, log-probs, histories and cv scores are BIT-identical to running each fit alone."""
import copy

import numpy as np
import pytest
import torch

import gold

pytestmark = pytest.mark.gpu


def _engines(c, K, dropouts, B=None):
    from oracle import transformer_ref as tr
    from slnlp import synth, tf_engine as te
    out = []
    for f in range(K):
        cfg = te.make_config(c["E"], c["H"], c["N"], c["F"], c["Vs"], c["Vt"], B or c["B"], c["S"], 1, 1, dropouts[f], 3)
        sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(tr.param_shapes(c["E"], c["H"], c["N"], c["F"], c["Vs"], c["Vt"]), seed=10 + f).items()}
        e = te.TransformerEngine(cfg, seed=100 + f)
        e.load_state(sd)
        e.set_lr(0.01 * (f + 1))
        out.append(e)
    return out


@pytest.mark.parametrize("name,dropouts", [("tiny", (0.0, 0.0, 0.0)), ("tiny", (0.1, 0.3, 0.2)), ("cfg1", (0.1, 0.5, 0.1, 0.3)),
                                            ("cfg2", (0.1, 0.1)), ("cfg2", (0.1, 0.3, 0.2, 0.1))])
def test_lockstep_steps_are_bit_identical_to_solo_steps(name, dropouts):
    """Engines with their own weights / lr / dropout rate / seed and their own data: train steps (full and ragged last
    batch) and an eval pass in lockstep == the same calls on each engine alone."""
    from slnlp import synth
    from slnlp.lockstep import LockstepGroup
    g, c, sd, X, L, y = gold.tf_case(name)
    K, B, S = len(dropouts), c["B"], c["S"]
    rows = 2 * B + max(1, B // 3)                                  # two full batches + a ragged one
    data = []
    for f in range(K):
        Xn, _, yn = synth.make_batch(rows, S, c["Vs"], c["Vt"], seed=50 + f, min_len=c["min_len"])
        data.append((torch.from_numpy(Xn).cuda(), torch.from_numpy(yn).cuda()))
    solo, lock = _engines(c, K, dropouts), _engines(c, K, dropouts)
    # ---- solo reference: per engine, 3 train steps then an eval forward over all rows
    want = []
    for e, (Xd, yd) in zip(solo, data):
        losses, logps = [], []
        for r in range(0, rows, B):
            lp = e.train_step(Xd[r:r + B], yd[r:r + B], 0.9, 0.5).clone()
            losses.append(e.scalars[0].clone()); logps.append(lp)
        ev = torch.cat([e.forward(Xd[r:r + B], yd[r:r + B]).clone() for r in range(0, rows, B)])
        torch.cuda.synchronize()
        want.append((torch.stack(losses).cpu(), torch.cat(logps).cpu(), ev.cpu(), e.params.clone().cpu(), e.momentum.clone().cpu(), int(e.rng[1])))
    # ---- the same in lockstep
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        grp = LockstepGroup(lock)
        grp.set_data(0, [d[0] for d in data], [d[1] for d in data], B)
        grp.set_data(1, [d[0] for d in data], [d[1] for d in data], B)
        grp.epoch(0, B, True, 0.9, 0.5)
        torch.cuda.synchronize()
        train_out = [(grp.loss[0][f].clone().cpu(), grp.logp[0][f].clone().cpu()) for f in range(K)]
        grp.epoch(1, B, False)
        torch.cuda.synchronize()
    n_train, n_eval = grp.num_launches(0, B, True), grp.num_launches(1, B, False)
    print(f"[{name} K={K}] launches per lockstep step: train {n_train}, eval {n_eval} (for all {K} fits)")
    assert 0 < n_eval < n_train
    for f in range(K):
        losses, logps, ev, params, mom, step = want[f]
        assert torch.equal(train_out[f][0], losses), f
        assert torch.equal(train_out[f][1], logps), f
        assert torch.equal(grp.logp[1][f].cpu(), ev), f
        assert torch.equal(lock[f].params.cpu(), params) and torch.equal(lock[f].momentum.cpu(), mom), f
        assert int(lock[f].rng[1]) == step == 3
    grp.close()
    # the plans are usable on their own again
    lp = lock[0].forward(data[0][0][:B], data[0][1][:B]).cpu()
    assert torch.equal(lp, solo[0].forward(data[0][0][:B], data[0][1][:B]).cpu())


def _net(ds, **kw):
    from slnlp.net import NeuralNetClassifier
    args = dict(module="model.Transformer", module__dropout=0.1, module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y,
                module__batch_first=True, module__embedding_size=32, module__num_heads=4, module__num_layers=2, module__hidden_size=64,
                criterion="torch.nn.CrossEntropyLoss", criterion__ignore_index=1, optimizer="torch.optim.SGD", optimizer__momentum=0.9,
                optimizer__nesterov=False, lr=0.05, max_epochs=6, batch_size=20, device="cuda", gradient_clipping={"gradient_clip_value": 0.5},
                scoring=["neg_log_loss", "accuracy", "f1_weighted"], use_graph=False,
                lr_scheduler={"policy": "ReduceLROnPlateau", "factor": 0.2, "patience": 1},
                early_stopping={"patience": 2, "threshold": 1e-4, "threshold_mode": "rel"})
    args.update(kw)
    return NeuralNetClassifier(**args)


def test_fit_lockstep_equals_one_fit_at_a_time():
    """fit_lockstep == partial_fit per estimator: identical histories (every metric, lr, batch rows), identical weights; one
    fit stops early (lr ~ 0 -> no improvement) and leaves the group while the others go on."""
    from slnlp.data import synthetic_dataset
    from slnlp.lockstep import fit_lockstep
    ds = synthetic_dataset(150, seq_len=12, src_vocab=64, n_labels=6, seed=5, min_len=3)
    parts = [ds[np.arange(0, 130)], ds[np.arange(10, 140)], ds[np.arange(20, 150)]]      # one fold-like subset per fit
    variants = [dict(lr=0.05, module__dropout=0.1), dict(lr=1e-7, module__dropout=0.3), dict(lr=0.02, module__dropout=0.1)]

    def build():
        nets = []
        for i, kw in enumerate(variants):
            torch.manual_seed(40 + i)
            nets.append(_net(ds, **kw).initialize())
        return nets
    seq = build()
    for n, d in zip(seq, parts):
        n.partial_fit(d)
    lock = build()
    fit_lockstep(lock, parts)
    strip = lambda h: [{k: v for k, v in row.items() if k != "dur"} for row in h]
    assert len(seq[1].history) < 6 and len(seq[0].history) == 6          # the lr ~ 0 fit stopped early, alone
    for a, b in zip(seq, lock):
        assert strip(a.history) == strip(b.history)
        sa, sb = a.module_.state_dict(), b.module_.state_dict()
        assert all(torch.equal(sa[k], sb[k]) for k in sa)
        assert np.array_equal(a.predict_proba(parts[0]), b.predict_proba(parts[0]))


def test_sharded_grid_lockstep_equals_sequential():
    """ShardedGridSearchCV(lockstep=k): work units of shape-compatible tasks, cv_results_ bit-identical to one fit at a time."""
    from slnlp.data import synthetic_dataset
    from slnlp.grid import ShardedGridSearchCV
    ds = synthetic_dataset(100, seq_len=10, src_vocab=50, n_labels=3, seed=9, min_len=3)
    grid = {"lr": [0.1, 0.01], "module__dropout": [0.1, 0.4], "module__num_layers": [1, 2]}
    factory = lambda: _net(ds, max_epochs=2, early_stopping=None, lr_scheduler=None, scoring=["neg_log_loss"])
    res = {}
    for k in (1, 5):
        gs = ShardedGridSearchCV(factory, grid, cv=5, refit=False, device="cuda:0", lockstep=k).fit(ds)
        res[k] = gs
    assert res[5].n_units_ < res[1].n_units_ == res[1].n_tasks_ == 40
    for key in ("mean_test_score", "std_test_score", "split0_test_score", "split4_test_score"):
        assert np.array_equal(res[1].cv_results_[key], res[5].cv_results_[key]), key
    assert res[1].best_index_ == res[5].best_index_ and np.isfinite(res[5].best_score_)


def test_lockstep_rejects_mismatched_plans():
    from slnlp.lockstep import LockstepGroup
    g, c, sd, X, L, y = gold.tf_case("tiny")
    a = _engines(c, 1, (0.0,))[0]
    c2 = dict(c, F=32)
    b = _engines(c2, 1, (0.0,))[0]
    with pytest.raises(RuntimeError, match="shape of plan 0"):
        LockstepGroup([a, b])
    d = _engines(c, 1, (0.2,))[0]
    with pytest.raises(RuntimeError, match="dropout on/off"):
        LockstepGroup([a, d])
    with pytest.raises(RuntimeError, match="listed twice"):
        LockstepGroup([a, a])


def _rnn_engines(c, K, dropouts):
    from oracle import rnn_ref as rr
    from slnlp import synth, rnn_engine as re_
    out = []
    for f in range(K):
        cfg = re_.make_config(c["rnn_type"], c["E"], c["Hd"], c["N"], c["Vs"], c["Vt"], c["B"], c["S"], 1, 1, 0, dropouts[f], 3)
        sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(rr.param_shapes(c["rnn_type"], c["E"], c["Hd"], c["N"], c["Vs"], c["Vt"]), seed=20 + f).items()}
        e = re_.RnnEngine(cfg, seed=200 + f)
        e.load_state(sd)
        e.set_lr(0.02 * (f + 1))
        out.append(e)
    return out


@pytest.mark.parametrize("rnn_type,name,dropouts", [("lstm", "tiny", (0.0, 0.0, 0.0)), ("gru", "tiny", (0.2, 0.1, 0.4)), ("lstm", "mid", (0.1, 0.3)),
                                                     ("gru", "mid", (0.1, 0.3)), ("lstm", "cfg3", (0.1, 0.1)),
                                                     ("gru", "cfg3", (0.1, 0.3, 0.2, 0.1))])
def test_rnn_lockstep_steps_are_bit_identical_to_solo_steps(rnn_type, name, dropouts):
    """slnlp_rnn_lockstep_*: LSTM / GRU encoder-decoder fits with their own weights, lr, dropout rate, seed, data and sequence
    lengths -- train steps (full and ragged last batch) and an eval pass in lockstep == the same calls on each engine alone."""
    from slnlp import synth
    from slnlp.lockstep import LockstepGroup
    g, c, sd, X, L, y = gold.rnn_case(rnn_type, name)
    K, B, S = len(dropouts), c["B"], c["S"]
    rows = 2 * B + max(1, B // 3)
    data = []
    for f in range(K):
        Xn, Ln, yn = synth.make_batch(rows, S, c["Vs"], c["Vt"], seed=70 + f, min_len=c["min_len"])
        data.append((torch.from_numpy(Xn).cuda(), torch.from_numpy(Ln).cuda(), torch.from_numpy(yn).cuda()))
    solo, lock = _rnn_engines(c, K, dropouts), _rnn_engines(c, K, dropouts)
    want = []
    for e, (Xd, Ld, yd) in zip(solo, data):
        losses, logps = [], []
        for r in range(0, rows, B):
            lp = e.train_step(Xd[r:r + B], yd[r:r + B], Ld[r:r + B], 0.9, 0.5).clone()
            losses.append(e.scalars[0].clone()); logps.append(lp)
        ev = torch.cat([e.forward(Xd[r:r + B], yd[r:r + B], Ld[r:r + B]).clone() for r in range(0, rows, B)])
        torch.cuda.synchronize()
        want.append((torch.stack(losses).cpu(), torch.cat(logps).cpu(), ev.cpu(), e.params.clone().cpu(), e.momentum.clone().cpu(), int(e.rng[1])))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        grp = LockstepGroup(lock)
        assert grp.kind == "rnn"
        for slot in (0, 1):
            grp.set_data(slot, [d[0] for d in data], [d[2] for d in data], B, [d[1] for d in data])
        grp.epoch(0, B, True, 0.9, 0.5)
        torch.cuda.synchronize()
        train_out = [(grp.loss[0][f].clone().cpu(), grp.logp[0][f].clone().cpu()) for f in range(K)]
        grp.epoch(1, B, False)
        torch.cuda.synchronize()
    n_train, n_eval = grp.num_launches(0, B, True), grp.num_launches(1, B, False)
    print(f"[{rnn_type} {name} K={K}] launches per lockstep step: train {n_train}, eval {n_eval} (for all {K} fits)")
    assert 0 < n_eval < n_train
    for f in range(K):
        losses, logps, ev, params, mom, step = want[f]
        assert torch.equal(train_out[f][0], losses), f
        assert torch.equal(train_out[f][1], logps), f
        assert torch.equal(grp.logp[1][f].cpu(), ev), f
        assert torch.equal(lock[f].params.cpu(), params) and torch.equal(lock[f].momentum.cpu(), mom), f
        assert int(lock[f].rng[1]) == step == 3
    grp.close()
    d0 = data[0]
    lp = lock[0].forward(d0[0][:B], d0[2][:B], d0[1][:B]).cpu()                 # the plans are usable on their own again
    assert torch.equal(lp, solo[0].forward(d0[0][:B], d0[2][:B], d0[1][:B]).cpu())


@pytest.mark.parametrize("module", ["model.EncoderDecoderLSTMAttn", "model.EncoderDecoderGRUAttn"])
def test_rnn_fit_lockstep_and_grid_equal_one_fit_at_a_time(module):
    """The estimator level for the RNN modules: fit_lockstep == partial_fit per estimator (histories, weights), and
    ShardedGridSearchCV(lockstep=5) == lockstep=1 on cv_results_."""
    from slnlp.data import synthetic_dataset
    from slnlp.grid import ShardedGridSearchCV
    from slnlp.lockstep import fit_lockstep, lockstep_supported
    ds = synthetic_dataset(120, seq_len=10, src_vocab=40, n_labels=4, seed=11, min_len=3)
    from slnlp.net import NeuralNetClassifier

    def _rnn_net(**kw):
        args = dict(module=module, module__dropout=0.1, module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y, module__batch_first=True,
                    module__embedding_size=32, module__hidden_size=32, module__num_layers=2, criterion="torch.nn.CrossEntropyLoss",
                    criterion__ignore_index=1, optimizer="torch.optim.SGD", optimizer__momentum=0.9, lr=0.05, max_epochs=3, batch_size=20,
                    device="cuda", gradient_clipping={"gradient_clip_value": 0.5}, scoring=["neg_log_loss", "accuracy"], use_graph=False)
        args.update(kw)
        return NeuralNetClassifier(**args)
    parts = [ds[np.arange(0, 100)], ds[np.arange(20, 120)]]
    variants = [dict(lr=0.05, module__dropout=0.1), dict(lr=0.02, module__dropout=0.3)]

    def build():
        nets = []
        for i, kw in enumerate(variants):
            torch.manual_seed(60 + i)
            nets.append(_rnn_net(**kw).initialize())
        return nets
    seq = build()
    assert all(lockstep_supported(n) for n in seq)
    for n, d in zip(seq, parts):
        n.partial_fit(d)
    lock = build()
    fit_lockstep(lock, parts)
    strip = lambda h: [{k: v for k, v in row.items() if k != "dur"} for row in h]
    for a, b in zip(seq, lock):
        assert strip(a.history) == strip(b.history)
        sa, sb = a.module_.state_dict(), b.module_.state_dict()
        assert all(torch.equal(sa[k], sb[k]) for k in sa)
    grid = {"lr": [0.1, 0.01], "module__dropout": [0.1, 0.4]}
    factory = lambda: _rnn_net(max_epochs=2, scoring=["neg_log_loss"])
    res = {k: ShardedGridSearchCV(factory, grid, cv=5, refit=False, device="cuda:0", lockstep=k).fit(ds) for k in (1, 5)}
    assert res[5].n_units_ < res[1].n_units_ == 20
    for key in ("mean_test_score", "std_test_score", "split0_test_score", "split4_test_score"):
        assert np.array_equal(res[1].cv_results_[key], res[5].cv_results_[key]), key
