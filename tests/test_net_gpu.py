"""GPU: the skorch-shaped estimator (slnlp.net) on the HIP path.
G8: a mini-fit trajectory -- same data, same initial weights, dropout 0 -- run through the
estimator's loop must reproduce the loss trajectory of the CPU oracle stepping over the same batches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(module__embedding_size=32, module__num_heads=4, module__num_layers=2, module__hidden_size=64)


def make_net(ds, dropout=0.0, **kw):
    from slnlp.net import NeuralNetClassifier
    args = dict(module="model.Transformer", module__dropout=dropout, module__src_vocab=ds.vocab_X,
                module__tgt_vocab=ds.vocab_y, module__batch_first=True, **CFG,
                criterion="torch.nn.CrossEntropyLoss", criterion__ignore_index=1,
                optimizer="torch.optim.SGD", optimizer__momentum=0.9, optimizer__nesterov=False,
                lr=0.05, max_epochs=3, batch_size=20, device="cuda",
                gradient_clipping={"gradient_clip_value": 0.5})
    args.update(kw)
    return NeuralNetClassifier(**args)


def test_minifit_trajectory_matches_oracle():
    from oracle import train_ref, transformer_ref as tr
    from slnlp.data import synthetic_dataset
    ds = synthetic_dataset(120, seq_len=12, src_vocab=64, n_labels=6, seed=5, min_len=3)
    torch.manual_seed(7)
    net = make_net(ds, scoring=["neg_log_loss", "accuracy"])
    net.initialize()
    sd0 = {k: v.detach().cpu().clone() for k, v in net.module_.state_dict().items() if not k.endswith(".pe")}
    net.partial_fit(ds)
    hist = net.history
    assert len(hist) == 3 and {"train_loss", "valid_loss", "valid_loss_best", "lr", "dur", "train_accuracy",
                               "valid_neg_log_loss"} <= set(hist[0])
    # oracle over the same internal split / batch order
    idx_tr, idx_va = net._train_split(ds)
    tr_ds, va_ds = ds[idx_tr], ds[idx_va]
    fwd = lambda p, X, y, L: tr.forward(p, X, y, num_heads=4, num_layers=2)
    trn = train_ref.Trainer(sd0, fwd, pad_tgt=1, lr=0.05, momentum=0.9, max_norm=0.5)
    X, y = torch.from_numpy(tr_ds.ids), torch.from_numpy(tr_ds.y)
    Xv, yv = torch.from_numpy(va_ds.ids), torch.from_numpy(va_ds.y)
    for ep in range(3):
        tot, n = 0.0, 0
        for i in range(0, len(tr_ds), 20):
            loss, _, _ = trn.step(X[i:i + 20], y[i:i + 20], None)
            tot += float(loss) * len(X[i:i + 20]); n += len(X[i:i + 20])
        with torch.no_grad():
            vl = 0.0
            for i in range(0, len(va_ds), 20):
                vl += float(train_ref.cross_entropy_on_logprobs(fwd(trn.sd, Xv[i:i + 20], yv[i:i + 20], None), yv[i:i + 20], 1)) * len(Xv[i:i + 20])
        print(f"epoch {ep}: train {hist[ep]['train_loss']:.5f} vs oracle {tot / n:.5f}; valid {hist[ep]['valid_loss']:.5f} vs {vl / len(va_ds):.5f}")
        assert abs(hist[ep]["train_loss"] - tot / n) < 2e-3 * (tot / n)
        assert abs(hist[ep]["valid_loss"] - vl / len(va_ds)) < 2e-3 * (vl / len(va_ds))
    proba = net.predict_proba(ds)
    assert proba.shape == (120, len(ds.vocab_y)) and np.allclose(proba.sum(1), 1.0, atol=1e-5)
    assert net.predict(ds).shape == (120,)
    assert hist[-1]["train_loss"] < hist[0]["train_loss"]


def test_fused_step_equals_autograd_path():
    """The fused hipGraph step and the torch-optimizer path around the autograd Function agree."""
    from slnlp.data import synthetic_dataset
    ds = synthetic_dataset(80, seq_len=12, src_vocab=64, n_labels=6, seed=6, min_len=3)
    losses = []
    for opt_kw in ({}, {"optimizer__nesterov": False, "optimizer__weight_decay": 1e-30}):   # 2nd: forces the generic path
        torch.manual_seed(11)
        net = make_net(ds, max_epochs=2, **opt_kw).fit(ds)
        losses.append([h["train_loss"] for h in net.history])
    assert np.allclose(losses[0], losses[1], rtol=2e-3), losses


def test_callbacks_semantics(tmp_path):
    from slnlp.data import synthetic_dataset
    ds = synthetic_dataset(100, seq_len=10, src_vocab=50, n_labels=5, seed=8, min_len=3)
    torch.manual_seed(3)
    net = make_net(ds, max_epochs=40, lr=1e-6, early_stopping={"patience": 3, "threshold": 1e-4, "threshold_mode": "rel"},
                   lr_scheduler={"policy": "ReduceLROnPlateau", "factor": 0.2, "patience": 1},
                   checkpoint_dir=str(tmp_path))
    net.fit(ds)
    assert len(net.history) < 40                                   # lr ~ 0: no improvement -> early stop
    assert net.history[-1]["lr"] < 1e-6                            # plateau scheduler cut the lr
    assert (tmp_path / "params.pt").exists() and (tmp_path / "history.json").exists()
    sd = torch.load(tmp_path / "params.pt")
    assert "src_pos_encoding.pe" in sd and "transformer.decoder.layers.1.norm3.bias" in sd


def test_sklearn_gridsearch_drives_the_estimator():
    from sklearn.model_selection import GridSearchCV
    from slnlp.data import synthetic_dataset
    from slnlp.net import ScoringWrapper
    ds = synthetic_dataset(90, seq_len=10, src_vocab=50, n_labels=3, seed=9, min_len=3)
    net = make_net(ds, max_epochs=2)
    gs = GridSearchCV(net, {"lr": [0.1, 0.01], "module__num_layers": [1, 2]}, cv=3, refit=True,
                      scoring=ScoringWrapper("neg_log_loss", ds.labels()), error_score="raise")
    gs.fit(ds, ds.y)
    assert len(gs.cv_results_["params"]) == 4 and np.isfinite(gs.best_score_)
    assert gs.best_estimator_.predict(ds).shape == (90,)


@pytest.mark.parametrize("module", ["model.EncoderDecoderLSTMAttn", "model.EncoderDecoderGRUAttn"])
def test_rnn_modules_through_the_estimator(module):
    """config-enc-dec-{lstm,gru}-attn.yaml style fit: same estimator, RNN module, fused step."""
    from slnlp.data import synthetic_dataset
    from slnlp.net import NeuralNetClassifier
    ds = synthetic_dataset(120, seq_len=12, src_vocab=64, n_labels=6, seed=5, min_len=3)
    torch.manual_seed(3)
    net = NeuralNetClassifier(module=module, module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y,
                              module__batch_first=True, module__embedding_size=24, module__hidden_size=32,
                              module__num_layers=2, module__dropout=0.1, criterion__ignore_index=1,
                              optimizer__momentum=0.9, lr=0.5, max_epochs=12, batch_size=20,
                              gradient_clipping={"gradient_clip_value": 0.5}, scoring=["accuracy"])
    net.fit(ds)
    h = net.history
    assert net._fused and h[-1]["train_loss"] < h[0]["train_loss"] - 0.05
    assert np.allclose(net.predict_proba(ds).sum(1), 1.0, atol=1e-5)


def test_sharded_grid_concurrent_fits_equal_sequential():
    """ShardedGridSearchCV(fits_per_gpu=2): two fits at a time on two streams give the same cv_results_ as one at
    a time (per-task seeding; no cross-talk between plans, streams or the thread-local error state)."""
    from slnlp.data import synthetic_dataset
    from slnlp.grid import ShardedGridSearchCV
    ds = synthetic_dataset(90, seq_len=10, src_vocab=50, n_labels=3, seed=9, min_len=3)
    factory = lambda: make_net(ds, max_epochs=2, dropout=0.1)
    pg = {"lr": [0.1, 0.01], "module__num_layers": [1, 2]}
    out = []
    for k in (1, 2):
        gs = ShardedGridSearchCV(factory, pg, cv=3, refit=False, device="cuda", fits_per_gpu=k).fit(ds)
        out.append(gs.cv_results_["mean_test_score"])
    print(out)
    assert np.array_equal(out[0], out[1]) and np.isfinite(out[0]).all()


def test_cli_end_to_end_on_a_synthetic_corpus(tmp_path):
    """python -m slnlp.cli: reference-style YAML -> ingest -> balancing -> split -> sharded grid search -> test metrics,
    with the reference's artefact files in workdir."""
    import json
    import pandas as pd
    from test_pipeline_cpu import FIELDS, make_corpus
    from slnlp import cli
    corpus = tmp_path / "corpus"
    corpus.mkdir()
    make_corpus(str(corpus), per_label=(8, 9, 7, 10, 8, 9))
    cfg = tmp_path / "config.yaml"
    cfg.write_text(f"""
debug: False
cuda: True
seed: 1
workdir: '{tmp_path}/work/{{model}}'
verbose: 0
cv: 2
lr:
scoring: [neg_log_loss, accuracy, f1_weighted]
max_epochs: 3
batch_size: 16
test_size: 0.15
early_stopping: {{patience: 30, threshold: 1e-4, threshold_mode: rel}}
gradient_clipping: {{gradient_clip_value: 0.5}}
lr_scheduler: {{policy: ReduceLROnPlateau, factor: 0.2, patience: 5}}
model: model.Transformer
model_args: {{embedding_size: , hidden_size: 32, num_layers: 1, dropout: 0.1, num_heads: 2}}
criterion: torch.nn.CrossEntropyLoss
optimizer: torch.optim.SGD
optimizer_args: {{nesterov: False, momentum: 0.9}}
grid_args:
  lr: [0.1, 0.01]
  model_args: {{embedding_size: [16, 32]}}
dataset_args:
  dataset_dir: {corpus}
  fields: {FIELDS}
  samples_min_freq: 2
  composition_strategy: as_words
  balance_dataset: True
""")
    cli.main(["--config", str(cfg)])
    work = tmp_path / "work" / "model.Transformer"
    for f in ("config.yaml", "grid_search_grid_params.csv", "grid_search_output.json", "grid_search_results.csv",
              "test_output.json", "params.pt", "optimizer.pt", "criterion.pt", "history.json"):
        assert (work / f).exists(), f
    out = json.load(open(work / "grid_search_output.json"))
    assert set(out) == {"best_score", "best_params", "best_index", "scoring"} and set(out["best_params"]) == {"lr", "module__embedding_size"}
    res = pd.read_csv(work / "grid_search_results.csv", index_col=0)
    assert len(res) == 4 and {"mean_test_score", "rank_test_score", "split0_test_score", "split1_test_score"} <= set(res.columns)
    test_out = json.load(open(work / "test_output.json"))
    assert set(test_out) == {"test_neg_log_loss", "test_accuracy", "test_f1_weighted"} and all(np.isfinite(v) for v in test_out.values())
    hist = json.load(open(work / "history.json"))
    assert len(hist) == 3 and {"train_f1_weighted", "valid_f1_weighted", "valid_accuracy", "lr"} <= set(hist[0])


def test_sharded_grid_concurrent_rnn_fits_equal_sequential():
    """Same as above for the recurrent models: two GRU fits at a time on two streams == one at a time."""
    from slnlp.data import synthetic_dataset
    from slnlp.grid import ShardedGridSearchCV
    from slnlp.net import NeuralNetClassifier
    ds = synthetic_dataset(90, seq_len=10, src_vocab=50, n_labels=3, seed=9, min_len=3)
    factory = lambda: NeuralNetClassifier(
        module="model.EncoderDecoderGRUAttn", module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y, module__batch_first=True,
        module__embedding_size=64, module__hidden_size=64, module__num_layers=2, module__dropout=0.1, criterion__ignore_index=1,
        optimizer__momentum=0.9, lr=0.1, max_epochs=2, batch_size=20, gradient_clipping={"gradient_clip_value": 0.5})
    pg = {"lr": [0.1, 0.01], "module__hidden_size": [64, 128]}
    out = []
    for k in (1, 2):
        gs = ShardedGridSearchCV(factory, pg, cv=2, refit=False, device="cuda", fits_per_gpu=k).fit(ds)
        out.append(gs.cv_results_["mean_test_score"])
    assert np.array_equal(out[0], out[1]) and np.isfinite(out[0]).all()


def test_checkpoint_is_a_torch_sgd_state_dict_and_resumes(tmp_path):
    """optimizer.pt of the fused path loads into a stock torch.optim.SGD over the module's parameters (momentum buffers
    per parameter, lr in the param group) and ``load_params`` restores weights + momentum + lr: a resumed fit continues
    the trajectory of an uninterrupted one.  history.json carries skorch's per-batch rows."""
    import json
    from slnlp.data import synthetic_dataset
    ds = synthetic_dataset(100, seq_len=10, src_vocab=50, n_labels=5, seed=8, min_len=3)
    torch.manual_seed(3)
    full = make_net(ds, max_epochs=4).fit(ds)
    torch.manual_seed(3)
    first = make_net(ds, max_epochs=2).fit(ds)
    first.save_params(str(tmp_path))
    sd = torch.load(tmp_path / "optimizer.pt")
    names = [n for n, _ in first.module_.named_parameters()]
    ref_opt = torch.optim.SGD(first.module_.parameters(), lr=123.0, momentum=0.9)
    ref_opt.load_state_dict(sd)                                       # the stock optimizer accepts it
    assert ref_opt.param_groups[0]["lr"] == pytest.approx(0.05) and ref_opt.param_groups[0]["momentum"] == 0.9
    mom = first.module_._shared_state()["momentum"]
    ent = {n: (shape, off) for n, shape, off in first.module_._entries}
    for i, n in enumerate(names):
        buf = sd["state"][i]["momentum_buffer"]
        shape, off = ent[n]
        assert tuple(buf.shape) == tuple(shape) and torch.equal(buf, mom[off:off + buf.numel()].view(*shape).cpu())
    assert float(sd["state"][names.index("linear.weight")]["momentum_buffer"].abs().sum()) > 0
    hist = json.load(open(tmp_path / "history.json"))
    assert len(hist) == 2 and {"train_loss", "train_batch_size"} <= set(hist[0]["batches"][0])
    assert {"valid_loss", "valid_batch_size"} <= set(hist[0]["batches"][-1])
    assert sum(b["train_batch_size"] for b in hist[0]["batches"] if "train_batch_size" in b) == 80
    # resume in a fresh estimator: epochs 3-4 reproduce the uninterrupted run
    torch.manual_seed(99)                                              # different initial weights: must be overwritten
    resumed = make_net(ds, max_epochs=2, warm_start=True).initialize()
    resumed.load_params(str(tmp_path))
    # the dropout step counter is not part of a skorch checkpoint; dropout is 0 here so the trajectory is deterministic
    resumed.partial_fit(ds)
    assert [h["epoch"] for h in resumed.history] == [1, 2, 3, 4]
    for a, b in zip(resumed.history[2:], full.history[2:]):
        assert a["train_loss"] == pytest.approx(b["train_loss"], rel=1e-5), (a["train_loss"], b["train_loss"])
        assert a["valid_loss"] == pytest.approx(b["valid_loss"], rel=1e-5)


def test_rccl_collectives_of_the_grid_on_one_gpu(tmp_path):
    """The nccl (= RCCL) branch of broadcast_dataset / all_gather executed on hardware: a one-rank RCCL group with
    ``force_collectives`` sends the packed dataset and the score rows through RCCL and must give the results of the
    collective-free run."""
    import torch.distributed as dist
    from slnlp.data import synthetic_dataset
    from slnlp.grid import ShardedGridSearchCV, broadcast_dataset
    ds = synthetic_dataset(60, seq_len=10, src_vocab=50, n_labels=3, seed=9, min_len=3)
    grid = {"lr": [0.1, 0.01], "module__num_layers": [1, 2]}
    plain = ShardedGridSearchCV(lambda: make_net(ds, max_epochs=1), grid, cv=2, refit=False, device="cuda:0").fit(ds)
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdv", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        got = broadcast_dataset(ds, "cuda:0", force=True)
        assert np.array_equal(got.ids, ds.ids) and np.array_equal(got.y, ds.y) and np.array_equal(got.lengths, ds.lengths)
        gs = ShardedGridSearchCV(lambda: make_net(ds, max_epochs=1), grid, cv=2, refit=False, device="cuda:0",
                                 force_collectives=True).fit(ds)
        assert dist.get_backend() == "nccl" and gs.rank_tasks_ == [8]
    finally:
        dist.destroy_process_group()
    assert np.array_equal(gs.cv_results_["mean_test_score"], plain.cv_results_["mean_test_score"])
    assert gs.best_index_ == plain.best_index_


# G8 bars per epoch.  A fit is a chaotic map (tools/gen_golden.py, FIT_CASE; tests/test_oracle_golden.py::G8_TOL): two fp32
# CPU implementations already drift apart by ~10x per epoch at lr 0.01.  lr 0.001 holds north_star's 1e-3 on every epoch.
G8_TOL = {0.001: [1e-3] * 5, 0.01: [1e-3, 1e-3, 3e-3, 1e-2, 3e-2]}


@pytest.mark.parametrize("lr", [0.001, 0.01])
def test_fit_trajectory_g8_vs_reference(lr):
    """G8 at the configs[0] shape: the estimator's 5-epoch fit on the HIP path against the same
    loop run around the REFERENCE module (tests/golden/fit_cfg1.npz) -- train / valid loss and the five epoch metrics of
    both splits, every epoch."""
    import gold
    from slnlp import synth
    from slnlp.data import synthetic_dataset
    from slnlp.net import NeuralNetClassifier
    from oracle import transformer_ref as tr
    g = gold.load("fit_cfg1")
    Vs, nl, E, H, N, F, S, n, bs, epochs = [int(v) for v in g["cfg"]]
    mom, clip = [float(v) for v in g["mom_clip"]]
    hist = g[f"history_lr{lr}"]
    ds = synthetic_dataset(n, seq_len=S, src_vocab=Vs, n_labels=nl, seed=1, min_len=8)
    net = NeuralNetClassifier(module="model.Transformer", module__dropout=0.0, module__src_vocab=ds.vocab_X,
                              module__tgt_vocab=ds.vocab_y, module__batch_first=True, module__embedding_size=E,
                              module__num_heads=H, module__num_layers=N, module__hidden_size=F, criterion__ignore_index=1,
                              optimizer__momentum=mom, optimizer__nesterov=False, lr=lr, max_epochs=epochs, batch_size=bs,
                              device="cuda", gradient_clipping={"gradient_clip_value": clip}, scoring=list(g["metrics"]))
    net.initialize()
    sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(tr.param_shapes(E, H, N, F, Vs, nl + 2), seed=1).items()}
    net.module_.load_state_dict({**net.module_.state_dict(), **sd})
    net.partial_fit(ds)
    cols = list(g["columns"])
    worst = [0.0] * epochs
    for ep in range(epochs):
        for j, k in enumerate(cols):
            ref, got = float(hist[ep, j]), net.history[ep][k]
            if "loss" in k:                      # train_loss, valid_loss, *_neg_log_loss
                worst[ep] = max(worst[ep], abs(got - ref) / abs(ref))
                assert abs(got - ref) <= G8_TOL[lr][ep] * abs(ref), (ep, k, got, ref)
            else:                                # arg-max metrics move in steps: allow near-tie samples to flip
                n_split = int(g["n_train"]) if k.startswith("train") else int(g["n_valid"])
                assert abs(got - ref) <= (2.5 if ep < 3 else 6.5) / n_split, (ep, k, got, ref)
        print(f"[lr {lr}] epoch {ep}: train {net.history[ep]['train_loss']:.5f} (ref {hist[ep, 0]:.5f})  valid "
              f"{net.history[ep]['valid_loss']:.5f} (ref {hist[ep, 1]:.5f})  valid acc {net.history[ep]['valid_accuracy']:.3f} "
              f"(ref {hist[ep, cols.index('valid_accuracy')]:.3f})  worst rel err {worst[ep]:.2e}")
    gold.check_summary(g, f"wfinal_lr{lr}", {k: v.detach().cpu() for k, v in net.module_.state_dict().items() if not k.endswith(".pe")},
                       2e-3 if lr == 0.001 else 3e-2)


def test_fused_adam_equals_torch_adam():
    """optimizer=torch.optim.Adam: the fused clip + Adam kernel (north_star's "fused SGD-momentum/Adam update") against
    torch.optim.Adam stepping the same module through the autograd bridge, end to end: the epoch losses agree to 1e-5 and
    optimizer.pt is a torch.optim.Adam state_dict.  (The update arithmetic itself is held to an ulp per step in
    test_kernels_gpu.py::test_clip_adam_vs_torch.  Single weights are NOT compared here: Adam divides by sqrt(v), so a
    parameter whose gradient is pure rounding noise -- the key bias of every attention block has an exactly-zero true
    gradient -- moves by a full lr step whose SIGN follows that noise, and the two paths differ by an ulp in d loss/d logits.)"""
    from slnlp.data import synthetic_dataset
    from slnlp.net import NeuralNetClassifier
    ds = synthetic_dataset(80, seq_len=12, src_vocab=64, n_labels=6, seed=6, min_len=3)
    nets = []
    for fused in (True, False):
        torch.manual_seed(11)
        net = NeuralNetClassifier(module="model.Transformer", module__dropout=0.0, module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y,
                                  module__batch_first=True, **CFG, criterion="torch.nn.CrossEntropyLoss", criterion__ignore_index=1,
                                  optimizer="torch.optim.Adam", optimizer__betas=(0.9, 0.99), optimizer__eps=1e-8, lr=3e-3, max_epochs=2,
                                  batch_size=20, device="cuda", gradient_clipping={"gradient_clip_value": 0.5})
        net.initialize()
        assert net._fused_kind == "adam"
        if not fused:                                   # force the stock-optimizer path around the autograd bridge
            net._fused, net._fused_kind = False, None
            net.optimizer_ = net._opt_cls(net.module_.parameters(), lr=net.lr, **net._opt_kwargs)
        net.partial_fit(ds)
        nets.append(net)
    la, lb = [h["train_loss"] for h in nets[0].history], [h["train_loss"] for h in nets[1].history]
    assert np.allclose(la, lb, rtol=1e-5), (la, lb)
    va, vb = [h["valid_loss"] for h in nets[0].history], [h["valid_loss"] for h in nets[1].history]
    assert np.allclose(va, vb, rtol=1e-4), (va, vb)
    ref = nets[1].optimizer_.state_dict()
    got = nets[0]._sgd_state_dict()
    assert got["param_groups"][0]["betas"] == (0.9, 0.99) and float(got["state"][0]["step"]) == float(ref["state"][0]["step"]) == 8.0
    for i in ref["state"]:
        for key in ("exp_avg", "exp_avg_sq"):
            a, b = got["state"][i][key], ref["state"][i][key].cpu()
            # 8 steps in, the noise-driven lr-sized moves described above have fed back into every gradient: the moments
            # agree to ~1e-3 of their mean size (measured 9e-4 on the embedding), the losses above to 1e-5
            assert a.shape == b.shape and float((a - b).abs().mean()) <= 5e-3 * max(1e-6, float(b.abs().mean())) + 1e-12, (i, key)
    torch.optim.Adam(nets[1].module_.parameters(), lr=1.0).load_state_dict(got)         # the stock optimizer accepts it


def test_recipe_init_on_the_device_trains_and_is_reproducible():
    """module__init="recipe": the weights are drawn on the GPU (no CPU modules, no host-to-device copy of the arena), the same
    torch seed gives the same weights, a fit on them descends, and ShardedGridSearchCV uses it for the CV fits only (the
    refit keeps the reference-identical stream)."""
    from slnlp.data import synthetic_dataset
    from slnlp.grid import ShardedGridSearchCV
    ds = synthetic_dataset(120, seq_len=12, src_vocab=64, n_labels=6, seed=5, min_len=3)
    nets = []
    for _ in range(2):
        torch.manual_seed(21)
        nets.append(make_net(ds, max_epochs=4, dropout=0.1, module__init="recipe").initialize())
    a, b = nets[0].module_.state_dict(), nets[1].module_.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a) and nets[0].module_._arena.is_cuda
    w = a["transformer.encoder.layers.0.linear1.weight"]
    bound = (6.0 / (w.shape[0] + w.shape[1])) ** 0.5
    assert float(w.abs().max()) <= bound and float(w.std()) > 0.5 * bound              # xavier-uniform, drawn
    assert float(a["transformer.encoder.layers.0.norm1.weight"].min()) == 1.0
    nets[0].partial_fit(ds)
    h = nets[0].history
    assert h[-1]["train_loss"] < h[0]["train_loss"]
    gs = ShardedGridSearchCV(lambda: make_net(ds, max_epochs=1, dropout=0.1), {"lr": [0.1, 0.01]}, cv=2, refit=True, device="cuda").fit(ds)
    assert gs.recipe_init and gs.best_estimator_.module_.init == "reference" and np.isfinite(gs.best_score_)


def test_concurrent_fits_at_working_sizes_do_not_influence_each_other():
    """fits_per_gpu=3 at E 512 / batch 50 / len 48 -- sizes at which kernels of three fits on three hardware queues used to
    read their own producer kernels' output stale (tools/probes/probe_concurrent3.py: backward results changed from
    run to run; the tiny shapes of test_sharded_grid_concurrent_fits_equal_sequential never showed it).  All estimators of a
    device now share one stream (slnlp.net.device_stream): three host threads, one kernel sequence, scores identical to one
    thread and from run to run."""
    from slnlp.data import synthetic_dataset
    from slnlp.grid import ShardedGridSearchCV
    from slnlp.net import NeuralNetClassifier, device_stream
    ds = synthetic_dataset(300, seq_len=48, src_vocab=3000, n_labels=50, seed=2, min_len=8)
    factory = lambda: NeuralNetClassifier(
        module="model.Transformer", module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y, module__batch_first=True,
        module__embedding_size=512, module__num_heads=8, module__num_layers=2, module__hidden_size=512, module__dropout=0.1,
        criterion__ignore_index=1, optimizer__momentum=0.9, optimizer__nesterov=False, lr=0.1, max_epochs=2, batch_size=50,
        device="cuda", gradient_clipping={"gradient_clip_value": 0.5}, scoring=["neg_log_loss"])
    grid = {"lr": [0.1, 0.03, 0.01]}
    runs = []
    for k, ls in ((1, 1), (3, 1), (3, 1), (3, 3)):
        gs = ShardedGridSearchCV(factory, grid, cv=3, refit=False, device="cuda", fits_per_gpu=k, lockstep=ls).fit(ds)
        runs.append(np.stack([gs.cv_results_[f"split{i}_test_score"] for i in range(3)]))
    for r in runs[1:]:
        assert np.array_equal(runs[0], r), (runs[0], r)
    a, b = factory().initialize(), factory().initialize()
    assert a._stream is b._stream is device_stream("cuda")
