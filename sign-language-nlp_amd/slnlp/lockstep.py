"""K fits of one shape advancing in lockstep through ONE launch sequence (libslnlp ``slnlp_tf_lockstep_*`` /
``slnlp_rnn_lockstep_*``).

The reference runs the (candidate x fold) fits of its grid one at a time per worker
(/root/reference/main.py:70-78, helper.py:490-526).  One batch-50 fit cannot fill an MI355X -- its decoder stages
are 50-row kernels -- and fits on separate streams only reach 1.26x.  Fits of one work unit (same shapes; own
weights, lr, dropout rate, seed and data) therefore share every kernel launch: a 50-row stage becomes a K x 50-row
stage at the same latency, the grouped GEMM launches carry K times the tiles.  Each fit's arithmetic is untouched,
so its history, weights and scores are bit-identical to a solo fit (tests/test_lockstep_gpu.py).

``LockstepGroup`` owns the C object; ``fit_lockstep`` is ``NeuralNetClassifier.partial_fit`` for K estimators at once
(the per-epoch callbacks are the same ``_FitRun`` code); ``fit_and_score_group`` is what ``ShardedGridSearchCV(lockstep=k)``
calls per work unit.
"""
import ctypes as C
import time

import numpy as np
import torch

from ._lib import check, load, ptr, stream_ptr

TRAIN, VALID, TEST = 0, 1, 2


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[ptr(t) for t in tensors])


class LockstepGroup:
    """The fits' engines -- all TransformerEngines or all RnnEngines, same configuration up to the dropout rate -- stepping
    together."""

    def __init__(self, engines):
        cfg = engines[0].cfg
        self.engines, self.K, self.device = list(engines), len(engines), engines[0].device
        self.kind = "rnn" if type(engines[0]).__name__ == "RnnEngine" else "tf"
        assert all(type(e) is type(engines[0]) for e in engines), "lockstep: one engine type per group"
        nbytes = int(self._fn("workspace_bytes")(C.byref(cfg), self.K))
        if nbytes < 0:
            raise RuntimeError("lockstep: bad configuration")
        self._alloc_stream = self._last_stream = torch.cuda.current_stream(self.device)
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        handles = (C.c_void_p * self.K)(*[e.handle for e in self.engines])
        out = C.c_void_p()
        check(self._fn("create")(handles, self.K, ptr(self.workspace), nbytes, self._sp(), C.byref(out)), f"{self.kind}_lockstep_create")
        self.handle = out
        check(self._fn("set_destroy_sync")(out, 0), "lockstep_set_destroy_sync")   # torch-allocated tables: see tf_engine.py
        self.data, self.logp, self.loss, self.rows = {}, {}, {}, {}

    def _fn(self, name):                                 # a method, not a closure over self: no reference cycle
        return getattr(load(), f"slnlp_{self.kind}_lockstep_{name}")

    def _sp(self):
        st = self._last_stream = torch.cuda.current_stream(self.device)
        return st.cuda_stream

    def close(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            ls, al = getattr(self, "_last_stream", None), getattr(self, "_alloc_stream", None)
            if ls is not None and al is not None and ls != al:
                ls.synchronize()                         # the tables go back to another stream's pool (tf_engine.__del__)
            self._fn("destroy")(h)

    __del__ = close

    def set_data(self, slot, Xs, ys, batch, lengths=None):
        """Per-fit datasets of one slot (device int64 [rows, S] / [rows], the same number of rows for every fit; RNN fits also
        pass the sequence lengths [rows]).  Allocates the slot's output buffers: ``logp[slot][f]`` [rows, Vt] and
        ``loss[slot][f]`` [ceil(rows / batch)]."""
        rows = int(Xs[0].shape[0])
        assert len(Xs) == len(ys) == self.K and all(x.shape[0] == rows and x.is_contiguous() for x in Xs)
        Vt = self.engines[0].cfg.Vt
        nb = (rows + batch - 1) // batch
        self.logp[slot] = [torch.empty(rows, Vt, dtype=torch.float32, device=self.device) for _ in range(self.K)]
        self.loss[slot] = [torch.zeros(nb, dtype=torch.float32, device=self.device) for _ in range(self.K)]
        self.rows[slot] = rows
        if self.kind == "rnn":
            assert lengths is not None and len(lengths) == self.K, "lockstep: RNN fits need the sequence lengths"
            lengths = [l.contiguous() for l in lengths]
            self.data[slot] = (list(Xs), list(ys), lengths)   # keep the tensors alive: the C side holds raw pointers
            check(self._fn("set_data")(self.handle, slot, _ptr_array(Xs), _ptr_array(ys), _ptr_array(lengths), rows,
                                       _ptr_array(self.logp[slot]), _ptr_array(self.loss[slot]), self._sp()), "rnn_lockstep_set_data")
        else:
            self.data[slot] = (list(Xs), list(ys))
            check(self._fn("set_data")(self.handle, slot, _ptr_array(Xs), _ptr_array(ys), rows,
                                       _ptr_array(self.logp[slot]), _ptr_array(self.loss[slot]), self._sp()), "tf_lockstep_set_data")

    def set_adam(self, exp_avg_sq, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """Train with the fused clip + Adam update from now on; ``exp_avg_sq``: one arena-shaped buffer per fit."""
        assert len(exp_avg_sq) == self.K
        self._v2 = list(exp_avg_sq)                      # keep the tensors alive: the C side holds raw pointers
        check(self._fn("set_adam")(self.handle, _ptr_array(self._v2), betas[0], betas[1], eps, weight_decay), f"{self.kind}_lockstep_set_adam")

    def _sync_versions(self):
        for e in self.engines:                           # Transformer: weight planes follow outside writes to the fp32 arena
            if hasattr(e, "sync_params_version"):
                e.sync_params_version()

    def step(self, slot, row0, B, step_index, train, momentum=0.9, max_norm=0.5):
        self._sync_versions()
        check(self._fn("step")(self.handle, slot, row0, B, step_index, int(train), momentum, max_norm, self._sp()),
              f"{self.kind}_lockstep_step")

    def epoch(self, slot, batch, train, momentum=0.9, max_norm=0.5):
        """One pass over the slot in dataset order; no host synchronisation.  Results: ``logp[slot]``, ``loss[slot]``."""
        self._sync_versions()
        check(self._fn("epoch")(self.handle, slot, batch, int(train), momentum, max_norm, self._sp()), f"{self.kind}_lockstep_epoch")

    def num_launches(self, slot, B, train):
        return int(self._fn("num_launches")(self.handle, slot, B, int(train)))

    def results(self, slot, f, batch):
        """What ``NeuralNetClassifier._run_epoch`` returns for fit ``f``: (batch-size weighted mean loss, log-probs [rows, Vt],
        [(batch loss, batch size)]).  Call after a synchronisation point."""
        rows = self.rows[slot]
        sizes = [min(batch, rows - r) for r in range(0, rows, batch)]
        per_batch = self.loss[slot][f].float().cpu()
        w = torch.tensor(sizes, dtype=torch.float32)
        mean = float((per_batch * w).sum() / w.sum())
        return mean, self.logp[slot][f], list(zip(per_batch.tolist(), sizes))


LOCKSTEP_MODULES = ("Transformer", "EncoderDecoderLSTMAttn", "EncoderDecoderGRUAttn")


def lockstep_supported(net):
    """A fused update (SGD-momentum or Adam) + CrossEntropyLoss on one of the path's three modules: what the lockstep launch
    sequences implement."""
    return getattr(net, "_fused_kind", None) in ("sgd", "adam") and type(net.module_).__name__ in LOCKSTEP_MODULES


def _adam_key(net):
    ok = net._opt_kwargs
    return (net._fused_kind,) + ((tuple(ok.get("betas", (0.9, 0.999))), float(ok.get("eps", 1e-8)), float(ok.get("weight_decay", 0.0)))
                                 if net._fused_kind == "adam" else ())


def fit_lockstep(nets, datasets):
    """``net.partial_fit(ds)`` for every (net, ds) pair, all fits advancing together.  The nets must be initialised,
    of one shape (lr and dropout rate may differ) and their datasets of one size; fits that stop early (EarlyStopping)
    leave the group, the others go on."""
    nets[0]._gate.enter(False)                          # fused fits share the device (slnlp.net: _DeviceGate)
    try:
        return _fit_lockstep_gated(nets, datasets)
    finally:
        nets[0]._gate.leave(False)


# tools/bench_grid_long.py sets EPOCH_LOG = [] to get one record per lockstep unit: how many fits were still training in each
# epoch and how long the epoch took (fits that stop early -- EarlyStopping, helper.py:240-250 -- leave the group, which is
# rebuilt from the fits that are left: `regroups`)
EPOCH_LOG = None


def _fit_lockstep_gated(nets, datasets):
    from .net import _FitRun, stream_sync
    import time
    K = len(nets)
    log = {"fits": K, "epochs": [], "regroups": 0} if EPOCH_LOG is not None else None
    stream = nets[0]._stream
    assert all(n._stream is stream for n in nets), "lockstep: the fits of a group share the device's stream"
    nets[0]._enter_stream()                             # the stream waits for whatever this thread queued elsewhere so far
    with torch.cuda.stream(stream):
        runs = [_FitRun(n, d) for n, d in zip(nets, datasets)]
    r0 = runs[0]
    assert all(lockstep_supported(n) for n in nets), "lockstep: fused SGD / Adam + CrossEntropyLoss on the model.* modules only"
    assert len({type(n.module_) for n in nets}) == 1, "lockstep: one module class per group"
    assert len({_adam_key(n) for n in nets}) == 1, "lockstep: one optimizer (and one set of Adam constants) per group"
    adam = _adam_key(nets[0]) if nets[0]._fused_kind == "adam" else None
    assert all((r.bs, r.momentum, r.max_norm, len(r.tr), (len(r.va) if r.va is not None else 0)) ==
               (r0.bs, r0.momentum, r0.max_norm, len(r0.tr), (len(r0.va) if r0.va is not None else 0)) for r in runs), \
        "lockstep: the fits of a group share batch size, momentum, clipping and split sizes"
    S = r0.Xtr.shape[1]
    with torch.cuda.stream(stream):
        engines = [n.module_.engine(r0.bs, S) for n in nets]
    active, group = [i for i in range(K) if not runs[i].done], None
    members = None
    with torch.cuda.stream(stream):
        while active:
            if members != active:                       # a fit left (or first epoch): regroup the ones still training
                if group is not None:
                    stream_sync(stream)
                    group.close()
                    if log is not None:
                        log["regroups"] += 1
                group = LockstepGroup([engines[i] for i in active])
                if adam is not None:
                    group.set_adam([nets[i].module_.adam_second_moment() for i in active], adam[1], adam[2], adam[3])
                group.set_data(TRAIN, [runs[i].Xtr for i in active], [runs[i].ytr for i in active], r0.bs,
                               [runs[i].Ltr for i in active])
                if r0.va is not None:
                    group.set_data(VALID, [runs[i].Xva for i in active], [runs[i].yva for i in active], r0.bs,
                                   [runs[i].Lva for i in active])
                members = list(active)
            for i in active:
                engines[i].set_lr(nets[i].lr_)
                nets[i].module_.train()
                runs[i].begin_epoch()
            t_epoch = time.perf_counter()
            group.epoch(TRAIN, r0.bs, True, r0.momentum, r0.max_norm)
            if r0.va is not None:
                group.epoch(VALID, r0.bs, False, r0.momentum, r0.max_norm)
            stream_sync(stream)                         # one host sync per epoch for all K fits
            nxt = []
            for j, i in enumerate(active):
                tr = group.results(TRAIN, j, r0.bs)
                va = group.results(VALID, j, r0.bs) if r0.va is not None else None
                if not runs[i].end_epoch(tr, va):
                    nxt.append(i)
            if log is not None:
                log["epochs"].append([len(active), time.perf_counter() - t_epoch])
            active = nxt
    stream_sync(stream)
    if group is not None:
        group.close()
    if log is not None:
        log["epochs_run"] = [len(n.history) for n in nets]
        EPOCH_LOG.append(log)
    return nets


def predict_proba_lockstep(nets, datasets):
    """``net.predict_proba(ds)`` for every pair through one launch sequence (eval-mode forward, softmax of the log-probs as
    skorch's predict_nonlinearity='auto' does)."""
    from .net import stream_sync
    bs = int(nets[0].batch_size)
    nets[0]._gate.enter(False)
    try:
        return _predict_proba_lockstep_gated(nets, datasets, bs, stream_sync)
    finally:
        nets[0]._gate.leave(False)


def _predict_proba_lockstep_gated(nets, datasets, bs, stream_sync):
    nets[0]._enter_stream()
    with torch.cuda.stream(nets[0]._stream):
        dev = [n._device_data(d) for n, d in zip(nets, datasets)]
        S = dev[0][0].shape[1]
        engines = [n.module_.engine(bs, S) for n in nets]
        for n in nets:
            n.module_.eval()
        group = LockstepGroup(engines)
        group.set_data(TEST, [d[0] for d in dev], [d[2] for d in dev], bs, [d[1] for d in dev])
        group.epoch(TEST, bs, False)
        out = [lp.clone() for lp in group.logp[TEST]]
        stream_sync(nets[0]._stream)
        # softmax on the host copies (torch's CPU op, as in NeuralNetClassifier.predict_proba: no torch arithmetic kernel runs
        # beside other fits on the GPU)
        out = [(torch.softmax(o.cpu(), dim=-1) if n.predict_nonlinearity == "auto" else o.cpu()).numpy() for n, o in zip(nets, out)]
        group.close()
    return out


def fit_and_score_group(estimator_factory, params_list, trains, tests, scoring="neg_log_loss", seeds=None):
    """``grid.default_fit_and_score`` for the tasks of one work unit: fresh estimators (seeded one after another, like the
    one-at-a-time path), one lockstep fit, one lockstep scoring pass over the test folds.  Falls back to one fit at a time for
    anything the lockstep sequence does not implement (other modules / optimizers) -- same results, just not merged."""
    from .grid import default_fit_and_score
    from .net import INIT_LOCK, ScoringWrapper, _CachedPredictor
    seeds = seeds or [None] * len(params_list)
    nets = []
    for params, seed in zip(params_list, seeds):
        net = estimator_factory().set_params(**params)
        if "checkpoint_dir" in net.get_params():
            net.set_params(checkpoint_dir=None)
        with INIT_LOCK:
            if seed is not None:
                torch.manual_seed(seed)
            net.initialize()
        nets.append(net)
    if not all(lockstep_supported(n) for n in nets) or len({type(n.module_) for n in nets}) != 1 or len({_adam_key(n) for n in nets}) != 1 or \
            len({len(t) for t in trains}) != 1 or len({len(t) for t in tests}) != 1:
        del nets
        # concurrent=True: no hipGraph capture -- other host threads may be launching on the device's shared stream
        return [default_fit_and_score(estimator_factory, p, tr, te, scoring, seed=s, concurrent=True)
                for p, tr, te, s in zip(params_list, trains, tests, seeds)]
    fit_lockstep(nets, trains)
    probas = predict_proba_lockstep(nets, tests)
    scores = []
    for net, train, test, proba in zip(nets, trains, tests, probas):
        wr = ScoringWrapper(scoring, train.labels() if scoring == "neg_log_loss" else None)
        scores.append(float(wr(_CachedPredictor(proba, net.classes_), None, test.y)))
    return scores
