"""skorch-shaped estimator around the HIP modules.

The reference trains through ``skorch.NeuralNetClassifier(**net_params)``
(/root/reference/main.py:44, helper.py:41-105) driven by sklearn.  skorch is
not available here, and its Python step loop is exactly the per-batch overhead
the hot path removes, so this class restates the pieces of skorch 0.10 the
reference configures, with the same parameter names
(``module__*``, ``optimizer__*``, ``criterion__*``, ``lr``, ``max_epochs``,
``batch_size``, ``device``, ``callbacks``-level settings) and the same fit-loop
semantics:

* internal 80/20 stratified split, first fold of ``StratifiedKFold(5)`` (skorch ``CVSplit(5)``);
* batches in dataset order (``shuffle`` is commented out, helper.py:75-76);
* per batch: forward -> CrossEntropyLoss(ignore_index=pad) -> backward ->
  clip_grad_norm_(gradient_clip_value) -> SGD(momentum)  == ONE hipGraph replay;
* per epoch: valid pass, ``EpochScoring`` metrics for train/valid, ``lr`` scoring,
  ``LRScheduler(ReduceLROnPlateau)`` on valid_loss, ``EarlyStopping`` (patience,
  relative threshold), ``Checkpoint`` on ``valid_loss_best`` (helper.py:197-273).

The compute path is HIP only; with no GPU ``fit`` / ``predict`` raise.
"""
import threading
import json
import os
import time
import importlib

import numpy as np
import torch

from . import metrics
from .data import TokenDataset


# module construction consumes torch's global CPU generator (initial weights): concurrent fits take turns
INIT_LOCK = threading.RLock()

# Which stream a fit runs on.  "thread" (default): one stream per host thread and device for FUSED fits -- fits whose every
# kernel is this library's (fused SGD / Adam step, slnlp.lockstep): the grid search's `fits_per_gpu` host threads feed separate
# hardware queues, so one unit's small launches (the decoder's [B, E] chain, the optimizer) run beside another's; measured +17 %
# folds/hr on bench.py's grid sample, scores bit-identical (bench.py prints their CRC-32).  "device": one stream per device for
# every estimator of the process (round 2's rule, SLNLP_STREAM_MODE=device).
# History: round 2 measured that fits on several queues changed each other's results and shipped the
# one-stream rule.  Round 3 found the cause -- packed fp32 VALU instructions compute wrongly when a workgroup of another kernel
# shares the CU -- and builds the library without them.  torch's own kernels (and rocBLAS) ARE built with packed fp32, so
# whatever runs them never shares the GPU with another fit here:
#   * a fit that steps through torch (another optimizer / criterion: `_fused` False) runs on the shared device stream and holds
#     the device EXCLUSIVELY for its fit / predict calls (`_DeviceGate`: fused fits of other threads hold it shared);
#   * the scoring softmax of `predict_proba` runs on the host copy of the log-probs (torch's CPU op, what the reference runs);
#   * what is left on the GPU from torch beside other fits moves or compares bits (copies, cat, argmax, gather): no fp32 arithmetic.
# The library-wide stream policy is never flipped from here: a thread that gets a stream of its own opts ITS steps out
# (slnlp_set_thread_stream_policy).
STREAM_MODE = os.environ.get("SLNLP_STREAM_MODE", "thread")      # "thread" | "device"
_DEVICE_STREAMS = {}
_DEVICE_STREAMS_LOCK = threading.Lock()


class _DeviceGate:
    """Readers-writer gate per device: fused fits enter shared, fits that run torch kernels enter exclusive (writer
    preference, re-entrant per thread for nested fit -> predict calls)."""
    def __init__(self):
        self._cv = threading.Condition()
        self._shared, self._excl_owner, self._excl_depth, self._excl_waiting = 0, None, 0, 0
        self._tl = threading.local()

    def enter(self, exclusive):
        me = threading.get_ident()
        with self._cv:
            if self._excl_owner == me:                       # nested call of the exclusive holder
                self._excl_depth += 1
                return
            depth = getattr(self._tl, "shared", 0)
            if not exclusive and depth:                      # nested shared call
                self._tl.shared = depth + 1
                return
            if exclusive and depth:
                # this thread already holds the gate SHARED (it is inside a fused fit): waiting for "no shared holders" would wait for
                # itself, for ever.  A torch-stepped fit / predict nested in a fused fit of the same thread has no legal order.
                raise RuntimeError("slnlp device gate: a fit or predict that steps through torch kernels (exclusive use of the GPU) was "
                                   "started from inside a fused fit of the same thread (shared use): finish the fused fit first, or run "
                                   "the other estimator from a thread of its own")
            if exclusive:
                self._excl_waiting += 1
                while self._excl_owner is not None or self._shared:
                    self._cv.wait()
                self._excl_waiting -= 1
                self._excl_owner, self._excl_depth = me, 1
            else:
                while self._excl_owner is not None or self._excl_waiting:
                    self._cv.wait()
                self._shared += 1
                self._tl.shared = 1

    def leave(self, exclusive):
        with self._cv:
            if self._excl_owner == threading.get_ident():
                self._excl_depth -= 1
                if self._excl_depth == 0:
                    self._excl_owner = None
                    self._cv.notify_all()
                return
            self._tl.shared -= 1
            if self._tl.shared == 0:
                self._shared -= 1
                self._cv.notify_all()


_GATES = {}


def device_gate(dev):
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _DEVICE_STREAMS_LOCK:
        g = _GATES.get(idx)
        if g is None:
            g = _GATES[idx] = _DeviceGate()
        return g


def stream_sync(stream):
    """Wait for what THIS thread has queued on the shared stream so far -- not for the whole device: with several host
    threads feeding one stream, torch.cuda.synchronize() would also wait for everything the others queue in the meantime."""
    ev = torch.cuda.Event()
    ev.record(stream)
    ev.synchronize()


def device_stream(dev, per_thread=None):
    """The stream a fit on `dev` runs on: this host thread's own (`per_thread`, default: STREAM_MODE == "thread") or the one
    every estimator of the process shares on that device.  A thread that gets a stream of its own also opts its library steps
    out of the one-sequence-per-device ordering (thread-scoped: slnlp_set_thread_stream_policy)."""
    dev = torch.device(dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if per_thread is None:
        per_thread = STREAM_MODE == "thread"
    key = (dev.index, threading.get_ident()) if per_thread else dev.index
    with _DEVICE_STREAMS_LOCK:
        st = _DEVICE_STREAMS.get(key)
        if st is None:
            st = _DEVICE_STREAMS[key] = torch.cuda.Stream(device=dev)
    if per_thread:
        from . import _lib
        _lib.load().slnlp_set_thread_stream_policy(0)
    return st


def release_thread_streams():
    """Drop the calling thread's per-thread streams (a grid worker thread about to exit) and put its library steps back under
    the process-wide stream policy."""
    me = threading.get_ident()
    with _DEVICE_STREAMS_LOCK:
        for key in [k for k in _DEVICE_STREAMS if isinstance(k, tuple) and k[1] == me]:
            del _DEVICE_STREAMS[key]
    from . import _lib
    _lib.load().slnlp_set_thread_stream_policy(-1)


_RESOLVED = {}


def _resolve(obj):
    """Dotted name -> object ("model.Transformer", "torch.optim.SGD"; helper.py resolves its config strings with
    pydoc.locate).  Not pydoc.locate itself: its safeimport() parks sys.exc_info() in a local, and that traceback <-> frame
    cycle keeps every CALLER frame -- the estimator being initialised, the whole list of a lockstep unit's estimators and
    their GPU arenas -- alive until the cyclic GC happens to run (measured: 1.5 GB per grid work unit, 170 GB peak)."""
    if not isinstance(obj, str):
        return obj
    if obj in _RESOLVED:
        return _RESOLVED[obj]
    parts = obj.split(".")
    found = None
    for i in range(len(parts) - 1, 0, -1):               # longest importable module prefix, then attributes
        try:
            found = importlib.import_module(".".join(parts[:i]))
        except ImportError:
            continue
        for name in parts[i:]:
            found = getattr(found, name, None)
            if found is None:
                break
        if found is not None:
            break
    if found is None and len(parts) == 1:
        try:
            found = importlib.import_module(obj)
        except ImportError:
            found = None
    if found is not None:
        _RESOLVED[obj] = found
    return found


class ScoringWrapper:
    """Named sklearn scorer with the extra keyword the reference gives each metric (helper.py:529-554): log-loss is told
    the full label set (a fold may miss classes), the precision / recall / F1 family gets ``zero_division=0``, accuracy
    takes nothing.  Exposes ``score`` (the name) and ``greater_is_better`` -- what the reference's EpochScoring and
    GridSearchCV wiring read (helper.py:255-268, 183-194)."""

    _EXTRA = {"neg_log_loss": lambda labels: {"labels": labels}, "accuracy": lambda labels: {}}

    def __init__(self, score_func, labels=None):
        from sklearn.metrics import get_scorer
        self.score = score_func
        base = get_scorer(score_func)
        extra = self._EXTRA.get(score_func, lambda labels: {"zero_division": 0})(labels)
        self.greater_is_better = base._sign > 0
        # a scorer object of the same kind with the merged keywords (get_scorer hands out a fresh copy per call)
        base._kwargs = {**base._kwargs, **extra}
        self.scorer = base

    def __call__(self, estimator, X, y_true, sample_weight=None):
        return self.scorer(estimator, X, y_true, sample_weight)

    def __repr__(self):
        return "%s('%s')" % (type(self).__name__, self.score)


from sklearn.base import BaseEstimator, ClassifierMixin  # noqa: E402  (sklearn >= 1.6 scorers require classifier tags)


class _CachedPredictor(ClassifierMixin, BaseEstimator):
    """What skorch's score caching gives EpochScoring: predictions already made during the epoch."""

    def __init__(self, proba, classes):
        self._proba, self.classes_ = proba, classes

    def predict_proba(self, X):
        return self._proba

    def predict(self, X):
        return self.classes_[self._proba.argmax(-1)]


class _FitRun:
    """One estimator's fit between epochs: the internal split, the device-resident data, and skorch's per-epoch callbacks
    (EpochScoring on the cached predictions, Checkpoint, LRScheduler(ReduceLROnPlateau), EarlyStopping; helper.py:197-273).
    ``partial_fit`` drives one of these with its own batch loop; ``slnlp.lockstep`` drives K of them with one launch
    sequence per step -- the epoch bookkeeping is this one piece of code either way."""

    def __init__(self, net, ds):
        self.net = net
        net.classes_ = np.arange(len(ds.vocab_y)) if ds.vocab_y is not None else np.arange(int(ds.y.max()) + 1)
        labels = net.labels if net.labels is not None else ds.labels()
        idx_tr, idx_va = net._train_split(ds)
        self.tr, self.va = ds[idx_tr], (ds[idx_va] if idx_va is not None else None)
        self.Xtr, self.Ltr, self.ytr = net._device_data(self.tr)
        if self.va is not None:
            self.Xva, self.Lva, self.yva = net._device_data(self.va)
        self.wrappers = [ScoringWrapper(s, labels) for s in (net.scoring or [])]
        # the fast metrics index the probability columns by class id: valid when the labels are exactly the columns
        self.fast_ok = labels is not None and list(labels) == list(range(len(net.classes_)))
        es, clip, sched = net.early_stopping, net.gradient_clipping, net.lr_scheduler
        self.es = es
        self.max_norm = float(clip["gradient_clip_value"]) if clip and clip.get("gradient_clip_value") else 0.0
        self.momentum = float(net._opt_kwargs.get("momentum", 0.0))
        self.plateau = None
        if sched:
            assert sched.get("policy", "ReduceLROnPlateau") == "ReduceLROnPlateau", "only ReduceLROnPlateau is wired"
            dummy = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=net.lr_)
            self.plateau = torch.optim.lr_scheduler.ReduceLROnPlateau(
                dummy, **{k: v for k, v in sched.items() if k not in ("policy", "monitor", "step_every")})
        self.best_valid, self.misses, self.dyn_thr = float("inf"), 0, float("inf")
        self.bs = int(net.batch_size)
        self.epochs_left = int(net.max_epochs)
        self.done = self.epochs_left <= 0

    def begin_epoch(self):
        self.t0 = time.time()

    def end_epoch(self, tr, va):
        """tr / va: (sample-weighted mean loss, log-probs [n, V] on the device, [(batch loss, batch size)]) of the epoch's
        train and valid passes (va None without a valid split).  Returns True when the fit is over."""
        net = self.net
        tr_loss, tr_logp, tr_batches = tr
        epoch = len(net.history) + 1
        row = {"epoch": epoch, "train_loss": tr_loss, "lr": net.lr_,
               "batches": [{"train_loss": l, "train_batch_size": n} for l, n in tr_batches]}   # skorch history layout
        if va is not None:
            va_loss, va_logp, va_batches = va
            row["batches"] += [{"valid_loss": l, "valid_batch_size": n} for l, n in va_batches]
            row["valid_loss"] = va_loss
            row["valid_loss_best"] = bool(va_loss < self.best_valid)
            self.best_valid = min(self.best_valid, va_loss)
        # EpochScoring on the epoch's cached predictions: the reference's five metrics from one device-side
        # reduction (slnlp/metrics.py, same numbers as the sklearn scorers); anything else through sklearn
        splits = [("train", tr_logp, self.ytr, self.tr)] + ([("valid", va_logp, self.yva, self.va)] if va is not None else [])
        names = [wr.score for wr in self.wrappers]
        fast = {sp: metrics.epoch_scores(names, lp, yd, part.y) if self.fast_ok and names else {} for sp, lp, yd, part in splits}
        proba = {}
        for wr in self.wrappers:
            for sp, lp, yd, part in splits:
                if wr.score in fast[sp]:
                    row[f"{sp}_{wr.score}"] = fast[sp][wr.score]
                    continue
                if sp not in proba:
                    proba[sp] = np.exp(lp.cpu().numpy())
                row[f"{sp}_{wr.score}"] = float(wr(_CachedPredictor(proba[sp], net.classes_), None, part.y))
        row["dur"] = time.time() - self.t0
        net.history.append(row)
        if net.verbose:
            print("  ".join(f"{k}={v:.4f}" if isinstance(v, float) else f"{k}={v}" for k, v in row.items() if k != "batches"))
        if net.checkpoint_dir and row.get("valid_loss_best"):
            net.save_params(net.checkpoint_dir)
        monitor = row.get("valid_loss", tr_loss)
        if self.plateau is not None:                         # LRScheduler(monitor=valid_loss, step_every=epoch)
            self.plateau.step(monitor)
            net._set_lr(self.plateau.optimizer.param_groups[0]["lr"])
        self.epochs_left -= 1
        if self.es:                                          # skorch EarlyStopping, lower_is_better
            es = self.es
            if monitor < self.dyn_thr:
                self.misses = 0
                thr = float(es.get("threshold", 1e-4))
                self.dyn_thr = monitor - (thr * monitor if es.get("threshold_mode", "rel") == "rel" else thr)
            else:
                self.misses += 1
            if self.misses == int(es.get("patience", 5)):
                if net.verbose:
                    print(f"Stopping since valid_loss has not improved in the last {self.misses} epochs.")
                self.done = True
        if self.epochs_left <= 0:
            self.done = True
        return self.done


class NeuralNetClassifier(ClassifierMixin, BaseEstimator):
    _OWN = ("module", "criterion", "optimizer", "lr", "max_epochs", "batch_size", "device", "warm_start", "verbose",
            "predict_nonlinearity", "scoring", "labels", "early_stopping", "gradient_clipping", "lr_scheduler",
            "checkpoint_dir", "train_split", "use_graph", "callbacks", "dataset")

    def __init__(self, module, criterion="torch.nn.CrossEntropyLoss", optimizer="torch.optim.SGD", lr=0.01,
                 max_epochs=10, batch_size=128, device="cuda", warm_start=False, verbose=0,
                 predict_nonlinearity="auto", scoring=None, labels=None, early_stopping=None,
                 gradient_clipping=None, lr_scheduler=None, checkpoint_dir=None, train_split=5, use_graph="auto",
                 callbacks=None, dataset=None, **kwargs):
        loc = locals()
        self._params = {k: loc[k] for k in self._OWN}
        for k, v in kwargs.items():
            if not any(k.startswith(p) for p in ("module__", "optimizer__", "criterion__", "iterator_train__",
                                                 "iterator_valid__", "callbacks__")):
                raise TypeError(f"NeuralNetClassifier: unexpected argument {k!r}")
            self._params[k] = v
        self.initialized_ = False
        self.history = []
        self._apply_callbacks(callbacks)

    # skorch callback objects (helper.py:197-273 builds Checkpoint, EarlyStopping, GradientNormClipping, LRScheduler and
    # EpochScoring instances) are translated, by class name and public attributes, into the settings this loop
    # implements natively; a callback it cannot honour is an error, never silently dropped.
    _COSMETIC_CALLBACKS = ("PrintLog", "ProgressBar", "EpochTimer", "PassthroughScoring")

    def _apply_callbacks(self, callbacks):
        if callbacks in (None, "disable", []):
            return
        scoring = list(self._params.get("scoring") or [])
        for cb in callbacks:
            obj = cb[1] if isinstance(cb, tuple) else cb
            kind = type(obj).__name__
            get = lambda k, d=None: getattr(obj, k, d)
            if kind == "EarlyStopping":
                if get("monitor", "valid_loss") != "valid_loss" or not get("lower_is_better", True):
                    raise ValueError("EarlyStopping: only monitor='valid_loss', lower_is_better=True is implemented")
                self._params["early_stopping"] = {"patience": get("patience", 5), "threshold": get("threshold", 1e-4),
                                                  "threshold_mode": get("threshold_mode", "rel")}
            elif kind == "GradientNormClipping":
                self._params["gradient_clipping"] = {"gradient_clip_value": get("gradient_clip_value")}
            elif kind == "LRScheduler":
                pol = get("policy", "ReduceLROnPlateau")
                pol = pol if isinstance(pol, str) else getattr(pol, "__name__", str(pol))
                if pol != "ReduceLROnPlateau" or get("monitor", "valid_loss") != "valid_loss":
                    raise ValueError("LRScheduler: only ReduceLROnPlateau on valid_loss is implemented")
                self._params["lr_scheduler"] = {"policy": pol, **dict(get("kwargs", {}) or {})}
            elif kind == "Checkpoint":
                if get("monitor", "valid_loss_best") != "valid_loss_best":
                    raise ValueError("Checkpoint: only monitor='valid_loss_best' is implemented")
                self._params["checkpoint_dir"] = get("dirname")
            elif kind == "EpochScoring":
                sc = get("scoring")
                name = sc if isinstance(sc, str) else getattr(sc, "score", None)
                if get("name") == "lr":
                    continue                                  # every history row carries lr already
                if not isinstance(name, str):
                    raise ValueError("EpochScoring: scoring must be a metric name or a ScoringWrapper")
                if name not in scoring:
                    scoring.append(name)                      # both splits are scored for every metric
            elif kind in self._COSMETIC_CALLBACKS:
                continue
            else:
                raise TypeError(f"NeuralNetClassifier: callback {kind!r} has no equivalent in the fused fit loop")
        if scoring:
            self._params["scoring"] = scoring

    # ------------------------------------------------------------ sklearn API
    def get_params(self, deep=True):
        return dict(self._params)

    def set_params(self, **params):
        for k, v in params.items():
            self._params[k] = v
        if params.get("callbacks") is not None:
            self._apply_callbacks(params["callbacks"])
        return self

    def __getattr__(self, name):
        p = self.__dict__.get("_params", {})
        if name in p:
            return p[name]
        raise AttributeError(name)

    def _sub(self, prefix):
        n = len(prefix) + 2
        return {k[n:]: v for k, v in self._params.items() if k.startswith(prefix + "__")}

    # ------------------------------------------------------------- lifecycle
    def initialize(self):
        dev = torch.device(self.device)
        if dev.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("slnlp.net: device %r -- the HIP path is the only compute path (no CPU fallback)" % (self.device,))
        kw = self._sub("module")
        kw.setdefault("device", dev)
        self.criterion_ = _resolve(self.criterion)(**self._sub("criterion"))
        self._opt_cls = _resolve(self.optimizer)
        ok = self._opt_kwargs = self._sub("optimizer")
        mod_cls = _resolve(self.module)
        ce = isinstance(self.criterion_, torch.nn.CrossEntropyLoss) and hasattr(mod_cls, "engine")
        self._fused_kind = None                   # which fused clip + update kernel replaces the torch optimizer
        if ce and self._opt_cls is torch.optim.SGD and not ok.get("nesterov", False) and not ok.get("weight_decay", 0) \
                and not ok.get("dampening", 0) and not ok.get("maximize", False):
            self._fused_kind = "sgd"
        elif ce and self._opt_cls is torch.optim.Adam and not ok.get("amsgrad", False) and not ok.get("maximize", False):
            self._fused_kind = "adam"
        self._fused = self._fused_kind is not None
        # a fused fit may take this host thread's own stream; one that steps through torch kernels stays on the device's shared
        # stream and holds the device exclusively while it runs (see STREAM_MODE above)
        self._stream = device_stream(dev, per_thread=(STREAM_MODE == "thread" and self._fused))
        self._gate = device_gate(dev)
        with torch.cuda.stream(self._stream):            # the weight draw / upload too: nothing of a fit runs on another queue
            self.module_ = mod_cls(**kw).to(dev)
        if not self._fused:
            self.optimizer_ = self._opt_cls(self.module_.parameters(), lr=self.lr, **ok)
        self.lr_ = float(self.lr)
        self.history = []
        self.initialized_ = True
        return self

    def _enter_stream(self):
        """Order the fit's stream behind what this thread queued on the ambient stream so far (the weight draw / upload of
        ``initialize``, the dataset upload): ``self._stream`` is a non-blocking stream, nothing else makes it wait."""
        self._stream.wait_stream(torch.cuda.current_stream(self._stream.device))

    # ------------------------------------------------------------------ data
    @staticmethod
    def _as_dataset(X, y=None):
        if isinstance(X, TokenDataset):
            return X
        if isinstance(X, dict):
            return TokenDataset(X["X"], X["lengths"], X["y"] if y is None else y)
        raise TypeError("X must be a slnlp.data.TokenDataset (ids, lengths and labels travel together: the "
                        "Transformer consumes y as decoder input, transformer.py:65)")

    def _device_data(self, ds):
        dev = self.module_._arena.device if hasattr(self.module_, "_arena") else torch.device(self.device)
        return (torch.from_numpy(ds.ids).to(dev), torch.from_numpy(ds.lengths).to(dev), torch.from_numpy(ds.y).to(dev))

    # ------------------------------------------------------------------- fit
    def fit(self, X, y=None, **fit_params):
        if not (self.warm_start and self.initialized_):
            self.initialize()
        return self.partial_fit(X, y, **fit_params)

    def partial_fit(self, X, y=None, **fit_params):
        if not self.initialized_:
            self.initialize()
        self._gate.enter(not self._fused)
        try:
            return self._partial_fit_gated(X, y)
        finally:
            self._gate.leave(not self._fused)

    def _partial_fit_gated(self, X, y):
        self._enter_stream()
        with torch.cuda.stream(self._stream):
            run = _FitRun(self, self._as_dataset(X, y))
            for _ in range(int(self.max_epochs)):
                run.begin_epoch()
                self.module_.train()
                tr = self._run_epoch(run.Xtr, run.Ltr, run.ytr, run.bs, True, run.momentum, run.max_norm)
                va = None
                if run.va is not None:
                    self.module_.eval()
                    va = self._run_epoch(run.Xva, run.Lva, run.yva, run.bs, False, run.momentum, run.max_norm)
                if run.end_epoch(tr, va):
                    break
        stream_sync(self._stream)
        return self

    def _train_split(self, ds):
        ts = self.train_split
        if not ts:
            return np.arange(len(ds)), None
        from sklearn.model_selection import KFold, StratifiedKFold
        idx = np.arange(len(ds))
        try:
            tr, va = next(iter(StratifiedKFold(n_splits=int(ts)).split(idx, ds.y)))
        except ValueError:                                   # a class with fewer members than folds
            tr, va = next(iter(KFold(n_splits=int(ts)).split(idx)))
        return tr, va

    def _set_lr(self, lr):
        self.lr_ = float(lr)
        if not self._fused:
            for g in self.optimizer_.param_groups:
                g["lr"] = self.lr_

    def _run_epoch(self, X, L, y, bs, train, momentum, max_norm):
        """One pass in dataset order.  Returns (sample-weighted mean loss, log-probs [N,V] on the device,
        [(batch loss, batch size)])."""
        n = X.shape[0]
        losses, sizes, outs = [], [], []
        for i in range(0, n, bs):
            xb, lb, yb = X[i:i + bs], L[i:i + bs], y[i:i + bs]
            if train and self._fused:
                eng = self.module_.engine(xb.shape[0], xb.shape[1])
                eng.set_lr(self.lr_)
                if self._fused_kind == "adam":
                    ok = self._opt_kwargs
                    logp = eng.train_step_adam(xb, yb, self.module_.adam_second_moment(), tuple(ok.get("betas", (0.9, 0.999))),
                                               float(ok.get("eps", 1e-8)), float(ok.get("weight_decay", 0.0)), max_norm, lengths=lb)
                else:
                    logp = eng.step(xb, yb, lb, momentum, max_norm, graph=self.use_graph if self.use_graph == "auto" else bool(self.use_graph))
                losses.append(eng.scalars[0].clone())
            elif train:
                self.optimizer_.zero_grad()
                logp = self.module_(X=xb, y=yb, lengths=lb)
                loss = self.criterion_(logp, yb)
                loss.backward()
                if max_norm:
                    torch.nn.utils.clip_grad_norm_(self.module_.parameters(), max_norm)
                self.optimizer_.step()
                losses.append(loss.detach())
            else:
                with torch.no_grad():
                    logp = self.module_(X=xb, y=yb, lengths=lb)
                    if hasattr(self.module_, "engine") and isinstance(self.criterion_, torch.nn.CrossEntropyLoss):
                        losses.append(self.module_.engine(xb.shape[0], xb.shape[1]).scalars[0].clone())
                    else:
                        losses.append(self.criterion_(logp, yb))
            sizes.append(xb.shape[0])
            outs.append(logp.detach().clone())
        per_batch = torch.stack(losses).float().cpu()                       # one sync per epoch
        w = torch.tensor(sizes, dtype=torch.float32)
        mean = float((per_batch * w).sum() / w.sum())
        return mean, torch.cat(outs), list(zip(per_batch.tolist(), sizes))

    # --------------------------------------------------------------- predict
    def predict_proba(self, X):
        """softmax of the module output -- the module returns log-probs and skorch's
        ``predict_nonlinearity='auto'`` applies softmax for CrossEntropyLoss."""
        if not self.initialized_:
            raise RuntimeError("This NeuralNetClassifier instance is not initialized yet.")
        ds = self._as_dataset(X)
        self.module_.eval()
        outs = []
        self._gate.enter(not self._fused)
        try:
            self._enter_stream()
            with torch.cuda.stream(self._stream), torch.no_grad():
                Xd, Ld, yd = self._device_data(ds)
                for i in range(0, len(ds), int(self.batch_size)):
                    outs.append(self.module_(X=Xd[i:i + self.batch_size], y=yd[i:i + self.batch_size], lengths=Ld[i:i + self.batch_size]))
                out = torch.cat(outs)
            stream_sync(self._stream)
        finally:
            self._gate.leave(not self._fused)
        out = out.cpu()
        # the nonlinearity on the host copy (torch's CPU softmax -- the op the reference's CPU path runs): torch's GPU kernels
        # are built with packed fp32 and must not run beside other fits' kernels (STREAM_MODE above); [N, V] is tiny
        return (torch.softmax(out, dim=-1) if self.predict_nonlinearity == "auto" else out).numpy()

    def predict(self, X):
        return self.classes_[self.predict_proba(X).argmax(-1)]

    def score(self, X, y=None):
        ds = self._as_dataset(X)
        return float((self.predict(ds) == (ds.y if y is None else np.asarray(y))).mean())

    # ------------------------------------------------------------ checkpoint
    def _sgd_state_dict(self):
        """The fused update's momentum arena as a ``torch.optim.SGD.state_dict()``: one ``momentum_buffer`` per parameter,
        in ``module.parameters()`` order, plus the param group with the current lr -- what skorch's Checkpoint writes as
        optimizer.pt (helper.py:211-213) and what ``torch.optim.SGD.load_state_dict`` reads back."""
        params = dict(self.module_.named_parameters())
        opt = self._opt_cls([p for n, p in params.items()], lr=self.lr_, **self._opt_kwargs)
        st = self.module_._shared_state()
        mom = st["momentum"]
        adam = self._fused_kind == "adam"
        step = None
        if adam:
            step = float(st["scalars"][2])
            v2 = self.module_.adam_second_moment()
        for name, shape, off in self.module_._entries:
            if name in self.module_._dead_params:
                continue                                   # never receives a gradient: torch keeps no state for it
            n = 1
            for d in shape:
                n *= d
            view = lambda arena: arena[off:off + n].view(*shape).detach().cpu().clone()
            if adam:
                opt.state[params[name]].update(step=torch.tensor(step), exp_avg=view(mom), exp_avg_sq=view(v2))
            else:
                opt.state[params[name]]["momentum_buffer"] = view(mom)
        return opt.state_dict()

    def _load_sgd_state_dict(self, sd):
        names = [n for n, _ in self.module_.named_parameters()]
        ent = {n: (shape, off) for n, shape, off in self.module_._entries}
        st = self.module_._shared_state()
        step = None
        for idx, state in sd.get("state", {}).items():
            shape, off = ent[names[int(idx)]]
            buf = state.get("momentum_buffer", state.get("exp_avg"))
            if buf is not None:
                st["momentum"][off:off + buf.numel()].copy_(buf.reshape(-1).to(st["momentum"].device, torch.float32))
            if state.get("exp_avg_sq") is not None:
                v2 = self.module_.adam_second_moment()
                v2[off:off + buf.numel()].copy_(state["exp_avg_sq"].reshape(-1).to(v2.device, torch.float32))
                step = float(state.get("step", 0.0))
        if step is not None:                           # the device-side Adam step count (shared by every plan of the module)
            st["scalars"][2] = step
        groups = sd.get("param_groups") or [{}]
        if "lr" in groups[0]:
            self._set_lr(groups[0]["lr"])

    def save_params(self, dirname):
        """skorch ``Checkpoint`` artefacts: params.pt (state_dict), optimizer.pt (a torch.optim state_dict in either
        mode), criterion.pt, history.json."""
        os.makedirs(dirname, exist_ok=True)
        torch.save({k: v.detach().cpu() for k, v in self.module_.state_dict().items()}, os.path.join(dirname, "params.pt"))
        torch.save(self._sgd_state_dict() if self._fused else self.optimizer_.state_dict(), os.path.join(dirname, "optimizer.pt"))
        torch.save(self.criterion_.state_dict(), os.path.join(dirname, "criterion.pt"))
        with open(os.path.join(dirname, "history.json"), "w") as f:
            json.dump(self.history, f, indent=1)

    def load_params(self, dirname):
        """Restore what ``save_params`` / skorch's Checkpoint wrote: weights, optimizer state (momentum buffers + lr) and,
        when present, the history -- training resumes where the checkpoint was taken."""
        if not self.initialized_:
            self.initialize()
        self.module_.load_state_dict(torch.load(os.path.join(dirname, "params.pt")))
        opt_file = os.path.join(dirname, "optimizer.pt")
        if os.path.exists(opt_file):
            sd = torch.load(opt_file)
            if self._fused:
                self._load_sgd_state_dict(sd)
            else:
                self.optimizer_.load_state_dict(sd)
                self.lr_ = float(self.optimizer_.param_groups[0]["lr"])
        hist = os.path.join(dirname, "history.json")
        if os.path.exists(hist):
            with open(hist) as f:
                self.history = json.load(f)
        return self
