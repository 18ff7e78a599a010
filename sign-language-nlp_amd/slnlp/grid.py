"""Cross-validated grid search sharded over the GPUs of one node.

The reference farms the (candidate x fold) fits of ``GridSearchCV`` out to dask
workers, one per GPU, pickling the whole dataset with every task; dask hands a
worker its next task when it falls idle (/root/reference/main.py:62-95,
helper.py:108-180, 490-526; 324 candidates x 5 folds for config-transformer.yaml).
Here: one process per GPU (``torch.distributed``, RCCL on GPUs / gloo in CPU tests);

* rank 0 broadcasts the integer dataset ONCE (``X[N,S]``, ``lengths[N]``, ``y[N]``),
* every rank derives the identical work list (``ParameterGrid`` x ``StratifiedKFold(cv)``, no shuffle), sorted
  by estimated cost, longest first,
* **dynamic distribution with admission control** (``WorkCounter``): the ``world`` longest units are dealt one per rank;
  after that a rank that falls idle pulls the index of its next work unit from ONE shared counter -- ``store.add`` on
  the process group's rendezvous store across ranks, a locked integer inside one process -- so a rank whose fits stop
  early (``EarlyStopping``) simply takes more units; the extra ``fits_per_gpu`` host threads of a busy rank prefetch a
  unit only while plenty are left and the rank stays within its fair share of the estimated work.
  ``schedule="static"`` keeps the round-robin deal ``i % world``,
* one ``all_gather`` of the per-task score rows at the end; every rank computes ``mean_test_score`` /
  ``rank_test_score`` / ``best_index_`` identically and rank 0 refits the best candidate on the whole training
  set (``refit=True``).  The refit needs every score, so it cannot start before the last fit ends; it is one
  fit out of 1 621.

There is no per-step collective: fits are independent.  Every task is seeded by its
index, so ``cv_results_`` does not depend on the world size, on ``fits_per_gpu`` or on which rank ran what.

A **work unit** is a list of tasks.  With ``lockstep=k`` tasks that share every shape-defining parameter (and the
fold sizes) are packed k to a unit and handed to ``fit_and_score_group`` in one call, which advances them through ONE
launch sequence (slnlp.lockstep); otherwise a unit is one task.

A task that raises does not strand the other ranks in the collective: its row carries NaN and an error flag, a shared
abort flag stops every rank from pulling further units (the reference searches with ``error_score='raise'``), the gather
completes, and then EVERY rank raises.
"""
import itertools
import threading
import time

import numpy as np

from .data import TokenDataset

_FIT_CALLS = itertools.count()       # distinguishes the store keys of successive fit() calls in one process group


def parameter_grid(param_grid):
    """sklearn.model_selection.ParameterGrid order: keys sorted, last key varies fastest."""
    keys = sorted(param_grid)
    return [dict(zip(keys, vals)) for vals in itertools.product(*[param_grid[k] for k in keys])]


# estimate_cost: fixed work per batch and per layer, in the units of the arithmetic term (multiply-adds per sequence).  Measured on
# MI355X (bench.py's grid sample, one lockstep-15 unit alone on the GPU, profiles/r04_grid_calibration.json): E512 N4 / E512 N2 /
# E128 N4 / E128 N2 units take 4.9 / 2.9 / 2.4 / 1.45 s where the arithmetic alone says 1 : 0.5 : 0.13 : 0.065 -- a step is a chain of
# dependent launches whose length grows with the number of layers, not with their width.  Least squares over the four classes:
COST_PER_STEP, COST_PER_LAYER = 13.7e6, 9.5e6


def estimate_cost(params, seq_len=48, n_samples=1, defaults=None):
    """Relative cost of one fit -- what orders the work list, sizes the work units and feeds the admission control: samples x
    epochs x (fixed + layers x (fixed + per-sequence arithmetic ~ S x (E^2 + E x F) + S^2 x E)) (SURVEY.md section 8e;
    attention's S^2 term matters once S > E).  The two fixed terms are the launch-bound floor of a step (above): without them a
    small model's fit is under-estimated five-fold against a large one's."""
    d = dict(defaults or {})
    d.update(params)
    E = d.get("module__embedding_size", 128)
    F = d.get("module__hidden_size", 256)
    N = d.get("module__num_layers", 2)
    epochs = d.get("max_epochs", 1)
    arith = seq_len * (E * E + E * F) + seq_len * seq_len * E
    return float(n_samples) * float(epochs) * (COST_PER_STEP + float(N) * (COST_PER_LAYER + arith))


def build_tasks(param_grid, y, cv, seq_len=48, defaults=None):
    from sklearn.model_selection import StratifiedKFold
    cands = parameter_grid(param_grid)
    folds = list(StratifiedKFold(n_splits=cv).split(np.zeros(len(y)), y))
    tasks = [(ci, fi) for ci in range(len(cands)) for fi in range(len(folds))]
    cost = lambda t: estimate_cost(cands[tasks[t][0]], seq_len, len(folds[tasks[t][1]][0]), defaults)
    order = sorted(range(len(tasks)), key=lambda t: (-cost(t), t))
    return cands, folds, tasks, order


SHAPE_KEYS_EXCLUDED = ("lr", "module__dropout")


def estimate_fit_bytes(params, seq_len, defaults=None, lockstep=1):
    """Device bytes one resident fit of this candidate needs (plan workspace + parameter / gradient / momentum arenas + its
    share of a ``lockstep``-wide group's table workspace), asked from the library's host-side size queries; None when the
    module is not one of this package's."""
    d = dict(defaults or {})
    d.update(params)
    try:
        import ctypes as C
        from ._lib import load
        from .net import _resolve
        name = getattr(_resolve(d.get("module")), "__name__", "")
        Vs, Vt, B = len(d["module__src_vocab"]), len(d["module__tgt_vocab"]), int(d.get("batch_size", 50))
        E, F, N = int(d["module__embedding_size"]), int(d["module__hidden_size"]), int(d["module__num_layers"])
        if name == "Transformer":
            from . import tf_engine as te
            cfg = te.make_config(E, int(d["module__num_heads"]), N, F, Vs, Vt, B, int(seq_len), 1, 1, 0.1, 3)
            ws, arena = int(load().slnlp_tf_workspace_bytes(C.byref(cfg))), te.layout(cfg)[1]
            if lockstep > 1:
                ws += int(load().slnlp_tf_lockstep_workspace_bytes(C.byref(cfg), int(lockstep))) // int(lockstep)
        elif name.startswith("EncoderDecoder"):
            from . import rnn_engine as re_
            cfg = re_.make_config("lstm" if "LSTM" in name else "gru", E, F, N, Vs, Vt, B, int(seq_len), 1, 1, 0, 0.1, 3)
            ws, arena = int(load().slnlp_rnn_workspace_bytes(C.byref(cfg))), re_.layout(cfg)[1]
            if lockstep > 1:
                ws += int(load().slnlp_rnn_lockstep_workspace_bytes(C.byref(cfg), int(lockstep))) // int(lockstep)
        else:
            return None
        return ws + 3 * 4 * int(arena) if ws > 0 else None
    except Exception:
        return None


def build_units(cands, folds, tasks, order, lockstep=1, cap=None, task_cost=None, max_unit_cost=None, min_width=4):
    """Pack the cost-ordered task list into work units of up to ``lockstep`` tasks that can advance through one
    launch sequence: same candidate shapes (every parameter except lr / dropout rate) and the same train-fold
    size (=> the same number and sizes of batches).  ``lockstep <= 1``: one task per unit.  ``cap(candidate index)``: an upper
    bound on the unit size for that candidate (device memory), or None.

    ``max_unit_cost`` (with ``task_cost(task)``): a ceiling on a unit's estimated cost -- how the search keeps at least a
    couple of units per host thread and GPU when the grid is spread over many GPUs (``unit_cost_ceiling``): a compatible group
    whose cost exceeds it is cut into the fewest EQUAL parts that fit under it, never narrower than ``min_width`` tasks (a
    narrow lockstep unit fills the GPU less well).  Groups are cut evenly (15 tasks under a limit of 4 -> 4 + 4 + 4 + 3), and
    with ``task_cost`` the units are returned by estimated cost, longest first (ties: position of the first task) -- the
    order the ``WorkCounter`` hands them out in; without it a unit keeps the position of its first task."""
    if lockstep <= 1:
        return [[t] for t in order]
    groups = {}
    for t in order:
        ci, fi = tasks[t]
        drop = cands[ci].get("module__dropout")
        key = (tuple(sorted((k, repr(v)) for k, v in cands[ci].items() if k not in SHAPE_KEYS_EXCLUDED)),
               None if drop is None else bool(drop > 0),          # dropout on / off changes the launch sequence, its rate does not
               len(folds[fi][0]), len(folds[fi][1]))
        groups.setdefault(key, []).append(t)
    units = []
    for members in groups.values():
        limit = lockstep
        if cap is not None:
            limit = min(limit, max(1, min(cap(tasks[t][0]) for t in members)))
        if max_unit_cost is not None and task_cost is not None:
            dearest = max(task_cost(t) for t in members)
            if dearest > 0:
                limit = min(limit, max(min(min_width, limit), int(max_unit_cost // dearest)))
        parts = -(-len(members) // limit)
        base, extra = divmod(len(members), parts)
        at = 0
        for p in range(parts):
            n = base + (1 if p < extra else 0)
            units.append(members[at:at + n])
            at += n
    pos = {t: i for i, t in enumerate(order)}
    if task_cost is not None:
        units.sort(key=lambda u: (-sum(task_cost(t) for t in u), pos[u[0]]))
    else:
        units.sort(key=lambda u: pos[u[0]])
    return units


# unit_cost_ceiling: aim for at least this many work units per host thread of every GPU.  slnlp.grid_sim on bench.py's sample with
# the unit durations and the concurrency gains measured on one MI355X (profiles/r04_grid_calibration.json): strong-scaling
# efficiency at 8 GPUs 0.82 without cutting (32 units: every GPU ends with its one long unit alone on the card), 0.90 at 2 per
# thread, 0.95 at 3, 0.97 at 6 (124 units of 4-5 fits; 0.98 at 2 and 4 GPUs) -- the tail of the schedule shrinks with the last units.
# Narrow units cost nothing once four run side by side: lockstep 5 x 4 threads = lockstep 15 x 4 threads = 27.2 k folds/hr measured.
UNITS_PER_THREAD = 6.0


def unit_cost_ceiling(total_cost, world, fits_per_gpu, units_per_thread=UNITS_PER_THREAD):
    """The estimated cost above which a lockstep group is cut into several units: ``total / (world x fits_per_gpu x
    units_per_thread)``.  With 4 units per GPU (bench.py's 32 lockstep-15 units on 8 GPUs) every GPU ends its run with its one
    long unit alone on the card -- measured on one GPU, a single resident unit delivers about 60 % of what three or four side
    by side do -- so the long groups are cut until every host thread has a couple of units to work
    through and the tail of the schedule is a short unit, not a long one."""
    return float(total_cost) / (max(1, world) * max(1, fits_per_gpu) * units_per_thread)


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def comm_device(device="cpu"):
    """Where collective payloads must live: the GPU for RCCL ("nccl"), host memory for gloo."""
    dist, _, _ = _dist()
    if dist is not None and dist.get_backend() == "nccl":
        import torch
        dev = torch.device(device)
        return dev if dev.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
    return "cpu"


def broadcast_dataset(ds, device="cpu", src=0, force=False):
    """Rank ``src`` holds the dataset; everyone else receives it (ONE broadcast of a packed int64 buffer
    over RCCL/xGMI on GPUs: ~4 MB at N=10k, latency-bound).  ``force``: run the collectives even in a
    one-rank group (exercises the RCCL path on a single GPU)."""
    import torch
    dist, rank, world = _dist()
    if dist is None or (world == 1 and not force):
        return ds
    device = comm_device(device)
    meta = torch.zeros(4, dtype=torch.int64, device=device)
    if rank == src:
        meta[:] = torch.tensor([len(ds), ds.ids.shape[1], len(ds.vocab_X) if ds.vocab_X else 0,
                                len(ds.vocab_y) if ds.vocab_y else 0])
    dist.broadcast(meta, src)
    n, s, vx, vy = [int(v) for v in meta.cpu()]
    buf = torch.empty(n * (s + 2), dtype=torch.int64, device=device)
    if rank == src:
        buf[:] = torch.from_numpy(np.concatenate([ds.ids.ravel(), ds.lengths, ds.y])).to(device)
    dist.broadcast(buf, src)
    if rank == src and not force:
        return ds
    from model.util import Vocab
    b = buf.cpu().numpy()
    keep = ds if rank == src else None
    return TokenDataset(b[:n * s].reshape(n, s), b[n * s:n * s + n], b[n * s + n:],
                        keep.vocab_X if keep else (Vocab(vx) if vx else None),
                        keep.vocab_y if keep else (Vocab(vy) if vy else None))


class WorkCounter:
    """Hands out the unit indices 0 .. n_units-1, each exactly once across every worker thread of every rank.

    ``schedule="dynamic"`` with **admission control** (the reference gives each GPU exactly ONE dask worker, and dask hands a
    worker its next task when it falls idle -- /root/reference/helper.py:503-519, main.py:70-78; a rank here has
    ``fits_per_gpu`` host threads whose units all serialise on the rank's one stream, so a unit taken early by an extra
    thread is a unit another, idle GPU cannot run):

    * the first ``world`` units -- the longest: the list is sorted longest first -- are dealt one per rank (unit r to rank r);
    * after that a rank with nothing in flight always gets the next unit from the shared counter (``store.add`` on the process
      group's rendezvous store; a locked integer inside one process);
    * a rank that already runs k units (on its other host threads) gets another one only while that keeps its estimated load --
      the sum of ``unit_costs`` of what it has taken -- within its fair share ``sum(unit_costs) / world`` (+ ``slack`` mean unit
      costs): prefetching for the host threads never takes what the longest-first deal would give to a less loaded rank.
      Otherwise its extra threads wait (polling the counter with a back-off); they leave when the list is exhausted.  Round 3
      also required more than ``world * (k - 1)`` units to be left (``reserve=True`` keeps that rule for comparison): with 32
      units on 8 ranks it made every GPU finish its longest unit alone -- slnlp.grid_sim prices that at 0.82 of ideal; the
      work list is now cut finer instead (``unit_cost_ceiling``) and the fair share alone admits a unit (0.97).
    * the decision reads the shared counter and this rank's load without a global lock: two ranks may both pass the test for
      the same next unit, and the second then receives the one after it -- the estimate it was admitted on is off by one unit
      of a cost-sorted list, which the fair-share bound tolerates.

    ``abort()`` (first failed unit, any rank) makes every later ``acquire`` on every rank return None, so a broken grid stops
    at once instead of burning through the remaining units (the reference searches with ``error_score='raise'``,
    helper.py:162,193).  ``static``: the round-robin deal ``i % world`` without a counter.
    """

    POLL_S, POLL_MAX_S = 0.02, 0.2      # a waiting host thread asks again after 20 ms, backing off to 200 ms (one store request each)
    ABORT = 1 << 40                       # added to the shared counter by abort(): one request returns {units taken, aborted}

    def __init__(self, key, n_units, static=False, unit_costs=None, rank=None, world=None, store=None, reserve=False,
                 slack=0.0):
        """``rank`` / ``world`` / ``store`` given explicitly: a counter outside a process group (slnlp.grid_sim drives one per
        simulated rank against a shared in-memory store; tests)."""
        if rank is None:
            dist, self.rank, self.world = _dist()
        else:
            self.rank, self.world = int(rank), int(world)
        self._lock, self._n, self._store, self._key = threading.Lock(), 0, store, key
        self.n_units, self.static = int(n_units), static
        self._costs = [float(c) for c in unit_costs] if unit_costs is not None else [1.0] * self.n_units
        self._fair, self._load = sum(self._costs) / max(1, self.world), 0.0
        self._in_flight, self._dealt, self._aborted = 0, False, False
        # reserve: round 3's extra rule (a rank running k units pulls only while more than world x (k - 1) are left) -- off
        # since round 4: it left every GPU to finish its longest unit alone (grid_sim: 0.92 of ideal at 8 GPUs; 0.97 without).
        # slack: how far beyond its fair share (in units of the mean unit cost) a busy rank may prefetch.
        self._reserve = bool(reserve)
        self._slack = float(slack) * (sum(self._costs) / max(1, self.n_units))
        if self.world > 1 and store is None:
            try:
                from torch.distributed.distributed_c10d import _get_default_store
                self._store = _get_default_store()
            except Exception:                      # no rendezvous store reachable: fall back to the static deal
                self.static = True

    # ---- shared state: units handed out beyond the initial deal; the abort flag rides in the same counter (bit 40)
    def _taken(self, inc):
        if self._store is not None and not self.static:
            v = int(self._store.add(self._key, inc))
        else:
            with self._lock:
                self._n += inc
                v = self._n
        if v >= self.ABORT:
            self._aborted = True
        return v % self.ABORT

    def aborted(self):
        if not self._aborted and self._store is not None:
            if self.static:
                self._aborted = int(self._store.add(self._key + "/abort", 0)) > 0
            else:
                self._taken(0)
        return self._aborted

    def abort(self):
        self._aborted = True
        if self._store is not None:
            self._store.add(self._key + "/abort" if self.static else self._key, 1 if self.static else self.ABORT)

    def release(self):
        with self._lock:
            self._in_flight -= 1

    def cost(self, i):
        return self._costs[i]

    def acquire(self):
        """Index of this thread's next unit, or None when there is nothing (more) to run.  Pair with release()."""
        if self.static:
            if self.aborted():
                return None
            with self._lock:
                i = self._n
                self._n += 1
                self._in_flight += 1
            i = i * self.world + self.rank
            if i >= self.n_units:
                self.release()
                return None
            return i
        wait = self.POLL_S
        while True:
            i = self.try_acquire()
            if i is not WorkCounter.WAIT:
                return i
            time.sleep(wait)
            wait = min(self.POLL_MAX_S, wait * 1.5)

    WAIT = object()

    def try_acquire(self):
        """One admission decision: a unit index, None (nothing more to run) or WAIT (ask again later)."""
        dealt = min(self.world, self.n_units)       # units 0 .. dealt-1: one per rank
        while True:
            if self._aborted:
                return None
            with self._lock:
                if not self._dealt:
                    self._dealt = True
                    if self.rank < dealt:
                        self._in_flight += 1
                        self._load += self._costs[self.rank]
                        return self.rank
                k = self._in_flight
                if k == 0:
                    self._in_flight += 1            # reserve: an idle rank always pulls
            if k == 0:
                return self._pull(dealt)
            nxt = dealt + self._taken(0)            # the unit the counter would hand out now (one request: it carries the abort flag)
            left = self.n_units - nxt
            if left <= 0 or self._aborted:
                return None
            if (not self._reserve or left > self.world * (k - 1)) and \
                    self._load + self._costs[nxt] <= self._fair * (1 + 1e-9) + self._slack:
                with self._lock:
                    if self._in_flight != k:
                        continue                    # another thread of this rank moved meanwhile: look again
                    self._in_flight += 1
                return self._pull(dealt)
            return WorkCounter.WAIT

    def _pull(self, dealt):                         # (the caller has reserved the in-flight slot)
        i = dealt + self._taken(1) - 1
        if i >= self.n_units or self._aborted:
            self.release()
            return None
        with self._lock:
            self._load += self._costs[i]
        return i


def default_fit_and_score(estimator_factory, params, train, test, scoring="neg_log_loss", seed=None, concurrent=False):
    """sklearn ``_fit_and_score`` for one task on this rank's GPU: fresh estimator, fit on the
    train fold, score on the test fold.  Module construction draws the initial weights from torch's global CPU
    generator (like the reference's modules), so it is seeded and serialised under a lock."""
    import torch
    from .net import INIT_LOCK, ScoringWrapper
    net = estimator_factory().set_params(**params)
    if "checkpoint_dir" in net.get_params():
        net.set_params(checkpoint_dir=None)      # CV fits leave no files; the refit of the best candidate does
    if concurrent and "use_graph" in net.get_params():
        # hipGraph capture on one host thread makes device-wide calls of the other threads fail
        # ("operation not permitted when stream is capturing"): concurrent fits use plain stream launches
        net.set_params(use_graph=False)
    with INIT_LOCK:
        if seed is not None:
            torch.manual_seed(seed)
        net.initialize()
    if concurrent and hasattr(net.module_, "_make_engine"):
        # persistent kernels (all their workgroups co-resident, spinning at device-wide barriers) need the GPU to
        # themselves: concurrent fits must use the per-timestep launches
        net.module_.persistent_kernels = False
    net.partial_fit(train)
    return float(ScoringWrapper(scoring, train.labels() if scoring == "neg_log_loss" else None)(net, test, test.y))


def _recipe_init_factory(factory):
    """The CV fits' estimators draw their initial weights from the device-side recipe (model/transformer.py: the same
    distributions as the reference's modules, a generator seeded per task) instead of building the torch modules on the CPU
    under the global RNG lock -- 0.2-0.4 s per fit at E 512 ... 1024, serialised over the host threads.  The reference does
    not define a CV fit's initial weights either (they depend on the dask worker that happens to run it); the refit of the
    best candidate keeps the reference-identical stream.  Only for the model.* modules of this package, and only when the
    caller did not choose ``module__init`` itself."""
    def make():
        net = factory()
        try:
            from model._arena_module import ArenaModule
            from .net import _resolve
            if issubclass(_resolve(net.module), ArenaModule) and "module__init" not in net.get_params():
                net.set_params(module__init="recipe")
        except Exception:               # not one of ours (or a test double): leave it alone
            pass
        return net
    return make


class ShardedGridSearchCV:
    def __init__(self, estimator_factory, param_grid, cv=5, scoring="neg_log_loss", refit=True, fit_and_score=None,
                 device="cpu", verbose=0, fits_per_gpu=1, seed=1, schedule="dynamic", lockstep=1,
                 fit_and_score_group=None, force_collectives=False, recipe_init=True, memory_fraction=0.4,
                 units_per_thread=UNITS_PER_THREAD, min_unit_width=4):
        self.estimator_factory, self.param_grid, self.cv = estimator_factory, param_grid, cv
        self.recipe_init = bool(recipe_init)
        self.memory_fraction = float(memory_fraction)     # of the device's memory that resident fits may take (unit size cap)
        self._task_factory = _recipe_init_factory(estimator_factory) if recipe_init else estimator_factory
        self.scoring, self.refit, self.verbose, self.device = scoring, refit, verbose, device
        self.fit_and_score = fit_and_score or default_fit_and_score
        self.fits_per_gpu, self.seed = int(fits_per_gpu), seed
        assert schedule in ("dynamic", "static")
        self.schedule, self.lockstep, self.fit_and_score_group = schedule, int(lockstep), fit_and_score_group
        self.force_collectives = force_collectives
        # lockstep groups are cut until every host thread of every GPU has about this many units (0 / None: never cut), but not
        # into units narrower than min_unit_width fits -- build_units, unit_cost_ceiling
        self.units_per_thread, self.min_unit_width = units_per_thread, int(min_unit_width)

    def _defaults(self):
        try:
            return self.estimator_factory().get_params()
        except Exception:
            return {}

    def fit(self, dataset):
        import inspect
        import torch
        dist, rank, world = _dist()
        ds = broadcast_dataset(dataset, self.device, force=self.force_collectives)
        self._defaults_cache = self._defaults()
        cands, folds, tasks, order = build_tasks(self.param_grid, ds.y, self.cv, ds.ids.shape[1], self._defaults_cache)
        group_fn = self.fit_and_score_group
        if self.lockstep > 1 and group_fn is None:
            from .lockstep import fit_and_score_group as group_fn
        cap = None
        if group_fn and self.lockstep > 1 and str(self.device).startswith("cuda") and torch.cuda.is_available():
            # fits_per_gpu units of up to `lockstep` fits are resident at a time: keep them inside a fraction of the device's
            # memory (E 1024 / 6 layers is 3.7 GB per fit: 45 of them are 166 GB).  Same arithmetic on every rank.
            budget = self.memory_fraction * torch.cuda.get_device_properties(torch.device(self.device)).total_memory
            defaults, S, per = self._defaults(), ds.ids.shape[1], {}

            def cap(ci):
                if ci not in per:
                    b = estimate_fit_bytes(cands[ci], S, defaults, min(self.lockstep, 64))
                    per[ci] = self.lockstep if not b else int(budget // (max(1, self.fits_per_gpu) * b))
                return per[ci]
        task_cost = lambda t: estimate_cost(cands[tasks[t][0]], ds.ids.shape[1], len(folds[tasks[t][1]][0]), self._defaults_cache)
        ceiling = unit_cost_ceiling(sum(task_cost(t) for t in range(len(tasks))), world, self.fits_per_gpu, self.units_per_thread) \
            if self.units_per_thread else None
        units = build_units(cands, folds, tasks, order, self.lockstep if group_fn else 1, cap, task_cost, ceiling, self.min_unit_width)
        # what a unit is, for logs and the schedule simulator: its candidates' shape-defining numeric parameters
        self.unit_shapes_ = [{k.replace("module__", ""): v for k, v in cands[tasks[u[0]][0]].items()
                              if k not in SHAPE_KEYS_EXCLUDED and isinstance(v, (int, float))} for u in units]
        call = next(_FIT_CALLS)
        counter = WorkCounter(f"slnlp/grid/{call}/next", len(units), static=self.schedule == "static",
                              unit_costs=[sum(task_cost(t) for t in u) for u in units])
        rows = torch.full((len(tasks), 3), float("nan"), dtype=torch.float64)      # score, seconds, error flag
        rows[:, 2] = 0.0
        mine, errors = [], []
        self.unit_log_ = []          # this rank's units: (unit index, fits, estimated cost, start s, end s) -- what grid_sim replays
        if dist is not None and world > 1:
            # ranks reach this point at different times (imports, the first fold split): start the clock -- and the race for
            # the dynamically scheduled units -- together, so rank_seconds_ are comparable across ranks
            dist.all_reduce(torch.zeros(1, device=comm_device(self.device)))
        t_start = time.time()
        sig = inspect.signature(self.fit_and_score).parameters
        takes_seed, takes_conc = "seed" in sig, "concurrent" in sig
        is_cuda = str(self.device).startswith("cuda")
        lock = threading.Lock()

        def run_unit(unit):
            if is_cuda and torch.device(self.device).index is not None:
                torch.cuda.set_device(torch.device(self.device))     # the current device is per host thread
            t0 = time.time()
            seeds = [self.seed + t if self.seed is not None else None for t in unit]
            try:
                if len(unit) > 1 or (group_fn is not None and self.lockstep > 1):
                    scores = group_fn(self._task_factory, [cands[tasks[t][0]] for t in unit],
                                      [ds[folds[tasks[t][1]][0]] for t in unit], [ds[folds[tasks[t][1]][1]] for t in unit],
                                      self.scoring, seeds=seeds)
                else:
                    ci, fi = tasks[unit[0]]
                    kw = {"seed": seeds[0]} if (takes_seed and seeds[0] is not None) else {}
                    if takes_conc and self.fits_per_gpu > 1:
                        kw["concurrent"] = True
                    scores = [self.fit_and_score(self._task_factory, cands[ci], ds[folds[fi][0]], ds[folds[fi][1]],
                                                 self.scoring, **kw)]
                dt = (time.time() - t0) / len(unit)
                for t, s in zip(unit, scores):
                    rows[t, 0], rows[t, 1] = float(s), dt         # each task owns its row
            except Exception as e:                                  # keep the collective alive; raise after it
                import traceback
                for t in unit:
                    rows[t, 0], rows[t, 1], rows[t, 2] = float("nan"), time.time() - t0, 1.0
                with lock:
                    errors.append((unit, e, traceback.format_exc()))
                counter.abort()                                     # every rank stops pulling units (error_score='raise')
            if self.verbose:
                for t in unit:
                    print(f"[rank {rank}] task {t} cand {tasks[t][0]} fold {tasks[t][1]}: score {float(rows[t, 0]):.4f} "
                          f"({float(rows[t, 1]):.2f}s)", flush=True)

        def worker():
            try:
                worker_loop()
            finally:
                if is_cuda and self.fits_per_gpu > 1:       # a host thread that ends gives its stream back (slnlp.net)
                    from .net import release_thread_streams
                    release_thread_streams()

        def worker_loop():
            while True:
                i = counter.acquire()
                if i is None:
                    return
                with lock:
                    mine.extend(units[i])
                u_t0 = time.time()
                try:
                    run_unit(units[i])
                finally:
                    counter.release()
                    with lock:
                        self.unit_log_.append((i, len(units[i]), counter.cost(i), round(u_t0 - t_start, 3), round(time.time() - t_start, 3)))

        if self.fits_per_gpu > 1:
            threads = [threading.Thread(target=worker) for _ in range(self.fits_per_gpu)]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
        else:
            worker()
        self.local_seconds_ = time.time() - t_start
        if dist is not None and (world > 1 or self.force_collectives):   # each task has exactly one owner: combine by all_gather
            mine_mask = torch.zeros(len(tasks), dtype=torch.bool)
            mine_mask[mine] = True
            send = torch.where(mine_mask[:, None], rows, torch.zeros_like(rows))
            send = torch.cat([send, torch.tensor([[self.local_seconds_, float(len(mine)), 0.0]], dtype=torch.float64)])
            send = send.to(comm_device(self.device))
            gathered = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(gathered, send)
            gathered = [g.cpu() for g in gathered]
            rows = torch.stack([g[:-1] for g in gathered]).sum(0)
            self.rank_seconds_ = [float(g[-1, 0]) for g in gathered]
            self.rank_tasks_ = [int(g[-1, 1]) for g in gathered]
        else:
            self.rank_seconds_, self.rank_tasks_ = [self.local_seconds_], [len(mine)]
        if float(rows[:, 2].sum()) > 0:
            failed = [int(t) for t in torch.nonzero(rows[:, 2]).flatten()]
            detail = f"\n--- first failure on this rank ---\n{errors[0][2]}" if errors else ""
            raise RuntimeError(f"ShardedGridSearchCV: {len(failed)} task(s) failed (task ids {failed[:8]}...)" + detail) \
                from (errors[0][1] if errors else None)
        scores = rows[:, 0].numpy().reshape(len(cands), len(folds))
        self.cv_results_ = {
            "params": cands,
            "mean_test_score": scores.mean(1), "std_test_score": scores.std(1),
            **{f"split{f}_test_score": scores[:, f] for f in range(len(folds))},
            "mean_fit_time": rows[:, 1].numpy().reshape(len(cands), len(folds)).mean(1),
        }
        mean = self.cv_results_["mean_test_score"]
        self.cv_results_["rank_test_score"] = (np.argsort(np.argsort(-mean, kind="stable"), kind="stable") + 1).astype(np.int32)
        self.best_index_ = int(np.argmax(mean))            # first maximum, like sklearn's rank 1
        self.best_score_ = float(mean[self.best_index_])
        self.best_params_ = cands[self.best_index_]
        self.n_tasks_, self.tasks_of_rank_, self.n_units_ = len(tasks), sorted(mine), len(units)
        if self.refit and rank == 0:
            self.best_estimator_ = self.estimator_factory().set_params(**self.best_params_)
            if hasattr(self.best_estimator_, "fit"):
                t0 = time.time()
                self.best_estimator_.fit(ds)
                self.refit_time_ = time.time() - t0
        return self
