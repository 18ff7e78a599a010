"""Cross-validated grid search sharded over the GPUs of one node.

The reference farms the (candidate x fold) fits of ``GridSearchCV`` out to dask
workers, one per GPU, pickling the whole dataset with every task
(/root/reference/main.py:62-95, helper.py:108-180, 490-526; 324 candidates x 5
folds for config-transformer.yaml).  Here: one process per GPU
(``torch.distributed``, RCCL on GPUs / gloo in CPU tests);

* rank 0 broadcasts the integer dataset ONCE (``X[N,S]``, ``lengths[N]``, ``y[N]``),
* every rank derives the identical task list (``ParameterGrid`` x ``StratifiedKFold(cv)``,
  no shuffle) and takes the tasks ``i % world == rank`` of the list sorted by
  estimated cost (longest first), so no scheduler and no per-task traffic,
* one ``all_gather`` of the per-task score rows at the end; every rank computes
  ``mean_test_score`` / ``rank_test_score`` / ``best_index_`` identically and
  rank 0 refits the best candidate on the whole training set (``refit=True``).

There is no per-step collective: fits are independent.

``fits_per_gpu=k`` runs k of the rank's fits at a time, one host thread and one HIP stream each: a single
batch-50 fit leaves most of the GPU idle during its decoder (tgt length 1) phases, and k = 4 fits were
measured at 1.5x the aggregate seq/s of one (tools/bench_concurrent.py).  Every task is seeded by its index,
so results do not depend on the world size or on k.
"""
import itertools
import time

import numpy as np

from .data import TokenDataset


def parameter_grid(param_grid):
    """sklearn.model_selection.ParameterGrid order: keys sorted, last key varies fastest."""
    keys = sorted(param_grid)
    return [dict(zip(keys, vals)) for vals in itertools.product(*[param_grid[k] for k in keys])]


def estimate_cost(params):
    """Relative cost of one fit for LPT ordering: layers x (E^2 + E*F)."""
    E = params.get("module__embedding_size", 128)
    F = params.get("module__hidden_size", 256)
    N = params.get("module__num_layers", 2)
    return float(N) * (E * E + E * F)


def build_tasks(param_grid, y, cv):
    from sklearn.model_selection import StratifiedKFold
    cands = parameter_grid(param_grid)
    folds = list(StratifiedKFold(n_splits=cv).split(np.zeros(len(y)), y))
    tasks = [(ci, fi) for ci in range(len(cands)) for fi in range(len(folds))]
    order = sorted(range(len(tasks)), key=lambda t: (-estimate_cost(cands[tasks[t][0]]), t))
    return cands, folds, tasks, order


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def broadcast_dataset(ds, device="cpu", src=0):
    """Rank ``src`` holds the dataset; everyone else receives it (ONE broadcast of a packed int64 buffer
    over RCCL/xGMI on GPUs: ~4 MB at N=10k, latency-bound)."""
    import torch
    dist, rank, world = _dist()
    if world == 1:
        return ds
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
    if rank == src:
        return ds
    from model.util import Vocab
    b = buf.cpu().numpy()
    return TokenDataset(b[:n * s].reshape(n, s), b[n * s:n * s + n], b[n * s + n:], Vocab(vx) if vx else None,
                        Vocab(vy) if vy else None)


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


class ShardedGridSearchCV:
    def __init__(self, estimator_factory, param_grid, cv=5, scoring="neg_log_loss", refit=True, fit_and_score=None,
                 device="cpu", verbose=0, fits_per_gpu=1, seed=1):
        self.estimator_factory, self.param_grid, self.cv = estimator_factory, param_grid, cv
        self.scoring, self.refit, self.verbose, self.device = scoring, refit, verbose, device
        self.fit_and_score = fit_and_score or default_fit_and_score
        self.fits_per_gpu, self.seed = int(fits_per_gpu), seed

    def fit(self, dataset):
        import torch
        dist, rank, world = _dist()
        ds = broadcast_dataset(dataset, self.device)
        cands, folds, tasks, order = build_tasks(self.param_grid, ds.y, self.cv)
        mine = [order[i] for i in range(rank, len(order), world)]
        rows = torch.full((len(tasks), 2), float("nan"), dtype=torch.float64)
        t_start = time.time()
        import inspect
        sig = inspect.signature(self.fit_and_score).parameters
        takes_seed, takes_conc = "seed" in sig, "concurrent" in sig
        is_cuda = str(self.device).startswith("cuda")

        def run_task(t):
            if is_cuda and torch.device(self.device).index is not None:
                torch.cuda.set_device(torch.device(self.device))     # the current device is per host thread
            ci, fi = tasks[t]
            tr, te = folds[fi]
            t0 = time.time()
            kw = {"seed": self.seed + t} if (takes_seed and self.seed is not None) else {}
            if takes_conc and self.fits_per_gpu > 1:
                kw["concurrent"] = True
            score = self.fit_and_score(self.estimator_factory, cands[ci], ds[tr], ds[te], self.scoring, **kw)
            rows[t, 0], rows[t, 1] = score, time.time() - t0         # each task owns its row
            if self.verbose:
                print(f"[rank {rank}] task {t} cand {ci} fold {fi}: score {float(rows[t, 0]):.4f} ({float(rows[t, 1]):.2f}s)",
                      flush=True)

        if self.fits_per_gpu > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(self.fits_per_gpu) as pool:
                list(pool.map(run_task, mine))                       # longest-first order; re-raises a task's exception
        else:
            for t in mine:
                run_task(t)
        self.local_seconds_ = time.time() - t_start
        if world > 1:                                     # each task has exactly one owner: combine by all_gather
            mine_mask = torch.zeros(len(tasks), dtype=torch.bool)
            mine_mask[mine] = True
            send = torch.where(mine_mask[:, None], rows, torch.zeros_like(rows)).to(self.device)
            gathered = [torch.empty_like(send) for _ in range(world)]
            dist.all_gather(gathered, send)
            rows = torch.stack([g.cpu() for g in gathered]).sum(0)
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
        self.n_tasks_, self.tasks_of_rank_ = len(tasks), mine
        if self.refit and rank == 0:
            self.best_estimator_ = self.estimator_factory().set_params(**self.best_params_)
            if hasattr(self.best_estimator_, "fit"):
                self.best_estimator_.fit(ds)
        return self
