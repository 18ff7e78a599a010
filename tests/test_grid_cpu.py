"""CPU (gloo, world_size 2): the grid-search sharding -- dataset broadcast, task
partition, all_gather of scores, best-candidate selection -- with the ORACLE as
the per-task fit engine (tests may use the oracle; the product default is the
HIP estimator).  Results must not depend on the number of ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from slnlp import grid
from slnlp.data import synthetic_dataset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GRID = {"lr": [0.1, 0.01], "module__embedding_size": [16, 32], "module__num_layers": [1, 2]}
CV = 3


def oracle_fit_and_score(factory, params, train, test, scoring):
    """Tiny Transformer trained for a few oracle steps on the train fold; score = -CE on the test fold."""
    from oracle import train_ref, transformer_ref as tr
    from slnlp import synth
    E, N, H, F = params["module__embedding_size"], params["module__num_layers"], 2, 32
    Vs, Vt = len(train.vocab_X), len(train.vocab_y)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(tr.param_shapes(E, H, N, F, Vs, Vt), seed=1).items()}
    fwd = lambda p, X, y, L: tr.forward(p, X, y, num_heads=H, num_layers=N)
    trn = train_ref.Trainer(sd, fwd, lr=params["lr"])
    X, y = torch.from_numpy(train.ids), torch.from_numpy(train.y)
    for i in range(0, len(train), 16):
        trn.step(X[i:i + 16], y[i:i + 16], None)
    Xt, yt = torch.from_numpy(test.ids), torch.from_numpy(test.y)
    with torch.no_grad():
        return -float(train_ref.cross_entropy_on_logprobs(fwd(trn.sd, Xt, yt, None), yt, 1))


def _run(rank, world, port, out, schedule="dynamic"):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = synthetic_dataset(48, seq_len=8, src_vocab=40, n_labels=4, seed=3, min_len=3) if rank == 0 else None
    gs = grid.ShardedGridSearchCV(lambda: None, GRID, cv=CV, fit_and_score=oracle_fit_and_score, refit=False, schedule=schedule)
    gs.fit(ds)
    out[rank] = (gs.cv_results_["mean_test_score"].tolist(), gs.best_index_, gs.best_params_, sorted(gs.tasks_of_rank_))
    dist.barrier()
    dist.destroy_process_group()


def skewed_fit_and_score(factory, params, train, test, scoring):
    """Deliberately unpredictable fit times (what EarlyStopping does to the real grid): the estimated cost says
    nothing about them.  Score = a pure function of the task, so results are checkable."""
    import time
    slow = params["lr"] == 0.1 and params["module__num_layers"] == 2
    time.sleep(0.30 if slow else 0.02)
    return -float(params["lr"]) * params["module__embedding_size"] - 0.001 * len(test)


def _run_skewed(rank, world, port, out, schedule):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = synthetic_dataset(48, seq_len=8, src_vocab=40, n_labels=4, seed=3, min_len=3) if rank == 0 else None
    gs = grid.ShardedGridSearchCV(lambda: None, GRID, cv=CV, fit_and_score=skewed_fit_and_score, refit=False, schedule=schedule)
    gs.fit(ds)
    out[rank] = (gs.cv_results_["mean_test_score"].tolist(), gs.rank_seconds_, gs.rank_tasks_, sorted(gs.tasks_of_rank_))
    dist.barrier()
    dist.destroy_process_group()


ADMIT_GRID = {"lr": [0.1, 0.01], "module__embedding_size": [64, 16]}     # x cv 2 = 8 units: 4 long (E 64) sort first, 4 short
_ONE_STREAM = __import__("threading").Lock()                              # a rank's units serialise on its one stream


def one_stream_fit_and_score(factory, params, train, test, scoring):
    """A fit whose device work (the sleep) is serialised per process, like the kernels of a rank's fits_per_gpu host threads
    on the rank's single stream: a unit taken early by an extra thread only queues up behind the rank's first one."""
    import time
    with _ONE_STREAM:
        time.sleep(0.40 if params["module__embedding_size"] == 64 else 0.05)
    return -float(params["lr"]) * params["module__embedding_size"] - 0.001 * len(test)


def _run_admission(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = synthetic_dataset(48, seq_len=8, src_vocab=40, n_labels=4, seed=3, min_len=3) if rank == 0 else None
    gs = grid.ShardedGridSearchCV(lambda: None, ADMIT_GRID, cv=2, fit_and_score=one_stream_fit_and_score, refit=False, fits_per_gpu=3)
    gs.fit(ds)
    out[rank] = (gs.cv_results_["mean_test_score"].tolist(), gs.rank_seconds_, gs.rank_tasks_, sorted(gs.tasks_of_rank_), gs.best_index_)
    dist.barrier()
    dist.destroy_process_group()


def failing_fit_and_score(factory, params, train, test, scoring):
    if params["lr"] == 0.01 and params["module__embedding_size"] == 32 and params["module__num_layers"] == 2:
        raise ValueError("boom")
    return -1.0


def _run_failing(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = synthetic_dataset(48, seq_len=8, src_vocab=40, n_labels=4, seed=3, min_len=3) if rank == 0 else None
    gs = grid.ShardedGridSearchCV(lambda: None, GRID, cv=CV, fit_and_score=failing_fit_and_score, refit=False)
    try:
        gs.fit(ds)
        out[rank] = "no error"
    except RuntimeError as e:
        out[rank] = str(e)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_task_list_and_partition():
    ds = synthetic_dataset(48, seq_len=8, src_vocab=40, n_labels=4, seed=3, min_len=3)
    cands, folds, tasks, order = grid.build_tasks(GRID, ds.y, CV)
    assert len(cands) == 8 and len(folds) == CV and len(tasks) == 24
    from sklearn.model_selection import ParameterGrid
    assert cands == list(ParameterGrid(GRID))                          # sklearn candidate order
    costs = [grid.estimate_cost(cands[tasks[t][0]]) for t in order]
    assert costs == sorted(costs, reverse=True)                        # longest first
    for world in (1, 2, 3, 8):
        owned = [t for r in range(world) for t in order[r::world]]
        assert sorted(owned) == list(range(len(tasks)))                # every task exactly once


def test_sharded_grid_world2_equals_world1():
    ds = synthetic_dataset(48, seq_len=8, src_vocab=40, n_labels=4, seed=3, min_len=3)
    single = grid.ShardedGridSearchCV(lambda: None, GRID, cv=CV, fit_and_score=oracle_fit_and_score, refit=False).fit(ds)
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    for schedule in ("dynamic", "static"):
        out.clear()
        mp.spawn(_run, args=(world, _free_port(), out, schedule), nprocs=world, join=True)
        assert len(out) == world
        for r in range(world):
            mean, best, params, mine = out[r]
            assert np.allclose(mean, single.cv_results_["mean_test_score"], rtol=0, atol=1e-12)   # same on every rank
            assert best == single.best_index_ and params == single.best_params_
        assert sorted(out[0][3] + out[1][3]) == list(range(single.n_tasks_))
        assert set(out[0][3]).isdisjoint(out[1][3])


def test_dynamic_schedule_balances_skewed_fit_times():
    """Work counter on the process group's store: with fit times the cost estimate cannot see, both ranks finish
    within 15 % of the mean (the static deal leaves one rank with most of the slow fits), and the results are still
    those of one rank."""
    ds = synthetic_dataset(48, seq_len=8, src_vocab=40, n_labels=4, seed=3, min_len=3)
    single = grid.ShardedGridSearchCV(lambda: None, GRID, cv=CV, fit_and_score=skewed_fit_and_score, refit=False).fit(ds)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_run_skewed, args=(2, _free_port(), out, "dynamic"), nprocs=2, join=True)
    mean, secs, ntasks, mine0 = out[0]
    assert np.allclose(mean, single.cv_results_["mean_test_score"], rtol=0, atol=1e-12)
    assert out[0][1] == out[1][1] and sum(ntasks) == single.n_tasks_       # every rank knows every rank's time
    assert max(secs) / (sum(secs) / len(secs)) <= 1.15, secs
    assert sorted(out[0][3] + out[1][3]) == list(range(single.n_tasks_))


def test_admission_control_deals_long_units_one_per_rank_with_host_threads():
    """world 4 x fits_per_gpu 3 x 8 skewed units: without admission control the 12 host threads race for the
    8 units and one rank can take three of the four long ones, which then serialise on its stream.  With it every rank gets
    exactly one long unit, the short ones are pulled as ranks fall idle, and the ranks finish together."""
    ds = synthetic_dataset(48, seq_len=8, src_vocab=40, n_labels=4, seed=3, min_len=3)
    single = grid.ShardedGridSearchCV(lambda: None, ADMIT_GRID, cv=2, fit_and_score=one_stream_fit_and_score, refit=False).fit(ds)
    cands, folds, tasks, order = grid.build_tasks(ADMIT_GRID, ds.y, 2)
    long_tasks = {t for t in range(len(tasks)) if cands[tasks[t][0]]["module__embedding_size"] == 64}
    assert set(order[:4]) == long_tasks                                   # longest first
    world = 4
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_run_admission, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == world
    owned = []
    for r in range(world):
        mean, secs, ntasks, mine, best = out[r]
        assert np.allclose(mean, single.cv_results_["mean_test_score"], rtol=0, atol=1e-12) and best == single.best_index_
        assert len(long_tasks & set(mine)) == 1, (r, mine)                # one long unit per rank
        owned += mine
    assert sorted(owned) == list(range(len(tasks)))
    secs = out[0][1]
    assert max(secs) / (sum(secs) / len(secs)) <= 1.15, secs
    assert max(secs) < 0.40 + 2 * 0.05 + 0.35, secs                       # nobody ran two long units back to back


def test_work_counter_admission_single_process():
    """Inside one process (world 1) the fair share is everything: a host thread always gets the next unit while units are left
    (round 3's reserve rule, kept behind reserve=True, makes a fourth thread wait for the last unit)."""
    c = grid.WorkCounter("k", 4)
    assert [c.acquire() for _ in range(4)] == [0, 1, 2, 3] and c.acquire() is None
    c = grid.WorkCounter("k1", 4, reserve=True)
    assert [c.acquire() for _ in range(3)] == [0, 1, 2]    # the deal; 3 left > 1 x 0; 2 left > 1 x 1
    assert c.try_acquire() is grid.WorkCounter.WAIT        # three in flight: the 1 unit left is not more than 1 x 2
    c.release(); c.release(); c.release()
    assert c.acquire() == 3 and c.acquire() is None
    c = grid.WorkCounter("k2", 3)
    assert c.acquire() == 0
    c.abort()
    assert c.acquire() is None


def _counters(world, costs, **kw):
    from slnlp.grid_sim import MemoryStore
    store = MemoryStore()
    return [grid.WorkCounter("k", len(costs), unit_costs=costs, rank=r, world=world, store=store, **kw) for r in range(world)], store


def test_admission_prefetch_stays_within_the_fair_share_of_estimated_work():
    """8 ranks, 32 units of estimated cost 2 / 1 / .15 / .08 (8 each, longest first): a rank's extra host threads may take ONE
    medium unit next to the dealt long one (3.0 <= the fair share 3.23) but not a second one, whatever the number of units
    still left; then light units while they fit under the share, and an idle rank always pulls.  One store request per
    decision carries the abort flag too."""
    costs = [2.0] * 8 + [1.0] * 8 + [0.15] * 8 + [0.08] * 8
    counters, store = _counters(8, costs)
    assert [c.try_acquire() for c in counters] == list(range(8))              # the deal: one long unit per rank
    assert counters[0].try_acquire() == 8                                       # prefetch: 2 + 1 <= 25.84 / 8
    assert counters[0].try_acquire() is grid.WorkCounter.WAIT                   # a second medium unit would exceed the share
    assert [c.try_acquire() for c in counters[1:]] == list(range(9, 16))
    assert [c.try_acquire() for c in counters] == list(range(16, 24))           # 3.15 <= 3.23
    assert counters[0].try_acquire() == 24                                      # 3.23 <= 3.23: the reserve rule is gone
    assert counters[0].try_acquire() is grid.WorkCounter.WAIT                   # 3.31 > 3.23
    for _ in range(4):
        counters[3].release()                                                   # (rank 3 had three units in flight)
    counters[3]._in_flight = 0
    assert counters[3].try_acquire() == 25                                      # an idle rank always pulls
    n_req = store["k"]
    counters[5].abort()
    assert all(c.try_acquire() is None for c in counters) and store["k"] >= grid.WorkCounter.ABORT
    assert list(store) == ["k"], "units taken and the abort flag share one key: one request per admission decision"


def test_simulated_schedule_scales_on_the_bench_sample():
    """the 8-GPU schedule proven on the CPU before anyone measures it.  slnlp.grid_sim drives the REAL WorkCounter
    (one per simulated rank, shared in-memory store) for 1 / 2 / 4 / 8 GPUs x 4 host threads in virtual time over bench.py's grid
    sample, with the per-unit solo durations and the concurrency gains measured on one MI355X
    (profiles/r04_grid_calibration.json).  The simulated 1-GPU makespan reproduces the measured run; strong-scaling efficiency
    T(1) / (N T(N)), worst of 8 thread-race orders: >= 0.97 at 2 and 4 GPUs, >= 0.94 at 8.  Round 3's rule and unit list (32 units,
    reserve rule) are priced beside it: 0.82 at 8 GPUs."""
    import json, os, bench
    from slnlp import grid_sim as gs
    cal = json.load(open(os.path.join(ROOT, "profiles", "r04_grid_calibration.json")))
    per_fit = {}
    for u in cal["solo_units_lockstep15_one_thread"]:
        sh = u["shape"]
        per_fit.setdefault((sh["embedding_size"], sh["hidden_size"], sh["num_heads"], sh["num_layers"]), []).append(u["seconds"] / u["fits"])
    per_fit = {k: float(np.mean(v)) for k, v in per_fit.items()}
    gain = gs.gain_from_throughputs({int(k): v for k, v in cal["throughput_by_threads_lockstep15_kfolds_per_hr"].items() if k.isdigit()})
    assert 1.4 < gain(4) < 1.5 and gain(8) == gain(4)
    ds = synthetic_dataset(bench.GRID_SAMPLES, seq_len=48, src_vocab=300, n_labels=200, seed=1, min_len=8)
    defaults = {"max_epochs": bench.GRID_EPOCHS}
    cands, folds, tasks, order = grid.build_tasks(bench.GRID_SAMPLE, ds.y, bench.GRID_CV, 48, defaults)
    tc = lambda t: grid.estimate_cost(cands[tasks[t][0]], 48, len(folds[tasks[t][1]][0]), defaults)
    total = sum(tc(t) for t in range(len(tasks)))
    shape = lambda u: tuple(cands[tasks[u[0]][0]][k] for k in ("module__embedding_size", "module__hidden_size", "module__num_heads", "module__num_layers"))

    def units_at(world, per_thread):
        ceiling = grid.unit_cost_ceiling(total, world, 4, per_thread) if per_thread else None
        units = grid.build_units(cands, folds, tasks, order, 15, None, tc, ceiling, 4)
        assert sorted(t for u in units for t in u) == list(range(len(tasks))) and all(len(u) >= 3 for u in units)
        costs = [sum(tc(t) for t in u) for u in units]
        assert costs == sorted(costs, reverse=True)                            # handed out longest first, by SUMMED cost
        return [per_fit[shape(u)] * len(u) for u in units], costs

    def makespan(world, per_thread, **counter_kw):
        work, costs = units_at(world, per_thread)
        saved = gs.WorkCounter
        if counter_kw:
            import functools
            gs.WorkCounter = functools.partial(saved, **counter_kw)
            gs.WorkCounter.WAIT = saved.WAIT
        try:
            return max(gs.simulate(work, costs, world, 4, gain, seed=s)["makespan"] for s in range(8)), len(work)
        finally:
            gs.WorkCounter = saved

    t1, n1 = makespan(1, grid.UNITS_PER_THREAD)
    measured = [r for r in cal["runs"] if r["lockstep"] == 15 and r["fits_per_gpu"] == 4][0]["seconds"]
    assert abs(t1 - measured) / measured < 0.05, (t1, measured)               # the model reproduces the measured 1-GPU run
    eff = {}
    for world in (2, 4, 8):
        tn, n = makespan(world, grid.UNITS_PER_THREAD)
        eff[world] = t1 / (world * tn)
        assert n >= 2 * 4 * world                                              # at least a couple of units per host thread
    assert eff[2] >= 0.97 and eff[4] >= 0.97 and eff[8] >= 0.94, eff
    t1_old, _ = makespan(1, 0, reserve=True)
    t8_old, n8_old = makespan(8, 0, reserve=True)
    assert n8_old == 32 and t1_old / (8 * t8_old) < 0.88, t1_old / (8 * t8_old)   # what round 3 shipped


def test_simulated_schedule_with_coarse_units_still_deals_one_long_unit_per_rank():
    """The case admission control was built for (world 4 x 3 host threads x 8 skewed units, nothing to cut): every rank runs
    exactly one long unit and the ranks finish together, in the simulator as in the gloo test above."""
    from slnlp import grid_sim as gs
    work, costs = [4.0] * 4 + [0.5] * 4, [8.0] * 4 + [1.0] * 4
    for seed in range(6):
        r = gs.simulate(work, costs, 4, 3, lambda k: 1.0, seed=seed)           # one stream per rank: no gain from concurrency
        assert all(sum(1 for u in us if u < 4) == 1 for us in r["rank_units"]), r["rank_units"]
        assert max(r["rank_seconds"]) <= 4.5 + 1e-9


def test_bench_grid_sample_is_cut_finer_as_the_gpu_count_grows():
    """bench.py's strong-scaling sample: 480 fits in four cost classes (E512 N4 / E512 N2 / E128 N4 / E128 N2).  On one GPU the
    lockstep groups stay almost whole; on 8 GPUs x 4 host threads they are cut until every thread has about six units, none
    narrower than 4 fits (a 15-fit group -> 4 + 4 + 4 + 3), all shape-compatible, handed out by summed cost."""
    import bench
    ds = synthetic_dataset(bench.GRID_SAMPLES, seq_len=48, src_vocab=300, n_labels=200, seed=1, min_len=8)
    defaults = {"max_epochs": bench.GRID_EPOCHS}
    cands, folds, tasks, order = grid.build_tasks(bench.GRID_SAMPLE, ds.y, bench.GRID_CV, 48, defaults)
    tc = lambda t: grid.estimate_cost(cands[tasks[t][0]], 48, len(folds[tasks[t][1]][0]), defaults)
    total = sum(tc(t) for t in range(len(tasks)))
    whole = grid.build_units(cands, folds, tasks, order, 15, None, tc, None)
    assert len(cands) == 96 and len(tasks) == 480 and len(whole) == 32 and all(len(u) == 15 for u in whole)
    shape = lambda u: (cands[tasks[u[0]][0]]["module__embedding_size"], cands[tasks[u[0]][0]]["module__num_layers"])
    assert [shape(u) for u in whole] == [(512, 4)] * 8 + [(512, 2)] * 8 + [(128, 4)] * 8 + [(128, 2)] * 8
    # the launch-bound floor in the estimate: a small model's unit is a third of a large one's, not a twentieth (measured: 0.30)
    cost = lambda u: sum(tc(t) for t in u)
    assert 0.2 < cost(whole[-1]) / cost(whole[0]) < 0.4
    n_units = {}
    for world in (1, 2, 4, 8):
        units = grid.build_units(cands, folds, tasks, order, 15, None, tc, grid.unit_cost_ceiling(total, world, 4), 4)
        assert sorted(t for u in units for t in u) == list(range(len(tasks)))
        assert all(3 <= len(u) <= 15 for u in units)
        assert all(len({shape([t]) for t in u}) == 1 for u in units)
        n_units[world] = len(units)
    assert n_units[1] <= 40 and n_units[8] >= 96 and n_units[1] <= n_units[2] <= n_units[4] <= n_units[8], n_units


def test_failing_task_raises_on_every_rank_instead_of_hanging():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_run_failing, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in range(2):
        assert "task(s) failed" in out[r], out[r]
    with pytest.raises(RuntimeError, match="boom|failed"):
        ds = synthetic_dataset(48, seq_len=8, src_vocab=40, n_labels=4, seed=3, min_len=3)
        grid.ShardedGridSearchCV(lambda: None, GRID, cv=CV, fit_and_score=failing_fit_and_score, refit=False).fit(ds)


def test_build_units_packs_shape_compatible_tasks():
    ds = synthetic_dataset(50, seq_len=8, src_vocab=40, n_labels=4, seed=3, min_len=3)
    g = {"lr": [0.1, 0.01], "module__dropout": [0.1, 0.3], "module__embedding_size": [16, 32]}
    cands, folds, tasks, order = grid.build_tasks(g, ds.y, 5)
    assert [[t] for t in order] == grid.build_units(cands, folds, tasks, order, 1)
    units = grid.build_units(cands, folds, tasks, order, 4)
    assert sorted(t for u in units for t in u) == list(range(len(tasks)))
    for u in units:
        assert 1 <= len(u) <= 4
        assert len({cands[tasks[t][0]]["module__embedding_size"] for t in u}) == 1      # one shape per unit
        assert len({len(folds[tasks[t][1]][0]) for t in u}) == 1                         # one batch schedule per unit
    assert len(units) < len(tasks) / 2


def test_build_units_respects_a_per_candidate_cap():
    """cap(candidate) bounds a unit's size (device memory): large candidates get narrower units, nothing is lost or repeated."""
    cands = [{"lr": lr, "module__embedding_size": E} for E in (1024, 128) for lr in (0.1, 0.01, 0.001)]
    folds = [(np.arange(80), np.arange(20))] * 5
    tasks = [(ci, fi) for ci in range(len(cands)) for fi in range(5)]
    order = list(range(len(tasks)))
    units = grid.build_units(cands, folds, tasks, order, lockstep=15, cap=lambda ci: 4 if cands[ci]["module__embedding_size"] == 1024 else 99)
    assert sorted(t for u in units for t in u) == order
    big = [u for u in units if cands[tasks[u[0]][0]]["module__embedding_size"] == 1024]
    small = [u for u in units if cands[tasks[u[0]][0]]["module__embedding_size"] == 128]
    assert [len(u) for u in big] == [4, 4, 4, 3] and [len(u) for u in small] == [15]
    assert all(len({cands[tasks[t][0]]["module__embedding_size"] for t in u}) == 1 for u in units)


def test_fits_per_gpu_threads_same_results_and_seed_passed():
    """fits_per_gpu=k: the rank's tasks run k at a time on host threads; scores, ranking and the per-task seeds
    are the same as with k = 1."""
    import threading
    ds = synthetic_dataset(48, seq_len=8, src_vocab=40, n_labels=4, seed=3, min_len=3)
    seen = {}

    def fs(factory, params, train, test, scoring, seed=None):
        seen.setdefault(threading.get_ident(), []).append(seed)
        return oracle_fit_and_score(factory, params, train, test, scoring)

    torch.set_num_threads(2)
    res = []
    for k in (1, 3):
        seen.clear()
        gs = grid.ShardedGridSearchCV(lambda: None, GRID, cv=CV, fit_and_score=fs, refit=False, fits_per_gpu=k, seed=7)
        gs.fit(ds)
        seeds = sorted(s for v in seen.values() for s in v)
        assert seeds == [7 + t for t in range(gs.n_tasks_)]          # one deterministic seed per task index
        assert len(seen) == (1 if k == 1 else min(k, gs.n_tasks_)) or k > 1
        res.append((gs.cv_results_["mean_test_score"].tolist(), gs.best_index_))
    assert res[0] == res[1]


def test_estimate_fit_bytes_asks_the_library_and_orders_candidates():
    """The device-memory estimate behind the lockstep unit cap: host-side size queries of the library (no GPU), larger for larger
    models, None for a module that is not one of this package's."""
    from slnlp.data import synthetic_dataset
    ds = synthetic_dataset(20, seq_len=48, src_vocab=300, n_labels=20, seed=1, min_len=8)
    base = dict(module="model.Transformer", module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y, batch_size=50, module__num_heads=4)
    size = lambda E, F, N, **kw: grid.estimate_fit_bytes(dict(module__embedding_size=E, module__hidden_size=F, module__num_layers=N), 48, dict(base, **kw))
    small, mid, big = size(128, 128, 2), size(512, 512, 2), size(1024, 512, 6)
    assert 0 < small < mid < big and big > 2 * 2**30
    assert size(512, 512, 4, module="model.EncoderDecoderLSTMAttn") > 0
    assert size(128, 128, 2, module="torch.nn.Linear") is None


# ---- 8 ranks x 5 host threads against rank 0's store (what `bench.py --gpus 8` starts on an 8-GPU node), rehearsed on the CPU:
# the bench's own grid sample, lockstep units of up to 15 fits cut by the cost ceiling, a fake group fit whose duration follows
# the estimated cost (with a deterministic +-30 % the estimate cannot see).  40 threads poll / add on ONE TCP store.
def _sample_grid():
    import bench
    return bench.GRID_SAMPLE, bench.GRID_CV


def rehearsal_group(factory, params_list, trains, tests, scoring="neg_log_loss", seeds=None):
    """Stands in for slnlp.lockstep.fit_and_score_group: sleeps for the unit's estimated cost, scores = a pure function of the task."""
    import time, zlib
    cost = sum(grid.estimate_cost(p, trains[0].ids.shape[1], len(tr)) for p, tr in zip(params_list, trains))
    jitter = 0.7 + 0.6 * (zlib.crc32(repr(sorted(params_list[0].items())).encode()) % 1000) / 1000.0
    dt = min(0.5, REHEARSAL_S_PER_COST * cost * jitter)
    with _SLEPT_LOCK:
        _SLEPT[0] += dt
    time.sleep(dt)
    return [-float(p["lr"]) * p["module__embedding_size"] - 0.001 * len(te) - 0.01 * p["module__num_layers"] for p, te in zip(params_list, tests)]


REHEARSAL_S_PER_COST = 2.5e-12
_SLEPT, _SLEPT_LOCK = [0.0], __import__("threading").Lock()       # what this process' units slept in total


def _run_rehearsal(rank, world, port, out, threads):
    import time
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G, cv = _sample_grid()
    ds = synthetic_dataset(400, seq_len=48, src_vocab=300, n_labels=20, seed=1, min_len=8) if rank == 0 else None
    gs = grid.ShardedGridSearchCV(lambda: None, G, cv=cv, refit=False, fits_per_gpu=threads, lockstep=15, fit_and_score_group=rehearsal_group,
                                  recipe_init=False)
    t0 = time.perf_counter()
    gs.fit(ds)
    out[rank] = (gs.cv_results_["mean_test_score"].tolist(), gs.rank_seconds_, gs.rank_tasks_, gs.n_units_, time.perf_counter() - t0, gs.best_index_, _SLEPT[0])
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_times_five_host_threads_on_one_store_neither_deadlock_nor_dominate():
    """the N = 8 leg of bench.py has never run -- 8 ranks x 5 host threads pulling work units from rank 0's
    rendezvous store.  Here on the CPU (gloo; the fits are sleeps that follow the cost estimate): every rank ends with the one-rank
    results, every unit is run exactly once, the ranks finish within 25 % of each other, and the whole search -- dataset
    broadcast, 40 threads' store traffic, all_gather -- stays within 1.6 x the ideal makespan of its sleeps (sum / 40 threads)."""
    G, cv = _sample_grid()
    ds = synthetic_dataset(400, seq_len=48, src_vocab=300, n_labels=20, seed=1, min_len=8)
    single = grid.ShardedGridSearchCV(lambda: None, G, cv=cv, refit=False, fits_per_gpu=2, lockstep=15, fit_and_score_group=rehearsal_group,
                                      recipe_init=False).fit(ds)
    world, threads = 8, 5
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_run_rehearsal, args=(world, _free_port(), out, threads), nprocs=world, join=True)
    assert len(out) == world
    for r in range(world):
        mean, secs, ntasks, n_units, wall, best, slept = out[r]
        assert np.allclose(mean, single.cv_results_["mean_test_score"], rtol=0, atol=1e-12) and best == single.best_index_
    mean, secs, ntasks, n_units, wall, best, _ = out[0]
    assert sum(ntasks) == single.n_tasks_ == 480 and n_units >= world * threads          # cut finer than one unit per thread
    assert max(secs) / (sum(secs) / len(secs)) <= 1.25, secs
    ideal = sum(out[r][6] for r in range(world)) / (world * threads)                      # every thread asleep all the time
    print(f"[rehearsal] {n_units} units, rank seconds {[round(v, 2) for v in secs]}, ideal makespan {ideal:.2f} s, wall (rank 0, with broadcast + gather) {wall:.2f} s")
    assert max(secs) <= 1.6 * ideal + 0.3, (secs, ideal)                                   # (a dead-locked or serialised store shows here)


def _run_world1_group(rank, world, port, out):
    import zlib
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = synthetic_dataset(48, seq_len=8, src_vocab=40, n_labels=4, seed=3, min_len=3)
    gs = grid.ShardedGridSearchCV(lambda: None, GRID, cv=CV, fit_and_score=oracle_fit_and_score, refit=False, force_collectives=True).fit(ds)
    out[rank] = "%08x" % zlib.crc32(np.asarray(gs.cv_results_["mean_test_score"], dtype=np.float64).tobytes())
    dist.destroy_process_group()


def test_one_rank_under_a_launcher_scores_like_one_rank_alone():
    """`python bench.py --gpus 1` (BENCH) and the same command under torch.distributed.run with one process (SCALE, N = 1) take
    different routes through the grid leg -- no process group against a world-1 group with its broadcast, store counter and
    all_gather -- and must report the same scores_crc32 (bench.py's checksum of every candidate's mean test score)."""
    import zlib
    ds = synthetic_dataset(48, seq_len=8, src_vocab=40, n_labels=4, seed=3, min_len=3)
    alone = grid.ShardedGridSearchCV(lambda: None, GRID, cv=CV, fit_and_score=oracle_fit_and_score, refit=False).fit(ds)
    crc = "%08x" % zlib.crc32(np.asarray(alone.cv_results_["mean_test_score"], dtype=np.float64).tobytes())
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_run_world1_group, args=(1, _free_port(), out), nprocs=1, join=True)
    assert out[0] == crc


def test_simulated_schedule_with_the_unit_duration_spread_measured_to_early_stopping():
    """a grid run to EarlyStopping (profiles/r05_grid_long.json: 120 fits, max_epochs 80, patience 30; a third of the
    fits stop early, 26 regroupings) says how far a unit's duration strays from its estimated cost: 0.70 ... 2.57 of the median
    ratio.  The simulated 8-GPU schedule of the bench sample is re-run with every unit's work scaled by those measured ratios
    (dealt out cyclically, several phases): the dynamic counter absorbs the spread -- strong-scaling efficiency stays >= 0.90 at
    8 GPUs and >= 0.93 at 2 / 4 -- while a static deal of the same units loses much more."""
    import json, os, bench
    from slnlp import grid_sim as gs
    cal = json.load(open(os.path.join(ROOT, "profiles", "r04_grid_calibration.json")))
    long_run = json.load(open(os.path.join(ROOT, "profiles", "r05_grid_long.json")))
    ratios = long_run["unit_seconds_over_estimated_cost_normalised"]["values"]
    assert len(ratios) >= 20 and min(ratios) < 0.8 and max(ratios) > 1.5
    per_fit = {}
    for u in cal["solo_units_lockstep15_one_thread"]:
        sh = u["shape"]
        per_fit.setdefault((sh["embedding_size"], sh["hidden_size"], sh["num_heads"], sh["num_layers"]), []).append(u["seconds"] / u["fits"])
    per_fit = {k: float(np.mean(v)) for k, v in per_fit.items()}
    gain = gs.gain_from_throughputs({int(k): v for k, v in cal["throughput_by_threads_lockstep15_kfolds_per_hr"].items() if k.isdigit()})
    ds = synthetic_dataset(bench.GRID_SAMPLES, seq_len=48, src_vocab=300, n_labels=200, seed=1, min_len=8)
    defaults = {"max_epochs": bench.GRID_EPOCHS}
    cands, folds, tasks, order = grid.build_tasks(bench.GRID_SAMPLE, ds.y, bench.GRID_CV, 48, defaults)
    tc = lambda t: grid.estimate_cost(cands[tasks[t][0]], 48, len(folds[tasks[t][1]][0]), defaults)
    total = sum(tc(t) for t in range(len(tasks)))
    shape = lambda u: tuple(cands[tasks[u[0]][0]][k] for k in ("module__embedding_size", "module__hidden_size", "module__num_heads", "module__num_layers"))

    def units_at(world, phase):
        units = grid.build_units(cands, folds, tasks, order, 15, None, tc, grid.unit_cost_ceiling(total, world, 4, grid.UNITS_PER_THREAD), 4)
        costs = [sum(tc(t) for t in u) for u in units]
        work = [per_fit[shape(u)] * len(u) * ratios[(i * 7 + phase) % len(ratios)] for i, u in enumerate(units)]
        return work, costs

    eff, eff_static = {}, {}
    for world in (2, 4, 8):
        worst, worst_static = 1.0, 1.0
        for phase in range(4):
            w1, c1 = units_at(1, phase)
            wn, cn = units_at(world, phase)
            t1 = max(gs.simulate(w1, c1, 1, 4, gain, seed=s)["makespan"] for s in range(4))
            tn = max(gs.simulate(wn, cn, world, 4, gain, seed=s)["makespan"] for s in range(4))
            ts = gs.simulate(wn, cn, world, 4, gain, seed=0, static=True)["makespan"]
            worst, worst_static = min(worst, t1 / (world * tn)), min(worst_static, t1 / (world * ts))
        eff[world], eff_static[world] = worst, worst_static
    print(f"[grid_sim with the measured spread] dynamic {eff}, static {eff_static}")
    assert eff[2] >= 0.93 and eff[4] >= 0.93 and eff[8] >= 0.90, eff
    assert eff_static[8] < eff[8]
