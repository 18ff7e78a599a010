"""Discrete-event simulation of ShardedGridSearchCV's schedule: N simulated GPUs x ``fits_per_gpu`` host threads pull work
units through the REAL ``WorkCounter`` (one instance per simulated rank, a shared in-memory store in place of the process
group's rendezvous store), in virtual time.  Pure host logic, no GPU: it is how the 8-GPU schedule is checked before anyone
measures it (the reference leaves the same decision to dask's idle-worker dispatch, /root/reference/helper.py:503-519,
main.py:70-78).

Model of one GPU (calibrated on measured 1-GPU runs of bench.py's grid sample):

* a unit u has a solo duration ``work[u]`` seconds (nothing else on the GPU);
* with k units resident the GPU completes ``gain(k)`` solo-seconds of work per second, shared equally: every resident unit
  advances at ``gain(k) / k``.  ``gain`` is the measured aggregate throughput with k host threads over the throughput with one
  (one unit's small launches leave most CUs idle; a second and third unit fill them; beyond four nothing is gained).

The simulator answers: given unit works, a gain curve and the admission rules of ``WorkCounter``, when does every rank finish?
``efficiency = T(1 GPU) / (N x T(N GPUs))``.
"""
import random

from .grid import WorkCounter


class MemoryStore(dict):
    """The two operations WorkCounter needs of a rendezvous store."""
    def add(self, key, inc):
        self[key] = self.get(key, 0) + inc
        return self[key]


def gain_from_throughputs(throughput_by_threads):
    """{threads: measured folds/hr} -> gain(k), linear between the measured points, flat beyond the last."""
    pts = sorted(throughput_by_threads.items())
    base = pts[0][1] / 1.0 if pts[0][0] == 1 else None
    assert base, "the curve needs the one-thread throughput"

    def gain(k):
        if k <= pts[0][0]:
            return pts[0][1] / base
        for (k0, v0), (k1, v1) in zip(pts, pts[1:]):
            if k <= k1:
                return (v0 + (v1 - v0) * (k - k0) / (k1 - k0)) / base
        return pts[-1][1] / base
    return gain


def simulate(work, costs, world, fits_per_gpu, gain, seed=0, static=False, start_skew=0.0):
    """Run the schedule.  ``work[u]``: solo seconds of unit u (list order = the order WorkCounter hands them out);
    ``costs[u]``: the ESTIMATE the admission control sees.  Returns {"makespan", "rank_seconds", "rank_units",
    "timeline": [(rank, unit, start, end)]}.  ``seed`` shuffles the order in which racing host threads reach the counter;
    ``start_skew``: rank r's threads start r x start_skew seconds late (process start-up skew)."""
    rng = random.Random(seed)
    store = MemoryStore()
    counters = [WorkCounter("sim", len(work), static=static, unit_costs=costs, rank=r, world=world, store=store)
                for r in range(world)]
    running = [dict() for _ in range(world)]            # rank -> {unit: [remaining work, start time]}
    threads_left = [fits_per_gpu] * world               # host threads that have not been told "nothing more to run"
    done_at = [0.0] * world
    units_of = [[] for _ in range(world)]
    timeline = []
    now = 0.0

    def admit():
        progressed = True
        while progressed:
            progressed = False
            ranks = [r for r in range(world) if now >= r * start_skew]
            rng.shuffle(ranks)
            for r in ranks:
                idle = threads_left[r] - len(running[r])
                for _ in range(idle):
                    got = counters[r].try_acquire() if not static else counters[r].acquire()
                    if got is WorkCounter.WAIT:
                        break
                    if got is None:
                        threads_left[r] -= 1
                        continue
                    running[r][got] = [float(work[got]), now]
                    units_of[r].append(got)
                    progressed = True

    admit()
    while any(running):
        # next completion over all ranks
        best = None
        for r in range(world):
            k = len(running[r])
            if not k:
                continue
            rate = gain(k) / k
            for u, (rem, _) in running[r].items():
                t = rem / rate
                if best is None or t < best[0]:
                    best = (t, r, u)
        dt, r_fin, u_fin = best
        skew_next = min([r * start_skew - now for r in range(world) if r * start_skew > now] or [float("inf")])
        if skew_next < dt:                              # a late rank joins before the next completion
            dt, r_fin = skew_next, None
        for r in range(world):
            k = len(running[r])
            if k:
                rate = gain(k) / k
                for st in running[r].values():
                    st[0] -= rate * dt
        now += dt
        if r_fin is not None:
            _, start = running[r_fin].pop(u_fin)
            counters[r_fin].release()
            timeline.append((r_fin, u_fin, start, now))
            done_at[r_fin] = now
        admit()
    assert sorted(u for us in units_of for u in us) == list(range(len(work))), "a unit was lost or handed out twice"
    return {"makespan": max(done_at), "rank_seconds": done_at, "rank_units": units_of, "timeline": timeline}


def efficiency(work_by_world, costs_by_world, world, fits_per_gpu, gain, seeds=range(5)):
    """Strong-scaling efficiency T(1) / (world x T(world)), worst over ``seeds``.  ``work_by_world[w]`` / ``costs_by_world[w]``:
    the unit list at world size w (the unit split may depend on it)."""
    t1 = max(simulate(work_by_world[1], costs_by_world[1], 1, fits_per_gpu, gain, seed=s)["makespan"] for s in seeds)
    tn = max(simulate(work_by_world[world], costs_by_world[world], world, fits_per_gpu, gain, seed=s)["makespan"] for s in seeds)
    return t1 / (world * tn), t1, tn
