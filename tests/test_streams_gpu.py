"""One kernel sequence per device (include/slnlp.h, slnlp_set_stream_policy; csrc/launch.hpp StepScope).

Kernels of this library running on two hardware queues at once have been measured to read each other's producer output stale
on MI355X / ROCm 7.2: with three fits on three streams the backward results changed from run to run.
The library therefore serialises its step entry points per device -- host threads enqueue whole steps in turn and a step on
another stream waits for the previous stream's tail -- so callers of the C API that bring their own streams get the same
bits as a fit running alone.  The unserialised mode stays reachable for probes and is expected to misbehave."""
import threading

import pytest
import torch

import gold
from slnlp._lib import load

pytestmark = pytest.mark.gpu

STEPS, FITS = 6, 3


def _make(c, sd, seed):
    from slnlp import tf_engine as te
    cfg = te.make_config(c["E"], c["H"], c["N"], c["F"], c["Vs"], c["Vt"], c["B"], c["S"], 1, 1, 0.1, 3)
    eng = te.TransformerEngine(cfg, seed=seed)
    eng.load_state(sd)
    eng.set_lr(0.05)
    return eng


def _solo(c, sd, X, y):
    outs = []
    st = torch.cuda.Stream()
    for f in range(FITS):
        eng = _make(c, sd, 11 + f)
        with torch.cuda.stream(st):
            for _ in range(STEPS):
                eng.train_step(X, y)
        torch.cuda.synchronize()
        outs.append(eng.params.clone())
    return outs


def _threads_on_own_streams(c, sd, X, y):
    engs = [_make(c, sd, 11 + f) for f in range(FITS)]
    streams = [torch.cuda.Stream() for _ in range(FITS)]
    torch.cuda.synchronize()
    go = threading.Barrier(FITS)

    def work(f):
        go.wait()
        with torch.cuda.stream(streams[f]):
            for _ in range(STEPS):
                engs[f].train_step(X, y)

    ths = [threading.Thread(target=work, args=(f,)) for f in range(FITS)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return [e.params.clone() for e in engs]


def test_engines_on_separate_streams_get_the_bits_of_a_fit_running_alone():
    g, c, sd, X, L, y = gold.tf_case("cfg2")
    X, y = X.cuda(), y.cuda()
    ref = _solo(c, sd, X, y)
    for _ in range(3):
        got = _threads_on_own_streams(c, sd, X, y)
        for f in range(FITS):
            assert torch.equal(ref[f], got[f]), f"fit {f} on its own stream differs from the same fit running alone"


@pytest.mark.xfail(strict=False, reason="kernels of several fits on several hardware queues: measured nondeterministic on MI355X / "
                                        "ROCm 7.2 (DESIGN.md section 6); this is the mode slnlp_set_stream_policy(1) exists to prevent")
def test_unserialised_streams_are_the_documented_hazard():
    g, c, sd, X, L, y = gold.tf_case("cfg2")
    X, y = X.cuda(), y.cuda()
    ref = _solo(c, sd, X, y)
    load().slnlp_set_stream_policy(0)
    try:
        for _ in range(4):
            got = _threads_on_own_streams(c, sd, X, y)
            for f in range(FITS):
                assert torch.equal(ref[f], got[f])
    finally:
        load().slnlp_set_stream_policy(1)
