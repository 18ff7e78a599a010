"""Fits on several streams (hardware queues) of one GPU get the bits of a fit running alone.

Round 2 measured that they did not (three fits on three streams: backward results changed from run to run) and shipped a
one-queue rule.  Round 3 found the cause -- packed fp32 VALU instructions return wrong values when a workgroup of another kernel
shares the CU -- and builds the library without them; these tests are the canaries that hold every shipped
kernel to it.  Both modes are tested: the library's default (slnlp_set_stream_policy(1): step entry points of different streams
are ordered, for callers of the C API that bring their own streams) and the overlapping mode slnlp.net uses for the host threads
of a grid search (one stream per thread, policy 0)."""
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


def test_overlapping_streams_get_the_same_bits():
    g, c, sd, X, L, y = gold.tf_case("cfg2")
    X, y = X.cuda(), y.cuda()
    ref = _solo(c, sd, X, y)
    load().slnlp_set_stream_policy(0)
    try:
        for _ in range(4):
            got = _threads_on_own_streams(c, sd, X, y)
            for f in range(FITS):
                assert torch.equal(ref[f], got[f]), f"fit {f} beside two fits on other queues differs from the same fit running alone"
    finally:
        load().slnlp_set_stream_policy(1)


@pytest.mark.parametrize("rnn_type", ["lstm", "gru"])
def test_overlapping_rnn_fits_get_the_same_bits(rnn_type):
    from slnlp import rnn_engine as re_
    g, c, sd, X, L, y = gold.rnn_case(rnn_type, "cfg3")
    X, L, y = X.cuda(), L.cuda(), y.cuda()

    def make(seed):
        cfg = re_.make_config(c["rnn_type"], c["E"], c["Hd"], c["N"], c["Vs"], c["Vt"], c["B"], c["S"], 1, 1, 0, 0.1, 3)
        eng = re_.RnnEngine(cfg, seed=seed)
        eng.load_state(sd)
        eng.set_lr(0.01)
        return eng

    def run(engs, streams, together):
        go = threading.Barrier(len(engs) if together else 1)

        def work(f):
            go.wait()
            with torch.cuda.stream(streams[f]):
                for _ in range(STEPS):
                    engs[f].train_step(X, y, L)

        if together:
            ths = [threading.Thread(target=work, args=(f,)) for f in range(len(engs))]
            [t.start() for t in ths]
            [t.join() for t in ths]
        else:
            for f in range(len(engs)):
                work(f)
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return [e.params.clone() for e in engs]

    streams = [torch.cuda.Stream() for _ in range(FITS)]
    ref = run([make(21 + f) for f in range(FITS)], streams, together=False)
    load().slnlp_set_stream_policy(0)
    try:
        for _ in range(3):
            got = run([make(21 + f) for f in range(FITS)], streams, together=True)
            for f in range(FITS):
                assert torch.equal(ref[f], got[f]), f"{rnn_type} fit {f} beside two fits on other queues differs from running alone"
    finally:
        load().slnlp_set_stream_policy(1)


def test_grid_scores_do_not_depend_on_the_stream_mode(monkeypatch):
    """The grid search's host threads on one shared stream and on a stream each (the default): the same scores, bit for bit."""
    import numpy as np
    from slnlp import net
    from slnlp.data import synthetic_dataset
    from slnlp.grid import ShardedGridSearchCV
    import bench
    ds = synthetic_dataset(600, seq_len=48, src_vocab=3000, n_labels=200, seed=1, min_len=8)
    grid = {"lr": [0.1, 0.01], "module__dropout": [0.1, 0.2], "module__embedding_size": [256, 512], "module__num_heads": [4, 8]}
    scores = {}
    for mode in ("device", "thread", "thread"):
        monkeypatch.setattr(net, "STREAM_MODE", mode)
        gs = ShardedGridSearchCV(bench.grid_factory(ds, "cuda:0", 2), grid, cv=3, refit=False, device="cuda:0", fits_per_gpu=3,
                                 lockstep=3).fit(ds)
        scores.setdefault(mode, []).append(np.asarray(gs.cv_results_["mean_test_score"], dtype=np.float64).tobytes())
    load().slnlp_set_stream_policy(1)
    assert scores["device"][0] == scores["thread"][0] == scores["thread"][1]


@pytest.mark.parametrize("module", ["model.EncoderDecoderLSTMAttn", "model.EncoderDecoderGRUAttn"])
def test_rnn_grid_scores_do_not_depend_on_the_stream_mode(monkeypatch, module):
    """The same for the encoder-decoder RNN estimators at working sizes (E 256, hidden 256, 2 layers, batch 50, len 48): three
    host threads on a stream each against one shared stream, twice."""
    import numpy as np
    from slnlp import net
    from slnlp.data import synthetic_dataset
    from slnlp.grid import ShardedGridSearchCV
    from slnlp.net import NeuralNetClassifier
    ds = synthetic_dataset(300, seq_len=48, src_vocab=3000, n_labels=50, seed=3, min_len=8)
    factory = lambda: NeuralNetClassifier(
        module=module, module__dropout=0.1, module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y, module__batch_first=True,
        module__embedding_size=256, module__hidden_size=256, module__num_layers=2, criterion="torch.nn.CrossEntropyLoss",
        criterion__ignore_index=1, optimizer="torch.optim.SGD", optimizer__momentum=0.9, lr=0.05, max_epochs=2, batch_size=50,
        device="cuda:0", gradient_clipping={"gradient_clip_value": 0.5}, scoring=["neg_log_loss"], use_graph=False)
    grid = {"lr": [0.05, 0.02, 0.01], "module__dropout": [0.1, 0.3]}
    scores = {}
    for mode in ("device", "thread", "thread"):
        monkeypatch.setattr(net, "STREAM_MODE", mode)
        gs = ShardedGridSearchCV(factory, grid, cv=3, refit=False, device="cuda:0", fits_per_gpu=3, lockstep=2).fit(ds)
        scores.setdefault(mode, []).append(np.asarray(gs.cv_results_["mean_test_score"], dtype=np.float64).tobytes())
    load().slnlp_set_stream_policy(1)
    assert scores["device"][0] == scores["thread"][0] == scores["thread"][1]


def test_a_fit_that_steps_through_torch_kernels_never_shares_the_gpu(monkeypatch):
    """torch's own kernels are built WITH packed fp32, so a fit whose step goes through them (another
    optimizer: criterion / clip_grad_norm_ / optimizer.step on the GPU) must not run beside other fits' MFMA kernels.  slnlp.net
    gives it the device's shared stream and the device exclusively (_DeviceGate): next to two host threads running fused fits
    on streams of their own it gets the bits of the same fit running alone, and the gate never had a fused fit inside while
    it ran.  The library-wide stream policy is not touched by any of this (thread-scoped opt-out only)."""
    from slnlp import net
    from slnlp.data import synthetic_dataset
    from slnlp.net import NeuralNetClassifier
    monkeypatch.setattr(net, "STREAM_MODE", "thread")
    ds = synthetic_dataset(300, seq_len=48, src_vocab=3000, n_labels=50, seed=5, min_len=8)

    def make(opt, **kw):
        return NeuralNetClassifier(
            module="model.Transformer", module__dropout=0.1, module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y,
            module__batch_first=True, module__embedding_size=256, module__num_heads=4, module__num_layers=2, module__hidden_size=256,
            criterion="torch.nn.CrossEntropyLoss", criterion__ignore_index=1, optimizer=opt, lr=0.02, max_epochs=2, batch_size=50,
            device="cuda:0", gradient_clipping={"gradient_clip_value": 0.5}, scoring=["neg_log_loss"], use_graph=False, **kw)

    def fit_torch_stepper():
        with net.INIT_LOCK:
            torch.manual_seed(3)
            n = make("torch.optim.Adagrad").initialize()
        assert not n._fused and n._stream is net.device_stream("cuda:0", per_thread=False)
        n.partial_fit(ds)
        return torch.cat([p.detach().flatten() for p in n.module_.parameters()]).clone(), n.predict_proba(ds)

    ref_w, ref_p = fit_torch_stepper()
    gate = net.device_gate("cuda:0")
    overlaps, stop = [], threading.Event()
    orig_enter = gate.enter

    def spy_enter(exclusive):
        orig_enter(exclusive)
        if exclusive and gate._shared:
            overlaps.append(gate._shared)
    monkeypatch.setattr(gate, "enter", spy_enter)

    def fused_worker(seed):
        try:
            while not stop.is_set():
                with net.INIT_LOCK:
                    torch.manual_seed(seed)
                    n = make("torch.optim.SGD", optimizer__momentum=0.9).initialize()
                assert n._fused and n._stream is not net.device_stream("cuda:0", per_thread=False)
                n.partial_fit(ds)
        finally:
            net.release_thread_streams()

    ths = [threading.Thread(target=fused_worker, args=(7 + i,)) for i in range(2)]
    [t.start() for t in ths]
    try:
        for _ in range(2):
            w, p = fit_torch_stepper()
            assert torch.equal(w, ref_w), "a torch-stepped fit beside fused fits differs from the same fit alone"
            assert (p == ref_p).all()
    finally:
        stop.set()
        [t.join() for t in ths]
    assert not overlaps
    import ctypes as C
    assert load().slnlp_set_thread_stream_policy(-1) == 0       # (this thread: back to the process-wide default)
