"""CPU checks of the host-side contract pieces around the hot path: the reference's ``model.util`` helpers against the
golden masks, module copy / pickle keeping the parameter-arena aliasing, the skorch-callback translation, the scorer
wrapper, the reference's ``fit(X=..., y=train.y().to_array())`` spelling, and bench.py's rank spawning."""
import copy
import io
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

import gold
from slnlp import synth


def test_model_util_helpers_match_reference_goldens():
    """generate_mask / generate_padding_mask / resolve_lengths (util.py:11-69) vs tests/golden/masks_pe.npz (written by the
    reference's own functions, tools/gen_golden.py::gen_masks_pe)."""
    from model import util
    g = np.load(os.path.join(gold.GOLD, "masks_pe.npz"))
    Xn, Ln, _ = synth.make_batch(50, 48, 3000, 202, seed=1)
    src = torch.from_numpy(Xn).transpose(0, 1)                       # [len, B], the layout transformer.py:64 produces
    v = util.Vocab(3000)
    assert np.array_equal(util.generate_mask(torch.zeros(1, 50, dtype=torch.long)).numpy(), g["mask1"])
    assert np.array_equal(util.generate_mask(src).numpy(), g["mask48"])
    assert np.array_equal(util.generate_mask(torch.from_numpy(Xn), batch_first=True).numpy(), g["mask48"])
    assert np.array_equal(util.generate_padding_mask(src, v).numpy(), g["padmask"])
    assert np.array_equal(util.resolve_lengths(torch.from_numpy(Xn), v).numpy(), Ln)
    assert util.get_pad_idx(v) == 1 and util.get_bos_idx(v) == 0    # '<bos>' is not in the vocab -> <unk>
    # the table the modules register as `pe` is the reference's, bit for bit
    from slnlp import tf_engine as te
    assert np.array_equal(te.positional_table(64, 128).numpy(), g["pe128"])
    for E in (512, 1024):
        assert np.array_equal(te.positional_table(64, E).numpy()[g["pe_rows"]], g[f"pe{E}"])


def _tiny_module():
    import model
    torch.manual_seed(5)
    return model.Transformer(embedding_size=32, num_heads=4, num_layers=1, hidden_size=64, dropout=0.1,
                             src_vocab=model.util.Vocab(40), tgt_vocab=model.util.Vocab(12), device=torch.device("cpu"),
                             batch_first=True)


def _aliases_arena(m):
    base = m._arena.untyped_storage().data_ptr()
    return all(p.untyped_storage().data_ptr() == base for p in m.parameters())


def test_copy_and_pickle_keep_parameters_aliasing_the_arena():
    m = _tiny_module()
    next(m.parameters()).grad = torch.ones_like(next(m.parameters()))
    assert _aliases_arena(m)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    clones = {"deepcopy": copy.deepcopy(m), "pickle": pickle.loads(pickle.dumps(m)),
              "torch.save": torch.load(buf, weights_only=False)}
    import sklearn.base
    for how, c in clones.items():
        assert c is not m and _aliases_arena(c), how
        assert c._arena.untyped_storage().data_ptr() != m._arena.untyped_storage().data_ptr(), how
        assert list(c.state_dict()) == list(m.state_dict()), how
        assert all(torch.equal(a, b) for a, b in zip(c.state_dict().values(), m.state_dict().values())), how
        assert torch.equal(next(c.parameters()).grad, next(m.parameters()).grad), how
        with torch.no_grad():                                     # an in-place edit of a parameter is seen by the arena
            c.get_parameter("linear.bias").fill_(7.0)
        off = dict((n, o) for n, _, o in c._entries)["linear.bias"]
        assert float(c._arena[off]) == 7.0 and float(m._arena[off]) != 7.0, how
        c.load_state_dict(m.state_dict())                         # and load_state_dict writes through
        assert float(c._arena[off]) == float(m._arena[off]), how
    v = pickle.loads(pickle.dumps(m.src_vocab))
    assert len(v) == 40 and v.stoi["<pad>"] == 1 and v.stoi["never seen"] == 0


def test_rnn_module_copy_keeps_aliasing():
    import model
    m = model.EncoderDecoderGRUAttn(src_vocab=model.util.Vocab(30), tgt_vocab=model.util.Vocab(9), batch_first=True,
                                    embedding_size=16, hidden_size=16, num_layers=2, dropout=0.1)
    c = copy.deepcopy(m)
    assert _aliases_arena(c) and list(c.state_dict()) == list(m.state_dict())


def test_skorch_style_callbacks_are_translated_or_rejected():
    from slnlp.net import NeuralNetClassifier, ScoringWrapper
    mk = lambda cls_name, **kw: type(cls_name, (), kw)()
    cbs = [("checkpoint", mk("Checkpoint", monitor="valid_loss_best", dirname="/tmp/x")),
           ("early_stopping", mk("EarlyStopping", patience=30, threshold=1e-4, threshold_mode="rel", monitor="valid_loss",
                                 lower_is_better=True)),
           ("gradient_clipping", mk("GradientNormClipping", gradient_clip_value=0.5)),
           ("lr_scoring", mk("EpochScoring", scoring=lambda net, X, y=None: 0.0, name="lr", on_train=False)),
           ("lr_scheduler", mk("LRScheduler", policy="ReduceLROnPlateau", monitor="valid_loss", step_every="epoch",
                               kwargs={"factor": 0.2, "patience": 5})),
           ("score_valid_accuracy", mk("EpochScoring", scoring=ScoringWrapper("accuracy"), name="valid_accuracy", on_train=False)),
           ("score_train_f1", mk("EpochScoring", scoring="f1_weighted", name="train_f1_weighted", on_train=True)),
           mk("PrintLog")]
    net = NeuralNetClassifier(module="model.Transformer", callbacks=cbs)                 # helper.py:197-273's list
    p = net.get_params()
    assert p["checkpoint_dir"] == "/tmp/x" and p["gradient_clipping"] == {"gradient_clip_value": 0.5}
    assert p["early_stopping"] == {"patience": 30, "threshold": 1e-4, "threshold_mode": "rel"}
    assert p["lr_scheduler"] == {"policy": "ReduceLROnPlateau", "factor": 0.2, "patience": 5}
    assert p["scoring"] == ["accuracy", "f1_weighted"]
    with pytest.raises(TypeError, match="no equivalent"):
        NeuralNetClassifier(module="model.Transformer", callbacks=[mk("WandbLogger")])
    with pytest.raises(ValueError, match="ReduceLROnPlateau"):
        NeuralNetClassifier(module="model.Transformer", callbacks=[mk("LRScheduler", policy="StepLR")])
    with pytest.raises(ValueError, match="valid_loss"):
        NeuralNetClassifier(module="model.Transformer").set_params(callbacks=[mk("EarlyStopping", monitor="train_loss")])


def test_scoring_wrapper_keywords_and_sign():
    from sklearn.base import BaseEstimator, ClassifierMixin
    from sklearn.metrics import f1_score, log_loss
    from slnlp.net import ScoringWrapper

    class Fixed(ClassifierMixin, BaseEstimator):
        def __init__(self, proba=None):
            self.proba = proba
            self.classes_ = np.arange(4)

        def fit(self, X, y):
            return self

        def predict_proba(self, X):
            return self.proba

        def predict(self, X):
            return self.proba.argmax(1)
    rs = np.random.RandomState(0)
    proba = rs.dirichlet(np.ones(4), size=12)
    y = rs.randint(0, 3, size=12)                                 # class 3 never occurs: labels= must carry it
    est = Fixed(proba)
    nll = ScoringWrapper("neg_log_loss", labels=[0, 1, 2, 3])
    assert np.isclose(nll(est, None, y), -log_loss(y, proba, labels=[0, 1, 2, 3]))
    # reference quirk kept (helper.py:548-550 reads the scorer's sign): the negated log-loss scorer has sign -1, so the
    # wrapper says "lower is better" for neg_log_loss and EpochScoring tracks its best epoch that way
    assert nll.greater_is_better is False
    f1 = ScoringWrapper("f1_weighted")
    assert np.isclose(f1(est, None, y), f1_score(y, proba.argmax(1), average="weighted", zero_division=0))
    acc = ScoringWrapper("accuracy")
    assert f1.greater_is_better and acc.greater_is_better
    assert np.isclose(acc(est, None, y), (proba.argmax(1) == y).mean())
    assert (nll.score, f1.score, repr(acc)) == ("neg_log_loss", "f1_weighted", "ScoringWrapper('accuracy')")
    # wrappers are independent: building one must not leak its keywords into the next scorer of the same name
    assert "labels" not in ScoringWrapper("neg_log_loss").scorer._kwargs or ScoringWrapper("neg_log_loss").scorer._kwargs["labels"] is None


def test_reference_fit_call_spelling():
    """main.py:77: ``gs.fit(X=train_data.X(), y=train_data.y().to_array())``."""
    from slnlp.data import synthetic_dataset
    ds = synthetic_dataset(20, seq_len=8, src_vocab=30, n_labels=3, seed=2, min_len=2)
    assert ds.X() is ds
    y = ds.y().to_array()
    assert type(y) is np.ndarray and np.array_equal(y, np.asarray(ds.y)) and np.array_equal(ds[np.arange(5)].y, y[:5])
    assert torch.from_numpy(np.asarray(ds.y)).dtype == torch.int64
    from slnlp.net import NeuralNetClassifier
    assert NeuralNetClassifier._as_dataset(ds.X(), y) is ds


def test_bench_spawns_one_fresh_process_per_rank(monkeypatch):
    sys.path.insert(0, os.path.dirname(gold.GOLD.rstrip("/")).rsplit("/tests", 1)[0])
    import bench
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None):
            started.append((cmd, env))

        def wait(self):
            return 0
    import subprocess
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    assert bench.spawn_ranks(2) == 0
    assert len(started) == 2
    ports = {env["MASTER_PORT"] for _, env in started}
    assert len(ports) == 1 and all(env["MASTER_ADDR"] == "127.0.0.1" and env["WORLD_SIZE"] == "2" for _, env in started)
    assert [env["RANK"] for _, env in started] == ["0", "1"] and [env["LOCAL_RANK"] for _, env in started] == ["0", "1"]
    assert all(cmd[0] == sys.executable and cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "2", "--steps", "3"] for cmd, _ in started)
    assert not torch.cuda.is_initialized()                          # the parent never touched the GPU


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the shipped package may import it (only tests/, __graft_entry__.smoke()
    and bench.py's parity / cpu_baseline legs do)."""
    import re
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sign-language-nlp_amd")
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)", re.M)
    offenders = []
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                path = os.path.join(root, f)
                if pat.search(open(path, encoding="utf-8").read()):
                    offenders.append(path)
    assert not offenders, offenders


@pytest.mark.parametrize("cls_name,kw", [("Transformer", dict(embedding_size=64, num_heads=4, num_layers=2, hidden_size=96)),
                                         ("EncoderDecoderLSTMAttn", dict(embedding_size=32, hidden_size=48, num_layers=2)),
                                         ("EncoderDecoderGRUAttn", dict(embedding_size=32, hidden_size=48, num_layers=2))])
def test_recipe_init_draws_the_reference_distributions(cls_name, kw):
    """init="recipe" (what the grid's CV fits use): no torch modules are built and the global RNG is not consumed; every
    tensor has the distribution the reference-identical init gives it (same zeros / ones, same spread and range), the draw is
    a pure function of the torch seed, and the parameters still alias the arena."""
    import model
    from slnlp.data import synthetic_dataset
    ds = synthetic_dataset(50, seq_len=12, src_vocab=64, n_labels=6, seed=5, min_len=3)
    cls = getattr(model, cls_name)
    mk = lambda **extra: cls(src_vocab=ds.vocab_X, tgt_vocab=ds.vocab_y, batch_first=True, dropout=0.1, **kw, **extra)
    torch.manual_seed(3)
    ref = mk().state_dict()
    torch.manual_seed(3)
    before = torch.random.get_rng_state()
    rec_mod = mk(init="recipe")
    assert torch.equal(torch.random.get_rng_state(), before), "recipe init must not consume the global CPU stream"
    rec = rec_mod.state_dict()                                        # materialises the draw (here: on the CPU)
    assert list(ref) == list(rec)
    for k, a in ref.items():
        b = rec[k]
        assert a.shape == b.shape, k
        if k.endswith(".pe"):
            continue
        a, b = a.float(), b.float()
        const = float(a.std()) == 0.0 if a.numel() > 1 else True
        if const and a.numel() > 1:
            assert torch.equal(a, b), k                               # biases at 0, LayerNorm at (1, 0)
        elif a.numel() >= 256:
            assert abs(float(a.std()) - float(b.std())) <= 0.12 * float(a.std()), (k, float(a.std()), float(b.std()))
            assert abs(float(b.mean())) <= 4 * float(a.std()) / a.numel() ** 0.5 + 1e-3, k
            if "embed" not in k:                                      # uniform tensors: same bound
                assert float(b.abs().max()) <= float(a.abs().max()) * 1.1, k
        assert torch.equal(a == 0, b == 0) or not k.endswith("embed.weight"), k      # the zero padding rows
    torch.manual_seed(3)
    again = mk(init="recipe").state_dict()
    assert all(torch.equal(rec[k], again[k]) for k in rec)
    torch.manual_seed(4)
    other = mk(init="recipe").state_dict()
    assert any(not torch.equal(rec[k], other[k]) for k in rec)
    p0 = next(rec_mod.parameters())
    assert p0.untyped_storage().data_ptr() == rec_mod._arena.untyped_storage().data_ptr()


def test_no_kernel_of_the_built_library_needs_more_than_256_registers():
    """The code object's own metadata, no compiler run: every kernel's register allocation (VGPRs + AGPRs, `.vgpr_count` on the
    unified file) stays within 256 (spill-free occupancy; layernorm_bwd once took 347 = 256 + 91 AGPRs)."""
    import importlib.util
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(ROOT, "sign-language-nlp_amd", "lib", "libslnlp.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    spec = importlib.util.spec_from_file_location("kernel_registers", os.path.join(ROOT, "tools", "kernel_registers.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    ks = kr.library_kernels(lib)
    assert len(ks) > 50, "no kernels found in the library's code objects"
    over = [(v, n) for v, a, n in ks if v > 256]
    assert not over, f"kernels above 256 registers: {over}"


def test_the_built_library_contains_no_packed_fp32_instructions():
    """v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 return wrong values in 16-lane pieces
    of a wave when a workgroup of another kernel shares the CU (MI355X / ROCm 7.2; DESIGN.md section 6: two independent victim
    kernels, each flipped from nondeterministic to bit-stable by compiling without them).  The Makefile builds with
    -target-feature -packed-fp32-ops; this disassembles every code object of the built library and holds it to that."""
    import importlib.util
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(ROOT, "sign-language-nlp_amd", "lib", "libslnlp.so")
    spec = importlib.util.spec_from_file_location("check_no_packed_fp32", os.path.join(ROOT, "tools", "check_no_packed_fp32.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    if not os.path.exists(lib) or not os.path.exists(chk.OBJDUMP):
        pytest.skip("library or llvm-objdump not present")
    lines, hits = chk.packed_fp32_hits(lib)          # (the Makefile runs the same check after linking)
    assert lines > 100000, "disassembly looks empty"
    assert not hits, f"{len(hits)} packed fp32 instructions in the built library (first: {hits[0]})"


def test_device_gate_orders_shared_and_exclusive_holders_and_refuses_the_self_deadlock():
    """slnlp.net._DeviceGate: fused fits hold the GPU shared, a fit that steps through torch kernels exclusively -- shared holders
    overlap, an exclusive holder is alone, both nest per thread, and the one request that can never be granted (exclusive from a
    thread that already holds the gate shared) raises instead of waiting for itself."""
    import threading, time
    import pytest
    from slnlp.net import _DeviceGate
    g = _DeviceGate()
    inside, peak, excl_alone = [0], [0], [True]
    lock = threading.Lock()

    def shared():
        g.enter(False)
        g.enter(False)                                   # nested shared call
        with lock:
            inside[0] += 1
            peak[0] = max(peak[0], inside[0])
        time.sleep(0.05)
        with lock:
            inside[0] -= 1
        g.leave(False)
        g.leave(False)

    def exclusive():
        g.enter(True)
        g.enter(False)                                   # the exclusive holder's nested (predict) call
        with lock:
            excl_alone[0] = excl_alone[0] and inside[0] == 0
        time.sleep(0.02)
        with lock:
            excl_alone[0] = excl_alone[0] and inside[0] == 0
        g.leave(False)
        g.leave(True)

    ts = [threading.Thread(target=shared) for _ in range(3)] + [threading.Thread(target=exclusive)] + [threading.Thread(target=shared) for _ in range(2)]
    for t in ts: t.start()
    for t in ts: t.join(10)
    assert not any(t.is_alive() for t in ts)
    assert peak[0] >= 2 and excl_alone[0]
    g.enter(False)
    with pytest.raises(RuntimeError, match="device gate"):
        g.enter(True)
    g.leave(False)
    g.enter(True); g.leave(True)                         # ... and the gate is still usable
