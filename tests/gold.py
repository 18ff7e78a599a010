"""Helpers shared by the oracle-vs-golden (CPU) and HIP-vs-oracle (GPU) tests."""
import os

import numpy as np
import torch

from slnlp import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))


def tf_case(name):
    g = load("tf_" + name)
    Vs, Vt, E, H, N, F, B, S, min_len = [int(v) for v in g["cfg"]]
    cfg = dict(Vs=Vs, Vt=Vt, E=E, H=H, N=N, F=F, B=B, S=S, min_len=min_len)
    from oracle import transformer_ref as tr
    shapes = tr.param_shapes(E, H, N, F, Vs, Vt)
    assert [k for k, _ in shapes] == list(g["param_order"]), "state_dict order drifted"
    sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(shapes, seed=1).items()}
    X, L, y = [torch.from_numpy(a) for a in synth.make_batch(B, S, Vs, Vt, seed=1, min_len=min_len)]
    return g, cfg, sd, X, L, y


def rnn_case(rnn_type, name):
    g = load(f"rnn_{rnn_type}_{name}")
    Vs, Vt, E, Hd, N, B, S, min_len = [int(v) for v in g["cfg"]]
    cfg = dict(Vs=Vs, Vt=Vt, E=E, Hd=Hd, N=N, B=B, S=S, min_len=min_len, rnn_type=rnn_type)
    from oracle import rnn_ref as rr
    shapes = rr.param_shapes(rnn_type, E, Hd, N, Vs, Vt)
    assert [k for k, _ in shapes] == list(g["param_order"]), "state_dict order drifted"
    sd = {k: torch.from_numpy(v) for k, v in synth.make_weights(shapes, seed=1).items()}
    X, L, y = [torch.from_numpy(a) for a in synth.make_batch(B, S, Vs, Vt, seed=1, min_len=min_len)]
    return g, cfg, sd, X, L, y


def rel_err(a, b):
    """max |a-b| / max |b|  -- the 'rel' of north_star's 1e-3 logits/loss bar."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def check_summary(g, prefix, named, tol, tol_head=None):
    """Compare {name: tensor|None} with a golden 'summarize' record."""
    names = list(g[prefix + "_names"])
    tol_head = 5 * tol if tol_head is None else tol_head   # single elements carry more fp32 noise than norms
    worst = 0.0
    for i, k in enumerate(names):
        v = named.get(k)
        if g[prefix + "_isnone"][i]:
            assert v is None or float(v.abs().max()) == 0.0, f"{k}: reference grad is None"
            continue
        assert v is not None, k
        n_ref = float(g[prefix + "_norm"][i])
        n = float(v.double().norm())
        scale = max(n_ref, 1e-12)
        e1 = abs(n - n_ref) / scale
        head = v.flatten()[:16].double().numpy()
        href = g[prefix + "_head"][i][:head.size].astype(np.float64)
        e2 = float(np.abs(head - href).max()) / max(float(np.abs(href).max()), n_ref / max(v.numel(), 1) ** 0.5, 1e-12)
        worst = max(worst, e1, e2)
        assert e1 < tol and e2 < tol_head, f"{prefix} {k}: norm err {e1:.2e}, head err {e2:.2e}"
    return worst
