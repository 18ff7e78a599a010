"""Soak: N train steps of a workload twice from the same state; every parameter must be bit-identical at the end and
the loss finite all along (split-K last-arriver, side streams, Philox counters: nothing may depend on timing)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sign-language-nlp_amd")]
import numpy as np, torch
import bench
from slnlp import synth, tf_engine as te, rnn_engine as re_

def run(wl, steps):
    c = dict(bench.WORKLOADS[wl], precision=3)
    cfg, sd = bench.build_sd(c, seed=1)
    eng = (re_.RnnEngine if "rnn" in c else te.TransformerEngine)(cfg, seed=7)
    eng.load_state(sd); eng.set_lr(0.01)
    B, S = c["B"], c["S"]
    Xn, Ln, yn = synth.make_batch(40 * B, S, c["Vs"], c["Vt"], seed=2)
    X, L, y = [torch.from_numpy(a).cuda() for a in (Xn, Ln, yn)]
    losses = []
    for i in range(steps):
        j = (i % 40) * B
        eng.step(X[j:j + B], y[j:j + B], L[j:j + B], 0.9, 0.5, graph=False)
        if i % 50 == 0 or i == steps - 1:
            torch.cuda.synchronize(); losses.append(eng.loss)
    torch.cuda.synchronize()
    return eng.params.clone(), losses

for wl, steps in (("cfg2", 600), ("cfg3", 150), ("cfg3gru", 150), ("cfg1", 1000)):
    a, la = run(wl, steps)
    b, lb = run(wl, steps)
    same = torch.equal(a, b)
    print(f"{wl}: {steps} steps x 2  identical={same}  finite={bool(torch.isfinite(a).all())}  loss {la[0]:.4f} -> {la[-1]:.4f}  (second run {lb[-1]:.4f})", flush=True)
    assert same and np.isfinite(la).all()
print("soak ok")
