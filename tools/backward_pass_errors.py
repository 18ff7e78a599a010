"""What the two-pass gradient products cost in accuracy: the reference's golden training trajectories (tests/golden/tf_*.npz, written
by the reference itself) replayed under (wgrad passes, dgrad passes) = (3, 3), (2, 3), (2, 2) -- slnlp_set_backward_passes -- with
the errors the GPU tests bound: per-tensor gradient norms / heads at step 0, loss and pre-clip gradient norm of every step, every
weight tensor after the last step.

    python tools/backward_pass_errors.py [cfg2 cfg5 ...]
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sign-language-nlp_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import gold
from slnlp import tf_engine as te
from slnlp._lib import load, check


def summary_err(g, prefix, named):
    worst_norm = worst_head = 0.0
    for i, k in enumerate(list(g[prefix + "_names"])):
        v = named.get(k)
        if g[prefix + "_isnone"][i] or v is None:
            continue
        n_ref = float(g[prefix + "_norm"][i])
        scale = max(n_ref, 1e-12)
        worst_norm = max(worst_norm, abs(float(v.double().norm()) - n_ref) / scale)
        head = v.flatten()[:16].double().numpy()
        href = g[prefix + "_head"][i][:head.size].astype(np.float64)
        worst_head = max(worst_head, float(np.abs(head - href).max()) / max(float(np.abs(href).max()), n_ref / max(v.numel(), 1) ** 0.5, 1e-12))
    return worst_norm, worst_head


import ctypes as _C
_w, _d = _C.c_int32(0), _C.c_int32(0)
load().slnlp_get_backward_passes(_C.byref(_w), _C.byref(_d))
_SAVED = (int(_w.value), int(_d.value))          # the default this process started with: restored at the end (new plans copy it)
out = []
for name in (sys.argv[1:] or ["cfg1", "cfg2", "cfg5"]):
    g, c, sd, X, L, y = gold.tf_case(name)
    for passes in ((3, 3), (2, 3), (2, 2)):
        check(load().slnlp_set_backward_passes(*passes), "set_backward_passes")
        cfg = te.make_config(c["E"], c["H"], c["N"], c["F"], c["Vs"], c["Vt"], c["B"], c["S"], 1, 1, 0.0, 3)
        eng = te.TransformerEngine(cfg, seed=0)
        eng.load_state(sd)
        eng.set_lr(0.01)
        Xc, yc = X.cuda(), y.cuda()
        eng.forward(Xc, yc, train=True)
        eng.backward()
        gn, gh = summary_err(g, "grad0", {k: v.cpu() for k, v in eng.views(eng.grads).items()})
        loss_e, norm_e = [], []
        for s in range(len(g["losses"])):
            eng.train_step(Xc, yc, momentum=0.9, max_norm=0.5)
            torch.cuda.synchronize()
            loss_e.append(abs(eng.loss - g["losses"][s]) / g["losses"][s])
            norm_e.append(abs(eng.grad_norm - g["grad_norms"][s]) / g["grad_norms"][s])
        wn, wh = summary_err(g, "wfinal", {k: v.cpu() for k, v in eng.views().items()})
        rec = {"case": name, "wgrad_passes": passes[0], "dgrad_passes": passes[1], "steps": len(loss_e),
               "grad0_worst_tensor_norm_err": gn, "grad0_worst_head_err": gh, "loss_rel_err_max": max(loss_e),
               "grad_norm_rel_err_step0": norm_e[0], "grad_norm_rel_err_max": max(norm_e), "wfinal_worst_norm_err": wn, "wfinal_worst_head_err": wh}
        out.append(rec)
        print(json.dumps({k: (float("%.3g" % v) if isinstance(v, float) else v) for k, v in rec.items()}), flush=True)
check(load().slnlp_set_backward_passes(*_SAVED), "set_backward_passes")        # leave the process default as it was found
