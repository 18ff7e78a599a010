"""Locate the first differing workspace buffer between identical solo train steps (multi-stream plan)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import gold
from test_lockstep_gpu import _engines
from slnlp import synth, _lib

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
tries = int(sys.argv[2]) if len(sys.argv) > 2 else 6
drop = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
g, c, sd, X, L, y = gold.tf_case(name)
B, S = c["B"], c["S"]
Xn, _, yn = synth.make_batch(B, S, c["Vs"], c["Vt"], seed=50, min_len=c["min_len"])
Xd, yd = torch.from_numpy(Xn).cuda(), torch.from_numpy(yn).cuda()

def layout(cfg):
    buf = C.create_string_buffer(1 << 16)
    _lib.check(_lib.load().slnlp_tf_debug_layout(C.byref(cfg), buf, len(buf)), "layout")
    rows = [l.split() for l in buf.value.decode().strip().split("\n")]
    return [(n, int(o)) for n, o in rows]

def run():
    e = _engines(c, 1, (drop,))[0]
    e.forward(Xd, yd, train=True)
    e.backward()
    torch.cuda.synchronize()
    return e

ref = run()
lay = layout(ref.cfg)
act_end = dict(lay)["wp.hi"]            # compare activations / gradients only (before the weight planes)
for t in range(tries):
    e = run()
    same_g = torch.equal(e.grads, ref.grads)
    a, b = ref.workspace[:act_end], e.workspace[:act_end]
    ne = (a != b)
    n = int(ne.sum())
    print(f"try {t}: grads equal {same_g}; differing workspace bytes {n}", flush=True)
    if n:
        idx = torch.nonzero(ne).flatten()
        offs = [o for _, o in lay]
        names = [nm for nm, _ in lay]
        import bisect
        seen = {}
        for i in idx[:: max(1, len(idx) // 200000)].tolist():
            k = bisect.bisect_right(offs, i) - 1
            seen.setdefault(names[k], []).append(i - offs[k])
        for nm in names:
            if nm in seen:
                v = seen[nm]
                fl = [x // 4 for x in v]
                E = c["E"]
                print(f"   {nm}: {len(v)} sampled bytes differ; float index range {min(fl)}..{max(fl)}; rows {min(fl)//E}..{max(fl)//E} cols {min(x % E for x in fl)}..{max(x % E for x in fl)} (if [rows,E])")
        break
