"""Which part of ONE train step changes when another stream keeps the GPU busy?  Same engine, same batch, same weights:
step once alone, reset, step once next to a load generator; compare log-probs and every gradient tensor bit for bit."""
import os, sys, threading, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
warnings.filterwarnings("ignore")
import torch
import bench
from slnlp import synth, tf_engine as te
dev = torch.device("cuda", 0)
c = dict(E=512, H=8, N=2, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.1, precision=3)
cfg, sd = bench.build_sd(c, seed=1)
Xn, _, yn = synth.make_batch(c["B"], c["S"], c["Vs"], c["Vt"], seed=1)
X, y = torch.from_numpy(Xn).to(dev), torch.from_numpy(yn).to(dev)
e = te.TransformerEngine(cfg, device=dev, seed=1)
st = torch.cuda.Stream()
stop = threading.Event()
kind = sys.argv[1] if len(sys.argv) > 1 else "matmul"

def load_gen():
    torch.cuda.set_device(dev)
    s2 = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev)
    big = torch.empty(1 << 28, device=dev)
    with torch.cuda.stream(s2):
        while not stop.is_set():
            if kind == "matmul":
                for _ in range(20): a @ a
            else:
                for _ in range(20): big.add_(1.0)
            s2.synchronize()

def one_step():
    e.load_state(sd); e.grads.zero_(); e.momentum.zero_(); e.rng[1] = 0; e.set_lr(0.0)
    e.sync_params_version() if hasattr(e, "sync_params_version") else None
    import ctypes
    from slnlp._lib import load
    load().slnlp_tf_params_changed(e.handle)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        lp = e.forward(X, y, train=True).clone()
        e.backward()
        st.synchronize()
    return lp.cpu(), {k: v.clone().cpu() for k, v in e.views(e.grads).items()}

ref = [one_step() for _ in range(3)]
assert all(torch.equal(ref[0][0], r[0]) and all(torch.equal(ref[0][1][k], r[1][k]) for k in r[1]) for r in ref), "solo not deterministic"
print("solo: 3 identical steps")
th = threading.Thread(target=load_gen); th.start()
import time; time.sleep(0.5)
bad_total = {}
for rep in range(6):
    lp, g = one_step()
    bad = [k for k in g if not torch.equal(g[k], ref[0][1][k])]
    print(f"loaded ({kind}) rep {rep}: logp {'same' if torch.equal(lp, ref[0][0]) else 'DIFFERS'}; grads differing: {len(bad)} of {len(g)}", bad[:8], flush=True)
    for k in bad: bad_total[k] = bad_total.get(k, 0) + 1
stop.set(); th.join()
print("tensors that ever differed:", sorted(bad_total, key=lambda k: -bad_total[k])[:40])
