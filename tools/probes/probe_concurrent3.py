"""Narrow down: 3 engines of this library in 3 threads -- what is the smallest piece whose result depends on the others?"""
import os, sys, threading, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
warnings.filterwarnings("ignore")
import torch
import bench
from slnlp import synth, tf_engine as te
from slnlp._lib import load
dev = torch.device("cuda", 0)
c = dict(E=512, H=8, N=2, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.1, precision=3)
mode = sys.argv[1]
engs, data, sds = {}, {}, {}
for s in (1, 2, 3):
    cfg, sd = bench.build_sd(c, seed=s)
    Xn, _, yn = synth.make_batch(4 * c["B"], c["S"], c["Vs"], c["Vt"], seed=s)
    data[s] = (torch.from_numpy(Xn).to(dev), torch.from_numpy(yn).to(dev))
    engs[s] = te.TransformerEngine(cfg, device=dev, seed=s)
    sds[s] = sd
if os.environ.get("ONE_STREAM"):
    _st = torch.cuda.Stream()
    streams = {s: _st for s in (1, 2, 3)}
else:
    streams = {s: torch.cuda.Stream() for s in (1, 2, 3)}

def work(s, out, bar=None):
    torch.cuda.set_device(dev)
    e, (X, y) = engs[s], data[s]
    B = c["B"]
    if bar: bar.wait()
    with torch.cuda.stream(streams[s]):
        if mode == "fwd":
            r = torch.cat([e.forward(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B], train=True).clone() for i in range(4)])
        elif mode == "fwdbwd":
            for i in range(4):
                e.forward(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B], train=True); e.backward()
            r = e.grads.clone()
        else:
            for i in range(4):
                e.train_step(X[i * B:(i + 1) * B], y[i * B:(i + 1) * B], 0.9, 0.5)
            r = e.params.clone()
        streams[s].synchronize()
    out[s] = r.cpu()

def reset():
    for s, e in engs.items():
        e.load_state(sds[s]); e.grads.zero_(); e.momentum.zero_(); e.rng[1] = 0; e.set_lr(0.05)
        load().slnlp_tf_params_changed(e.handle)
    torch.cuda.synchronize()

reset(); solo = {}
for s in (1, 2, 3): work(s, solo)
reset(); solo2 = {}
for s in (1, 2, 3): work(s, solo2)
print(mode, "solo repeat identical:", all(torch.equal(solo[s], solo2[s]) for s in solo))
for rep in range(4):
    reset(); conc, bar = {}, threading.Barrier(3)
    th = [threading.Thread(target=work, args=(s, conc, bar)) for s in (1, 2, 3)]
    [t.start() for t in th]; [t.join() for t in th]
    bad = [s for s in (1, 2, 3) if not torch.equal(solo[s], conc[s])]
    print(mode, "rep", rep, "concurrent == solo" if not bad else f"DIFFERS for {bad}, max |d| {max(float((solo[s]-conc[s]).abs().max()) for s in bad):.3e}", flush=True)
    if bad and mode != "fwd" and os.environ.get("VERBOSE"):
        s0 = bad[0]
        for name, shape, off in engs[s0].entries:
            n = 1
            for d in shape: n *= d
            a, b = solo[s0][off:off + n], conc[s0][off:off + n]
            if not torch.equal(a, b):
                dd = (a - b).abs()
                print(f"     fit {s0} {name} {tuple(shape)}: {int((dd > 0).sum())} elements differ, max |d| {float(dd.max()):.3e}, |ref| max {float(a.abs().max()):.3e}")
