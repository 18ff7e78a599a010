"""3 engines concurrently, ONE forward + backward each: which named workspace buffers differ from the solo run?"""
import bisect, ctypes as C, os, sys, threading, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
warnings.filterwarnings("ignore")
import torch
import bench
from slnlp import synth, tf_engine as te, _lib
dev = torch.device("cuda", 0)
c = dict(E=512, H=8, N=2, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=float(sys.argv[1]) if len(sys.argv) > 1 else 0.1, precision=3)
engs, data, sds = {}, {}, {}
for s in (1, 2, 3):
    cfg, sd = bench.build_sd(c, seed=s)
    Xn, _, yn = synth.make_batch(c["B"], c["S"], c["Vs"], c["Vt"], seed=s)
    data[s] = (torch.from_numpy(Xn).to(dev), torch.from_numpy(yn).to(dev))
    engs[s] = te.TransformerEngine(cfg, device=dev, seed=s)
    sds[s] = sd
streams = {s: torch.cuda.Stream() for s in (1, 2, 3)}
buf = C.create_string_buffer(1 << 16)
_lib.check(_lib.load().slnlp_tf_debug_layout(C.byref(engs[1].cfg), buf, len(buf)), "layout")
lay = [(l.split()[0], int(l.split()[1])) for l in buf.value.decode().strip().split("\n")]
names, offs = [n for n, _ in lay], [o for _, o in lay]
act_end = dict(lay)["wp.hi"]

def work(s, out, bar=None):
    torch.cuda.set_device(dev)
    e, (X, y) = engs[s], data[s]
    if bar: bar.wait()
    with torch.cuda.stream(streams[s]):
        e.forward(X, y, train=True); e.backward()
        streams[s].synchronize()
    out[s] = (e.workspace[:act_end].clone(), e.grads.clone())

def reset():
    for s, e in engs.items():
        e.load_state(sds[s]); e.grads.zero_(); e.momentum.zero_(); e.rng[1] = 0; e.workspace[:act_end].zero_()
        _lib.load().slnlp_tf_params_changed(e.handle)
    torch.cuda.synchronize()

reset(); solo = {}
for s in (1, 2, 3): work(s, solo)
for rep in range(5):
    reset(); conc, bar = {}, threading.Barrier(3)
    th = [threading.Thread(target=work, args=(s, conc, bar)) for s in (1, 2, 3)]
    [t.start() for t in th]; [t.join() for t in th]
    for s in (1, 2, 3):
        ne = solo[s][0] != conc[s][0]
        n = int(ne.sum())
        if not n:
            continue
        idx = torch.nonzero(ne).flatten().tolist()
        seen = {}
        for i in idx:
            k = bisect.bisect_right(offs, i) - 1
            seen.setdefault(names[k], []).append((i - offs[k]) // 4)
        print(f"rep {rep} fit {s}: grads equal {torch.equal(solo[s][1], conc[s][1])}; buffers differing:")
        for nm in names:
            if nm in seen:
                fl = sorted(set(seen[nm]))
                print(f"    {nm}: {len(fl)} floats; first {fl[:6]} .. last {fl[-1]}")
        break
    else:
        print(f"rep {rep}: all three identical to solo")
