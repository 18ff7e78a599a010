"""probe_concurrent4.py with the detail needed to name the kernel: 3 engines on 3 streams (slnlp_set_stream_policy(0): the
library's own serialisation off), ONE forward + backward each, workspace zeroed before.  For every fit that differs from its
solo run: the differing buffers in layout order with their rows / columns, and for the first few wrong elements the solo
value, the concurrent value and whether the concurrent value is 0 (= the zeroed workspace: the kernel's output never arrived
or its input was read before it was written)."""
import bisect, ctypes as C, os, sys, threading, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
warnings.filterwarnings("ignore")
import torch
import bench
from slnlp import synth, tf_engine as te, _lib
dev = torch.device("cuda", 0)
_lib.load().slnlp_set_stream_policy(0)
c = dict(E=512, H=8, N=int(sys.argv[4]) if len(sys.argv) > 4 else 2, F=512, Vs=3000, Vt=202, B=50, S=48,
         dropout=float(sys.argv[1]) if len(sys.argv) > 1 else 0.1, precision=3)
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # > 0: that many whole train steps (with the optimizer) instead of one forward + backward
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
act_end = dict(lay)["opt_partials"]     # activations and gradients only: the device tables behind them stay
E, F, M, B = c["E"], c["F"], c["B"] * c["S"], c["B"]

def width(name):          # row length (floats) of a buffer, for row / column reporting
    base = name.split(".")[-1]
    if base in ("qkv", "gqkv"): return 3 * E
    if base in ("h", "gh"): return F
    if base in ("st1", "st2", "st3", "st_mem", "st_fin"): return 2
    return E

def work(s, out, bar=None):
    torch.cuda.set_device(dev)
    e, (X, y) = engs[s], data[s]
    if bar: bar.wait()
    with torch.cuda.stream(streams[s]):
        if STEPS:
            for _ in range(STEPS): e.train_step(X, y)
        else:
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
reset(); solo2 = {}
for s in (1, 2, 3): work(s, solo2)
print("solo repeat identical:", all(torch.equal(solo[s][0], solo2[s][0]) and torch.equal(solo[s][1], solo2[s][1]) for s in solo), flush=True)
found = 0
for rep in range(REPS):
    reset(); conc, bar = {}, threading.Barrier(3)
    th = [threading.Thread(target=work, args=(s, conc, bar)) for s in (1, 2, 3)]
    [t.start() for t in th]; [t.join() for t in th]
    for s in (1, 2, 3):
        a8, b8 = solo[s][0], conc[s][0]
        ne = a8 != b8
        if not int(ne.sum()):
            continue
        found += 1
        af, bf = a8.view(torch.float32), b8.view(torch.float32)
        idx = torch.nonzero(af.view(torch.int32) != bf.view(torch.int32)).flatten().tolist()
        seen = {}
        for i in idx:
            k = bisect.bisect_right(offs, i * 4) - 1
            seen.setdefault(names[k], []).append(i - offs[k] // 4)
        print(f"rep {rep} fit {s}: grads equal {torch.equal(solo[s][1], conc[s][1])}; {len(idx)} floats differ in {len(seen)} buffers:")
        for nm in names:
            if nm not in seen:
                continue
            fl = sorted(seen[nm]); w = width(nm)
            rows = sorted({f // w for f in fl})
            cols = sorted({f % w for f in fl})
            base = offs[names.index(nm)] // 4
            runs, start, prev = [], fl[0], fl[0]
            for f in fl[1:]:
                if f != prev + 1:
                    runs.append((start, prev - start + 1)); start = f
                prev = f
            runs.append((start, prev - start + 1))
            zeros = sum(1 for f in fl if float(bf[base + f]) == 0.0)
            print(f"    {nm:14s} {len(fl):7d} floats, rows {rows[:8]}{'..' if len(rows) > 8 else ''} ({len(rows)} rows), cols {cols[0]}..{cols[-1]}, "
                  f"{len(runs)} runs (first: start {runs[0][0] % w} len {runs[0][1]}), concurrent value == 0 in {zeros}")
            if len(fl) <= 64 or nm == next(n for n in names if n in seen):
                for f in fl[:4]:
                    print(f"        [{f // w},{f % w}] solo {float(af[base + f]):+.6e} concurrent {float(bf[base + f]):+.6e}")
        if found >= 3:
            sys.exit(0)
        break
    else:
        print(f"rep {rep}: all three identical to solo", flush=True)
