"""Does a fit's result depend on OTHER fits running concurrently on other streams / host threads?  Engine level."""
import os, sys, threading, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
warnings.filterwarnings("ignore")
import torch
import bench
from slnlp import synth, tf_engine as te
dev = torch.device("cuda", 0)

def run(c, seed, steps, out, barrier=None):
    torch.cuda.set_device(dev)
    cfg, sd = bench.build_sd(c, seed=seed)
    st = torch.cuda.Stream()
    Xn, _, yn = synth.make_batch(steps * c["B"], c["S"], c["Vs"], c["Vt"], seed=seed)
    X, y = torch.from_numpy(Xn).to(dev), torch.from_numpy(yn).to(dev)
    e = te.TransformerEngine(cfg, device=dev, seed=seed)
    e.load_state(sd)
    e.set_lr(0.05)
    torch.cuda.synchronize()
    if barrier: barrier.wait()
    with torch.cuda.stream(st):
        for i in range(steps):
            e.train_step(X[i * c["B"]:(i + 1) * c["B"]], y[i * c["B"]:(i + 1) * c["B"]], 0.9, 0.5)
        st.synchronize()
    out[seed] = e.params.clone().cpu()

for name, c in (("E1024 H4 (hd 256)", dict(E=1024, H=4, N=2, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.1)),
                ("E1024 H16 (hd 64)", dict(E=1024, H=16, N=2, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.1)),
                ("E512 H2 (hd 256)", dict(E=512, H=2, N=2, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.1)),
                ("E512 H8 (hd 64)", dict(E=512, H=8, N=2, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.1)),
                ("E1024 H4 no dropout", dict(E=1024, H=4, N=2, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.0))):
    c = dict(c, precision=3)
    solo = {}
    for s in (1, 2, 3):
        run(c, s, 12, solo)
    for rep in range(2):
        conc, bar = {}, threading.Barrier(3)
        th = [threading.Thread(target=run, args=(c, s, 12, conc, bar)) for s in (1, 2, 3)]
        [t.start() for t in th]; [t.join() for t in th]
        bad = [s for s in (1, 2, 3) if not torch.equal(solo[s], conc[s])]
        print(name, "rep", rep, "concurrent == solo" if not bad else f"DIFFERS for fits {bad}: max |d| {max(float((solo[s]-conc[s]).abs().max()) for s in bad):.3e}", flush=True)
