"""Host cost of recording + merging a lockstep program (first step of a (slot, batch, train) combination) vs a replayed step."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
warnings.filterwarnings("ignore")
import torch
import bench
from slnlp import synth, tf_engine as te
from slnlp.lockstep import LockstepGroup
dev = torch.device("cuda", 0)
for name, c in (("E512 N2", dict(E=512, H=8, N=2, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.1, precision=3)),
                ("E128 N2", dict(E=128, H=4, N=2, F=256, Vs=3000, Vt=202, B=50, S=48, dropout=0.1, precision=3)),
                ("E512 N6", dict(E=512, H=8, N=6, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.1, precision=3))):
    K = 15
    engs, data = [], []
    t0 = time.perf_counter()
    for f in range(K):
        cfg, sd = bench.build_sd(c, seed=1 + f)
        e = te.TransformerEngine(cfg, device=dev, seed=1 + f); e.load_state(sd); e.set_lr(0.01)
        Xn, _, yn = synth.make_batch(4 * 50, 48, 3000, 202, seed=1 + f)
        engs.append(e); data.append((torch.from_numpy(Xn).to(dev), torch.from_numpy(yn).to(dev)))
    torch.cuda.synchronize(); t_build = time.perf_counter() - t0
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        t0 = time.perf_counter(); grp = LockstepGroup(engs); grp.set_data(0, [d[0] for d in data], [d[1] for d in data], 50); torch.cuda.synchronize(); t_grp = time.perf_counter() - t0
        ts = []
        for i in range(4):
            t0 = time.perf_counter(); grp.step(0, i * 50, 50, i, True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); grp.step(0, 0, 50, 0, False); torch.cuda.synchronize(); t_eval_first = time.perf_counter() - t0
        t0 = time.perf_counter(); grp.step(0, 0, 50, 0, False); torch.cuda.synchronize(); t_eval = time.perf_counter() - t0
        grp.close()
    print(f"{name} K={K}: engines {t_build*1e3:.0f} ms, group {t_grp*1e3:.0f} ms, first train step (record+merge+run) {ts[0]*1e3:.1f} ms, replayed {ts[2]*1e3:.1f} ms; "
          f"first eval step {t_eval_first*1e3:.1f} ms, replayed {t_eval*1e3:.1f} ms", flush=True)
    del engs, data, grp
