"""Same question as probe_concurrent3.py, but the three fits live in three PROCESSES (one stream each): does a fit's backward
still depend on the others?  Separates 'several hardware queues at once' from 'several streams inside one HIP process'."""
import hashlib, os, subprocess, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
        sys.path.insert(0, p)
    warnings.filterwarnings("ignore")
    import time, torch
    import bench
    from slnlp import synth, tf_engine as te
    s, start_at = int(sys.argv[2]), float(sys.argv[3])
    dev = torch.device("cuda", 0)
    c = dict(E=512, H=8, N=2, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.1, precision=3)
    cfg, sd = bench.build_sd(c, seed=s)
    Xn, _, yn = synth.make_batch(4 * c["B"], c["S"], c["Vs"], c["Vt"], seed=s)
    X, y = torch.from_numpy(Xn).to(dev), torch.from_numpy(yn).to(dev)
    e = te.TransformerEngine(cfg, device=dev, seed=s)
    e.load_state(sd); torch.cuda.synchronize()
    while time.time() < start_at:
        pass
    hs = []
    for rep in range(30):
        e.grads.zero_(); e.rng[1] = 0
        for i in range(4):
            e.forward(X[i * 50:(i + 1) * 50], y[i * 50:(i + 1) * 50], train=True); e.backward()
        torch.cuda.synchronize()
        hs.append(hashlib.md5(e.grads.cpu().numpy().tobytes()).hexdigest()[:8])
    print(s, len(set(hs)), hs[0], flush=True)
    sys.exit(0)
import time
def launch(seeds, delay):
    t = time.time() + delay
    ps = [subprocess.Popen([sys.executable, __file__, "child", str(s), str(t)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for s in seeds]
    return [p.communicate()[0].strip() for p in ps]
print("alone     :", [launch([s], 0)[0] for s in (1, 2, 3)])
print("together  :", launch([1, 2, 3], 25))
