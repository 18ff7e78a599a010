"""Per-workgroup timelines of the plane-GEMM launches INSIDE a lockstep train step (K cfg2 fits, the launches as the merged
program issues them, operands as the step leaves them): the timeline probe build (make PROBE=128) records every plane-GEMM
workgroup; one step's records are split into launches by their start times and summarised per launch size.

    SLNLP_PROBE_LIB=128 python tools/probes/probe_lockstep_timeline.py [K=15]
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sign-language-nlp_amd")]
os.environ.setdefault("SLNLP_PROBE_LIB", "128")
import numpy as np, torch
import bench
from slnlp import synth, tf_engine as te
from slnlp.lockstep import LockstepGroup
from slnlp._lib import load
lib = load()
lib.slnlp_probe_ts.restype = C.c_int
lib.slnlp_probe_ts.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((1 << 16, 8), dtype=np.uint64)
def read():
    n = lib.slnlp_probe_ts(buf.ctypes.data, buf.shape[0])
    assert n >= 0
    return buf[:n].astype(np.int64).copy()

K = int(sys.argv[1]) if len(sys.argv) > 1 else 15
dev = torch.device("cuda", 0)
c = dict(bench.WORKLOADS["cfg2"], precision=3)
B, S = c["B"], c["S"]
engs, data = [], []
for f in range(K):
    cfg, sd = bench.build_sd(c, seed=1 + f)
    e = te.TransformerEngine(cfg, device=dev, seed=1 + f)
    e.load_state(sd); e.set_lr(0.01)
    Xn, Ln, yn = synth.make_batch(B, S, c["Vs"], c["Vt"], seed=1 + f)
    engs.append(e)
    data.append((torch.from_numpy(Xn).to(dev), torch.from_numpy(yn).to(dev), torch.from_numpy(Ln).to(dev)))
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    grp = LockstepGroup(engs)
    grp.set_data(0, [d[0] for d in data], [d[1] for d in data], B, [d[2] for d in data])
    for _ in range(3): grp.epoch(0, B, True, 0.9, 0.5)          # records the program, warms up (ONE step per epoch: rows = B)
    torch.cuda.synchronize(); read()
    grp.epoch(0, B, True, 0.9, 0.5)
    torch.cuda.synchronize()
    t = read()
    grp.close()
print(f"{K} cfg2 fits in lockstep, one train step: {len(t)} plane-GEMM workgroups recorded")
t = t[np.argsort(t[:, 0])]
us = lambda a: a / 100.0
# a new launch starts where no recorded workgroup is alive: start of the next > latest end so far
launches, cur, end = [], [0], t[0, 4]
for i in range(1, len(t)):
    if t[i, 0] > end: launches.append(cur); cur = []
    cur.append(i); end = max(end, t[i, 4])
launches.append(cur)
q = lambda a: "%6.2f %6.2f %6.2f" % tuple(np.percentile(a, [10, 50, 90]))
groups = {}
for idx in launches:
    L = t[idx]
    groups.setdefault(len(L), []).append(L)
print(f"{len(launches)} launches; by workgroup count:")
for n in sorted(groups, key=lambda n: -n * len(groups[n])):
    Ls = groups[n]
    span = np.mean([us(L[:, 4].max() - L[:, 0].min()) for L in Ls])
    allw = np.concatenate(Ls)
    ph = [us(allw[:, i + 1] - allw[:, i]) for i in range(4)] + [us(allw[:, 4] - allw[:, 0])]
    ok = np.abs(ph[2]) < 1e6
    start = np.concatenate([us(L[:, 0] - L[:, 0].min()) for L in Ls])
    life_sum = np.mean([us((L[:, 4] - L[:, 0]).sum()) for L in Ls])
    if os.environ.get("PROBE_EPI"):       # a -DSLNLP_PROBE_EPI build: words 6 / 7 are stamps behind (1b) and in front of (2)'s stores
        e = [us(allw[:, 6] - allw[:, 3]), us(allw[:, 7] - allw[:, 6]), us(allw[:, 4] - allw[:, 7])]
        print(f"        epilogue phases: image + element-wise (1a, 1b) {q(e[0])} | barrier {q(e[1])} | row-major stores (2) {q(e[2])}")
    print(f"  {n:5d} workgroups x {len(Ls):2d} launches: span {span:7.2f} us | first K-step {q(ph[0])} | K loop {q(ph[1])} | meeting {q(ph[2][ok])} | "
          f"epilogue {q(ph[3][ok])} | life {q(ph[4])} | start p50/p90 {np.percentile(start, 50):6.1f} {np.percentile(start, 90):6.1f} | "
          f"sum of lives / span = {life_sum / span:5.1f} workgroups alive on average")
