"""How often does a solo (multi-stream) train step differ from the deterministic single-stream (lockstep K=1) result?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import gold
from test_lockstep_gpu import _engines
from slnlp import synth
from slnlp.lockstep import LockstepGroup

name, tries, drop, mode = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
g, c, sd, X, L, y = gold.tf_case(name)
B, S = c["B"], c["S"]
Xn, _, yn = synth.make_batch(B, S, c["Vs"], c["Vt"], seed=50, min_len=c["min_len"])
Xd, yd = torch.from_numpy(Xn).cuda(), torch.from_numpy(yn).cuda()
if mode == "zeros":
    torch.empty = torch.zeros

lock = _engines(c, 1, (drop,))
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    grp = LockstepGroup(lock)
    grp.set_data(0, [Xd], [yd], B)
    grp.epoch(0, B, True, 0.9, 0.5)
    torch.cuda.synchronize()
    ref = lock[0].grads.clone()
    grp.close()
bad = 0
for t in range(tries):
    e = _engines(c, 1, (drop,))[0]
    if mode == "stream":
        with torch.cuda.stream(st):
            e.train_step(Xd, yd, 0.9, 0.5)
    else:
        e.train_step(Xd, yd, 0.9, 0.5)
    torch.cuda.synchronize()
    bad += int(not torch.equal(e.grads, ref))
print(f"{name} p={drop} mode={mode} AMD_SERIALIZE_KERNEL={os.environ.get('AMD_SERIALIZE_KERNEL')}: {bad}/{tries} solo steps differ from the single-stream result", flush=True)
