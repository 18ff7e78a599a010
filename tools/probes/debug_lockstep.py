"""Determinism probe: the same first train step from identical fresh engines, solo (multi-stream plan) and lockstep."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import gold
from test_lockstep_gpu import _engines
from slnlp import synth
from slnlp.lockstep import LockstepGroup

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
g, c, sd, X, L, y = gold.tf_case(name)
B, S = c["B"], c["S"]
data = []
for f in range(K):
    Xn, _, yn = synth.make_batch(B, S, c["Vs"], c["Vt"], seed=50 + f, min_len=c["min_len"])
    data.append((torch.from_numpy(Xn).cuda(), torch.from_numpy(yn).cuda()))


def solo_run(drop):
    engs = _engines(c, K, drop)
    out = []
    for e, (Xd, yd) in zip(engs, data):
        e.train_step(Xd, yd, 0.9, 0.5)
        torch.cuda.synchronize()
        out.append((e.scalars[0].item(), e.grads.clone().cpu(), e))
    return out


def lock_run(drop):
    lock = _engines(c, K, drop)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        grp = LockstepGroup(lock)
        grp.set_data(0, [d[0] for d in data], [d[1] for d in data], B)
        grp.epoch(0, B, True, 0.9, 0.5)
        torch.cuda.synchronize()
        out = [(float(grp.loss[0][f][0]), lock[f].grads.clone().cpu(), lock[f]) for f in range(K)]
        grp.close()
    return out


def diff(a, b, tag):
    for f in range(K):
        e = a[f][2]
        ga, gb = e.views(a[f][1]), e.views(b[f][1])
        bad = [k for k in ga if not torch.equal(ga[k], gb[k])]
        worst = max(((float((ga[k] - gb[k]).abs().max() / ga[k].abs().max()), k) for k in bad), default=(0.0, ""))
        print(f"{tag} fit {f}: loss eq {a[f][0] == b[f][0]}; {len(bad)}/{len(ga)} grad tensors differ; last differing (first in backward): "
              f"{bad[-1] if bad else '-'}; worst rel {worst[0]:.2e} {worst[1]}", flush=True)


for p in (0.1, 0.0):
    drop = tuple([p] * K)
    s = [solo_run(drop) for _ in range(3)]
    l = [lock_run(drop) for _ in range(2)]
    diff(s[0], s[1], f"[p={p}] solo0 vs solo1")
    diff(s[0], s[2], f"[p={p}] solo0 vs solo2")
    diff(l[0], l[1], f"[p={p}] lock0 vs lock1")
    diff(s[0], l[0], f"[p={p}] solo0 vs lock0")
    diff(s[1], l[0], f"[p={p}] solo1 vs lock0")
