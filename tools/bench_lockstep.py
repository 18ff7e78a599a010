#!/usr/bin/env python3
"""Aggregate train seq/s of K fits advancing in lockstep (one launch sequence) vs one fit alone.

    python tools/bench_lockstep.py [--workload cfg2] [--ks 1,2,4,8] [--steps 30]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--ks", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--precision", type=int, default=3)
    args = ap.parse_args()
    from slnlp import synth, tf_engine as te, rnn_engine as re_
    from slnlp.lockstep import LockstepGroup
    dev = torch.device("cuda", 0)
    c = dict(bench.WORKLOADS[args.workload], precision=args.precision)
    B, S = c["B"], c["S"]
    rows = args.steps * B
    out = {"workload": args.workload, "steps": args.steps, "results": []}
    st = torch.cuda.Stream()
    for K in [int(k) for k in args.ks.split(",")]:
        engs, data = [], []
        for f in range(K):
            cfg, sd = bench.build_sd(c, seed=1 + f)
            e = (re_.RnnEngine if "rnn" in c else te.TransformerEngine)(cfg, device=dev, seed=1 + f)
            e.load_state(sd)
            e.set_lr(0.01)
            Xn, Ln, yn = synth.make_batch(rows, S, c["Vs"], c["Vt"], seed=1 + f)
            engs.append(e)
            data.append((torch.from_numpy(Xn).to(dev), torch.from_numpy(yn).to(dev), torch.from_numpy(Ln).to(dev)))
        with torch.cuda.stream(st):
            grp = LockstepGroup(engs)
            grp.set_data(0, [d[0] for d in data], [d[1] for d in data], B, [d[2] for d in data])
            grp.epoch(0, B, True, 0.9, 0.5)              # warm-up pass (records the program)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            grp.epoch(0, B, True, 0.9, 0.5)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            n = grp.num_launches(0, B, True)
            grp.close()
        r = {"K": K, "seq_per_s": round(K * rows / dt, 1), "ms_per_lockstep_step": round(dt / args.steps * 1e3, 3),
             "launches_per_step": n}
        out["results"].append(r)
        print(json.dumps(r), flush=True)
        del engs, data, grp
        torch.cuda.empty_cache()
    base = out["results"][0]["seq_per_s"] / out["results"][0]["K"]
    for r in out["results"]:
        r["vs_one_fit"] = round(r["seq_per_s"] / base, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
