"""A / B of library builds on ONE box (devices differ by up to 12 %: never compare numbers of two gpurun calls): every variant is a
library under sign-language-nlp_amd/lib/ (libslnlp.so, or libslnlp_probe<name>.so picked through SLNLP_PROBE_LIB=<name>); the
workloads run in fresh child processes, variants interleaved, `--rounds` times.

    python tools/ab_bench.py --variants ,old --workloads cfg2,cfg5,ls15 [--rounds 2]
        ""    = the product library;  old = lib/libslnlp_probeold.so (make VARIANT=old in a checkout of the old tree);
        NAME=VALUE = the product library with that environment knob (e.g. SLNLP_DEC_ROWS=0); join with "+"
workloads: cfg2 / cfg5 / cfg3 / cfg3gru (bench.py lines, ms per step), ls4 / ls15 (tools/bench_lockstep.py, ms per lockstep step),
           nodrop (cfg2 with dropout 0)
"""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--variants", default=",old")
ap.add_argument("--workloads", default="cfg2,cfg5,ls15")
ap.add_argument("--rounds", type=int, default=2)
a = ap.parse_args()


def run(variant, wl):
    env = dict(os.environ)
    env.pop("SLNLP_PROBE_LIB", None)
    for part in (variant.split("+") if variant else []):       # "old", "SLNLP_DEC_ROWS=0", "old+X=1": a library and / or environment knobs
        if "=" in part:
            k, v = part.split("=", 1)
            env[k] = v
        else:
            env["SLNLP_PROBE_LIB"] = part
    if wl.startswith("ls"):
        cmd = [sys.executable, "tools/bench_lockstep.py", "--workload", "cfg2", "--ks", wl[2:], "--steps", "12"]
    elif wl == "nodrop":
        cmd = [sys.executable, "bench.py", "--steps", "100", "--warmup", "20", "--no-grid", "--no-cpu-baseline", "--dropout", "0"]
    else:
        steps = {"cfg2": ("100", "20"), "cfg1": ("100", "20")}.get(wl, ("20", "5"))
        cmd = [sys.executable, "bench.py", "--workload", wl, "--steps", steps[0], "--warmup", steps[1], "--no-grid", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        return None, out.stderr[-400:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    if wl.startswith("ls"):
        return d["results"][0]["ms_per_lockstep_step"], ""
    return d["ms_per_step"], ""


res = {}
for r in range(a.rounds):
    for wl in a.workloads.split(","):
        for v in a.variants.split(","):
            ms, err = run(v, wl)
            res.setdefault((wl, v), []).append(ms)
            print(f"round {r} {wl:8s} {v or 'product':10s} {ms} {err}", flush=True)
print("--- ms per step: min over rounds (all)")
for (wl, v), ms in res.items():
    ok = [m for m in ms if m is not None]
    print(f"{wl:8s} {v or 'product':10s} {min(ok) if ok else None}   {ms}")
