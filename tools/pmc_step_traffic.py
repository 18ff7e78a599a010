"""HBM traffic of ONE train step from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected in
separate runs as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Sums the counters over the dispatches between
the last two sgd_kernel launches.  gfx950 correction from the guide: FETCH_SIZE under-reports wide streaming reads
by exactly 2x, WRITE_SIZE is exact; both are in KiB.
usage: pmc_step_traffic.py <fetch_dir> <write_dir> [out.json]"""
import csv, glob, json, sys

def step_sum(d, counter):
    rows = [r for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
    a, b = sgd[-2] + 1, sgd[-1] + 1
    per_kernel, per_shape = {}, {}
    for r in rows[a:b]:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("slnlp::", "")[:40]
        per_kernel[k] = per_kernel.get(k, 0.0) + float(r["Counter_Value"])
        shape = f"{k} x{int(r['Grid_Size']) // int(r['Workgroup_Size'])}"      # kernel + workgroups of the launch
        tot, cnt = per_shape.get(shape, (0.0, 0))
        per_shape[shape] = (tot + float(r["Counter_Value"]), cnt + 1)
    return sum(per_kernel.values()), per_kernel, b - a, per_shape

f, fk, n, fs = step_sum(sys.argv[1], "FETCH_SIZE")
w, wk, _, ws = step_sum(sys.argv[2], "WRITE_SIZE")
# average HBM bytes of ONE launch, per (kernel, workgroup count) -- bench.py's per-kernel roofline reads this
per_launch = {k: {"launches_per_step": c, "fetch_bytes": round(t * 2048 / c), "write_bytes": round(ws.get(k, (0.0, 1))[0] * 1024 / c)}
              for k, (t, c) in fs.items() if k.startswith("gemm_planes")}
out = {"dispatches_per_step": n, "fetch_bytes": f * 1024 * 2, "write_bytes": w * 1024, "hbm_bytes_per_step": f * 2048 + w * 1024,
       "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), WRITE_SIZE exact; KiB -> bytes",
       "top_fetch_MB": {k: round(v * 2048 / 1e6, 1) for k, v in sorted(fk.items(), key=lambda kv: -kv[1])[:8]},
       "top_write_MB": {k: round(v * 1024 / 1e6, 1) for k, v in sorted(wk.items(), key=lambda kv: -kv[1])[:8]},
       "per_launch": per_launch}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
