"""In-step vs back-to-back durations of the `roofline` kernel (grouped plane GEMM of one dY) from a rocprofv3 --kernel-trace CSV
of `python bench.py`: the trailing run of consecutive launches is bench.py's own HIP-event measurement, the rest ran inside
train steps.  usage: roofline_kernel_stats.py <kernel_trace.csv> <workgroups> [out.json]"""
import csv, json, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
wgs = int(sys.argv[2])
sel = [(i, r) for i, r in enumerate(rows) if "gemm_planes_kernel" in r["Kernel_Name"]
       and int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) == wgs]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for _, r in sel]
k = len(sel) - 1
while k > 0 and sel[k][0] - sel[k - 1][0] == 1:
    k -= 1
out = {"kernel": f"gemm_planes_kernel x{wgs}", "in_step": {"n": k, "avg_us": round(sum(d[:k]) / k, 2)},
       "back_to_back": {"n": len(d) - k, "avg_us": round(sum(d[k:]) / (len(d) - k), 2)}, "min_us": round(min(d), 2)}
print(json.dumps(out))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
