"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) count / avg / total, in dispatch order of first use."""
import csv, collections, glob, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True) if not path.endswith(".csv") else [path]
rows = [r for f in files for r in csv.DictReader(open(f))]
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("slnlp::", "")
    key = (n[:60], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))
    agg.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
order = sorted(agg.items(), key=lambda kv: -sum(kv[1])) if "--by-time" in sys.argv else agg.items()
for k, v in order:
    print(f"{k[0]:62s} grid {k[1]:5d}x{k[2]:<3d} n={len(v):5d} avg {sum(v)/len(v)/1e3:8.2f} us  min {min(v)/1e3:8.2f}  total {sum(v)/1e6:8.2f} ms ({100*sum(v)/tot:4.1f}%)")
print(f"TOTAL kernel time {tot/1e6:.2f} ms over {len(rows)} dispatches")
