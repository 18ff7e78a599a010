"""Average rocprofv3 --pmc counters per (kernel name, grid size).  usage: pmc_summary.py <dir> [substring]"""
import csv, glob, sys, collections
rows = [r for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
want = sys.argv[2] if len(sys.argv) > 2 else "gemm"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("slnlp::", "")[:50]
    if want not in name and want != "all": continue
    agg[(name, r.get("Grid_Size", r.get("Grid_Size_X", "?")))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), d in sorted(agg.items()):
    print(f"{k}  grid {g}")
    for c, v in sorted(d.items()):
        print(f"   {c:28s} avg {sum(v)/len(v):16.1f}  (n={len(v)})")
