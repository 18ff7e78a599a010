"""Average rocprofv3 --pmc counters per kernel name."""
import csv, glob, sys, collections
rows = [r for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "gemm" not in k and len(sys.argv) < 3: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
