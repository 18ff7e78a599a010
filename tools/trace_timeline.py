"""Timeline of ONE graph replay from a rocprofv3 --kernel-trace CSV: every dispatch between the last two
sgd_kernel launches, in start order, with queue id, start offset, duration and the idle gap on its own queue."""
import csv, glob, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True) if not path.endswith(".csv") else [path]
rows = sorted((r for f in files for r in csv.DictReader(open(f))), key=lambda r: int(r["Start_Timestamp"]))
sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
a, b = sgd[-2] + 1, sgd[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
last_end = {}
busy_main = 0
mainq = rows[b - 1]["Queue_Id"]
for r in rows[a:b]:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("slnlp::", "")[:44]
    g = f'{int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)}x{r["Grid_Size_Y"]}'
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    if q == mainq: busy_main += e - s
    print(f"q{q:<2s} +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  gap {gap:6.1f}  {n:44s} {g}")
print(f"replay wall {(int(rows[b-1]['End_Timestamp']) - t0) / 1e3:.1f} us; main queue q{mainq} busy {busy_main / 1e3:.1f} us; {b - a} dispatches")
