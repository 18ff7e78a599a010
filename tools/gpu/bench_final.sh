# the default bench line (+ 2-rank rehearsal) and the RNN workloads
mkdir -p gpurun_out/r02w
bash tools/gpu/bench_default_and_rehearsal.sh > gpurun_out/final_bench.txt 2>&1 || { tail -5 gpurun_out/final_bench.txt; exit 1; }
for w in cfg3 cfg3gru; do
  timeout -k 10 400 python bench.py --workload $w --steps 40 --warmup 10 --no-grid > gpurun_out/r02w/bench_$w.json 2> gpurun_out/r02w/bench_$w.err || { tail -5 gpurun_out/r02w/bench_$w.err; exit 1; }
done
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_bench_default.json").read().strip().splitlines()[-1])
print("cfg2", d["value"], d["ms_per_step"], "grid", d["grid"]["value"], "roofline", d["roofline"]["achieved"], d["roofline"]["frac"], [(r["fits"], r["value"]) for r in d["concurrent_fits"]["runs"]], d["cpu_baseline"]["value"], d.get("gpu_over_cpu"))
for w in ("cfg3","cfg3gru"):
    x=json.loads(open(f"gpurun_out/r02w/bench_{w}.json").read().strip().splitlines()[-1])
    print(w, x["value"], x["ms_per_step"], x["roofline_step"]["achieved"], x["cpu_baseline"]["value"], x.get("gpu_over_cpu"), [(r["fits"], r["value"]) for r in x["concurrent_fits"]["runs"]])
PY
