mkdir -p gpurun_out/r02w
for w in cfg1 e1024 cfg3 cfg3gru cfg5; do
  timeout -k 10 400 python bench.py --workload $w --steps 40 --warmup 10 --no-grid > gpurun_out/r02w/bench_$w.json 2> gpurun_out/r02w/bench_$w.err; rc=$?
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/r02w/bench_$w.err; exit $rc; fi
  python - "$w" <<'PY'
import json, sys
w = sys.argv[1]
d = json.loads(open(f"gpurun_out/r02w/bench_{w}.json").read().strip().splitlines()[-1])
print(w, d["value"], d["ms_per_step"], d.get("parity"), d.get("roofline_step", {}).get("achieved"), (d.get("cpu_baseline") or {}).get("value"))
PY
done
timeout -k 10 400 python bench.py --workload cfg5 --precision 8 --steps 40 --warmup 10 --no-grid --no-cpu-baseline > gpurun_out/r02w/bench_cfg5_p8.json 2> gpurun_out/r02w/bench_cfg5_p8.err || { tail -5 gpurun_out/r02w/bench_cfg5_p8.err; exit 1; }
tail -1 gpurun_out/r02w/bench_cfg5_p8.json | cut -c1-1800
