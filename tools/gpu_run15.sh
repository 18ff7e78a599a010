mkdir -p gpurun_out
for l in 10 15; do
timeout -k 10 400 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --fits-per-gpu 3 --lockstep $l > gpurun_out/r02_b15_$l.json 2> gpurun_out/r02_b15.err || { tail -5 gpurun_out/r02_b15.err; exit 1; }
python - $l <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r02_b15_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("lockstep", sys.argv[1], "grid:", d["grid"]["value"], d["grid"]["seconds"], "units", d["grid"]["work_units"])
PY
done
