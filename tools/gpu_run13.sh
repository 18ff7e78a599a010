mkdir -p gpurun_out
timeout -k 10 600 python tools/probe_determinism.py 2>&1 | grep -E "mean|vs|fold" > gpurun_out/r02_determinism.txt; cat gpurun_out/r02_determinism.txt | cut -c1-200
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02_full.log 2>&1; rc=$?
tail -3 gpurun_out/r02_full.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E |^FAILED|^ERROR" gpurun_out/r02_full.log | head -20 | cut -c1-300; exit $rc; fi
timeout -k 10 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r02_b13.json 2> gpurun_out/r02_b13.err || { tail -5 gpurun_out/r02_b13.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_b13.json").read().strip().splitlines()[-1])
print("cfg2:", d["value"], d["ms_per_step"], "grid:", d["grid"]["value"], d["grid"]["seconds"], d["grid"]["warmup_seconds"])
PY
