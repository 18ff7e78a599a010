mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_net_gpu.py tests/test_lockstep_gpu.py tests/test_model_dropin.py -q -x > gpurun_out/r02_t12.log 2>&1; rc=$?
tail -4 gpurun_out/r02_t12.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E " gpurun_out/r02_t12.log | head -30 | cut -c1-300; exit $rc; fi
timeout -k 10 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r02_b12.json 2> gpurun_out/r02_b12.err || { tail -5 gpurun_out/r02_b12.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_b12.json").read().strip().splitlines()[-1])
print("cfg2:", d["value"], d["ms_per_step"], "grid:", d["grid"]["value"], d["grid"]["seconds"], d["grid"]["warmup_seconds"])
PY
