mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_transformer_gpu.py tests/test_net_gpu.py -q -x -k "fp8 or precision8 or adam" > gpurun_out/r02_t5a.log 2>&1; rc=$?
tail -15 gpurun_out/r02_t5a.log; grep -E "fp8 vs|log-prob rel" gpurun_out/r02_t5a.log
if [ $rc -ge 124 ]; then exit $rc; fi
for p in 3 8; do timeout -k 10 300 python bench.py --workload cfg5 --precision $p --steps 20 --warmup 6 --no-grid --no-cpu-baseline > gpurun_out/r02_cfg5_p$p.json 2> gpurun_out/r02_cfg5_p$p.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r02_cfg5_p$p.json").read().strip().splitlines()[-1])
print("cfg5 precision $p:", d["value"], d["ms_per_step"], d["roofline_step"]["achieved"], d["parity"], d.get("roofline_fp8"), d["roofline"]["achieved"], d["roofline"]["us_per_launch_hip_events"])
PY
done
for k in 3 4; do timeout -k 10 400 python bench.py --steps 50 --warmup 20 --no-cpu-baseline --fits-per-gpu $k > gpurun_out/r02_b6_$k.json 2> gpurun_out/r02_b6_$k.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r02_b6_$k.json").read().strip().splitlines()[-1])
print("fits_per_gpu $k:", d["grid"]["value"], d["grid"]["seconds"])
PY
done
