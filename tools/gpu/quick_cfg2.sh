# cfg2: transformer / edge-shape / lockstep tests, the solo bench line and the lockstep sweep
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_transformer_gpu.py tests/test_edge_shapes_gpu.py tests/test_lockstep_gpu.py -q -x > gpurun_out/quick_tests.log 2>&1; rc=$?
tail -2 gpurun_out/quick_tests.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E " gpurun_out/quick_tests.log | head -20 | cut -c1-300; exit $rc; fi
timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-grid --no-cpu-baseline > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err || exit 1
python - <<'PY'
import json
d=json.loads(open("gpurun_out/quick_bench.json").read().strip().splitlines()[-1])
print("cfg2:", d["value"], d["ms_per_step"], d["parity"], "roofline", d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["us_per_launch_hip_events"])
PY
timeout -k 10 300 python tools/bench_lockstep.py --workload cfg2 --ks 1,4,8,16 2>/dev/null | tail -1 || exit 1
