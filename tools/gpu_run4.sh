mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_edge_shapes_gpu.py tests/test_errors_gpu.py tests/test_kernels_gpu.py -q -x -k "edge or long or error or bad or adam or rnn_bad" > gpurun_out/r02_t4a.log 2>&1; rc=$?
tail -15 gpurun_out/r02_t4a.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r02_t4.log 2>&1; rc=$?
tail -6 gpurun_out/r02_t4.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 500 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --fits-per-gpu 2 > gpurun_out/r02_b5.json 2> gpurun_out/r02_b5.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02_b5.json").read().strip().splitlines()[-1])
print("fits_per_gpu 2:", d["value"], d["grid"]["value"], d["grid"]["seconds"])
PY
