mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r02_full.log 2>&1; rc=$?
tail -3 gpurun_out/r02_full.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E |^FAILED|^ERROR" gpurun_out/r02_full.log | head -20 | cut -c1-300; exit $rc; fi
for f in 3 4 6; do
timeout -k 10 400 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --fits-per-gpu $f > gpurun_out/r02_b14_$f.json 2> gpurun_out/r02_b14.err || { tail -5 gpurun_out/r02_b14.err; exit 1; }
python - $f <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/r02_b14_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("fits_per_gpu", sys.argv[1], "grid:", d["grid"]["value"], d["grid"]["seconds"])
PY
done
timeout -k 10 300 python tools/probe_determinism.py 2>&1 | grep -E " vs " | cut -c1-100
