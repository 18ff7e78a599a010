# round 3, session 15: in-process 3-stream probe (full output) + the grid leg on one shared stream vs one stream per host thread
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s15; mkdir -p $O
timeout -k 10 200 python tools/probes/probe_concurrent5.py 0.1 8 > $O/conc5.txt 2>&1; tail -5 $O/conc5.txt
timeout -k 10 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/grid_device.txt 2>$O/grid_device.err && python - <<'PY'
import json; d=json.loads(open("gpurun_out/r03s15/grid_device.txt").read().strip().splitlines()[-1]); print("device", d["grid"]["value"], d["grid"]["seconds"], d["grid"]["scores_crc32"], d["grid"]["best_score"])
PY
SLNLP_STREAM_MODE=thread timeout -k 10 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/grid_thread.txt 2>$O/grid_thread.err && python - <<'PY'
import json; d=json.loads(open("gpurun_out/r03s15/grid_thread.txt").read().strip().splitlines()[-1]); print("thread", d["grid"]["value"], d["grid"]["seconds"], d["grid"]["scores_crc32"], d["grid"]["best_score"])
PY
SLNLP_STREAM_MODE=thread timeout -k 10 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/grid_thread2.txt 2>$O/grid_thread2.err && python - <<'PY'
import json; d=json.loads(open("gpurun_out/r03s15/grid_thread2.txt").read().strip().splitlines()[-1]); print("thread", d["grid"]["value"], d["grid"]["seconds"], d["grid"]["scores_crc32"], d["grid"]["best_score"])
PY
