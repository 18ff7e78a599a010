# round 3, session 52: 32-k ring from 300 tiles: geometry tests, lockstep 4 / 8 / 15, bench solo + grid
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s52; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_lockstep_gpu.py -m gpu -x -q 2>&1 | tail -2 | tee $O/pytest.txt &&
timeout -k 10 300 python tools/bench_lockstep.py --workload cfg2 --ks 4,8,15 --steps 12 2>&1 | grep '^{"K"' | tee $O/lockstep.txt &&
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_cfg2.json 2> $O/b.err && python - <<'PY'
import json; d=json.loads(open("gpurun_out/r03s52/bench_cfg2.json").read().strip().splitlines()[-1]); print("cfg2", d["value"], d["ms_per_step"], "grid", d["grid"]["value"], d["grid"]["seconds"], d["grid"]["scores_crc32"])
PY
