# round 3, session 46: product library built without packed fp32 instructions: GPU suite, probes, bench (solo + grid), cfg5, lockstep
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s46; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt &&
timeout -k 10 200 python tools/probes/probe_victim.py 3 2>&1 | grep -v amdgpu.ids | grep "beside" | tee $O/victim.txt &&
timeout -k 10 200 python tools/probes/probe_procs_together.py 12 2>&1 | tail -1 | cut -c1-150 | tee $O/procs.txt &&
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_cfg2.json 2> $O/b.err && python - <<'PY'
import json; d=json.loads(open("gpurun_out/r03s46/bench_cfg2.json").read().strip().splitlines()[-1]); print("cfg2", d["value"], d["ms_per_step"], "grid", d["grid"]["value"], d["grid"]["scores_crc32"], [r["value"] for r in d["concurrent_fits"]["runs"]])
PY
timeout -k 10 300 python bench.py --workload cfg5 --no-grid --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160 | tee $O/bench_cfg5.txt
timeout -k 10 300 python bench.py --workload cfg3 --no-grid --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160 | tee $O/bench_cfg3.txt
