# round 3, session 8: layernorm_bwd rebuilt under 256 registers -- is the multi-queue nondeterminism gone?
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s8; mkdir -p $O
echo "== victim probe"; timeout -k 10 300 python tools/probes/probe_victim.py 8 2>&1 | grep -v amdgpu.ids | tee $O/victim.txt
echo "== 3 processes together x3"
P="timeout -k 10 200 python tools/probes/probe_procs_together.py 12"
for i in 1 2 3; do $P 2>&1 | tail -1 | cut -c1-150; done
echo "== in-process, 3 streams, policy off"
timeout -k 10 200 python tools/probes/probe_concurrent5.py 0.1 8 2>&1 | grep -E "identical|fit [0-9]:" | head -10
echo "== pytest -m gpu"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -ne 0 ] && { echo "PYTEST RC $rc"; grep -E "^E|Error|FAILED" $O/pytest.log | head -20; }
echo "== bench cfg2 (no grid)"; timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-grid --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['launch'], d['roofline']['achieved'], d['roofline']['us_per_launch_hip_events'])"
