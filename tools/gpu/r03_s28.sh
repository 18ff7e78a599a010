# round 3, session 28: LN backward with single-wave workgroups: tests, determinism probes again, bench
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s28; mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt &&
timeout -k 10 200 python tools/probes/probe_victim.py 4 2>&1 | grep -v amdgpu.ids | tee $O/victim.txt &&
timeout -k 10 200 python tools/probes/probe_procs_together.py 12 2>&1 | tail -1 | cut -c1-150 | tee $O/procs.txt &&
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-grid --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee $O/bench.txt &&
timeout -k 10 200 python tools/bench_lockstep.py --workload cfg2 --ks 15 --steps 12 2>&1 | grep '^{"K"' | tee $O/lockstep.txt
