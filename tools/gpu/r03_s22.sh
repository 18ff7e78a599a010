# round 3, session 22: skinny tile: GPU suite, bench, lockstep K = 1, 4, 15 (merged launches must not lose)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s22; mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.txt &&
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-grid --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee $O/bench.txt &&
timeout -k 10 300 python tools/bench_lockstep.py --workload cfg2 --ks 1,4,15 --steps 20 2>&1 | grep -v amdgpu | tee $O/lockstep.txt
