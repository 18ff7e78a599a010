# round 3, session 57: one-launch embedding backward for <= 64 tokens: tests + bench
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s57; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt &&
for i in 1 2; do timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-grid --no-cpu-baseline 2>&1 | tail -1 | cut -c60-150 | tee -a $O/bench.txt; done
