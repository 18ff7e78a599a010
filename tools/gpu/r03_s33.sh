# round 3, session 33: reverted tree: full GPU suite (with the RNN stream-mode canaries) + bench
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s33; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.txt &&
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-grid --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee $O/bench.txt
