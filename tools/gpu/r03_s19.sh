# round 3, session 19: up-front fetch in the short-K two-group GEMM: tests + bench
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s19; mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt &&
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-grid --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 | tee $O/bench.txt
