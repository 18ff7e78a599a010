# round 3, session 21: skinny tile (16 x 16 straight from global memory): chain micro-bench, tests, bench
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s21; mkdir -p $O
timeout -k 10 100 python tools/bench_skinny_chain.py 50 512 2>&1 | grep -v amdgpu.ids | tee $O/chain.txt &&
timeout -k 10 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.txt &&
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-grid --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 | tee $O/bench.txt
