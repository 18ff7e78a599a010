# round 3, session 31: decoder LayerNorms inside their consumer GEMM's launch: tests + bench + lockstep
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s31; mkdir -p $O
timeout -k 10 200 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "layernorm_prologue or gemm_epilogues or test_gemm_layouts" 2>&1 | tail -5 | tee $O/pytest_k.txt &&
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.txt &&
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-grid --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee $O/bench.txt &&
timeout -k 10 200 python tools/bench_lockstep.py --workload cfg2 --ks 4,15 --steps 12 2>&1 | grep '^{"K"' | tee $O/lockstep.txt
