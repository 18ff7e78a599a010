# round 3, session 55: weight-gradient stream experiment (encoder): parity tests with it on, bench off / on
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s55; mkdir -p $O
SLNLP_TF_WGRAD_STREAM=1 timeout -k 10 400 python -m pytest tests/test_transformer_gpu.py tests/test_lockstep_gpu.py tests/test_streams_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_on.txt &&
for v in 0 1 0 1; do
  echo "wgrad stream $v: $(SLNLP_TF_WGRAD_STREAM=$v timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-grid --no-cpu-baseline --launch eager 2>&1 | tail -1 | cut -c60-160)" | tee -a $O/bench.txt
done
