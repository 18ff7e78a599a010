# round 3, session 32: LN-fused decoder: bench first, then the concurrency canary three times with its assertion detail
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s32; mkdir -p $O
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-grid --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330 | tee $O/bench.txt
for i in 1 2 3; do
  timeout -k 10 200 python -m pytest tests/test_net_gpu.py tests/test_streams_gpu.py -m gpu -q -k "concurrent_fits_at_working or overlapping or grid_scores" 2>&1 | grep -E "passed|failed|Error|assert|array" | head -12 | tee -a $O/canary.txt
done
