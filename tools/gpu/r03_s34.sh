# round 3, session 34: victim-2 experiments (LN-prologue GEMM variants), canary tests per probe library
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s34; mkdir -p $O
for x in 1536 256 512 768 1024 1280 1792; do
  for i in 1 2; do
    echo "probe $x run $i: $(SLNLP_PROBE_LIB=$x timeout -k 10 100 python -m pytest tests/test_net_gpu.py tests/test_streams_gpu.py -m gpu -q -k 'concurrent_fits_at_working or overlapping_streams' 2>&1 | grep -E '[0-9]+ (passed|failed)' | tail -1)" | tee -a $O/canary.txt
  done
done
