# round 3, session 44: victim-2 (a) in a library compiled WITHOUT packed fp32 instructions (8192), (b) with vmcnt(0) + nops in front of the write-out arithmetic (8448)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s44; mkdir -p $O
for x in 8192 8448 8192 8448 8192; do
  echo "probe $x: $(SLNLP_PROBE_LIB=$x timeout -k 10 100 python -m pytest tests/test_net_gpu.py tests/test_streams_gpu.py -m gpu -q -k 'concurrent_fits_at_working or overlapping_streams' 2>&1 | grep -E '[0-9]+ (passed|failed)' | tail -1)" | tee -a $O/canary.txt
done
