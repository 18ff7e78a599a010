# round 3, session 35: victim-2 control code in a library where NO kernel uses AGPRs (-mllvm -amdgpu-mfma-vgpr-form=1)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s35; mkdir -p $O
for x in 2048 1536 2048 1536 2048; do
  echo "probe $x: $(SLNLP_PROBE_LIB=$x timeout -k 10 100 python -m pytest tests/test_net_gpu.py tests/test_streams_gpu.py -m gpu -q -k 'concurrent_fits_at_working or overlapping_streams' 2>&1 | grep -E '[0-9]+ (passed|failed)' | tail -1)" | tee -a $O/canary.txt
done
