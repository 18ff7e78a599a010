# round 3, session 36: victim-2 with its workgroups alone on their CU (LDS padded to ~145 KB): does co-residency trigger it?
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s36; mkdir -p $O
for x in 2304 1536 2304 2304; do
  echo "probe $x: $(SLNLP_PROBE_LIB=$x timeout -k 10 100 python -m pytest tests/test_streams_gpu.py -m gpu -q -k 'overlapping_streams' 2>&1 | grep -E '[0-9]+ (passed|failed)' | tail -1)" | tee -a $O/canary.txt
done
