# round 3, session 59: the reproducer's table once more (repeatability; all aggressors but MFMA)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s59; mkdir -p $O
for m in 0 1 30 15 1 30 15; do timeout -k 10 60 tools/probes/packed_fp32_repro $m 5 2>&1 | tee -a $O/repro.txt; done; true
