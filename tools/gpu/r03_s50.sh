# round 3, session 50: library-free reproducer with a transposing-LDS-read aggressor
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s50; mkdir -p $O
for m in 0 16 17 18 31; do timeout -k 10 60 tools/probes/packed_fp32_repro $m 5 2>&1 | tee -a $O/repro.txt; done; true
