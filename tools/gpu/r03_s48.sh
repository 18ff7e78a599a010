# round 3, session 48: library-free reproducer attempt for the packed fp32 finding
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s48; mkdir -p $O
for m in 0 1 2 4 8 15; do timeout -k 10 60 tools/probes/packed_fp32_repro $m 4 2>&1 | tee -a $O/repro.txt; done; true
