# round 3, session 7: victim isolation for the multi-queue nondeterminism
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s7; mkdir -p $O
timeout -k 10 300 python tools/probes/probe_victim.py 8 2>&1 | grep -v amdgpu.ids | tee $O/victim.txt
