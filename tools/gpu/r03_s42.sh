# round 3, session 42: victim-2 library under the 3-stream workspace-diff probe: which buffers differ first
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s42; mkdir -p $O
SLNLP_PROBE_LIB=4096 timeout -k 10 200 python tools/probes/probe_concurrent5.py 0.1 10 2>&1 | grep -v "ALIVE\|amdgpu.ids" > $O/conc5.txt; head -60 $O/conc5.txt
