# round 3, session 49: which neighbour kernel triggers the packed fp32 victim (old library 900) + rehearsal / full grid with the final build
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s49; mkdir -p $O
SLNLP_PROBE_LIB=900 timeout -k 10 300 python tools/probes/probe_aggressor.py 3 2>&1 | grep -v amdgpu.ids | tee $O/aggressor.txt
