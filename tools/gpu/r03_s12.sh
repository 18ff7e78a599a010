set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s12; mkdir -p $O
timeout -k 10 300 python tools/probes/probe_victim2.py 6 2>&1 | grep -v amdgpu.ids | tee $O/victim2.txt
