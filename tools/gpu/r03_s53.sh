# round 3, session 53: the synthetic packed-fp32 victim beside library kernels
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s53; mkdir -p $O
timeout -k 10 200 python tools/probes/probe_pk_victim.py 4 2>&1 | grep -v amdgpu.ids | tee $O/pkv.txt
