# round 3, session 24: per-workgroup timelines of the plane GEMM (probe build 128)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s24; mkdir -p $O
SLNLP_PROBE_LIB=128 timeout -k 10 200 python tools/probes/probe_tile_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/timeline.txt
