# round 3, session 25: forward launches at the three geometries
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s25; mkdir -p $O
timeout -k 10 300 python tools/bench_plane_tiles.py fwd 2>&1 | grep -v amdgpu.ids | tee $O/fwd.txt
