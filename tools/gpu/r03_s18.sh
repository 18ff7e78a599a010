# round 3, session 18: ring depth of the 64 x 64 plane tile at cfg2-size launches
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s18; mkdir -p $O
timeout -k 10 200 python tools/bench_plane_tiles.py ring 2>&1 | grep -v amdgpu.ids | tee $O/ring.txt
