# round 3, session 60: tile-order group height (GR tile rows per XCD block) at the configs[4] in_proj gradient group, 32-k ring
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s60; mkdir -p $O
for gr in 4 1 2 8 16 32; do
  echo "GR=$gr: $(SLNLP_PROBE_LIB=512 SLNLP_PLANE_GR=$gr timeout -k 10 60 python tools/bench_plane_one.py 16384 3072 1024 6 0 10 2>&1 | tail -1)" | tee -a $O/gr.txt
done
for gr in 4 8 16; do
  echo "fwd-like cfg5 FFN grads GR=$gr: $(SLNLP_PROBE_LIB=512 SLNLP_PLANE_GR=$gr timeout -k 10 60 python tools/bench_plane_one.py 16384 1024 512 8 0 10 2>&1 | tail -1)" | tee -a $O/gr.txt
done
