# dgrad and wgrad of configs[4]'s in_proj gradient group ALONE (tools/bench_plane_one.py, ONLY=...) at several tiles / splits, and together
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/$1
O=gpurun_out/$1/only_sweep.txt; : > $O
for tile in 12832 256; do
  ONLY=dgrad python3 tools/bench_plane_one.py 16384 3072 1024 6 $tile 20 2>/dev/null | grep alone >> $O
  for split in 3 4 5 6 8; do ONLY=wgrad python3 tools/bench_plane_one.py 16384 3072 1024 $split $tile 20 2>/dev/null | grep alone >> $O; done
  for split in 4 6; do python3 tools/bench_plane_one.py 16384 3072 1024 $split $tile 20 2>/dev/null | grep tokens >> $O; done
done
# FFN gradient group 16384 x 1024 x 512 and 15 fits' tokens 36000 x 512 x 512
for shape in "16384 1024 512" "36000 512 512"; do
  for tile in 12832 256; do
    ONLY=dgrad python3 tools/bench_plane_one.py $shape 8 $tile 20 2>/dev/null | grep alone >> $O
    for split in 4 8 12 16; do ONLY=wgrad python3 tools/bench_plane_one.py $shape $split $tile 20 2>/dev/null | grep alone >> $O; done
    python3 tools/bench_plane_one.py $shape 8 $tile 20 2>/dev/null | grep tokens >> $O
  done
done
cat $O
