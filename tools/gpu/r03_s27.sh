# round 3, session 27: lockstep K = 15 with every plane launch forced to one geometry
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s27; mkdir -p $O
for t in 0 64 128 12832; do
  echo "SLNLP_PLANE_TILE=$t" | tee -a $O/lockstep.txt
  SLNLP_PLANE_TILE=$t timeout -k 10 200 python tools/bench_lockstep.py --workload cfg2 --ks 15 --steps 12 2>&1 | grep '^{"K"' | tee -a $O/lockstep.txt || exit 1
done
