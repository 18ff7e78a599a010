# round 3, session 51: 4- and 8-fit lockstep steps with every plane launch forced to one geometry (where is the 64-k / 32-k ring crossover)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s51; mkdir -p $O
for t in 0 128 12832; do
  echo "SLNLP_PLANE_TILE=$t" | tee -a $O/lockstep.txt
  SLNLP_PLANE_TILE=$t timeout -k 10 200 python tools/bench_lockstep.py --workload cfg2 --ks 2,4,8 --steps 12 2>&1 | grep '^{"K"' | tee -a $O/lockstep.txt || exit 1
done
