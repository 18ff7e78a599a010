# round 3, session 30: 2-rank rehearsal of bench.py on the 1-GPU box (ranks share the card, gloo) + configs[3] full grid refresh
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s30; mkdir -p $O
timeout -k 10 500 python bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline > $O/r03_bench_cfg2_gpus2_rehearsal.json 2> $O/g2.err || { tail -8 $O/g2.err; exit 1; }
cut -c1-250 $O/r03_bench_cfg2_gpus2_rehearsal.json
timeout -k 10 500 python tools/full_grid.py --lockstep 15 > $O/r03_full_grid_324x5.json 2> $O/fg.err || { tail -8 $O/fg.err; exit 1; }
cut -c1-400 $O/r03_full_grid_324x5.json
