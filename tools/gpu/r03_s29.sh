# round 3, session 29: the round's bench lines (full default run + the other workloads)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s29; mkdir -p $O
timeout -k 10 600 python bench.py > $O/r03_bench_cfg2.json 2> $O/cfg2.err || { tail -5 $O/cfg2.err; exit 1; }
cut -c1-200 $O/r03_bench_cfg2.json
for w in cfg1 cfg3 cfg3gru cfg5 e1024; do
  timeout -k 10 300 python bench.py --workload $w --no-grid > $O/r03_bench_$w.json 2> $O/$w.err || { tail -5 $O/$w.err; exit 1; }
  cut -c1-160 $O/r03_bench_$w.json
done
timeout -k 10 300 python bench.py --workload cfg5 --precision 8 --no-grid > $O/r03_bench_cfg5_p8.json 2> $O/p8.err || { tail -5 $O/p8.err; exit 1; }
cut -c1-160 $O/r03_bench_cfg5_p8.json
