# round 3, session 58: final tree: bench line + kernel trace summary + step traffic
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=/tmp/prof_raw; O=$GRAFT_REPO_ROOT/gpurun_out/r03s58; mkdir -p $R $O
timeout -k 10 600 python bench.py > $O/r03_bench_cfg2.json 2> $O/cfg2.err || { tail -5 $O/cfg2.err; exit 1; }
cut -c1-160 $O/r03_bench_cfg2.json
rocprofv3 --kernel-trace --stats --output-format csv -d $R/trace -- python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-grid > $O/bench_trace.json 2> $R/bench_trace.err || { tail -5 $R/bench_trace.err; exit 1; }
python3 tools/trace_summary.py $R/trace --by-time > $O/r03_bench_cfg2_kernel_trace_summary.txt
KT=$(ls $R/trace/*/*kernel_trace.csv | head -1)
python3 tools/roofline_kernel_stats.py $KT 496 $O/r03_bench_cfg2_roofline_kernel.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/fetch -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-grid --launch eager > /dev/null 2> $R/fetch.err || { tail -5 $R/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/write -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-grid --launch eager > /dev/null 2> $R/write.err || { tail -5 $R/write.err; exit 1; }
python3 tools/pmc_step_traffic.py $R/fetch $R/write $O/r03_pmc_cfg2_step_traffic.json | head -8
tail -1 $O/r03_bench_cfg2_kernel_trace_summary.txt
