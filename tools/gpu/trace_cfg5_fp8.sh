# rocprofv3 kernel trace of the configs[4] step with fp8 forward products (and the split-bf16 step next to it)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=/tmp/prof_raw; O=$GRAFT_REPO_ROOT/gpurun_out/prof; mkdir -p $R $O
rocprofv3 --kernel-trace --stats --output-format csv -d $R/p8 -- python3 bench.py --workload cfg5 --precision 8 --steps 6 --warmup 4 --no-grid --no-cpu-baseline > $O/cfg5_p8.json 2> $R/p8.err || { tail -5 $R/p8.err; exit 1; }
python3 tools/trace_summary.py $R/p8 --by-time > $O/r02_bench_cfg5_p8_kernel_trace_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/p3 -- python3 bench.py --workload cfg5 --precision 3 --steps 6 --warmup 4 --no-grid --no-cpu-baseline > $O/cfg5_p3.json 2> $R/p3.err || { tail -5 $R/p3.err; exit 1; }
python3 tools/trace_summary.py $R/p3 --by-time > $O/r02_bench_cfg5_kernel_trace_summary.txt
head -8 $O/r02_bench_cfg5_p8_kernel_trace_summary.txt | cut -c1-150; tail -1 $O/r02_bench_cfg5_p8_kernel_trace_summary.txt
head -6 $O/r02_bench_cfg5_kernel_trace_summary.txt | cut -c1-150; tail -1 $O/r02_bench_cfg5_kernel_trace_summary.txt
