cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=/tmp/prof_raw; O=$GRAFT_REPO_ROOT/gpurun_out/prof; mkdir -p $R $O
rocprofv3 --kernel-trace --stats --output-format csv -d $R/trace -- python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-grid > $O/bench_trace.json 2> $R/bench_trace.err || { tail -5 $R/bench_trace.err; exit 1; }
python3 tools/trace_summary.py $R/trace --by-time > $O/trace_now.txt
head -40 $O/trace_now.txt | cut -c1-150; tail -1 $O/trace_now.txt
