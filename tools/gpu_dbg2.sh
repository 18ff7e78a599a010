export TMPDIR=/tmp
O=/tmp/prof; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-grid > $O/bench_trace.json 2> $O/bench_trace.err
echo rc=$?; tail -20 $O/bench_trace.err; ls -R $O/trace | head -20
