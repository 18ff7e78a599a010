# round 3, session 17: kernel trace of the cfg2 bench step (where does the step's time go after the LN split)
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=/tmp/prof_raw; O=$GRAFT_REPO_ROOT/gpurun_out/r03s17; mkdir -p $R $O
rocprofv3 --kernel-trace --stats --output-format csv -d $R/trace -- python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-grid --launch eager > $O/bench_trace.json 2> $R/bench_trace.err || { tail -5 $R/bench_trace.err; exit 1; }
python3 tools/trace_summary.py $R/trace --by-time > $O/trace_summary.txt
KT=$(ls $R/trace/*/*kernel_trace.csv | head -1)
python3 tools/trace_timeline.py $KT > $O/timeline.txt 2>&1 || true
head -60 $O/trace_summary.txt
