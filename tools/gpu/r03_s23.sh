# round 3, session 23: kernel trace of a lockstep-15 cfg2 step (what the merged launches cost)
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=/tmp/prof_raw; O=$GRAFT_REPO_ROOT/gpurun_out/r03s23; mkdir -p $R $O
rocprofv3 --kernel-trace --stats --output-format csv -d $R/ls -- python3 tools/bench_lockstep.py --workload cfg2 --ks 15 --steps 6 > $O/lockstep_k15.json 2> $R/ls.err || { tail -5 $R/ls.err; exit 1; }
python3 tools/trace_summary.py $R/ls --by-time > $O/lockstep_k15_trace.txt
head -45 $O/lockstep_k15_trace.txt; tail -1 $O/lockstep_k15_trace.txt; cat $O/lockstep_k15.json | tail -1
