cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=/tmp/prof_raw; O=$GRAFT_REPO_ROOT/gpurun_out/prof; mkdir -p $R $O
rocprofv3 --kernel-trace --stats --output-format csv -d $R/ls -- python3 tools/bench_lockstep.py --workload cfg2 --ks 4 --steps 10 > $O/lockstep_k4.json 2> $R/ls.err || { tail -5 $R/ls.err; exit 1; }
python3 tools/trace_summary.py $R/ls --by-time > $O/ls_now.txt
head -8 $O/ls_now.txt | cut -c1-150; tail -1 $O/ls_now.txt
