cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=/tmp/prof_raw; O=$GRAFT_REPO_ROOT/gpurun_out/prof; mkdir -p $R $O
rocprofv3 --kernel-trace --stats --output-format csv -d $R/rnnls -- python3 tools/bench_lockstep.py --workload cfg3 --ks 8 --steps 6 > $O/rnn_ls.json 2> $R/rnnls.err || { tail -5 $R/rnnls.err; exit 1; }
python3 tools/trace_summary.py $R/rnnls --by-time > $O/r02_lockstep_cfg3_k8_kernel_trace_summary.txt
head -24 $O/r02_lockstep_cfg3_k8_kernel_trace_summary.txt | cut -c1-150; tail -1 $O/r02_lockstep_cfg3_k8_kernel_trace_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/rnn1 -- python3 tools/bench_lockstep.py --workload cfg3 --ks 1 --steps 6 > $O/rnn_ls1.json 2> $R/rnn1.err || { tail -5 $R/rnn1.err; exit 1; }
python3 tools/trace_summary.py $R/rnn1 --by-time > $O/r02_lockstep_cfg3_k1_kernel_trace_summary.txt
head -16 $O/r02_lockstep_cfg3_k1_kernel_trace_summary.txt | cut -c1-150; tail -1 $O/r02_lockstep_cfg3_k1_kernel_trace_summary.txt
