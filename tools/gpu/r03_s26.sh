# round 3, session 26: new geometry policy: lockstep K = 15 step + its timeline (what are the tiny copies), GPU kernel tests
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=/tmp/prof_raw; O=$GRAFT_REPO_ROOT/gpurun_out/r03s26; mkdir -p $R $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_lockstep_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt &&
timeout -k 10 300 python tools/bench_lockstep.py --workload cfg2 --ks 4,15 --steps 20 2>&1 | grep -v amdgpu | tee $O/lockstep.txt &&
rocprofv3 --kernel-trace --stats --output-format csv -d $R/ls -- python3 tools/bench_lockstep.py --workload cfg2 --ks 15 --steps 4 > $O/lockstep_k15.json 2> $R/ls.err || { tail -5 $R/ls.err; exit 1; }
python3 tools/trace_summary.py $R/ls --by-time > $O/lockstep_k15_trace.txt
python3 tools/trace_timeline.py $R/ls > $O/lockstep_k15_timeline.txt
head -12 $O/lockstep_k15_trace.txt; grep -c copyBuffer $O/lockstep_k15_timeline.txt; grep -n -B2 -A2 copyBuffer $O/lockstep_k15_timeline.txt | head -60
