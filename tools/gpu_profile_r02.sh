# round-2 profiles: kernel trace of the bench command, HBM traffic PMC passes, SQ counters of the roofline kernel, lockstep trace
set -o pipefail
mkdir -p gpurun_out/prof profiles
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof
BENCH="python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-grid"
rocprofv3 --kernel-trace --stats -d $O/trace -- $BENCH > $O/bench_trace.json 2> $O/bench_trace.err || { tail -5 $O/bench_trace.err; exit 1; }
python3 tools/trace_summary.py $O/trace --by-time > $O/r02_bench_cfg2_kernel_trace_summary.txt
KT=$(ls $O/trace/*/*kernel_trace.csv | head -1)
python3 tools/roofline_kernel_stats.py $KT 496 $O/r02_bench_cfg2_roofline_kernel.json
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-grid --launch eager > /dev/null 2> $O/fetch.err || { tail -5 $O/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE -d $O/write -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-grid --launch eager > /dev/null 2> $O/write.err || { tail -5 $O/write.err; exit 1; }
python3 tools/pmc_step_traffic.py $O/fetch $O/write $O/r02_pmc_cfg2_step_traffic.json > /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $O/sq -- python3 tools/bench_group.py > $O/bench_group.txt 2> $O/sq.err || { tail -5 $O/sq.err; }
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU -d $O/sq2 -- python3 tools/bench_group.py > /dev/null 2> $O/sq2.err || { tail -5 $O/sq2.err; }
python3 tools/mfma_util.py $O/sq 496 2400 512 512 3 $O/r02_mfma_util_sq.json > /dev/null || true
python3 tools/mfma_util.py $O/sq2 496 2400 512 512 3 $O/r02_mfma_util_sq2.json > /dev/null || true
rocprofv3 --kernel-trace --stats -d $O/ls -- python3 tools/bench_lockstep.py --workload cfg2 --ks 4 --steps 10 > $O/lockstep_k4.json 2> $O/ls.err || { tail -5 $O/ls.err; }
python3 tools/trace_summary.py $O/ls --by-time > $O/r02_lockstep_cfg2_k4_kernel_trace_summary.txt
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
ls -la $O | head -40
tail -3 $O/r02_bench_cfg2_kernel_trace_summary.txt; cat $O/r02_bench_cfg2_roofline_kernel.json
