# round-3 profiles: kernel trace of the bench command, HBM traffic PMC passes, SQ counters of the roofline kernel (cfg2 group) and of
# the large-launch kernel (configs[4] in_proj group), lockstep trace
# usage (GPU box): bash tools/gpu/profile_r03.sh      -> summaries under gpurun_out/prof/ (copy the keepers into profiles/)
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=/tmp/prof_raw            # raw traces stay on the box (tens of MB); only the summaries travel back
O=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $R $O
BENCH="python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-grid"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/trace -- $BENCH > $O/bench_trace.json 2> $R/bench_trace.err || { tail -5 $R/bench_trace.err; exit 1; }
python3 tools/trace_summary.py $R/trace --by-time > $O/r03_bench_cfg2_kernel_trace_summary.txt
KT=$(ls $R/trace/*/*kernel_trace.csv | head -1)
python3 tools/roofline_kernel_stats.py $KT 496 $O/r03_bench_cfg2_roofline_kernel.json
echo "[prof] trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/fetch -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-grid --launch eager > /dev/null 2> $R/fetch.err || { tail -5 $R/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/write -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-grid --launch eager > /dev/null 2> $R/write.err || { tail -5 $R/write.err; exit 1; }
python3 tools/pmc_step_traffic.py $R/fetch $R/write $O/r03_pmc_cfg2_step_traffic.json > /dev/null
echo "[prof] traffic done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $R/sq -- python3 tools/bench_group.py > $O/bench_group.txt 2> $R/sq.err || { tail -5 $R/sq.err; }
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $R/sq2 -- python3 tools/bench_group.py > /dev/null 2> $R/sq2.err || { tail -5 $R/sq2.err; }
python3 tools/mfma_util.py $R/sq 496 2400 512 512 3 $O/r03_mfma_util_sq.json > /dev/null || true
python3 tools/pmc_summary.py $R/sq gemm_planes > $O/r03_pmc_plane_gemm_sq_raw.txt 2>/dev/null || true
python3 tools/pmc_summary.py $R/sq2 gemm_planes > $O/r03_pmc_plane_gemm_sq2_raw.txt 2>/dev/null || true
echo "[prof] sq cfg2 done"
# the large launch (configs[4] in_proj gradient group, 128 x 128 tile): SQ + L2 + HBM counters
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA"; do
  d=$R/big_$(echo $c | cut -c1-8 | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/bench_plane_one.py 16384 3072 1024 6 0 3 > $O/bench_plane_big.txt 2> $d.err || tail -3 $d.err
  python3 tools/pmc_summary.py $d gemm_planes >> $O/r03_pmc_plane_gemm_cfg5_raw.txt 2>/dev/null || true
done
echo "[prof] big launch done"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/ls -- python3 tools/bench_lockstep.py --workload cfg2 --ks 4 --steps 10 > $O/lockstep_k4.json 2> $R/ls.err || { tail -5 $R/ls.err; }
python3 tools/trace_summary.py $R/ls --by-time > $O/r03_lockstep_cfg2_k4_kernel_trace_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/c5 -- python3 bench.py --workload cfg5 --steps 8 --warmup 4 --no-cpu-baseline --no-grid > $O/bench_cfg5_trace.json 2> $R/c5.err || { tail -5 $R/c5.err; }
python3 tools/trace_summary.py $R/c5 --by-time > $O/r03_bench_cfg5_kernel_trace_summary.txt
ls -la $O | head -40
head -14 $O/r03_bench_cfg2_kernel_trace_summary.txt; tail -2 $O/r03_bench_cfg2_kernel_trace_summary.txt; cat $O/r03_bench_cfg2_roofline_kernel.json
