# rocprofv3 profiles of a round: kernel traces, HBM-traffic PMC passes, SQ counters.  One script for every round:
#     bash tools/gpu/profile.sh <round tag, e.g. r04> <out dir> [section ...]
# sections (default: all): trace traffic traffic_more sq big lockstep cfg5 cfg5p8 rnn
#   trace     kernel trace of the default bench command -> <tag>_bench_cfg2_kernel_trace_summary.txt, <tag>_bench_cfg2_roofline_kernel.json
#   traffic   FETCH_SIZE / WRITE_SIZE passes of the eager cfg2 step -> <tag>_pmc_cfg2_step_traffic.json
#   traffic_more  the same for the other workloads' steps (TRAFFIC_WORKLOADS, default cfg5 cfg3 cfg1 e1024 cfg3gru) -> <tag>_pmc_<workload>_step_traffic.json
#   sq        SQ counters of the cfg2 dgrad + wgrad group -> <tag>_mfma_util_sq.json, <tag>_pmc_plane_gemm_sq*_raw.txt
#   big       configs[4] in_proj gradient group: SQ + L2 + HBM counters -> <tag>_pmc_plane_gemm_cfg5_raw.txt, <tag>_pmc_large_launch_traffic.json
#   lockstep  kernel trace of a 4-fit and a 15-fit lockstep step (cfg2)
#   cfg5 / cfg5p8   kernel trace of the configs[4] step (precision 3 / 8)
#   rnn       kernel traces of the cfg3 LSTM step and its 8-fit lockstep step
# Raw traces stay on the box (tens of MB); the summaries land in <out dir>: copy the keepers into profiles/.
# PMC passes never share a command with --sys-trace / runtime tracing, and the program itself follows "--" (no env / bash -c hop).
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; O=$GRAFT_REPO_ROOT/$2; shift; shift
SECTIONS=${@:-trace traffic traffic_more sq big lockstep cfg5 cfg5p8 rnn}
R=/tmp/prof_raw
mkdir -p $R $O
has() { case " $SECTIONS " in *" $1 "*) return 0 ;; *) return 1 ;; esac; }
trace() {   # trace <name> <summary file> <program...>
  local name=$1 out=$2; shift; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$name -- "$@" > $O/$name.stdout 2> $R/$name.err || { tail -5 $R/$name.err; return 1; }
  python3 tools/trace_summary.py $R/$name --by-time > $O/$out
  echo "[prof] $out"; head -14 $O/$out | cut -c1-160; tail -1 $O/$out
}
if has trace; then
  trace bench ${TAG}_bench_cfg2_kernel_trace_summary.txt python3 bench.py --steps 20 --warmup 20 --no-cpu-baseline --no-grid || exit 1
  KT=$(ls $R/bench/*/*kernel_trace.csv | head -1)
  python3 tools/roofline_kernel_stats.py $KT 496 $O/${TAG}_bench_cfg2_roofline_kernel.json
  tail -1 $O/bench.stdout > $O/${TAG}_bench_cfg2_traced.json
fi
if has traffic; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/fetch -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-grid --launch eager > /dev/null 2> $R/fetch.err || { tail -5 $R/fetch.err; exit 1; }
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/write -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-grid --launch eager > /dev/null 2> $R/write.err || { tail -5 $R/write.err; exit 1; }
  python3 tools/pmc_step_traffic.py $R/fetch $R/write $O/${TAG}_pmc_cfg2_step_traffic.json > /dev/null
  echo "[prof] traffic done"
fi
if has sq; then
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $R/sq -- python3 tools/bench_group.py > $O/bench_group.txt 2> $R/sq.err || tail -5 $R/sq.err
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $R/sq2 -- python3 tools/bench_group.py > /dev/null 2> $R/sq2.err || tail -5 $R/sq2.err
  python3 tools/mfma_util.py $R/sq 496 2400 512 512 3 $O/${TAG}_mfma_util_sq.json > /dev/null || true
  python3 tools/pmc_summary.py $R/sq gemm_planes > $O/${TAG}_pmc_plane_gemm_sq_raw.txt 2>/dev/null || true
  python3 tools/pmc_summary.py $R/sq2 gemm_planes > $O/${TAG}_pmc_plane_gemm_sq2_raw.txt 2>/dev/null || true
  echo "[prof] sq cfg2 done"
fi
if has big; then
  : > $O/${TAG}_pmc_plane_gemm_cfg5_raw.txt
  for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA"; do
    d=$R/big_$(echo $c | cut -c1-8 | tr ' ' '_')
    rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/bench_plane_one.py 16384 3072 1024 0 0 3 > $O/bench_plane_big.txt 2> $d.err || tail -3 $d.err
    python3 tools/pmc_summary.py $d gemm_planes >> $O/${TAG}_pmc_plane_gemm_cfg5_raw.txt 2>/dev/null || true
    python3 tools/pmc_summary.py $d plane_splitk_reduce >> $O/${TAG}_pmc_plane_gemm_cfg5_raw.txt 2>/dev/null || true   # (the weight gradient's K-slices meet there)
  done
  python3 tools/pmc_large_launch.py $O/${TAG}_pmc_plane_gemm_cfg5_raw.txt $O/${TAG}_pmc_large_launch_traffic.json || true
  echo "[prof] big launch done"
fi
if has lockstep; then
  trace ls4 ${TAG}_lockstep_cfg2_k4_kernel_trace_summary.txt python3 tools/bench_lockstep.py --workload cfg2 --ks 4 --steps 10
  trace ls15 ${TAG}_lockstep_cfg2_k15_kernel_trace_summary.txt python3 tools/bench_lockstep.py --workload cfg2 --ks 15 --steps 6
fi
if has traffic_more; then   # HBM traffic of the configs[4] and configs[2] steps too (their bench records carry roofline.traffic)
  for w in ${TRAFFIC_WORKLOADS:-cfg5 cfg3 cfg1 e1024 cfg3gru}; do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/fetch_$w -- python3 bench.py --workload $w --steps 3 --warmup 3 --no-cpu-baseline --no-grid --launch eager > /dev/null 2> $R/fetch_$w.err || { tail -5 $R/fetch_$w.err; exit 1; }
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/write_$w -- python3 bench.py --workload $w --steps 3 --warmup 3 --no-cpu-baseline --no-grid --launch eager > /dev/null 2> $R/write_$w.err || { tail -5 $R/write_$w.err; exit 1; }
    python3 tools/pmc_step_traffic.py $R/fetch_$w $R/write_$w $O/${TAG}_pmc_${w}_step_traffic.json > /dev/null
  done
  echo "[prof] traffic of the other workloads done"
fi
if has cfg5; then trace c5 ${TAG}_bench_cfg5_kernel_trace_summary.txt python3 bench.py --workload cfg5 --steps 8 --warmup 4 --no-cpu-baseline --no-grid; fi
if has cfg5p8; then trace c5p8 ${TAG}_bench_cfg5_p8_kernel_trace_summary.txt python3 bench.py --workload cfg5 --precision 8 --steps 8 --warmup 4 --no-cpu-baseline --no-grid; fi
if has rnn; then
  trace rnn ${TAG}_bench_cfg3_kernel_trace_summary.txt python3 bench.py --workload cfg3 --steps 10 --warmup 6 --no-cpu-baseline --no-grid
  trace rnnls ${TAG}_lockstep_cfg3_k8_kernel_trace_summary.txt python3 tools/bench_lockstep.py --workload cfg3 --ks 8 --steps 6
fi
ls -la $O | head -40
