# One parametrised session script for the MI355X box (replaces the per-session r03_s*.sh files):
#     gpurun -- 'bash tools/gpu/steps.sh <out-tag> <step> [<step> ...]'
# Every step writes under gpurun_out/<out-tag>/ and prints a short summary; a failing step ends the session (no GPU step after a
# timeout or a crash).  Steps (add new ones here instead of new files):
#   clock        held shader clock + per-workgroup timelines of the plane GEMM (probe build, tools/probes/probe_tile_timeline.py)
#   tiles_big    plane-GEMM geometries on the large gradient groups (tools/bench_plane_tiles.py big)
#   tiles_fwd    ... on the forward launches
#   tiles_quick  ... on the cfg2 launches + bit-identity of the geometries
#   planes       the plane-GEMM tests of tests/test_kernels_gpu.py
#   kernels      tests/test_kernels_gpu.py
#   quick        transformer / edge / lockstep tests
#   rnn          tests/test_rnn_gpu.py
#   streams      tests/test_streams_gpu.py (fits on overlapping queues)
#   net          estimator / lockstep / configs tests
#   suite        the whole -m gpu suite
#   bench        default bench line (cfg2) without grid / cpu baseline
#   bench_full   the default bench.py run, exactly as the driver runs it
#   records      every other workload's bench line with its cpu_baseline (cfg1, cfg3, cfg3gru, e1024, cfg5, cfg5 precision 8)
#   bench_cfg5   cfg5 (configs[4] shape) line, precision 3
#   bench_cfg5p8 ... precision 8
#   bench_rnn    cfg3 LSTM + GRU lines
#   lockstep     lockstep sweep cfg2 (K = 1, 4, 15)
#   lockstep_rnn lockstep sweep cfg3 / cfg3gru
#   grid         the bench's grid leg only (folds/hr + CRC)
#   rehearsal    2 ranks on the one GPU (gloo): bench.py --gpus 2
#   smoke        __graft_entry__.smoke()
#   fullgrid     configs[3]: all 324 candidates x cv 5 on one GPU (tools/full_grid.py)
#   passerr      golden-trajectory errors under (wgrad, dgrad) passes (3,3) / (2,3) / (2,2) (tools/backward_pass_errors.py)
#   gridcal      grid leg alone at (lockstep x threads) pairs with per-unit logs (tools/bench_grid.py; GRIDCAL="15x1 15x4 5x4")
#   profile      rocprofv3 kernel traces + PMC passes -> gpurun_out/<tag>/prof (tools/gpu/profile.sh <round>)
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
line() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print({k: d.get(k) for k in ("value", "ms_per_step", "parity", "launches_per_step")}, "roofline", {k: r.get(k) for k in ("achieved", "frac", "us_per_launch", "population")},
      "grid", (d.get("grid") or {}).get("value"), (d.get("grid") or {}).get("scores_crc32"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
}
for spec in "$@"; do
  echo "=== [$spec]"
  step=${spec%%@*}                       # "<step>@<w>,<d>": split-bf16 passes of the weight / data gradient products for this step
  if [ "$spec" != "$step" ]; then
    export PLANE_PASSES=${spec#*@}; export SLNLP_WGRAD_PASSES=${PLANE_PASSES%,*}; export SLNLP_DGRAD_PASSES=${PLANE_PASSES#*,}
  else
    unset PLANE_PASSES SLNLP_WGRAD_PASSES SLNLP_DGRAD_PASSES
  fi
  O=gpurun_out/$TAG/$(echo $spec | tr '@,' '__'); mkdir -p $O
  case $step in
    clock)       SLNLP_PROBE_LIB=128 timeout -k 10 400 python tools/probes/probe_tile_timeline.py > $O/clock.txt 2> $O/clock.err || { tail -5 $O/clock.err; exit 1; }; cat $O/clock.txt ;;
    tiles_big)   timeout -k 10 400 python tools/bench_plane_tiles.py big > $O/tiles_big.txt 2> $O/tiles_big.err || { tail -5 $O/tiles_big.txt $O/tiles_big.err; exit 1; }; cat $O/tiles_big.txt ;;
    tiles_fwd)   timeout -k 10 400 python tools/bench_plane_tiles.py fwd > $O/tiles_fwd.txt 2> $O/tiles_fwd.err || { tail -5 $O/tiles_fwd.txt $O/tiles_fwd.err; exit 1; }; cat $O/tiles_fwd.txt ;;
    tiles_quick) timeout -k 10 400 python tools/bench_plane_tiles.py quick > $O/tiles_quick.txt 2> $O/tiles_quick.err || { tail -5 $O/tiles_quick.txt $O/tiles_quick.err; exit 1; }; cat $O/tiles_quick.txt ;;
    kernels|quick|rnn|suite|planes|streams|net)
      KEXPR=""
      case $step in
        kernels) T="tests/test_kernels_gpu.py" ;;
        planes)  T="tests/test_kernels_gpu.py"; KEXPR="plane_tile_geometries or two_pass or gemm_planes" ;;
        quick)   T="tests/test_transformer_gpu.py tests/test_edge_shapes_gpu.py tests/test_lockstep_gpu.py" ;;
        rnn)     T="tests/test_rnn_gpu.py" ;;
        streams) T="tests/test_streams_gpu.py" ;;
        net)     T="tests/test_net_gpu.py tests/test_lockstep_gpu.py tests/test_configs_gpu.py" ;;
        suite)   T="tests -m gpu" ;;
      esac
      timeout -k 10 1100 python -m pytest $T -k "$KEXPR" -q -x > $O/$step.log 2>&1; rc=$?
      tail -3 $O/$step.log | cut -c1-300
      if [ $rc -ne 0 ]; then grep -E "^E |^FAILED|^ERROR" $O/$step.log | head -20 | cut -c1-300; exit $rc; fi ;;
    bench)       timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-grid --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }; line $O/bench.json ;;
    bench_full)  timeout -k 10 600 python bench.py > $O/bench_full.json 2> $O/bench_full.err || { tail -5 $O/bench_full.err; exit 1; }; line $O/bench_full.json ;;
    bench_cfg5)  timeout -k 10 400 python bench.py --workload cfg5 --steps 20 --warmup 5 --no-grid --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err || { tail -5 $O/bench_cfg5.err; exit 1; }; line $O/bench_cfg5.json ;;
    bench_cfg5p8) timeout -k 10 400 python bench.py --workload cfg5 --precision 8 --steps 20 --warmup 5 --no-grid --no-cpu-baseline > $O/bench_cfg5_p8.json 2> $O/bench_cfg5_p8.err || { tail -5 $O/bench_cfg5_p8.err; exit 1; }; line $O/bench_cfg5_p8.json ;;
    bench_rnn)   for w in cfg3 cfg3gru; do timeout -k 10 400 python bench.py --workload $w --steps 40 --warmup 10 --no-grid --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err || { tail -5 $O/bench_$w.err; exit 1; }; line $O/bench_$w.json; done ;;
    records)     # every workload's bench line WITH its cpu_baseline (the round's record-keeping runs -> profiles/<round>_bench_<workload>.json)
                 for w in cfg1 cfg3 cfg3gru e1024 cfg5; do timeout -k 10 500 python bench.py --workload $w --steps 40 --warmup 10 --no-grid > $O/bench_$w.json 2> $O/bench_$w.err || { tail -5 $O/bench_$w.err; exit 1; }; line $O/bench_$w.json; done
                 timeout -k 10 500 python bench.py --workload cfg5 --precision 8 --steps 40 --warmup 10 --no-grid > $O/bench_cfg5_p8.json 2> $O/bench_cfg5_p8.err || { tail -5 $O/bench_cfg5_p8.err; exit 1; }; line $O/bench_cfg5_p8.json ;;
    lockstep)    timeout -k 10 400 python tools/bench_lockstep.py --workload cfg2 --ks 1,4,15 --steps 12 > $O/lockstep.json 2> $O/lockstep.err || { tail -5 $O/lockstep.err; exit 1; }
                 tail -1 $O/lockstep.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['workload'], [(r['K'], r['seq_per_s'], r['ms_per_lockstep_step']) for r in d['results']])" ;;
    lockstep_rnn) for w in cfg3 cfg3gru; do timeout -k 10 400 python tools/bench_lockstep.py --workload $w --ks 1,4,16 --steps 12 > $O/lockstep_$w.json 2> $O/lockstep_$w.err || { tail -5 $O/lockstep_$w.err; exit 1; }
                 tail -1 $O/lockstep_$w.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['workload'], [(r['K'], r['seq_per_s'], r['ms_per_lockstep_step']) for r in d['results']])"; done ;;
    grid)        timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/grid.json 2> $O/grid.err || { tail -5 $O/grid.err; exit 1; }; line $O/grid.json ;;
    rehearsal)   timeout -k 10 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/rehearsal.json 2> $O/rehearsal.err || { tail -5 $O/rehearsal.err; exit 1; }; line $O/rehearsal.json ;;
    smoke)       timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -5 $O/smoke.txt; exit 1; }; tail -2 $O/smoke.txt ;;
    fullgrid)    timeout -k 10 600 python tools/full_grid.py --lockstep 15 > $O/full_grid_324x5.json 2> $O/fullgrid.err || { tail -8 $O/fullgrid.err; exit 1; }; cut -c1-400 $O/full_grid_324x5.json ;;
    passerr)     timeout -k 10 400 python tools/backward_pass_errors.py ${PASSERR:-cfg1 cfg2 cfg5} > $O/passerr.jsonl 2> $O/passerr.err || { tail -5 $O/passerr.err; exit 1; }; cat $O/passerr.jsonl ;;
    gridcal)     timeout -k 10 700 python tools/bench_grid.py ${GRIDCAL:-15x1 15x4 5x4} > $O/gridcal.jsonl 2> $O/gridcal.err || { tail -5 $O/gridcal.err; exit 1; }
                 python -c "
import json,sys
for l in open('$O/gridcal.jsonl'):
    d=json.loads(l); print(d['lockstep'], d['fits_per_gpu'], d['units_per_thread'], d['folds_per_hr'], d['seconds'], d['work_units'], d['scores_crc32'])" ;;
    profile)     bash tools/gpu/profile.sh ${PROFILE_TAG:-r05} $O/prof ${PROFILE_SECTIONS:-} || exit 1 ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
