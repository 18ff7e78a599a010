# round 3, session 4: the four plane-GEMM geometries (tests with each forced, benchmark) and the kernarg-path workaround
# (DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 made the 3-process probe deterministic in session 3: confirm, price)
set -o pipefail
O=gpurun_out/r03s4; mkdir -p $O
echo "== geometry tests"; timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "plane or group" > $O/pt_geo.log 2>&1; rc=$?; tail -3 $O/pt_geo.log; [ $rc -ne 0 ] && { grep -E "^E|Error|FAILED" $O/pt_geo.log | head -20; echo STOP; exit 1; }
for t in 12832 256128; do
  echo "== suite subset, SLNLP_PLANE_TILE=$t"; SLNLP_PLANE_TILE=$t timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_transformer_gpu.py tests/test_lockstep_gpu.py tests/test_edge_shapes_gpu.py -m gpu -q > $O/pt_$t.log 2>&1; rc=$?; tail -2 $O/pt_$t.log; [ $rc -ne 0 ] && grep -E "^E|Error|FAILED" $O/pt_$t.log | head -20
done
echo "== tile benchmark"; timeout -k 10 400 python tools/bench_plane_tiles.py 2>&1 | grep -v amdgpu.ids | tee $O/tiles.txt | tail -60
echo "== kernarg workaround: 3 processes together, baseline x3 / WA x3"
P="timeout -k 10 200 python tools/probes/probe_procs_together.py 12"
for i in 1 2 3; do $P 2>&1 | tail -1 | cut -c1-140; done
for i in 1 2 3; do DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 $P 2>&1 | tail -1 | cut -c1-140; done
echo "== in-process, 3 streams, policy off: baseline / WA"
timeout -k 10 200 python tools/probes/probe_concurrent5.py 0.1 6 2>&1 | grep -E "identical|fit [0-9]:" | head -8
DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 timeout -k 10 200 python tools/probes/probe_concurrent5.py 0.1 6 2>&1 | grep -E "identical|fit [0-9]:" | head -8
echo "== price of the workaround: cfg2 step"
timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-grid --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain   ', d['value'], d['ms_per_step'], d['config']['launch'])"
DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1 timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-grid --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WA=1    ', d['value'], d['ms_per_step'], d['config']['launch'])"
