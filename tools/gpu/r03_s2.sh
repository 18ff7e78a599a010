# round 3, session 2: the 128 x 128 plane tile (suite with the tile forced, tile benchmark) and the split-K bisect of the
# multi-queue nondeterminism (3 processes together, per SLNLP_SPLITK_MODE)
set -o pipefail
O=gpurun_out/r03s2; mkdir -p $O
echo "== pytest -m gpu (automatic tile)"; timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -ne 0 ] && { echo "PYTEST RC $rc"; grep -E "^E|Error|FAILED" $O/pytest.log | head -20; }
echo "== pytest -m gpu, SLNLP_PLANE_TILE=128"; SLNLP_PLANE_TILE=128 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_transformer_gpu.py tests/test_lockstep_gpu.py tests/test_edge_shapes_gpu.py -m gpu -q > $O/pytest128.log 2>&1; rc=$?; tail -3 $O/pytest128.log; [ $rc -ne 0 ] && { echo "PYTEST128 RC $rc"; grep -E "^E|Error|FAILED" $O/pytest128.log | head -30; }
echo "== tile benchmark"; timeout -k 10 300 python tools/bench_plane_tiles.py 2>&1 | tee $O/tiles.txt | tail -40
echo "== split-K bisect: three processes together"
for m in 0 1 2 3; do SLNLP_SPLITK_MODE=$m timeout -k 10 200 python tools/probes/probe_procs_together.py 20 2>&1 | tail -1; done
