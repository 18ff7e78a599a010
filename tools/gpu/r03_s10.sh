# round 3, session 10: cross-lane reductions without the LDS crossbar (DPP + readlane) -- victims, probes, the suite
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s10; mkdir -p $O
echo "== kernel tests first (the new reductions)"; timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_transformer_gpu.py tests/test_rnn_gpu.py -m gpu -x -q > $O/pt1.log 2>&1; rc=$?; tail -2 $O/pt1.log; [ $rc -ne 0 ] && { grep -E "^E|Error|FAILED" $O/pt1.log | head -20; exit 1; }
echo "== victim probe"; timeout -k 10 300 python tools/probes/probe_victim.py 6 2>&1 | grep -v amdgpu.ids | grep -v "output [0-9]:" | tee $O/victim.txt
echo "== 3 processes together x3"
P="timeout -k 10 200 python tools/probes/probe_procs_together.py 12"
for i in 1 2 3; do $P 2>&1 | tail -1 | cut -c1-150; done
echo "== in-process, 3 streams, policy off"
timeout -k 10 200 python tools/probes/probe_concurrent5.py 0.1 8 2>&1 | grep -E "identical|fit [0-9]:" | head -10
echo "== pytest -m gpu (rest)"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -ne 0 ] && { echo "PYTEST RC $rc"; grep -E "^E|Error|FAILED" $O/pytest.log | head -20; }
