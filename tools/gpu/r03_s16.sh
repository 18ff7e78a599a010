# round 3, session 16: stream tests with per-thread streams as the default; 3-stream in-process probe (probe fixed: it zeroed the device tables)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s16; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_streams_gpu.py tests/test_lockstep_gpu.py tests/test_net_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.txt &&
timeout -k 10 200 python tools/probes/probe_concurrent5.py 0.1 8 > $O/conc5.txt 2>&1 && tail -5 $O/conc5.txt
