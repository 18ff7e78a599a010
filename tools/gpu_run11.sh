mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_lockstep_gpu.py tests/test_rnn_gpu.py -q -x -s > gpurun_out/r02_t11.log 2>&1; rc=$?
grep -E "launches per lockstep|passed|failed|error" gpurun_out/r02_t11.log | tail -20 | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E " gpurun_out/r02_t11.log | head -30 | cut -c1-300; exit $rc; fi
