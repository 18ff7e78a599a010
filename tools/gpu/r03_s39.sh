# round 3, session 39: recurrent forward step with two wave groups: RNN tests, cfg3 solo, 16-fit lockstep
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s39; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_rnn_gpu.py tests/test_lockstep_gpu.py tests/test_streams_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt &&
timeout -k 10 200 python bench.py --workload cfg3 --steps 100 --warmup 10 --no-grid --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee $O/bench_lstm.txt &&
timeout -k 10 200 python bench.py --workload cfg3gru --steps 100 --warmup 10 --no-grid --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee $O/bench_gru.txt &&
timeout -k 10 300 python tools/bench_lockstep.py --workload cfg3 --ks 1,4,16 --steps 10 2>&1 | grep '^{"K"' | tee $O/lockstep.txt
