# lockstep + transformer + rnn tests, solo lines and lockstep sweeps of cfg2 / cfg3 / cfg3gru
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_lockstep_gpu.py tests/test_transformer_gpu.py tests/test_rnn_gpu.py tests/test_kernels_gpu.py -q > gpurun_out/quick_ls.log 2>&1; rc=$?
tail -2 gpurun_out/quick_ls.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E |^FAILED" gpurun_out/quick_ls.log | head -20 | cut -c1-300; exit $rc; fi
for w in cfg2 cfg3 cfg3gru; do timeout -k 10 300 python tools/bench_lockstep.py --workload $w --ks 1,4,8,16 --steps 12 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['workload'], [(r['K'], r['seq_per_s'], r['ms_per_lockstep_step']) for r in d['results']])" || exit 1; done
