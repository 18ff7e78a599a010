mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_lockstep_gpu.py -x -q > gpurun_out/r02_t2a.log 2>&1; rc=$?
tail -15 gpurun_out/r02_t2a.log
if [ $rc -ge 124 ]; then echo "lockstep tests timed out"; exit $rc; fi
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r02_t2.log 2>&1; rc=$?
tail -8 gpurun_out/r02_t2.log
if [ $rc -ge 124 ]; then echo "pytest timed out"; exit $rc; fi
timeout -k 10 300 python tools/bench_lockstep.py --workload cfg1 --ks 1,2,4,8,16 > gpurun_out/r02_ls_cfg1.json 2> gpurun_out/r02_ls_cfg1.err; rc=$?
tail -2 gpurun_out/r02_ls_cfg1.json; tail -3 gpurun_out/r02_ls_cfg1.err
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/bench_lockstep.py --workload cfg2 --ks 1,2,4,8 > gpurun_out/r02_ls_cfg2.json 2> gpurun_out/r02_ls_cfg2.err; rc=$?
tail -2 gpurun_out/r02_ls_cfg2.json; tail -3 gpurun_out/r02_ls_cfg2.err
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 50 --warmup 20 --no-grid --no-cpu-baseline > gpurun_out/r02_b3.json 2> gpurun_out/r02_b3.err
tail -c 600 gpurun_out/r02_b3.json
