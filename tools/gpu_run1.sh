mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02_t1.log 2>&1; rc=$?
tail -5 gpurun_out/r02_t1.log
if [ $rc -ge 124 ]; then echo "pytest timed out"; exit $rc; fi
timeout -k 10 400 python bench.py --steps 50 --warmup 20 > gpurun_out/r02_b1.json 2> gpurun_out/r02_b1.err; rc=$?
tail -c 1500 gpurun_out/r02_b1.json; tail -3 gpurun_out/r02_b1.err
if [ $rc -ge 124 ]; then echo "bench timed out"; exit $rc; fi
timeout -k 10 400 python bench.py --gpus 2 --steps 30 --warmup 20 > gpurun_out/r02_b2.json 2> gpurun_out/r02_b2.err; rc=$?
tail -c 1500 gpurun_out/r02_b2.json; tail -5 gpurun_out/r02_b2.err
exit $rc
