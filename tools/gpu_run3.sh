mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r02_t3.log 2>&1; rc=$?
tail -12 gpurun_out/r02_t3.log
if [ $rc -ge 124 ]; then echo "pytest timed out"; exit $rc; fi
timeout -k 10 500 python bench.py --steps 100 --warmup 20 > gpurun_out/r02_b4.json 2> gpurun_out/r02_b4.err; rc=$?
tail -c 3000 gpurun_out/r02_b4.json; tail -3 gpurun_out/r02_b4.err
