mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err; rc=$?
if [ $rc -ne 0 ]; then tail -20 gpurun_out/r02_bench_default.err; exit $rc; fi
tail -1 gpurun_out/r02_bench_default.json | cut -c1-3000
timeout -k 10 500 python bench.py --gpus 2 --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/r02_bench_g2.json 2> gpurun_out/r02_bench_g2.err; rc=$?
if [ $rc -ne 0 ]; then tail -20 gpurun_out/r02_bench_g2.err; exit $rc; fi
tail -1 gpurun_out/r02_bench_g2.json | cut -c1-1500
