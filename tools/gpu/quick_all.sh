# every GPU test, then the cfg2 / cfg3 / cfg1 bench lines (no grid, no CPU baseline)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/quick_all.log 2>&1; rc=$?
tail -2 gpurun_out/quick_all.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "^E |^FAILED" gpurun_out/quick_all.log | head -20 | cut -c1-300; exit $rc; fi
for w in cfg2 cfg3 cfg3gru cfg1; do
timeout -k 10 300 python bench.py --workload $w --steps 100 --warmup 20 --no-grid --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['parity'])" || exit 1
done
