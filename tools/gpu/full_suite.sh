mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r02_full.log 2>&1; rc=$?
tail -4 gpurun_out/r02_full.log | cut -c1-300
if [ $rc -ge 124 ]; then exit $rc; fi
if [ $rc -ne 0 ]; then grep -E "^E |^FAILED|^ERROR" gpurun_out/r02_full.log | head -20 | cut -c1-300; exit $rc; fi
