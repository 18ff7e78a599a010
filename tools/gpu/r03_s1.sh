# round 3, session 1: the GPU suite on the round's first library changes (StepScope, admission control, lockstep Adam),
# the library-free stale-read reproducer alone / as 3 processes / as 3 streams, the existing 3-process library probe,
# and the AQL packet headers HIP emits for back-to-back kernels.
set -o pipefail
O=gpurun_out/r03s1; mkdir -p $O
echo "== pytest -m gpu"; timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -8 $O/pytest.log; [ $rc -ne 0 ] && echo "PYTEST RC $rc"
R=tools/probes/stale_read_repro
echo "== reproducer: one process, one stream"
for m in "0 0" "1 0" "2 1"; do timeout -k 5 60 $R $m 4 || true; done
echo "== reproducer: three processes at once"
for m in "0 0" "1 0" "2 1"; do
  for k in 1 2 3; do timeout -k 5 60 $R $m 8 > $O/repro_p${k}.txt 2>&1 & done; wait
  cat $O/repro_p1.txt $O/repro_p2.txt $O/repro_p3.txt
done
echo "== reproducer: three streams in one process"
for m in "0 0" "1 0" "2 1"; do timeout -k 5 60 $R $m 8 3 || true; done
echo "== AQL headers (AMD_LOG_LEVEL=4)"
AMD_LOG_LEVEL=4 timeout -k 5 60 $R 0 0 0.02 > $O/aql.log 2>&1 || true
grep -i "dispatch header\|Dispatch Header" $O/aql.log | sed -E 's/.*(Dispatch Header[^,]*,? ?\([^)]*\)).*/\1/I' | sort | uniq -c | sort -rn | head -12
grep -ic "barrier" $O/aql.log || true
echo "== library probe: 3 processes (tools/probes/probe_concurrent_procs.py)"
timeout -k 10 400 python tools/probes/probe_concurrent_procs.py 2>&1 | tail -4
