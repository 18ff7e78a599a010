# round 3, session 37: victim-2 with LDS guard zones: is its LDS written by someone else?
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s37; mkdir -p $O
for i in 1 2 3; do
  SLNLP_PROBE_LIB=2560 timeout -k 10 100 python -m pytest tests/test_streams_gpu.py -m gpu -q -s -k 'overlapping_streams' > $O/run$i.txt 2>&1
  echo "run $i: $(grep -E '[0-9]+ (passed|failed)' $O/run$i.txt | tail -1); guard hits: $(grep -c 'LDS GUARD HIT' $O/run$i.txt)"; grep 'LDS GUARD HIT' $O/run$i.txt | head -5
done
