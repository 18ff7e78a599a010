# round 3, session 41: victim-2 with in-kernel corruption detectors (LDS statistics table, a VGPR and an SGPR sentinel across the K loop)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s41; mkdir -p $O
for i in 1 2 3; do
  SLNLP_PROBE_LIB=4096 timeout -k 10 100 python -m pytest tests/test_streams_gpu.py -m gpu -q -s -k 'overlapping_streams' > $O/run$i.txt 2>&1
  echo "run $i: $(grep -E '[0-9]+ (passed|failed)' $O/run$i.txt | tail -1); alive prints $(grep -c ALIVE $O/run$i.txt); detections $(grep -c DETECT $O/run$i.txt)"; grep DETECT $O/run$i.txt | head -6
done
