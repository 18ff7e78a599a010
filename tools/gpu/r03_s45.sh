# round 3, session 45: the round-2 LayerNorm backward (commit fcccc0d) under the victim probe, library built with (900) and without (904) packed fp32 instructions
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s45; mkdir -p $O
for x in 900 904 900 904; do
  echo "== probe $x" | tee -a $O/victim.txt
  SLNLP_PROBE_LIB=$x timeout -k 10 200 python tools/probes/probe_victim.py 3 2>&1 | grep -E "layernorm_bwd|chain" | tee -a $O/victim.txt
done
