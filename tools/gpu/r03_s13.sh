# is layernorm_bwd the ONLY victim?  Whole-fit probes with a library whose layernorm_bwd is the form that never differed in the
# victim probe (compile-time 4 rows per wave, no dgamma / dbeta accumulation: their gradients are simply zero in this build)
set -o pipefail
cd $GRAFT_REPO_ROOT
P="timeout -k 10 200 python tools/probes/probe_procs_together.py 12"
echo "== product library"; for i in 1 2; do $P 2>&1 | tail -1 | cut -c1-150; done
echo "== probe library 6"; for i in 1 2 3; do SLNLP_PROBE_LIB=6 $P 2>&1 | tail -1 | cut -c1-150; done
echo "== in-process, 3 streams, probe library 6"
SLNLP_PROBE_LIB=6 timeout -k 10 200 python tools/probes/probe_concurrent5.py 0.1 8 2>&1 | grep -E "identical|fit [0-9]:|^    [a-z]" | head -40
