# round 3, session 43: victim-2 library, workspace-diff probe with whole train steps / 6 layers
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s43; mkdir -p $O
echo "== 1 train step, N=2"; SLNLP_PROBE_LIB=4096 timeout -k 10 200 python tools/probes/probe_concurrent5.py 0.1 6 1 2 2>&1 | grep -v "ALIVE\|amdgpu.ids" | head -30 | tee $O/s1n2.txt
echo "== 3 train steps, N=2"; SLNLP_PROBE_LIB=4096 timeout -k 10 200 python tools/probes/probe_concurrent5.py 0.1 6 3 2 2>&1 | grep -v "ALIVE\|amdgpu.ids" | head -40 | tee $O/s3n2.txt
echo "== fwd+bwd, N=6"; SLNLP_PROBE_LIB=4096 timeout -k 10 200 python tools/probes/probe_concurrent5.py 0.1 6 0 6 2>&1 | grep -v "ALIVE\|amdgpu.ids" | head -40 | tee $O/s0n6.txt
