# round 3, session 20: where do the 50-row GEMM's microseconds go (dissection builds, dependent chain replayed from a graph)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s20; mkdir -p $O
for v in "" 16 32 48 64; do
  SLNLP_PROBE_LIB=$v timeout -k 10 100 python tools/bench_skinny_chain.py 50 512 2>&1 | grep -v amdgpu.ids | tee -a $O/chain.txt || exit 1
done
