# round 3, session 14: the shipped LayerNorm backward (row kernel without accumulators + table-driven partial kernel):
# full GPU suite, victim probe, whole-fit probes across processes and streams, short bench
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s14; mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.txt &&
timeout -k 10 200 python tools/probes/probe_victim.py 6 2>&1 | grep -v amdgpu.ids | tee $O/victim.txt &&
P="timeout -k 10 200 python tools/probes/probe_procs_together.py 12" &&
for i in 1 2 3; do $P 2>&1 | tail -1 | cut -c1-150 | tee -a $O/procs.txt; done &&
timeout -k 10 200 python tools/probes/probe_concurrent5.py 0.1 8 2>&1 | grep -E "identical|fit [0-9]:|^    [a-z]" | head -40 | tee $O/conc5.txt;
timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-grid --no-cpu-baseline 2>&1 | tail -1 | tee $O/bench.txt
