# round 3, session 40: the round's final default bench line (4 host threads in the grid leg) + GPU suite on the final tree
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s40; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt &&
timeout -k 10 600 python bench.py > $O/r03_bench_cfg2.json 2> $O/cfg2.err || { tail -5 $O/cfg2.err; exit 1; }
cut -c1-200 $O/r03_bench_cfg2.json
