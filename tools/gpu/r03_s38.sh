# round 3, session 38: grid leg vs host threads per GPU
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s38; mkdir -p $O
for t in 2 4 6; do
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fits-per-gpu $t > $O/grid_t$t.json 2> $O/grid_t$t.err || { tail -3 $O/grid_t$t.err; exit 1; }
  python - <<PY
import json; d=json.loads(open("gpurun_out/r03s38/grid_t$t.json").read().strip().splitlines()[-1])["grid"]; print("threads $t:", d["value"], d["seconds"], d["scores_crc32"])
PY
done
