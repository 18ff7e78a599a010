# round 3, session 61: grid leg after the admission change: 1 rank and the 2-rank rehearsal
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s61; mkdir -p $O
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/g1.json 2> $O/g1.err || { tail -3 $O/g1.err; exit 1; }
timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/g2.json 2> $O/g2.err || { tail -3 $O/g2.err; exit 1; }
python - <<'PY'
import json
for n in ("g1","g2"):
    g=json.loads(open(f"gpurun_out/r03s61/{n}.json").read().strip().splitlines()[-1])["grid"]
    print(n, g["value"], g["seconds"], g["rank_seconds"], g["rank_fits"], g["scores_crc32"])
PY
