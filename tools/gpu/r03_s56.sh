# round 3, session 56: final tree: smoke(), GPU suite, the driver's torch.distributed.run launch form with 2 ranks on the one GPU
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s56; mkdir -p $O
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt &&
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt &&
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/torchrun2.json 2> $O/torchrun2.err || { tail -5 $O/torchrun2.err; exit 1; }
tail -1 $O/torchrun2.json | cut -c1-160; python - <<'PY'
import json; d=json.loads(open("gpurun_out/r03s56/torchrun2.json").read().strip().splitlines()[-1]); g=d["grid"]; print(d["n_gpus"], d["ranks"], g["value"], g["rank_seconds"], g["rank_fits"], g["scores_crc32"])
PY
