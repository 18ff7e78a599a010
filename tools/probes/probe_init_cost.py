"""How much of the grid leg's wall time is the serialized reference-style weight init?  Runs bench.grid_folds_per_hour twice:
as is, and with model.transformer._reference_init memoised per shape (NOT a valid mode -- an upper bound for a faster init)."""
import functools, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
import torch
import bench
dev = torch.device("cuda", 0)
r0 = bench.grid_folds_per_hour(dev, 1, 0)
print("as is     :", r0["value"], r0["seconds"], flush=True)
import model.transformer as mt
orig = mt._reference_init
cache = {}
def memo(*a):
    if a not in cache:
        cache[a] = orig(*a)
    return cache[a]
mt._reference_init = memo
r1 = bench.grid_folds_per_hour(dev, 1, 0)
print("memoised  :", r1["value"], r1["seconds"], flush=True)
