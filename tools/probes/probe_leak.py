"""Does an estimator's GPU memory come back by reference counting alone (no gc.collect) after a lockstep unit?"""
import gc, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
warnings.filterwarnings("ignore")
import torch
import bench
from slnlp.data import synthetic_dataset
from slnlp.lockstep import fit_and_score_group
gc.disable()
dev = torch.device("cuda", 0)
ds = synthetic_dataset(300, seq_len=48, src_vocab=3000, n_labels=200, seed=1, min_len=8)
fac = bench.grid_factory(ds, dev, 1)
import numpy as np
idx = np.arange(300)
base = torch.cuda.memory_allocated()
for rep in range(3):
    sc = fit_and_score_group(lambda: fac().set_params(module__init="recipe"), [{"lr": 0.01, "module__embedding_size": 512}] * 3,
                             [ds[idx[:200]]] * 3, [ds[idx[200:]]] * 3, "neg_log_loss", seeds=[1, 2, 3])
    print("after unit", rep, "allocated MiB:", (torch.cuda.memory_allocated() - base) >> 20, "gc objects pending:", len(gc.get_objects()))
n = gc.collect()
print("gc.collect freed", n, "-> allocated MiB:", (torch.cuda.memory_allocated() - base) >> 20)
gc.set_debug(gc.DEBUG_SAVEALL)
sc = fit_and_score_group(lambda: fac().set_params(module__init="recipe"), [{"lr": 0.01, "module__embedding_size": 512}] * 2,
                         [ds[idx[:200]]] * 2, [ds[idx[200:]]] * 2, "neg_log_loss", seeds=[1, 2])
gc.collect()
from collections import Counter
cnt = Counter(type(o).__name__ for o in gc.garbage)
print("garbage types:", cnt.most_common(25))
ours = [o for o in gc.garbage if type(o).__module__.split(".")[0] in ("slnlp", "model")]
print("ours:", Counter(type(o).__name__ for o in ours))
import types
for o in gc.garbage:
    if isinstance(o, (types.FunctionType, types.MethodType)):
        print("func:", getattr(o, "__qualname__", o))
for o in gc.garbage:
    if type(o).__name__ == "cell":
        try:
            print("cell ->", type(o.cell_contents).__name__)
        except ValueError:
            pass
print("==== referrer chains")
ids = {id(o) for o in gc.garbage}
def owners(o, depth=0, seen=None):
    seen = seen or set()
    if depth > 6 or id(o) in seen:
        return
    seen.add(id(o))
    for r in gc.get_referrers(o):
        if id(r) not in ids or r is gc.garbage:
            continue
        desc = type(r).__name__
        if isinstance(r, types.FrameType):
            desc += f" {r.f_code.co_name} ({r.f_code.co_filename.split('/')[-1]}:{r.f_lineno})"
        elif isinstance(r, types.FunctionType):
            desc += f" {r.__qualname__}"
        elif isinstance(r, dict):
            desc += " keys=" + ",".join(str(k) for k in list(r)[:6])
        print("  " * depth + "<- " + desc)
        if not isinstance(r, (types.FrameType,)):
            owners(r, depth + 1, seen)
for o in gc.garbage:
    if type(o).__name__ in ("NeuralNetClassifier",):
        print("NET")
        owners(o)
        break
for o in gc.garbage:
    if isinstance(o, types.FrameType):
        print("frame:", o.f_code.co_name, o.f_code.co_filename.split('/')[-1], o.f_lineno)
