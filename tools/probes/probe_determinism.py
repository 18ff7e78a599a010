"""Are grid results reproducible run to run (same process, fresh estimators), with 1 and 3 host threads?"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
warnings.filterwarnings("ignore")
import numpy as np, torch
from slnlp.data import synthetic_dataset
from slnlp.grid import ShardedGridSearchCV
from slnlp.net import NeuralNetClassifier
dev = torch.device("cuda", 0)
ds = synthetic_dataset(1000, seq_len=48, src_vocab=3000, n_labels=200, seed=1, min_len=8)
grid = {"lr": [0.1, 0.01], "module__embedding_size": [1024, 128], "module__hidden_size": [512, 128], "module__num_layers": [2], "module__dropout": [0.1, 0.5], "module__num_heads": [4]}
factory = lambda: NeuralNetClassifier(
    module="model.Transformer", module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y, module__batch_first=True,
    module__embedding_size=512, module__num_heads=4, module__num_layers=2, module__hidden_size=256, module__dropout=0.1,
    criterion__ignore_index=1, optimizer__momentum=0.9, optimizer__nesterov=False, lr=0.01, max_epochs=2, batch_size=50,
    device=str(dev), gradient_clipping={"gradient_clip_value": 0.5}, scoring=["neg_log_loss"])
res = {}
for tag, kw in (("t1", dict(fits_per_gpu=1)), ("t1b", dict(fits_per_gpu=1)), ("t3", dict(fits_per_gpu=3)), ("t3b", dict(fits_per_gpu=3)),
                ("t3_ls1", dict(fits_per_gpu=3, lockstep=1)), ("ref_init", dict(fits_per_gpu=3, recipe_init=False)), ("ref_init_b", dict(fits_per_gpu=3, recipe_init=False))):
    kw.setdefault("lockstep", 5)
    gs = ShardedGridSearchCV(factory, grid, cv=5, refit=False, device=str(dev), **kw).fit(ds)
    res[tag] = np.stack([gs.cv_results_[f"split{i}_test_score"] for i in range(5)])
    print(tag, "mean", float(res[tag].mean()), flush=True)
for a, b in (("t1", "t1b"), ("t1", "t3"), ("t3", "t3b"), ("t1", "t3_ls1"), ("ref_init", "ref_init_b")):
    d = np.abs(res[a] - res[b])
    print(a, "vs", b, "identical" if np.array_equal(res[a], res[b]) else f"DIFFER: {int((d > 0).sum())} of {d.size} scores, max {d.max():.3e}", flush=True)
    if (d > 0).any():
        idx = np.argwhere(d > 0)[:6]
        print("   (fold, candidate):", [tuple(int(v) for v in i) for i in idx], [gs.cv_results_["params"][int(i[1])] for i in idx[:2]])
