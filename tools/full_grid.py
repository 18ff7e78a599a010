#!/usr/bin/env python3
"""configs[3]: EVERY candidate of config-transformer.yaml's grid_args (3 lr x 3 embedding x 3 hidden x 3 layers x 2 dropout x
2 heads = 324) x cv 5 = 1620 fits through ShardedGridSearchCV on the GPUs of this node, with the fit length bounded
(--epochs, --samples) so the run fits a GPU-box session.  What it shows: every shape of the full grid builds, trains and
scores on the HIP path (E 1024 / 6 layers / head_dim 16 ... 256), the work-unit packing and the dynamic schedule at the real
grid's skew (cost ratio ~300 : 1 between the largest and the smallest candidate), and folds/hr on it.

    python tools/full_grid.py [--epochs 2] [--samples 1000] [--lockstep 5] [--fits-per-gpu 3] > profiles/rNN_full_grid.json
"""
import argparse
import json
import os
import sys
import threading
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# /root/reference/config/config-transformer.yaml: grid_args (the values, restated)
FULL_GRID = {"lr": [0.1, 0.01, 0.001], "module__embedding_size": [1024, 512, 128], "module__hidden_size": [512, 256, 128],
             "module__num_layers": [6, 4, 2], "module__dropout": [0.5, 0.1], "module__num_heads": [8, 4]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--lockstep", type=int, default=5)
    ap.add_argument("--fits-per-gpu", type=int, default=3)
    ap.add_argument("--layers", default="6,4,2", help="subset of num_layers (smoke runs)")
    args = ap.parse_args()
    warnings.filterwarnings("ignore")
    from slnlp.data import synthetic_dataset
    from slnlp.grid import ShardedGridSearchCV
    from slnlp.net import NeuralNetClassifier
    dev = torch.device("cuda", 0)
    ds = synthetic_dataset(args.samples, seq_len=48, src_vocab=3000, n_labels=200, seed=1, min_len=8)
    grid = dict(FULL_GRID, module__num_layers=[int(v) for v in args.layers.split(",")])
    factory = lambda: NeuralNetClassifier(
        module="model.Transformer", module__src_vocab=ds.vocab_X, module__tgt_vocab=ds.vocab_y, module__batch_first=True,
        module__embedding_size=512, module__num_heads=4, module__num_layers=2, module__hidden_size=256, module__dropout=0.1,
        criterion__ignore_index=1, optimizer__momentum=0.9, optimizer__nesterov=False, lr=0.01, max_epochs=args.epochs, batch_size=50,
        device=str(dev), gradient_clipping={"gradient_clip_value": 0.5},
        scoring=["neg_log_loss", "accuracy", "precision_weighted", "recall_weighted", "f1_weighted"],
        lr_scheduler={"policy": "ReduceLROnPlateau", "factor": 0.2, "patience": 5},
        early_stopping={"patience": 30, "threshold": 1e-4, "threshold_mode": "rel"})
    gs = ShardedGridSearchCV(factory, grid, cv=5, refit=False, device=str(dev), fits_per_gpu=args.fits_per_gpu, lockstep=args.lockstep)
    t0 = time.perf_counter()
    stop = threading.Event()

    def heartbeat():                                   # a long run must show it is alive (gpurun's no-output watchdog)
        while not stop.wait(30.0):
            print(f"[full_grid] {time.perf_counter() - t0:7.1f} s, GPU memory {torch.cuda.memory_allocated(dev) / 2**30:.1f} GiB", file=sys.stderr, flush=True)
    th = threading.Thread(target=heartbeat, daemon=True)
    th.start()
    gs.fit(ds)
    dt = time.perf_counter() - t0
    stop.set()
    scores = np.asarray(gs.cv_results_["mean_test_score"])
    cand = gs.cv_results_["params"]
    by_E = {}
    for p, s in zip(cand, scores):
        by_E.setdefault(p["module__embedding_size"], []).append(float(s))
    out = {"grid": "config-transformer.yaml grid_args, all of it", "candidates": len(cand), "cv": 5, "fits": gs.n_tasks_, "work_units": gs.n_units_,
           "epochs_per_fit": args.epochs, "samples": args.samples, "lockstep": args.lockstep, "fits_per_gpu": args.fits_per_gpu,
           "seconds": round(dt, 1), "folds_per_hr": round(gs.n_tasks_ / dt * 3600.0, 0), "all_scores_finite": bool(np.isfinite(scores).all()),
           "best_params": gs.best_params_, "best_score": round(float(gs.best_score_), 5),
           "mean_score_by_embedding_size": {str(k): round(float(np.mean(v)), 4) for k, v in by_E.items()},
           "rank_seconds": [round(v, 1) for v in gs.rank_seconds_], "peak_gpu_memory_GiB": round(torch.cuda.max_memory_allocated(dev) / 2**30, 2)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
