"""A grid measurement in the regime the reference runs: fits trained TO EarlyStopping -- max_epochs 80,
EarlyStopping(patience 30, threshold 1e-4 rel), ReduceLROnPlateau(factor .2, patience 5) on valid_loss: helper.py:217-250,
config-transformer.yaml:10-27 (200 epochs there) -- on learnable synthetic data, so fits stop at different epochs and the lockstep
groups shrink and are rebuilt.  24 candidates (lr x dropout x embedding_size x num_layers) x cv 5 = 120 fits, lockstep 15,
5 host threads, one GPU.  Reports folds/hr, per-fit epochs, regroupings and the time the shrunken groups cost.

    python tools/bench_grid_long.py [--samples 1000] [--max-epochs 80] > profiles/r05_grid_long.json
"""
import argparse, json, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sign-language-nlp_amd")]
import numpy as np, torch
import bench
from slnlp import lockstep as ls
from slnlp import grid
from slnlp.data import synthetic_dataset
from slnlp.grid import ShardedGridSearchCV
from slnlp.net import NeuralNetClassifier

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=1000)
ap.add_argument("--max-epochs", type=int, default=80)
ap.add_argument("--threads", type=int, default=5)
ap.add_argument("--lockstep", type=int, default=15)
a = ap.parse_args()
GRID = {"lr": [0.1, 0.01, 0.001], "module__dropout": [0.5, 0.1], "module__embedding_size": [512, 128], "module__num_layers": [4, 2]}
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
import warnings
warnings.filterwarnings("ignore")
ds = synthetic_dataset(a.samples, seq_len=48, src_vocab=3000, n_labels=200, seed=1, min_len=8)


def factory(data, max_epochs):
    return lambda: NeuralNetClassifier(
        module="model.Transformer", module__src_vocab=data.vocab_X, module__tgt_vocab=data.vocab_y, module__batch_first=True,
        module__embedding_size=512, module__num_heads=8, module__num_layers=2, module__hidden_size=512, module__dropout=0.1,
        criterion__ignore_index=1, optimizer__momentum=0.9, optimizer__nesterov=False, lr=0.01, max_epochs=max_epochs, batch_size=50,
        device=str(dev), gradient_clipping={"gradient_clip_value": 0.5},
        early_stopping={"patience": 30, "threshold": 1e-4, "threshold_mode": "rel"},
        lr_scheduler={"policy": "ReduceLROnPlateau", "factor": 0.2, "patience": 5},
        scoring=["neg_log_loss", "accuracy", "precision_weighted", "recall_weighted", "f1_weighted"])


warm = ShardedGridSearchCV(factory(ds.truncated(200), 1), {"module__embedding_size": [512, 128], "module__num_layers": [4, 2]}, cv=2, refit=False,
                           device=str(dev), fits_per_gpu=1, schedule="static", lockstep=2)
warm.fit(ds.truncated(200))
ls.EPOCH_LOG = []
t0 = time.perf_counter()
gs = ShardedGridSearchCV(factory(ds, a.max_epochs), GRID, cv=5, refit=False, device=str(dev), fits_per_gpu=a.threads, lockstep=a.lockstep).fit(ds)
dt = time.perf_counter() - t0
log = ls.EPOCH_LOG
epochs_run = sorted(e for u in log for e in u["epochs_run"])
lost = total = 0.0
for u in log:
    full = [t for n, t in u["epochs"] if n == u["fits"]]
    per_fit_full = (np.median(full) / u["fits"]) if full else None       # seconds per fit-epoch at the unit's full width
    for n, t in u["epochs"]:
        total += t
        if per_fit_full is not None:
            lost += max(0.0, t - n * per_fit_full)
units = [{"unit": i, "fits": n, "estimated_cost": cost, "seconds": e - s} for (i, n, cost, s, e) in sorted(gs.unit_log_)]
ratio = np.array([u["seconds"] / u["estimated_cost"] for u in units])
ratio = ratio / np.median(ratio)
out = {"what": "grid search to EarlyStopping: %d candidates x cv 5 = %d fits, max_epochs %d, patience 30, ReduceLROnPlateau(0.2, 5), %d samples, lockstep %d, %d host threads, one MI355X"
               % (len(gs.cv_results_["params"]), gs.n_tasks_, a.max_epochs, a.samples, a.lockstep, a.threads),
       "folds_per_hr": round(gs.n_tasks_ / dt * 3600.0), "seconds": round(dt, 2), "work_units": gs.n_units_,
       "epochs_run": {"min": epochs_run[0], "p25": epochs_run[len(epochs_run) // 4], "median": epochs_run[len(epochs_run) // 2],
                      "p75": epochs_run[3 * len(epochs_run) // 4], "max": epochs_run[-1], "stopped_early": sum(1 for e in epochs_run if e < a.max_epochs)},
       "fit_epochs_total": int(sum(epochs_run)), "fit_epochs_if_none_stopped": gs.n_tasks_ * a.max_epochs,
       "regroupings": int(sum(u["regroups"] for u in log)), "units_that_regrouped": int(sum(1 for u in log if u["regroups"])),
       "epoch_seconds_total": round(total, 2), "epoch_seconds_lost_to_shrunken_groups": round(lost, 2),
       "lost_fraction": round(lost / total, 4) if total else None,
       "unit_seconds_over_estimated_cost_normalised": {"min": round(float(ratio.min()), 3), "p10": round(float(np.percentile(ratio, 10)), 3),
                                                       "median": 1.0, "p90": round(float(np.percentile(ratio, 90)), 3), "max": round(float(ratio.max()), 3),
                                                       "values": [round(float(r), 3) for r in ratio]},
       "scores_crc32": "%08x" % zlib.crc32(np.asarray(gs.cv_results_["mean_test_score"], dtype=np.float64).tobytes()),
       "best_params": {k: (float(v) if isinstance(v, float) else v) for k, v in gs.best_params_.items()}, "best_score": round(float(gs.best_score_), 5),
       "units": units,
       "unit_epochs": [{"fits": u["fits"], "regroups": u["regroups"], "active_per_epoch": [n for n, _ in u["epochs"]]} for u in log]}
print(json.dumps(out))
