"""Command-line driver with the reference's config surface (/root/reference/main.py:12-143, args.py:3-53,
helper.py:307-341,415-440): one YAML file (config/config-*.yaml of the reference works unchanged) plus overrides,
then dataset -> optional balancing -> 85/15 split -> cross-validated grid search -> test metrics, writing the same
artefacts into ``workdir``:

    config.yaml                    the merged arguments                       (helper.dump_args)
    grid_search_grid_params.csv    cross product of the grid                  (helper.save_param_grid)
    grid_search_output.json        best_score / best_params / best_index      (main.tune_hyperparams)
    grid_search_results.csv        cv_results_ as a table                     (helper.save_cv_results)
    test_output.json               test_<metric> of the refitted best model   (main.test_model)
    params.pt optimizer.pt criterion.pt history.json   of the refit           (skorch Checkpoint)

    python -m slnlp.cli --config config-transformer.yaml [--grid_args '{"lr": [0.1]}'] [--max_epochs 5] ...
    python -m torch.distributed.run --nproc-per-node 8 -m slnlp.cli --config ...      # one rank per GPU

What replaces what: dask workers -> one process per GPU (``ShardedGridSearchCV``); commons-python's argument loader
-> PyYAML + argparse; torchtext / imblearn -> ``slnlp.ingest`` / ``slnlp.balance``; the torch profiler dump is not
reproduced.  ``dataset_args.synthetic: {n: ..}`` (not in the reference) generates an ASL-Phono-shaped dataset when
the corpus is not on the machine.
"""
import argparse
import copy
import datetime
import itertools
import json
import os

import numpy as np

DICT_ARGS = ("early_stopping", "gradient_clipping", "lr_scheduler", "dataset_args", "model_args", "optimizer_args",
             "criterion_args", "grid_args")
SCALAR_ARGS = {"model": str, "optimizer": str, "criterion": str, "cv": int, "scoring": str, "verbose": int, "n_jobs": int,
               "workdir": str, "debug": lambda s: s.lower() in ("1", "true", "yes"),
               "cuda": lambda s: s.lower() in ("1", "true", "yes"), "seed": int, "lr": float, "max_epochs": int,
               "batch_size": int, "test_size": float}


def deep_merge(base, over):
    out = copy.deepcopy(base)
    for k, v in over.items():
        out[k] = deep_merge(out[k], v) if isinstance(v, dict) and isinstance(out.get(k), dict) else copy.deepcopy(v)
    return out


def load_config(path=None, overrides=None):
    import yaml
    cfg = {}
    if path:
        with open(path) as f:
            cfg = yaml.safe_load(f) or {}
    cfg = deep_merge(cfg, overrides or {})
    for k in ("model_args", "optimizer_args", "criterion_args", "grid_args", "dataset_args"):
        cfg.setdefault(k, {})
        if cfg[k] is None:
            cfg[k] = {}
    return cfg


def format_dir(workdir, **kwargs):
    """``workdir`` is a ``str.format`` template over the run's arguments plus ``{datetime:%Y-...}`` (the reference's
    config files use ``{model}`` and ``{datetime:...}``, config-transformer.yaml:4; helper.py:307-313).  No workdir ->
    the empty string."""
    if not workdir:
        return ""
    fields = dict(kwargs, datetime=datetime.datetime.now())
    return os.path.normpath(workdir.format(**fields))


def prefix_args(prefix, ensure_list=False, output=None, **kwargs):
    """Flatten nested argument dicts into skorch's double-underscore names: ``prefix_args("module", a={"b": v})`` ->
    ``{"module__a__b": v}`` (helper.py:325-341).  ``ensure_list`` wraps scalars in one-element lists, the form a
    parameter grid needs.  Iterative (explicit stack of (name, value) pairs), leaves in depth-first key order."""
    flat = {} if output is None else output
    stack = [(k if prefix is None else f"{prefix}__{k}", v) for k, v in reversed(list(kwargs.items()))]
    while stack:
        name, value = stack.pop()
        if isinstance(value, dict):
            stack.extend((f"{name}__{k}", v) for k, v in reversed(list(value.items())))
        elif ensure_list and not isinstance(value, list):
            flat[name] = [value]
        else:
            flat[name] = value
    return flat


def build_param_grid(grid_args):
    """helper.py:108-180 ``build_grid_params``: model_args -> module__*, optimizer_args -> optimizer__*,
    criterion_args -> criterion__*, everything else (lr, ...) by its own name."""
    g = dict(grid_args or {})
    grid = {}
    grid.update(prefix_args("module", ensure_list=True, **(g.pop("model_args", None) or {})))
    grid.update(prefix_args("optimizer", ensure_list=True, **(g.pop("optimizer_args", None) or {})))
    grid.update(prefix_args("criterion", ensure_list=True, **(g.pop("criterion_args", None) or {})))
    g.pop("training_args", None)
    grid.update(prefix_args(None, ensure_list=True, **g))
    return grid


def build_net_params(args, dataset, device):
    """helper.py:41-105 ``build_net_params`` for ``slnlp.net.NeuralNetClassifier``."""
    from model.util import get_pad_idx
    model_args = {k: v for k, v in (args.get("model_args") or {}).items() if v is not None}
    crit = dict(args.get("criterion_args") or {})
    crit["ignore_index"] = get_pad_idx(dataset.vocab_y)
    p = {"module": args["model"], "criterion": args.get("criterion", "torch.nn.CrossEntropyLoss"),
         "optimizer": args.get("optimizer", "torch.optim.SGD"), "device": device,
         "scoring": args.get("scoring"), "early_stopping": args.get("early_stopping"),
         "gradient_clipping": args.get("gradient_clipping"), "lr_scheduler": args.get("lr_scheduler"),
         "checkpoint_dir": args.get("workdir") or None}
    for k in ("lr", "max_epochs", "batch_size", "verbose"):
        if args.get(k) is not None:
            p[k] = args[k]
    if isinstance(p.get("scoring"), str):
        p["scoring"] = [p["scoring"]]
    p.update(prefix_args("module", batch_first=True, src_vocab=dataset.vocab_X, tgt_vocab=dataset.vocab_y, **model_args))
    p.update(prefix_args("optimizer", **(args.get("optimizer_args") or {})))
    p.update(prefix_args("criterion", **crit))
    return p


def load_dataset(args):
    da = dict(args.get("dataset_args") or {})
    if da.get("synthetic"):
        from .data import synthetic_dataset
        return synthetic_dataset(**da["synthetic"])
    from .ingest import build_dataset
    return build_dataset(**da)


def _jsonable(o):
    if isinstance(o, dict):
        return {str(k): _jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_jsonable(v) for v in o]
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, np.ndarray):
        return o.tolist()
    return o if isinstance(o, (str, int, float, bool, type(None))) else str(o)


def save_json(obj, path):
    with open(path, "w") as f:
        json.dump(_jsonable(obj), f, indent=2)


def save_param_grid(param_grid, phase, workdir):
    import pandas as pd
    cols = list(param_grid.keys())
    pd.DataFrame(list(itertools.product(*[param_grid[c] for c in cols])), columns=cols).to_csv(
        os.path.join(workdir, f"{phase}_grid_params.csv"))


def save_cv_results(cv_results, phase, workdir):
    import pandas as pd
    pd.DataFrame({k: (list(v) if not isinstance(v, list) else v) for k, v in cv_results.items()}).to_csv(
        os.path.join(workdir, f"{phase}_results.csv"))


def run(args):
    """main.run + tune_hyperparams + test_model.  Returns (grid search object, test metrics); rank 0 writes files."""
    import random

    import torch

    from . import grid as G
    from .balance import balance_dataset
    from .net import NeuralNetClassifier, ScoringWrapper

    seed = int(args.get("seed", 1))
    torch.manual_seed(seed); random.seed(seed); np.random.seed(seed)            # helper.setup_seed
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("slnlp.cli: needs an MI355X -- the HIP path is the only compute path")
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            # the only collectives are the dataset broadcast and the score all_gather, hours apart on a full grid:
            # the default 10-minute watchdog would abort ranks that wait for a slower one
            dist.init_process_group("nccl", device_id=torch.device(device), timeout=datetime.timedelta(hours=48))
    workdir = args.get("workdir") or "."
    if rank == 0:
        os.makedirs(workdir, exist_ok=True)
        import yaml
        with open(os.path.join(workdir, "config.yaml"), "w") as f:
            yaml.safe_dump(_jsonable(args), f)

    dataset = load_dataset(args) if rank == 0 else None
    if world > 1:
        dataset = G.broadcast_dataset(dataset, device)
    if args.get("debug"):
        dataset = dataset.truncated(int(args.get("cv", 5)) * 10)
    if (args.get("dataset_args") or {}).get("balance_dataset") is True:
        dataset = balance_dataset(dataset, seed)
    test_data, train_data = dataset.split(float(args.get("test_size", 0.15)), seed)

    net_params = build_net_params(args, dataset, device)
    factory = lambda: NeuralNetClassifier(**net_params)
    scoring = args.get("scoring") or "neg_log_loss"
    first_score = scoring[0] if isinstance(scoring, list) else scoring
    param_grid = build_param_grid(args.get("grid_args"))
    gs = G.ShardedGridSearchCV(factory, param_grid, cv=int(args.get("cv", 5)), scoring=first_score, refit=True,
                               device=device, verbose=int(args.get("verbose", 0) or 0),
                               fits_per_gpu=int(args.get("fits_per_gpu", 1)), seed=seed)
    gs.fit(train_data)
    test_output = None
    if rank == 0:
        phase = "grid_search"
        save_param_grid(param_grid, phase, workdir)
        save_json({"best_score": float(gs.best_score_), "best_params": gs.best_params_, "best_index": int(gs.best_index_),
                   "scoring": repr(ScoringWrapper(first_score, train_data.labels()))}, os.path.join(workdir, f"{phase}_output.json"))
        save_cv_results(gs.cv_results_, phase, workdir)
        metrics = scoring if isinstance(scoring, list) else [scoring]
        if "accuracy" not in metrics:
            metrics = ["accuracy", *metrics]                                    # main.test_model
        est = gs.best_estimator_
        test_output = {f"test_{m}": float(ScoringWrapper(m, test_data.labels())(est, test_data, test_data.y)) for m in metrics}
        save_json(test_output, os.path.join(workdir, "test_output.json"))
        # workdir/{params,optimizer,criterion}.pt + history.json are the refit's best-valid-loss checkpoint (skorch
        # Checkpoint(monitor="valid_loss_best", dirname=workdir), helper.py:211-213) and stay untouched; the weights
        # after the last epoch (not kept by the reference) go to a directory of their own
        est.save_params(os.path.join(workdir, "final"))
    if world > 1:
        _wait_for_rank0(rank, world)
    return gs, test_output


def _wait_for_rank0(rank, world):
    """Ranks > 0 are done after the score all_gather; rank 0 still refits and tests.  Wait on the rendezvous store (a
    host-side key, no collective: nothing for a watchdog to time out) so the process group is torn down together."""
    import torch.distributed as dist
    try:
        from torch.distributed.distributed_c10d import _get_default_store
        store = _get_default_store()
        if rank == 0:
            store.set("slnlp/cli/done", "1")
        else:
            store.wait(["slnlp/cli/done"], datetime.timedelta(hours=48))
    except Exception:
        dist.barrier()


def main(argv=None):
    ap = argparse.ArgumentParser(description="SL Transformer (MI355X path)")
    ap.add_argument("--config", "-c", help="YAML file (the reference's config/config-*.yaml work unchanged)")
    for k, t in SCALAR_ARGS.items():
        ap.add_argument(f"--{k}", type=t, default=None)
    for k in DICT_ARGS:
        ap.add_argument(f"--{k}", type=json.loads, default=None, help="JSON object merged over the file's value")
    ap.add_argument("--fits_per_gpu", type=int, default=None, help="concurrent fits per GPU (not in the reference)")
    ns = vars(ap.parse_args(argv))
    path = ns.pop("config")
    args = load_config(path, {k: v for k, v in ns.items() if v is not None})
    args["workdir"] = format_dir(args.get("workdir"), **{k: v for k, v in args.items() if k != "workdir"})
    run(args)


if __name__ == "__main__":
    main()
