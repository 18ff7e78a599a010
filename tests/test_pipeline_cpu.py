"""CPU: the callers / data formats either side of the hot path: ASL-Phono ingest without
torchtext, class balancing, the reference's config / grid-parameter conventions and the result artefacts."""
import collections
import json
import math
import os

import numpy as np
import pytest

from slnlp import balance, cli, ingest

FIELDS = ["orientation_dh", "orientation_ndh", "movement_dh", "movement_ndh", "handshape_dh", "handshape_ndh"]


def make_corpus(root, n_labels=6, per_label=(3, 5, 1, 4, 2, 6), seed=0):
    """Synthetic ASL-Phono directory: <gloss>-<k>.json with frames of phonology attributes (some null)."""
    rs = np.random.RandomState(seed)
    orient = ["left_back", "left_down_front", "right_up", "up_front", "down"]
    shapes = ["L", "B", "5", "flat_O"]
    for li in range(n_labels):
        for k in range(per_label[li]):
            frames = []
            for _ in range(rs.randint(2, 7)):
                ph = {f: ({"value": str(rs.choice(orient if not f.startswith("handshape") else shapes))} if rs.rand() > 0.3 else None)
                      for f in FIELDS + ["mouth_openness"]}
                frames.append({"phonology": ph})
            with open(os.path.join(root, f"gloss{li}-{k}.json"), "w") as f:
                json.dump({"label": f"gloss{li}", "frames": frames}, f)


def test_composition_strategies_match_the_documented_examples():
    row = {"orientation_dh": {"value": "left_back"}, "orientation_ndh": None, "movement_dh": {"value": "left_down_front"},
           "movement_ndh": None, "handshape_dh": {"value": "L"}, "handshape_ndh": None}
    assert ingest.compose_as_words([row], FIELDS) == ["lb--ldf--L-"]                      # dataset_builder.py:176-181
    assert ingest.compose_as_words_norm([row], FIELDS) == ["l_b-___-ldf-___-L-"]          # :190-195
    assert ingest.compose_sep_feat([row], FIELDS) == ["['lb', '', 'ldf', '', 'L', '']"]   # :213-218
    allv = ingest.compose_all_values([row], FIELDS)[0]
    assert allv.startswith("left_back           -                    -left_down_front     -") and len(allv) == 6 * 20 + 5


def test_ingest_vocab_order_padding_and_min_freq(tmp_path):
    make_corpus(str(tmp_path))
    ds = ingest.build_dataset(str(tmp_path), FIELDS, samples_min_freq=2, composition_strategy="as_words")
    # label gloss2 has a single sample -> dropped (prefix frequency < 2)
    assert len(ds) == 3 + 5 + 4 + 2 + 6 and all(not f.startswith("gloss2-") for f in ds.files)
    vx, vy = ds.vocab_X, ds.vocab_y
    assert vx.itos[:2] == ["<unk>", "<pad>"] and vy.itos[:2] == ["<unk>", "<pad>"] and vx.stoi["<bos>"] == 0
    freqs = [vx.freqs[w] for w in vx.itos[2:]]
    assert freqs == sorted(freqs, reverse=True)                                            # by frequency ...
    for a, b in zip(vx.itos[2:], vx.itos[3:]):
        assert vx.freqs[a] > vx.freqs[b] or a < b                                          # ... ties alphabetical
    assert vy.itos[2] == "gloss5" and len(vy) == 2 + 5                                     # most frequent label first
    S = ds.ids.shape[1]
    assert S == ds.lengths.max() and ds.ids.dtype == np.int64
    for row, n in zip(ds.ids, ds.lengths):
        assert (row[n:] == 1).all() and (row[:n] >= 2).all()                               # <pad> = 1 after the true length
    assert set(ds.y.tolist()) <= set(range(2, len(vy)))
    with pytest.raises(ValueError):
        ingest.build_dataset(str(tmp_path), FIELDS, composition_strategy="nope")
    with pytest.raises(FileNotFoundError):
        ingest.build_dataset(str(tmp_path / "missing"), FIELDS)


def test_balancing_targets_and_determinism():
    y = np.repeat(np.arange(5), [40, 3, 12, 1, 20])
    counts = dict(collections.Counter(y.tolist()))
    u = sum(counts.values()) / len(counts)
    under, over = balance.sampling_targets(counts)
    for k, v in counts.items():
        assert under[k] == min(v, round(u + math.log(v)))                                  # helper.py:362-377
        assert over[k] == max(under[k], round(u + math.log(under[k])))
    idx = balance.balance_indices(y, seed=1)
    got = collections.Counter(y[idx].tolist())
    assert dict(got) == over
    assert np.array_equal(idx, balance.balance_indices(y, seed=1)) and not np.array_equal(idx, balance.balance_indices(y, seed=2))
    assert len(set(idx[y[idx] == 0].tolist())) == over[0]                                  # under-sampling draws without replacement


def test_balancing_sample_order_is_imbalanced_learns():
    """The row order of the balanced set, pinned with indices worked out BY HAND from imbalanced-learn 0.8's published
    algorithm for dict sampling strategies (RandomUnderSampler: RandomState(seed).choice(range(n_c), target_c, replace=False) for
    every class in sorted order, blocks concatenated; RandomOverSampler: a fresh RandomState(seed), per sorted class
    choice(rows of c, needed_c, replace=True) appended) -- not with slnlp.balance itself.  y: classes of 12 / 2 / 5 / 1 / 8 rows,
    mean 5.6 -> under-sampling targets 8 / 2 / 5 / 1 / 8, over-sampling targets 8 / 6 / 7 / 6 / 8 (helper.py:355-377)."""
    y = np.array([0] * 12 + [1] * 2 + [2] * 5 + [3] + [4] * 8)
    want = [7, 10, 2, 5, 0, 1, 11, 8,          # class 0: 8 of its 12 rows, drawn without replacement
            13, 12,                            # class 1 keeps both rows -- permuted, as choice(range(2), 2, replace=False) returns them
            15, 18, 16, 17, 14, 19,            # class 2 (5 rows, permuted), class 3 (1 row)
            21, 25, 22, 26, 23, 27, 24, 20,    # class 4 (8 rows, permuted)
            12, 13, 12, 13,                    # over-sampling: class 1 needs 4 more
            17, 17,                            # class 2 needs 2
            19, 19, 19, 19, 19]                # class 3 needs 5
    assert balance.balance_indices(y, 7).tolist() == want


def test_config_merge_grid_names_and_workdir(tmp_path):
    cfg_file = tmp_path / "c.yaml"
    cfg_file.write_text("seed: 1\nworkdir: '%s/{model}/run'\nmodel: model.Transformer\nlr:\nmodel_args:\n  embedding_size:\n"
                        "optimizer_args: {momentum: 0.9, nesterov: false}\n"
                        "grid_args:\n  lr: [0.1, 0.01]\n  model_args: {embedding_size: [128, 512], num_heads: [8, 4]}\n" % tmp_path)
    args = cli.load_config(str(cfg_file), {"grid_args": {"lr": [0.5]}, "max_epochs": 3})
    assert args["grid_args"]["lr"] == [0.5] and args["grid_args"]["model_args"]["num_heads"] == [8, 4] and args["max_epochs"] == 3
    grid = cli.build_param_grid(args["grid_args"])
    assert grid == {"module__embedding_size": [128, 512], "module__num_heads": [8, 4], "lr": [0.5]}   # helper.py:108-180
    assert cli.format_dir(args["workdir"], **{k: v for k, v in args.items() if k != "workdir"}) == os.path.normpath(f"{tmp_path}/model.Transformer/run")
    assert cli.prefix_args("optimizer", **args["optimizer_args"]) == {"optimizer__momentum": 0.9, "optimizer__nesterov": False}


def test_artefact_writers(tmp_path):
    import pandas as pd
    grid = {"lr": [0.1, 0.01], "module__num_layers": [2, 4, 6]}
    cli.save_param_grid(grid, "grid_search", str(tmp_path))
    df = pd.read_csv(tmp_path / "grid_search_grid_params.csv", index_col=0)
    assert list(df.columns) == ["lr", "module__num_layers"] and len(df) == 6                 # helper.save_param_grid
    cv = {"params": [{"lr": 0.1}, {"lr": 0.01}], "mean_test_score": np.array([-1.0, -2.0]), "rank_test_score": np.array([1, 2], dtype=np.int32)}
    cli.save_cv_results(cv, "grid_search", str(tmp_path))
    assert list(pd.read_csv(tmp_path / "grid_search_results.csv", index_col=0)["rank_test_score"]) == [1, 2]
    cli.save_json({"best_score": np.float64(-1.0), "best_params": {"lr": 0.1}, "best_index": np.int64(0)}, str(tmp_path / "o.json"))
    assert json.load(open(tmp_path / "o.json")) == {"best_score": -1.0, "best_params": {"lr": 0.1}, "best_index": 0}


def test_split_is_torch_random_split(tmp_path):
    import torch
    from slnlp.data import synthetic_dataset
    ds = synthetic_dataset(40, seq_len=6, src_vocab=20, n_labels=3, seed=2, min_len=2)
    test, train = ds.split(0.15, seed=1)
    perm = torch.randperm(40, generator=torch.Generator().manual_seed(1)).numpy()           # asl_dataset.py:240-244
    assert len(test) == 6 and np.array_equal(test.ids, ds.ids[perm[:6]]) and np.array_equal(train.y, ds.y[perm[6:]])


def test_fast_epoch_metrics_equal_the_sklearn_scorers():
    """slnlp/metrics.py forms the reference's five EpochScoring metrics (config-transformer.yaml:9) from a per-sample
    reduction; the numbers must be the ones the sklearn scorers (helper.py:529-554 wrapper) return."""
    import warnings
    import torch
    from slnlp import metrics
    from slnlp.net import ScoringWrapper, _CachedPredictor
    for seed, (N, V) in enumerate([(3000, 202), (37, 16), (500, 202)]):
        rng = np.random.RandomState(seed)
        y = rng.randint(2, V, N)
        logits = (rng.randn(N, V) * 3).astype(np.float32)
        logits[np.arange(N)[::3], y[::3]] += 6.0                        # a third of the samples classified correctly
        logp = torch.log_softmax(torch.from_numpy(logits), -1)
        if seed == 1:                                                   # p == 1 and p == 0: both clip bounds of log_loss
            logp[0, :] = -1e3
            logp[0, 3] = 0.0
        fast = metrics.epoch_scores(list(metrics.FAST), logp, torch.from_numpy(y))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for name in metrics.FAST:
                ref = float(ScoringWrapper(name, list(range(V)))(_CachedPredictor(np.exp(logp.numpy()), np.arange(V)), None, y))
                assert fast[name] == ref, (name, fast[name], ref)
