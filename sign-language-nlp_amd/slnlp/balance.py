"""Class balancing as index resampling (no imbalanced-learn dependency).

The reference (/root/reference/helper.py:355-388) chains imblearn's ``RandomUnderSampler(replacement=False)`` and
``RandomOverSampler`` with per-class targets smoothed around the mean class size u:

    under:  n_c -> min(n_c, round(u + ln n_c))          over (on the result):  n_c -> max(n_c, round(u + ln n_c))

This module computes the same targets and draws the samples the way imbalanced-learn 0.8 does with DICT sampling
strategies (which is what the reference passes: every class is a key): one ``numpy.random.RandomState(seed)`` per sampler;
classes visited in sorted order; the under-sampler calls ``choice(range(n_c), size=target_c, replace=False)`` for EVERY
class -- also one that keeps all its rows, whose rows come back permuted -- and concatenates the class blocks in sorted-class
order; the over-sampler appends, per class, ``choice(rows of c, size=needed_c, replace=True)`` (the same stream as
``randint(0, n_c, needed_c)``) after all rows of the under-sampled set.  imbalanced-learn is not installed in the build
image: tests/test_pipeline_cpu.py pins the order with indices worked out from that published algorithm by hand.
"""
import collections
import math

import numpy as np


def sampling_targets(counts):
    """{class: n} -> (under targets, over targets) per helper.py:362-377."""
    u = sum(counts.values()) / len(counts)
    smooth = lambda v: int(round(u + math.log(v)))
    under = {k: min(v, smooth(v)) for k, v in counts.items()}
    over = {k: max(v, smooth(v)) for k, v in under.items()}
    return under, over


def balance_indices(y, seed):
    """Row indices of the balanced dataset (under-sampling, then over-sampling)."""
    y = np.asarray(y)
    counts = dict(collections.Counter(y.tolist()))
    under, over = sampling_targets(counts)
    rs = np.random.RandomState(seed)                       # RandomUnderSampler(random_state=seed)
    keep = []
    for c in np.unique(y):
        members = np.flatnonzero(y == c)
        keep.append(members[rs.choice(range(len(members)), size=under[c], replace=False)])   # every class is in the dict
    idx = np.concatenate(keep)
    y_u = y[idx]
    rs = np.random.RandomState(seed)                       # RandomOverSampler(random_state=seed)
    extra = []
    for c in np.unique(y_u):
        members = np.flatnonzero(y_u == c)
        need = over[c] - len(members)
        if need > 0:
            extra.append(members[rs.randint(low=0, high=len(members), size=need)])
    if extra:
        idx = np.concatenate([idx, idx[np.concatenate(extra)]])
    return idx


def balance_dataset(dataset, seed):
    """TokenDataset -> balanced TokenDataset (helper.py:355 ``balance_dataset``)."""
    return dataset[balance_indices(dataset.y, seed)]
