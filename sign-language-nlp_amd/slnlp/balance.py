"""Class balancing as index resampling (no imbalanced-learn dependency).

The reference (/root/reference/helper.py:355-388) chains imblearn's ``RandomUnderSampler(replacement=False)`` and
``RandomOverSampler`` with per-class targets smoothed around the mean class size u:

    under:  n_c -> min(n_c, round(u + ln n_c))          over (on the result):  n_c -> max(n_c, round(u + ln n_c))

This module computes the same targets and draws the samples the way imbalanced-learn 0.8 does (one
``numpy.random.RandomState(seed)`` per sampler; classes visited in sorted order; under-sampling draws without
replacement inside each class and keeps class blocks in sorted-class order; over-sampling appends, per class, draws
with replacement after all original rows).  imbalanced-learn is not installed in the build image, so the sample-level
agreement is by construction from its published algorithm, not pinned by a test against it.
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
        pick = rs.choice(range(len(members)), size=under[c], replace=False) if under[c] < len(members) else slice(None)
        keep.append(members[pick])
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
