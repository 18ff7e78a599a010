"""The reference's five EpochScoring metrics (config/config-transformer.yaml:9, helper.py:529-554) from ONE pass over
the epoch's log-probs, which are still on the device.

skorch scores each metric by calling an sklearn scorer on the cached predictions, per metric and per data split:
ten sklearn calls per epoch on an [N, V] probability matrix (neg_log_loss alone binarises the labels into another
[N, V] matrix).  Here the device reduces the epoch to what the metrics need -- the arg-max class and the log-prob of the
true class per sample (N values each) -- and the scores are formed on the host with the arithmetic sklearn uses, so the
numbers are the ones the sklearn scorers return (tests/test_pipeline_cpu.py compares them).  A scorer name that is
not listed in ``FAST`` goes through sklearn as before.
"""
import numpy as np
import torch

FAST = ("accuracy", "precision_weighted", "recall_weighted", "f1_weighted", "neg_log_loss")


def reduce_epoch(logp, y):
    """Device side: (pred int64 [N], picked float32 [N]) as numpy.  ``logp`` [N, V] log-probs, ``y`` [N] class ids."""
    pred = logp.argmax(1)                                   # first maximum, like numpy
    picked = logp.gather(1, y.view(-1, 1)).view(-1).float()
    return pred.cpu().numpy(), picked.cpu().numpy()


def _prf(y_true, pred, n_classes):
    true_sum = np.bincount(y_true, minlength=n_classes)
    pred_sum = np.bincount(pred, minlength=n_classes)
    tp_sum = np.bincount(y_true[pred == y_true], minlength=n_classes)
    present = (true_sum + pred_sum) > 0                     # sklearn scores the labels that occur in y_true or y_pred
    return tp_sum[present], pred_sum[present], true_sum[present]


def _divide(num, den):                                      # sklearn _prf_divide with zero_division=0
    den = den.astype(np.float64)
    mask = den == 0.0
    den[mask] = 1.0
    out = num / den
    out[mask] = 0.0
    return out


def scores_from_reduction(names, y_true, pred, picked, n_classes):
    """Host side.  ``names`` must all be in ``FAST``."""
    out = {}
    prf = None
    for name in names:
        if name == "accuracy":                              # accuracy_score: average of (y_true == y_pred)
            out[name] = float(np.average(y_true == pred))
        elif name == "neg_log_loss":
            # log_loss(labels = all classes): probabilities are float32 exp(log-prob), clipped to [eps, 1 - eps] of
            # float32, the log is taken in float64 (xlogy of an int64 indicator and a float32 probability)
            p = np.exp(picked.astype(np.float32))
            eps = np.finfo(np.float32).eps
            p = np.clip(p, eps, 1 - eps)
            out[name] = -float(np.average(-np.log(p.astype(np.float64))))
        else:
            if prf is None:
                prf = _prf(y_true, pred, n_classes)
            tp, ps, ts = prf
            if name == "precision_weighted":
                per_class = _divide(tp, ps)
            elif name == "recall_weighted":
                per_class = _divide(tp, ts)
            elif name == "f1_weighted":                     # (1 + b^2) tp / (b^2 true + pred), b = 1
                per_class = _divide(2.0 * tp, 1.0 * ts + ps)
            else:
                raise KeyError(name)
            out[name] = float(np.average(per_class, weights=ts)) if ts.sum() > 0 else 0.0
    return out


def epoch_scores(names, logp, y, y_host=None):
    """{name: score} for the ``FAST`` names among ``names``; ``logp`` / ``y`` are device tensors of one epoch."""
    names = [n for n in names if n in FAST]
    if not names:
        return {}
    pred, picked = reduce_epoch(logp, y)
    y_true = np.asarray(y_host if y_host is not None else y.cpu().numpy()).astype(np.int64)
    return scores_from_reduction(names, y_true, pred, picked, int(logp.shape[1]))
