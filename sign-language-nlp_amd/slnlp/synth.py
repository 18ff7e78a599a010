"""Synthetic ASL-Phono-shaped data and seed-recipe weights.

No dataset or checkpoint can be fetched (no network), so benches, golden
fixtures and parity tests all draw inputs from the deterministic numpy recipes
below.  Layout follows what the reference hands to the model
(/root/reference/helper.py:293-304 ``collate_data``): ``X int64[B,S]`` padded
with ``<pad>``=1 after ``lengths[b]`` tokens, ``y int64[B]``.  Vocabulary ids
follow torchtext-0.6 ordering: ``<unk>``=0, ``<pad>``=1, real tokens from 2
(SURVEY.md section 8d).
"""
import math

import numpy as np

PAD_IDX = 1
UNK_IDX = 0


def make_batch(n, seq_len=48, src_vocab=3000, tgt_vocab=202, seed=1, min_len=8):
    """-> (X int64[n,seq_len], lengths int64[n], y int64[n]) numpy arrays."""
    rs = np.random.RandomState(seed)
    lengths = rs.randint(min(min_len, seq_len), seq_len + 1, size=n).astype(np.int64)
    X = rs.randint(2, src_vocab, size=(n, seq_len)).astype(np.int64)
    X[np.arange(seq_len)[None, :] >= lengths[:, None]] = PAD_IDX
    y = rs.randint(2, tgt_vocab, size=n).astype(np.int64)
    return X, lengths, y


def make_weights(shapes, seed=1):
    """Seed-recipe weights for a list of ``(name, shape)`` in a fixed order.

    Embeddings N(0,1); matrices xavier-uniform; LayerNorm gains 1+0.1 N(0,1);
    every bias 0.1 N(0,1) (non-zero so bias paths are exercised).  The recipe is
    pure numpy so fixtures only need to store the seed, not the weights.
    """
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in shapes:
        shape = tuple(int(s) for s in shape)
        if name.endswith("embedding.weight") or name.endswith("embed.weight"):
            w = rs.standard_normal(shape)
        elif len(shape) == 2:
            a = math.sqrt(6.0 / (shape[0] + shape[1]))
            w = rs.uniform(-a, a, size=shape)
        elif "norm" in name and name.endswith("weight"):
            w = 1.0 + 0.1 * rs.standard_normal(shape)
        else:
            w = 0.1 * rs.standard_normal(shape)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def make_dropout_bits(shape, p, seed):
    """Host-side keep mask (1 = keep) -- used only by tests to compare dropout
    paths with identical masks on both sides."""
    rs = np.random.RandomState(seed)
    return (rs.random_sample(shape) >= p).astype(np.float32)
