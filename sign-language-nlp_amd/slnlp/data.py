"""In-memory token dataset with the tensor layout the reference hands to the
model (/root/reference/helper.py:293-304 ``collate_data``): ``X int64[N,S]``
padded with ``<pad>``, ``lengths int64[N]``, ``y int64[N]``.

Stands in for the reference's ``AslDataset`` / ``AslSliceDataset``
(dataset/asl_dataset.py:13-303) only as far as the hot path needs: it is
indexable by integer arrays (so sklearn CV splitters and ``_safe_indexing``
work, asl_dataset.py:261-271), carries the labels with the inputs (the
Transformer consumes ``y`` as decoder input even at predict time,
asl_dataset.py:48-55 / transformer.py:65) and exposes ``vocab_X`` / ``vocab_y``.
"""
import numpy as np


class LabelArray(np.ndarray):
    """``dataset.y`` as the int64 array everything here indexes -- and callable, so the reference's spelling
    ``train_data.y().to_array()`` (main.py:77, asl_dataset.py:200-208) reads the same labels."""

    def __call__(self):
        return self

    def to_array(self):
        return np.asarray(self)


class TokenDataset:
    def __init__(self, X, lengths, y, vocab_X=None, vocab_y=None):
        self.ids = np.ascontiguousarray(X, dtype=np.int64)
        self.lengths = np.ascontiguousarray(lengths, dtype=np.int64)
        self.y = np.ascontiguousarray(y, dtype=np.int64).view(LabelArray)
        assert self.ids.ndim == 2 and len(self.ids) == len(self.lengths) == len(self.y)
        self.vocab_X, self.vocab_y = vocab_X, vocab_y
        self.batch_first = True

    def __len__(self):
        return len(self.y)

    @property
    def shape(self):          # makes sklearn index with X[ndarray] instead of a python loop
        return (len(self.y),)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            return (self.ids[idx].tolist(), int(self.lengths[idx])), int(self.y[idx])   # asl_dataset item layout
        if isinstance(idx, tuple):        # sklearn's _array_indexing passes (indices, Ellipsis)
            idx = idx[0]
        idx = np.asarray(idx) if not isinstance(idx, slice) else idx
        return TokenDataset(self.ids[idx], self.lengths[idx], self.y[idx], self.vocab_X, self.vocab_y)

    def X(self):
        return self

    def to_array(self):
        return np.asarray(self.y)

    def labels(self):
        """Every index of the target vocabulary, specials included (asl_dataset.py:210-213 returns
        ``vocab_y.stoi.values()``); it feeds sklearn's log_loss ``labels=`` and must match the
        model's V output columns."""
        if self.vocab_y is not None:
            return list(range(len(self.vocab_y)))
        return sorted(set(self.y.tolist()))

    def truncated(self, n):
        return self[np.arange(min(n, len(self)))]

    def split(self, test_size=0.15, seed=1):
        """(test, train) like ``AslDataset.split(lengths=0.15, seed)`` (asl_dataset.py:220-253): that is
        ``torch.utils.data.random_split`` with ``Generator().manual_seed(seed)``, i.e. one ``torch.randperm`` whose
        first ``round(test_size * N)`` entries are the test set and the rest the train set, both in permutation order."""
        import torch
        n_test = int(round(len(self) * test_size)) if isinstance(test_size, float) else int(test_size)
        gen = torch.Generator().manual_seed(seed) if seed else None
        perm = torch.randperm(len(self), generator=gen).numpy()
        return self[perm[:n_test]], self[perm[n_test:]]


def collate_data(data):
    """helper.py:293-304: list of ((ids, len), label) -> ({"X","lengths","y"}, y) int64 tensors."""
    import torch
    X, y = zip(*data)
    if len(X[0]) == 3:
        X, X_lengths, _ = zip(*X)
    elif len(X[0]) == 2:
        X, X_lengths = zip(*X)
    X = torch.tensor(X, dtype=torch.long)
    X_lengths = torch.tensor(X_lengths, dtype=torch.long)
    y = torch.tensor(y, dtype=torch.long)
    return {"X": X, "lengths": X_lengths, "y": y}, y


def synthetic_dataset(n, seq_len=48, src_vocab=3000, n_labels=200, seed=1, min_len=8, with_vocab=True):
    """Synthetic ASL-Phono-shaped dataset: ``n_labels`` glosses (+ <unk>, <pad>) and learnable
    structure -- each label draws its tokens from a label-specific slice of the vocabulary.
    ``with_vocab=False``: arrays only (tools/gen_golden.py runs with the reference's ``model`` package imported)."""
    from . import synth
    if with_vocab:
        from model.util import Vocab
    else:
        Vocab = lambda size: None
    rs = np.random.RandomState(seed)
    tgt_vocab = n_labels + 2
    X, lengths, _ = synth.make_batch(n, seq_len, src_vocab, tgt_vocab, seed=seed, min_len=min_len)
    y = rs.randint(2, tgt_vocab, size=n).astype(np.int64)
    span = max(4, (src_vocab - 2) // 8)
    base = 2 + (y * 37) % (src_vocab - 2 - span)
    sig = base[:, None] + rs.randint(0, span, size=X.shape)
    use = rs.random_sample(X.shape) < 0.6
    X = np.where((X != synth.PAD_IDX) & use, sig, X)
    return TokenDataset(X, lengths, y, Vocab(src_vocab), Vocab(tgt_vocab))
