"""Vocabulary helpers and mask builders with the reference's names and results
(/root/reference/model/util/util.py).  The masks are NOT used by the HIP path
(the kernels derive causal / padding masks from the token ids on the device);
they exist so code written against the reference's ``model.util`` keeps working.
"""
import torch

# /root/reference/dataset/constant/tokens.py:1-4
BOS_WORD = '<bos>'
EOS_WORD = '<eos>'
UNK_WORD = '<unk>'
PAD_WORD = '<pad>'


def get_pad_idx(vocab):
    """util.py:5-6"""
    return vocab.stoi[PAD_WORD]


def get_bos_idx(vocab):
    """util.py:8-9 -- on a torchtext-0.6 vocab without ``<bos>`` the defaultdict
    ``stoi`` answers 0 (= ``<unk>``)."""
    return vocab.stoi[BOS_WORD]


def generate_mask(data, batch_first=False):
    """util.py:11-42: bool [len, len], True where key j > query i (blocked)."""
    features_dim = 1 if batch_first else 0
    if data.ndim == 1:
        data = data.unsqueeze(features_dim)
    size = data.size(features_dim)
    i = torch.arange(size).unsqueeze(1)
    j = torch.arange(size).unsqueeze(0)
    return j > i


def generate_padding_mask(data, vocab):
    """util.py:45-61: (data == <pad>) transposed to [B, len]."""
    mask = (data == get_pad_idx(vocab)).bool()
    if mask.ndim < 2:
        mask = mask.unsqueeze(-1)
    else:
        mask = mask.transpose(0, 1)
    return mask


def resolve_lengths(data, vocab, dim=-1):
    """util.py:64-69"""
    pad_idx = get_pad_idx(vocab)
    if data.ndim < 2:
        data = data.unsqueeze(dim)
    return data.size(dim) - data.eq(pad_idx).sum(dim=dim)


class _Stoi(dict):
    """token -> index; unknown tokens answer 0 (``<unk>``) like torchtext-0.6's defaultdict.  Module-level so that a
    ``Vocab`` (and any module / estimator holding one) pickles."""

    def __missing__(self, key):
        return 0


class Vocab:
    """Minimal stand-in for the torchtext-0.6 ``Vocab`` the reference passes as
    ``src_vocab`` / ``tgt_vocab``: the model only needs ``.stoi[str]`` and
    ``len()`` (transformer.py:29-30, util.py:5-9).  Specials first: ``<unk>``=0,
    ``<pad>``=1; unknown keys resolve to 0 like torchtext's defaultdict."""

    def __init__(self, tokens_or_size):
        if isinstance(tokens_or_size, int):
            toks = [f"t{i}" for i in range(tokens_or_size - 2)]
        else:
            toks = [t for t in tokens_or_size if t not in (UNK_WORD, PAD_WORD)]
        self.itos = [UNK_WORD, PAD_WORD] + toks
        self.stoi = _Stoi({t: i for i, t in enumerate(self.itos)})

    def __len__(self):
        return len(self.itos)
