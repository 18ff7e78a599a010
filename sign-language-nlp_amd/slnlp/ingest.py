"""ASL-Phono ingest without torchtext: a directory of per-sample JSON files -> ``TokenDataset``.

Restates what the reference builds with torchtext 0.6 ``Field`` / ``TabularDataset``
(/root/reference/dataset/builder/dataset_builder.py:66-223, dataset/asl_dataset.py:129-178):

* a sample file holds ``{"label": <gloss>, "frames": [{"phonology": {<field>: {"value": str} | null, ...}}, ...]}``;
  samples whose file-name prefix (stem up to the first ``-``) occurs fewer than ``samples_min_freq`` times are
  dropped (dataset_builder.py:67-83);
* every frame becomes ONE source token, composed from the selected ``fields`` by one of four strategies
  (dataset_builder.py:138-223); the label string is whitespace-tokenised, its first token is the class;
* vocabularies list the specials ``<unk>`` (0), ``<pad>`` (1) first, then tokens by descending frequency, ties in
  alphabetical order (torchtext.vocab.Vocab); unknown strings map to ``<unk>`` = 0 -- which is why the RNN models'
  ``stoi['<bos>']`` is 0;
* all samples are padded ONCE to the dataset-global maximum length with ``<pad>`` and carry their true length
  (``Field.process`` over the whole dataset, asl_dataset.py:157-169).
"""
import collections
import json
import os

import numpy as np

from .data import TokenDataset

PAD_WORD, UNK_WORD = "<pad>", "<unk>"       # dataset/constant/tokens.py


class Vocab:
    """torchtext-0.6-shaped vocabulary: ``itos`` list, ``stoi`` mapping with unknown -> 0, ``freqs`` counter."""

    def __init__(self, counter, specials=(UNK_WORD, PAD_WORD)):
        self.freqs = collections.Counter(counter)
        self.itos = list(specials)
        words = sorted((w for w in self.freqs if w not in specials))          # alphabetical ...
        words.sort(key=lambda w: -self.freqs[w])                               # ... then stable by frequency, descending
        self.itos += words
        index = {w: i for i, w in enumerate(self.itos)}

        class _Stoi(dict):
            def __missing__(self, key):
                return 0                                                       # defaultdict(unk_index)

        self.stoi = _Stoi(index)

    def __len__(self):
        return len(self.itos)


def _initials(data):
    """'left_down_front' -> 'ldf'; null -> ''  (compose_as_words.compose_field)"""
    return "".join(k[0] for k in str(data["value"]).split("_")) if data else ""


def compose_all_values(rows, fields):
    return ["-".join(f"{(row[x]['value'] if row[x] else ''):<20}" for x in fields) for row in rows]


def compose_as_words(rows, fields):
    return ["-".join(_initials(row[f]) for f in fields) for row in rows]


def compose_as_words_norm(rows, fields):
    def one(field, data):
        values = str(data["value"]) if data else ""
        if field.startswith("orientation") or field.startswith("movement"):
            v = values.split("_")
            return (("l" if "left" in v else "r" if "right" in v else "_") +
                    ("u" if "up" in v else "d" if "down" in v else "_") +
                    ("f" if "front" in v else "b" if "back" in v else "_"))
        return values
    return ["-".join(one(f, row[f]) for f in fields) for row in rows]


def compose_sep_feat(rows, fields):
    return [str([_initials(row[f]) for f in fields]) for row in rows]


STRATEGIES = {"all_values": compose_all_values, "as_words": compose_as_words,
              "as_words_norm": compose_as_words_norm, "as_sep_feat": compose_sep_feat}


def read_samples(dataset_dir, samples_min_freq=1):
    """[(file name, label string, [phonology dict per frame])] of the samples that pass the prefix-frequency filter,
    in sorted file order."""
    if not os.path.isdir(dataset_dir):
        raise FileNotFoundError(f"Invalid dataset directory: {dataset_dir!r}")            # dataset_builder.py:76
    files = sorted(f for f in os.listdir(dataset_dir) if f.endswith(".json"))
    prefix = lambda f: os.path.splitext(f)[0].split("-")[0]
    count = collections.Counter(prefix(f) for f in files)
    out = []
    for f in files:
        if count[prefix(f)] < samples_min_freq:
            continue
        with open(os.path.join(dataset_dir, f)) as fh:
            d = json.load(fh)
        frames = [fr.get("phonology") or {} for fr in d.get("frames", [])]
        out.append((f, "" if d.get("label") is None else str(d["label"]), frames))
    return out


def build_dataset(dataset_dir, fields, samples_min_freq=1, composition_strategy="as_words", **_ignored):
    """``dataset_args`` of the reference config -> TokenDataset (ids padded to the global max length, lengths, labels)
    with ``vocab_X`` / ``vocab_y``.  ``reuse_transient`` / ``balance_dataset`` are accepted and ignored here."""
    if composition_strategy not in STRATEGIES:
        raise ValueError(f"Unknown composition strategy: '{composition_strategy}'")       # dataset_builder.py:145
    compose = STRATEGIES[composition_strategy]
    samples = read_samples(dataset_dir, samples_min_freq)
    if not samples:
        raise ValueError(f"no sample in {dataset_dir!r} passes samples_min_freq={samples_min_freq}")
    src = [compose([{f: fr.get(f) for f in fields} for fr in frames], fields) for _, _, frames in samples]
    tgt = [label.split() for _, label, _ in samples]
    vocab_X = Vocab(collections.Counter(t for s in src for t in s))
    vocab_y = Vocab(collections.Counter(t for s in tgt for t in s))
    S = max(len(s) for s in src)
    ids = np.full((len(src), S), vocab_X.stoi[PAD_WORD], dtype=np.int64)
    for i, s in enumerate(src):
        ids[i, :len(s)] = [vocab_X.stoi[t] for t in s]
    lengths = np.array([len(s) for s in src], dtype=np.int64)
    y = np.array([vocab_y.stoi[t[0]] if t else vocab_y.stoi[PAD_WORD] for t in tgt], dtype=np.int64)
    ds = TokenDataset(ids, lengths, y, vocab_X, vocab_y)
    ds.files = [f for f, _, _ in samples]
    return ds
