"""Drop-in for the reference's ``EncoderDecoderAttnBaseBkp`` and its two wrappers
(/root/reference/model/base/encoder_decoder_attn_bkp.py:330-413,
model/encoder_decoder_{lstm,gru}_attn.py:4-6), running on the MI355X HIP path.

Same constructor kwargs (``src_vocab, tgt_vocab, batch_first, rnn_type, embedding_size=256,
hidden_size=512, num_layers=1, dropout=0.1, **kwargs``), same ``forward(X, y, lengths) ->
log-probs [B, V]``, same ``state_dict()`` keys (``model.encoder.rnn.weight_ih_l0`` ...,
``model.decoder.pre_output_layer.weight`` -- a dead weight that never receives a gradient,
``model.generator.proj.weight``).  Quirks kept on purpose: the decoder is
unrolled for exactly one step fed ``<bos>`` (index 0 on a torchtext-0.6 target vocab), so the
output does not depend on ``y``; padded encoder outputs hold float(pad_idx)."""
import math

import torch
import torch.nn as nn

from . import util
from ._arena_module import ArenaModule


def _reference_init(rnn_type, E, Hd, N, dropout, Vs, Vt, pad_src, pad_tgt):
    """Same torch modules in the reference's construction order (bkp.py:358-381; call arguments are
    evaluated left to right, so BahdanauAttention is built before the Decoder's own members)
    -> identical initial weights under one torch seed."""
    cls = {"gru": nn.GRU, "lstm": nn.LSTM}[rnn_type]
    drop = dropout if N > 1 else 0.
    enc = cls(input_size=E, hidden_size=Hd, num_layers=N, batch_first=True, bidirectional=True, dropout=drop)
    key = nn.Linear(2 * Hd, Hd, bias=False)
    query = nn.Linear(Hd, Hd, bias=False)
    energy = nn.Linear(Hd, 1, bias=False)
    dec = cls(input_size=E + 2 * Hd, hidden_size=Hd, num_layers=N, batch_first=True, dropout=drop)
    bridge = nn.Linear(2 * Hd, Hd, bias=True)
    pre = nn.Linear(3 * Hd + E, Hd, bias=False)
    src = nn.Embedding(Vs, E, padding_idx=pad_src)
    trg = nn.Embedding(Vt, E, padding_idx=pad_tgt)
    gen = nn.Linear(Hd, Vt, bias=False)
    sd = {}
    for k, v in enc.state_dict().items():
        sd["model.encoder.rnn." + k] = v
    sd["model.decoder.attention.key_layer.weight"] = key.weight
    sd["model.decoder.attention.query_layer.weight"] = query.weight
    sd["model.decoder.attention.energy_layer.weight"] = energy.weight
    for k, v in dec.state_dict().items():
        sd["model.decoder.rnn." + k] = v
    sd.update({"model.decoder.bridge.weight": bridge.weight, "model.decoder.bridge.bias": bridge.bias,
               "model.decoder.pre_output_layer.weight": pre.weight, "model.src_embed.weight": src.weight,
               "model.trg_embed.weight": trg.weight, "model.generator.proj.weight": gen.weight})
    return {k: v.detach() for k, v in sd.items()}


def _init_recipe(entries, E, Hd, pad_src, pad_tgt):
    """The distributions ``_reference_init`` draws (torch defaults: nn.LSTM / nn.GRU U(+-1/sqrt(Hd)) for every tensor,
    nn.Linear U(+-1/sqrt(fan_in)), embeddings N(0, 1) with a zero padding row) without building the torch modules."""
    rec = {}
    for name, shape, _ in entries:
        if ".rnn." in name:
            rec[name] = ("uniform", 1.0 / math.sqrt(Hd))
        elif name == "model.src_embed.weight":
            rec[name] = ("normal", 1.0, pad_src)
        elif name == "model.trg_embed.weight":
            rec[name] = ("normal", 1.0, pad_tgt)
        elif name == "model.decoder.bridge.bias":
            rec[name] = ("uniform", 1.0 / math.sqrt(2 * Hd))
        else:                                                       # bias-free Linear weights [out, in]
            rec[name] = ("uniform", 1.0 / math.sqrt(shape[1]))
    return rec


class EncoderDecoderAttnBase(ArenaModule):

    MAX_OUTPUT_LEN = 1                      # bkp.py:332

    RNN_TYPES = ("gru", "lstm")

    _dead_params = frozenset(["model.decoder.pre_output_layer.weight"])   # grad is None in the reference

    def __init__(self,
                 src_vocab,
                 tgt_vocab,
                 batch_first,
                 rnn_type,
                 embedding_size=256,
                 hidden_size=512,
                 num_layers=1,
                 dropout=0.1,
                 **kwargs):
        super(EncoderDecoderAttnBase, self).__init__()
        assert (rnn_type in self.RNN_TYPES), "Invalid `rnn_type`."
        self.batch_first = batch_first
        self.src_vocab = src_vocab
        self.tgt_vocab = tgt_vocab
        self.rnn_type = rnn_type
        self.embedding_size, self.hidden_size, self.num_layers = embedding_size, hidden_size, num_layers
        self.dropout_p = float(dropout)
        self.device = kwargs.get("device")
        self.precision = int(kwargs.get("precision", 3))
        self.seed = int(kwargs.get("dropout_seed", torch.initial_seed() & 0x7FFFFFFF))

        from slnlp import rnn_engine as re_
        pad_src, pad_tgt = util.get_pad_idx(src_vocab), util.get_pad_idx(tgt_vocab)
        self._cfg_args = dict(rnn_type=rnn_type, E=embedding_size, Hd=hidden_size, N=num_layers,
                              Vs=len(src_vocab), Vt=len(tgt_vocab), pad_src=pad_src, pad_tgt=pad_tgt,
                              bos_idx=util.get_bos_idx(tgt_vocab),      # 0 when '<bos>' is not in the vocab
                              dropout=self.dropout_p if num_layers > 1 else 0.0, precision=self.precision)
        entries, total = re_.layout(re_.make_config(B=1, S=1, **self._cfg_args))
        self.init = kwargs.get("init", "reference")             # see model/transformer.py
        assert self.init in ("reference", "recipe"), "init must be 'reference' or 'recipe'"
        if self.init == "recipe":
            self._setup_arena(entries, total, None, _init_recipe(entries, embedding_size, hidden_size, pad_src, pad_tgt), torch.initial_seed())
        else:
            init = _reference_init(rnn_type, embedding_size, hidden_size, num_layers, self.dropout_p,
                                   len(src_vocab), len(tgt_vocab), pad_src, pad_tgt)
            self._setup_arena(entries, total, init)

    def _state_order(self, views):
        return [(n, views[n], True) for n in self._param_names]

    def _make_engine(self, B, S, shared):
        from slnlp import rnn_engine as re_
        cfg = re_.make_config(B=B, S=S, **self._cfg_args)
        eng = re_.RnnEngine(cfg, device=self._arena.device, seed=self.seed, params=self._arena,
                            grads=shared["grads"], momentum=shared["momentum"], rng=shared["rng"], lr=shared["lr"], scalars=shared["scalars"])
        if getattr(self, "persistent_kernels", False):        # opt-in (never when several fits share the GPU)
            eng.set_persistent(True)
        return eng

    def forward(self, X, y, lengths, **kwargs):
        dev = self._arena.device
        X = X.to(dev, torch.int64).contiguous()                 # [B, S]: the reference's RNNs are batch_first
        y = y.to(dev, torch.int64).reshape(-1).contiguous()
        lengths = lengths.to(dev, torch.int64).reshape(-1).contiguous()
        return self._run((X, y, lengths))

    def extra_repr(self):
        return (f"rnn_type={self.rnn_type}, embedding_size={self.embedding_size}, hidden_size={self.hidden_size}, "
                f"num_layers={self.num_layers}, dropout={self.dropout_p}, precision={self.precision}")


def _recurrent_variant(name, rnn_type, ref):
    """The reference ships one two-line subclass per cell type (model/encoder_decoder_{lstm,gru}_attn.py); both are generated here."""
    def __init__(self, **kwargs):
        EncoderDecoderAttnBase.__init__(self, rnn_type=rnn_type, **kwargs)
    doc = f"Bastings encoder-decoder with Bahdanau attention over a bidirectional {rnn_type.upper()} ({ref})."
    return type(name, (EncoderDecoderAttnBase,), {"__init__": __init__, "__doc__": doc, "__module__": __name__})


EncoderDecoderLSTMAttn = _recurrent_variant("EncoderDecoderLSTMAttn", "lstm", "/root/reference/model/encoder_decoder_lstm_attn.py:4-6")
EncoderDecoderGRUAttn = _recurrent_variant("EncoderDecoderGRUAttn", "gru", "/root/reference/model/encoder_decoder_gru_attn.py:4-6")
