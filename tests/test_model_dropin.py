"""Drop-in surface of ``model.Transformer``: CPU part (constructor kwargs,
state_dict keys/shapes, init parity with the torch modules the reference builds,
loud failure without GPU) + GPU part (forward/backward through autograd with a
stock torch optimizer matches the golden training trajectory)."""
import numpy as np
import pytest
import torch

import gold


def make(c, dropout=0.0, **kw):
    import model
    return model.Transformer(embedding_size=c["E"], num_heads=c["H"], num_layers=c["N"], hidden_size=c["F"],
                             dropout=dropout, src_vocab=model.util.Vocab(c["Vs"]), tgt_vocab=model.util.Vocab(c["Vt"]),
                             device=torch.device("cpu"), batch_first=True, **kw)


def test_state_dict_keys_match_reference_and_init_parity():
    g, c, sd, X, L, y = gold.tf_case("tiny")
    torch.manual_seed(123)
    m = make(c, dropout=0.1)
    keys = list(m.state_dict().keys())
    want = list(g["param_order"])
    assert [k for k in keys if not k.endswith(".pe")] == want
    assert keys[1] == "src_pos_encoding.pe" and keys[3] == "tgt_pos_encoding.pe"
    assert tuple(m.state_dict()["src_pos_encoding.pe"].shape) == (5000, 1, c["E"])
    # same construction order as transformer.py:32-47 -> identical initial weights under one seed
    torch.manual_seed(123)
    src = torch.nn.Embedding(c["Vs"], c["E"]); tgt = torch.nn.Embedding(c["Vt"], c["E"])
    tr = torch.nn.Transformer(d_model=c["E"], nhead=c["H"], num_encoder_layers=c["N"], num_decoder_layers=c["N"],
                              dim_feedforward=c["F"], dropout=0.1)
    lin = torch.nn.Linear(c["E"], c["Vt"])
    msd = m.state_dict()
    assert torch.equal(msd["src_embedding.weight"], src.weight) and torch.equal(msd["tgt_embedding.weight"], tgt.weight)
    assert torch.equal(msd["linear.weight"], lin.weight) and torch.equal(msd["linear.bias"], lin.bias)
    for k, v in tr.state_dict().items():
        assert torch.equal(msd["transformer." + k], v), k
    # every parameter is a view of one arena; load_state_dict writes through
    m.load_state_dict({**msd, **{k: v for k, v in sd.items()}})
    assert torch.equal(m.state_dict()["linear.weight"], sd["linear.weight"])
    off = dict((n, o) for n, _, o in m._entries)["linear.weight"]
    assert torch.equal(m._arena[off:off + sd["linear.weight"].numel()].view_as(sd["linear.weight"]), sd["linear.weight"])
    assert m.to(torch.device("cpu")) is m and m.device == torch.device("cpu")


def test_cpu_forward_raises_no_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g, c, sd, X, L, y = gold.tf_case("tiny")
    m = make(c)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(X=X, y=y, lengths=L)
    with pytest.raises(AssertionError, match="required"):
        m(X=None, y=y)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "cfg1"])
def test_autograd_training_with_torch_optimizer_matches_golden(name):
    """The skorch recipe with stock components around the drop-in module:
    CrossEntropyLoss(ignore_index) + clip_grad_norm_ + torch.optim.SGD."""
    g, c, sd, X, L, y = gold.tf_case(name)
    m = make(c).to(torch.device("cuda"))
    m.load_state_dict({**m.state_dict(), **sd})
    m.eval()
    with torch.no_grad():
        logp = m(X=X.cuda(), y=y.cuda(), lengths=L.cuda())
    assert gold.rel_err(logp.cpu().numpy(), g["logp"]) < 1e-3
    assert np.array_equal(logp.argmax(-1).cpu().numpy(), g["argmax"])
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
    crit = torch.nn.CrossEntropyLoss(ignore_index=1)
    for s in range(len(g["losses"])):
        opt.zero_grad()
        out = m(X=X.cuda(), y=y.cuda(), lengths=L.cuda())
        loss = crit(out, y.cuda())
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.5)
        opt.step()
        assert abs(float(loss) - g["losses"][s]) < 1e-3 * g["losses"][s], s
        assert abs(float(norm) - g["grad_norms"][s]) < 2e-3 * g["grad_norms"][s], s
    gold.check_summary(g, "wfinal", {k: v.detach().cpu() for k, v in m.named_parameters()}, 1e-3)
    # q/k rows of the decoder self-attention get exactly-zero grads, like the reference
    w = dict(m.named_parameters())["transformer.decoder.layers.0.self_attn.in_proj_weight"]
    assert float(w.grad[: 2 * c["E"]].abs().max()) == 0.0


@pytest.mark.gpu
def test_seq_first_input_and_growing_batch():
    import model
    g, c, sd, X, L, y = gold.tf_case("cfg1")
    m = model.Transformer(embedding_size=c["E"], num_heads=c["H"], num_layers=c["N"], hidden_size=c["F"], dropout=0.0,
                          src_vocab=model.util.Vocab(c["Vs"]), tgt_vocab=model.util.Vocab(c["Vt"]),
                          batch_first=False).to("cuda")
    m.load_state_dict({**m.state_dict(), **sd})
    m.eval()
    a = m(X=X[:10].T.cuda(), y=y[:10].cuda())            # [S,B] input, small batch first
    b = m(X=X.T.cuda(), y=y.cuda())                       # then a larger one (plan is rebuilt)
    assert gold.rel_err(b.cpu().numpy(), g["logp"]) < 1e-3
    assert gold.rel_err(a.cpu().numpy(), g["logp"][:10]) < 1e-3


# ----------------------------------------------------------------------------- RNN drop-ins
def make_rnn(rnn_type, c, dropout=0.0):
    import model
    cls = model.EncoderDecoderLSTMAttn if rnn_type == "lstm" else model.EncoderDecoderGRUAttn
    return cls(src_vocab=model.util.Vocab(c["Vs"]), tgt_vocab=model.util.Vocab(c["Vt"]), batch_first=True,
               embedding_size=c["E"], hidden_size=c["Hd"], num_layers=c["N"], dropout=dropout)


@pytest.mark.parametrize("rnn_type", ["lstm", "gru"])
def test_rnn_state_dict_and_init_parity(rnn_type):
    g, c, sd, X, L, y = gold.rnn_case(rnn_type, "tiny")
    torch.manual_seed(5)
    m = make_rnn(rnn_type, c, dropout=0.2)
    assert list(m.state_dict().keys()) == list(g["param_order"])
    torch.manual_seed(5)
    cls = torch.nn.LSTM if rnn_type == "lstm" else torch.nn.GRU
    enc = cls(input_size=c["E"], hidden_size=c["Hd"], num_layers=c["N"], batch_first=True, bidirectional=True, dropout=0.2)
    key = torch.nn.Linear(2 * c["Hd"], c["Hd"], bias=False)
    msd = m.state_dict()
    for k, v in enc.state_dict().items():
        assert torch.equal(msd["model.encoder.rnn." + k], v), k
    assert torch.equal(msd["model.decoder.attention.key_layer.weight"], key.weight)
    assert float(msd["model.src_embed.weight"][1].abs().max()) == 0.0       # nn.Embedding(padding_idx) zeroes the pad row
    import model
    with pytest.raises(AssertionError, match="rnn_type"):
        model.encoder_decoder_attn.EncoderDecoderAttnBase(src_vocab=model.util.Vocab(8), tgt_vocab=model.util.Vocab(8),
                                                          batch_first=True, rnn_type="rnn")


@pytest.mark.gpu
@pytest.mark.parametrize("rnn_type", ["lstm", "gru"])
def test_rnn_autograd_training_matches_golden(rnn_type):
    g, c, sd, X, L, y = gold.rnn_case(rnn_type, "tiny")
    m = make_rnn(rnn_type, c).to(torch.device("cuda"))
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        logp = m(X=X.cuda(), y=y.cuda(), lengths=L.cuda())
    assert gold.rel_err(logp.cpu().numpy(), g["logp"]) < 1e-3
    assert np.array_equal(logp.argmax(-1).cpu().numpy(), g["argmax"])
    m.train()
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
    crit = torch.nn.CrossEntropyLoss(ignore_index=1)
    for s in range(len(g["losses"])):
        opt.zero_grad()
        loss = crit(m(X=X.cuda(), y=y.cuda(), lengths=L.cuda()), y.cuda())
        loss.backward()
        params = [p for p in m.parameters() if p.grad is not None]
        norm = torch.nn.utils.clip_grad_norm_(params, 0.5)
        opt.step()
        assert abs(float(loss) - g["losses"][s]) < 1e-3 * g["losses"][s], s
        assert abs(float(norm) - g["grad_norms"][s]) < 2e-3 * g["grad_norms"][s], s
    assert dict(m.named_parameters())["model.decoder.pre_output_layer.weight"].grad is None   # dead weight, as in the reference
    gold.check_summary(g, "wfinal", {k: v.detach().cpu() for k, v in m.named_parameters()}, 1e-3)


@pytest.mark.gpu
def test_autograd_path_redraws_dropout_masks_and_shares_state_across_lengths():
    """Stock-optimizer loop (no fused update): every training step must draw new dropout masks (the fused SGD kernel is what
    advances the step counter otherwise), and plans for different sequence lengths share ONE rng / momentum / gradient
    set owned by the module."""
    g, c, sd, X, L, y = gold.tf_case("tiny")
    m = make(c, dropout=0.5).to("cuda")
    m.load_state_dict({**m.state_dict(), **sd})
    m.train()
    Xd, yd = X.cuda(), y.cuda()
    outs = []
    for _ in range(3):
        lp = m(X=Xd, y=yd)
        lp.sum().backward()                       # no optimizer step: the weights stay put, only the masks can differ
        outs.append(lp.detach().cpu())
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])
    e_full = m.engine(Xd.shape[0], Xd.shape[1])
    assert int(e_full.rng[1]) == 3
    lp = m(X=Xd[:, :7].contiguous(), y=yd)        # a second sequence length -> a second plan
    lp.sum().backward()
    e_short = m.engine(Xd.shape[0], 7)
    assert e_short is not e_full and int(e_full.rng[1]) == 4
    for k in ("rng", "lr", "grads", "momentum"):
        assert getattr(e_short, k).data_ptr() == getattr(e_full, k).data_ptr(), k
    m.eval()
    with torch.no_grad():
        assert torch.equal(m(X=Xd, y=yd), m(X=Xd, y=yd))      # eval: no dropout
