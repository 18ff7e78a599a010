"""Error conventions of the C ABI: argument validation returns a status code that the binding
turns into RuntimeError carrying slnlp_last_error(); nothing aborts, nothing is launched, and the plan stays usable.
Also the inputs torch would reject or that a kernel must not fault on (ids outside the vocabulary)."""
import numpy as np
import pytest
import torch

import gold

pytestmark = pytest.mark.gpu


def _engine(c, sd=None, **over):
    from slnlp import tf_engine as te
    k = dict(c, **over)
    cfg = te.make_config(k["E"], k["H"], k["N"], k["F"], k["Vs"], k["Vt"], k["B"], k["S"])
    eng = te.TransformerEngine(cfg)
    if sd is not None:
        eng.load_state(sd)
    return eng


@pytest.mark.parametrize("over, msg", [
    (dict(S=5001), "seq_len"),            # beyond the reference's 5000-row positional table
    (dict(S=4000, B=50), "tokens"),       # B * S above the embedding backward's 65536-token chunk table
    (dict(E=30, H=4), "not divisible"),   # head_dim must divide
    (dict(E=2048, H=8), "E="),            # arena / kernel limit
    (dict(B=0), "batch"),
])
def test_bad_config_is_an_error_code_not_an_abort(over, msg):
    g, c, sd, X, L, y = gold.tf_case("tiny")
    with pytest.raises(RuntimeError, match=msg):
        _engine(c, **over)


def test_batch_larger_than_the_plan_and_missing_forward_are_rejected_and_the_plan_survives():
    g, c, sd, X, L, y = gold.tf_case("tiny")
    eng = _engine(c, sd)
    with pytest.raises(RuntimeError, match="needs a prior forward"):
        eng.backward()
    big_X, big_y = torch.cat([X, X]).cuda(), torch.cat([y, y]).cuda()
    with pytest.raises(RuntimeError, match="batch"):
        eng.forward(big_X, big_y)
    logp = eng.forward(X.cuda(), y.cuda()).cpu()                 # same plan, still correct
    assert gold.rel_err(logp.numpy(), g["logp"]) < 1e-3


def test_out_of_vocabulary_ids_do_not_fault():
    """torch raises IndexError for an id >= len(vocab); the kernels must at least stay inside their buffers: such a token
    embeds as zeros (+ positional encoding) and gets no gradient row."""
    g, c, sd, X, L, y = gold.tf_case("tiny")
    eng = _engine(c, sd)
    Xb = X.clone()
    Xb[0, 0] = c["Vs"] + 7
    Xb[1, 2] = -3
    logp = eng.train_step(Xb.cuda(), y.cuda(), 0.9, 0.5).cpu()
    torch.cuda.synchronize()
    assert torch.isfinite(logp).all() and np.isfinite(eng.loss)


def test_rnn_bad_config_and_lengths():
    from slnlp import rnn_engine as re_
    with pytest.raises(RuntimeError):
        re_.RnnEngine(re_.make_config("lstm", 32, 30, 2, 64, 16, 4, 12, 1, 1, 0, 0.0, 3))       # Hd % 4 != 0
    with pytest.raises(RuntimeError):
        re_.RnnEngine(re_.make_config("lstm", 32, 32, 2, 64, 16, 4, 2049, 1, 1, 0, 0.0, 3))     # S > 2048 (Bahdanau scores in LDS)
