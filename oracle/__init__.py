"""CPU oracle for the sign-language-nlp hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU restatement (plain torch fp32, explicit
matmul / softmax / layer-norm arithmetic, no ``nn.Transformer`` / ``nn.LSTM``)
of the reference's model forward, loss, gradient clipping and SGD update:

* ``transformer_ref``  -- /root/reference/model/transformer.py:60-109 plus the
  ``torch.nn.Transformer`` arithmetic it delegates to.
* ``rnn_ref``          -- /root/reference/model/base/encoder_decoder_attn_bkp.py
* ``train_ref``        -- the skorch step configured by
  /root/reference/helper.py:61-70,227-229 and config/config-transformer.yaml:36-43.

Parity pin: every function here is checked against golden vectors captured
from the reference itself (``tools/gen_golden.py`` imports the reference's
``model`` package in the build container and writes ``tests/golden/*.npz``);
see ``tests/test_oracle_golden.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package, and only as the checker / the
timed CPU baseline.  The product path (``sign-language-nlp_amd/``) never
imports it and has no CPU fallback.
"""
