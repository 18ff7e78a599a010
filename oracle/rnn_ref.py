"""Oracle (test infrastructure): functional fp32 restatement of the reference
``model.EncoderDecoder{LSTM,GRU}Attn`` forward.

Follows /root/reference/model/base/encoder_decoder_attn_bkp.py:
``EncoderDecoderAttnBaseBkp.forward`` :388-413, ``Encoder.forward`` :102-159
(packed bidirectional RNN, pad positions filled with float(pad_idx)),
``Decoder.forward/forward_step/init_hidden`` :202-285 (exactly ONE step,
MAX_OUTPUT_LEN=1 :332), ``BahdanauAttention.forward`` :304-327 and
``Generator`` :69-76.  The LSTM/GRU cell arithmetic restates torch.nn.LSTM/GRU
(gate order i,f,g,o / r,z,n), which the reference delegates to at :95-100,186-190.

``sd`` is keyed like the reference ``state_dict()`` (``model.encoder.rnn.*`` ...).
"""
import torch

from .transformer_ref import dropout


def lstm_cell(x_proj, h, c, w_hh, b_hh):
    """x_proj = x @ W_ih^T + b_ih (pre-computed), returns (h', c')."""
    g = x_proj + h @ w_hh.T + b_hh
    Hd = h.shape[-1]
    i = torch.sigmoid(g[..., 0:Hd])
    f = torch.sigmoid(g[..., Hd:2 * Hd])
    gg = torch.tanh(g[..., 2 * Hd:3 * Hd])
    o = torch.sigmoid(g[..., 3 * Hd:])
    c2 = f * c + i * gg
    return o * torch.tanh(c2), c2


def gru_cell(x_proj, h, w_hh, b_hh):
    hp = h @ w_hh.T + b_hh
    Hd = h.shape[-1]
    r = torch.sigmoid(x_proj[..., 0:Hd] + hp[..., 0:Hd])
    z = torch.sigmoid(x_proj[..., Hd:2 * Hd] + hp[..., Hd:2 * Hd])
    n = torch.tanh(x_proj[..., 2 * Hd:] + r * hp[..., 2 * Hd:])
    return (1.0 - z) * n + z * h


def run_direction(x, lengths, sd, pre, sfx, rnn_type, reverse):
    """One direction of one layer over a padded batch with per-sequence
    lengths == pack_padded_sequence semantics (bkp.py:110-114): the state of
    sequence b only advances on steps t < lengths[b]; the reverse direction
    therefore starts at t = lengths[b]-1.  x [B,S,In] -> (out [B,S,Hd], h_final)."""
    B, S, _ = x.shape
    w_ih, w_hh = sd[pre + "weight_ih" + sfx], sd[pre + "weight_hh" + sfx]
    b_ih, b_hh = sd[pre + "bias_ih" + sfx], sd[pre + "bias_hh" + sfx]
    Hd = w_hh.shape[1]
    xp = x @ w_ih.T + b_ih
    h = torch.zeros(B, Hd, dtype=x.dtype)
    c = torch.zeros(B, Hd, dtype=x.dtype)
    outs = [None] * S
    steps = range(S - 1, -1, -1) if reverse else range(S)
    for t in steps:
        valid = (t < lengths).to(x.dtype).unsqueeze(1)  # [B,1]
        if rnn_type == "lstm":
            h2, c2 = lstm_cell(xp[:, t], h, c, w_hh, b_hh)
            c = valid * c2 + (1 - valid) * c
        else:
            h2 = gru_cell(xp[:, t], h, w_hh, b_hh)
        h = valid * h2 + (1 - valid) * h
        outs[t] = h2 * valid  # padded outputs: filled by the caller
    return torch.stack(outs, dim=1), h


def encoder(emb, lengths, sd, rnn_type, num_layers, pad_fill, p, masks):
    """bkp.py:102-132.  -> (output [B,S,2Hd] with pad positions = pad_fill,
    hidden [N,B,2Hd] = fwd || bwd final state per layer)."""
    B, S, _ = emb.shape
    x = emb
    finals = []
    valid = (torch.arange(S).unsqueeze(0) < lengths.unsqueeze(1)).unsqueeze(-1)
    for l in range(num_layers):
        pre = "model.encoder.rnn."
        of, hf = run_direction(x, lengths, sd, pre, f"_l{l}", rnn_type, False)
        ob, hb = run_direction(x, lengths, sd, pre, f"_l{l}_reverse", rnn_type, True)
        x = torch.cat([of, ob], dim=-1)
        finals.append(torch.cat([hf, hb], dim=-1))
        if l < num_layers - 1:
            x = dropout(x, p, masks, f"model.encoder.rnn.dropout{l}")
    out = torch.where(valid, x, torch.full_like(x, float(pad_fill)))
    return out, torch.stack(finals, dim=0)


def bahdanau(query, proj_key, value, mask, sd):
    """bkp.py:304-327.  query [B,1,Hd], proj_key [B,S,Hd], value [B,S,2Hd],
    mask [B,1,S] bool (True = valid)."""
    q = query @ sd["model.decoder.attention.query_layer.weight"].T
    e = torch.tanh(q + proj_key) @ sd["model.decoder.attention.energy_layer.weight"].T
    scores = e.squeeze(2).unsqueeze(1)              # [B,1,S]
    scores = scores.masked_fill(~mask, float("-inf"))
    alphas = torch.softmax(scores, dim=-1)
    return alphas @ value, alphas                   # [B,1,2Hd], [B,1,S]


def forward(sd, X, y, lengths, *, rnn_type, num_layers, pad_src=1, pad_tgt=1,
            bos_idx=0, p_drop=0.0, masks=None, taps=None):
    """X int64 [B,S], y int64 [B] (unused by the arithmetic: only <bos> is
    consumed, bkp.py:254 with max_len=1), lengths int64 [B] -> log-probs [B,V]."""
    B, S = X.shape
    src_mask = (X != pad_src).unsqueeze(1)          # bkp.py:404-406
    emb = sd["model.src_embed.weight"][X]           # [B,S,E]
    enc_out, enc_final = encoder(emb, lengths, sd, rnn_type, num_layers,
                                 pad_src, p_drop, masks)
    if taps is not None:
        taps["enc_out"], taps["enc_final"] = enc_out, enc_final
    # Decoder.init_hidden :268-280
    hidden = torch.tanh(enc_final @ sd["model.decoder.bridge.weight"].T
                        + sd["model.decoder.bridge.bias"])      # [N,B,Hd]
    cell = hidden
    proj_key = enc_out @ sd["model.decoder.attention.key_layer.weight"].T  # :246
    prev_embed = sd["model.trg_embed.weight"][
        torch.full((B, 1), bos_idx, dtype=torch.long)]          # [B,1,E]  :254
    ctx, alphas = bahdanau(hidden[-1].unsqueeze(1), proj_key, enc_out, src_mask, sd)
    if taps is not None:
        taps["alphas"], taps["context"] = alphas, ctx
    x = torch.cat([prev_embed, ctx], dim=2).squeeze(1)          # [B,E+2Hd]
    for l in range(num_layers):
        pre = "model.decoder.rnn."
        xp = x @ sd[pre + f"weight_ih_l{l}"].T + sd[pre + f"bias_ih_l{l}"]
        if rnn_type == "lstm":
            h, _ = lstm_cell(xp, hidden[l], cell[l], sd[pre + f"weight_hh_l{l}"],
                             sd[pre + f"bias_hh_l{l}"])
        else:
            h = gru_cell(xp, hidden[l], sd[pre + f"weight_hh_l{l}"],
                         sd[pre + f"bias_hh_l{l}"])
        x = h
        if l < num_layers - 1:
            x = dropout(x, p_drop, masks, f"model.decoder.rnn.dropout{l}")
    # bkp.py:40-46: the generator consumes decoder_states (``out``), NOT
    # pre_output -- pre_output_layer is dead weight.
    logits = x @ sd["model.generator.proj.weight"].T            # [B,V]
    if taps is not None:
        taps["dec_out"], taps["logits"] = x, logits
    return torch.log_softmax(logits, dim=-1)


def param_shapes(rnn_type, E, Hd, N, Vs, Vt):
    """Reference ``state_dict()`` order (bkp.py:358-381 construction order:
    encoder, decoder(attention, rnn, bridge, pre_output_layer), src_embed,
    trg_embed, generator)."""
    G = 4 if rnn_type == "lstm" else 3
    out = []
    for l in range(N):
        inp = E if l == 0 else 2 * Hd
        for sfx in (f"_l{l}", f"_l{l}_reverse"):
            out += [(f"model.encoder.rnn.weight_ih{sfx}", (G * Hd, inp)),
                    (f"model.encoder.rnn.weight_hh{sfx}", (G * Hd, Hd)),
                    (f"model.encoder.rnn.bias_ih{sfx}", (G * Hd,)),
                    (f"model.encoder.rnn.bias_hh{sfx}", (G * Hd,))]
    out += [("model.decoder.attention.key_layer.weight", (Hd, 2 * Hd)),
            ("model.decoder.attention.query_layer.weight", (Hd, Hd)),
            ("model.decoder.attention.energy_layer.weight", (1, Hd))]
    for l in range(N):
        inp = E + 2 * Hd if l == 0 else Hd
        out += [(f"model.decoder.rnn.weight_ih_l{l}", (G * Hd, inp)),
                (f"model.decoder.rnn.weight_hh_l{l}", (G * Hd, Hd)),
                (f"model.decoder.rnn.bias_ih_l{l}", (G * Hd,)),
                (f"model.decoder.rnn.bias_hh_l{l}", (G * Hd,))]
    out += [("model.decoder.bridge.weight", (Hd, 2 * Hd)),
            ("model.decoder.bridge.bias", (Hd,)),
            ("model.decoder.pre_output_layer.weight", (Hd, 3 * Hd + E)),
            ("model.src_embed.weight", (Vs, E)),
            ("model.trg_embed.weight", (Vt, E)),
            ("model.generator.proj.weight", (Vt, Hd))]
    return out
