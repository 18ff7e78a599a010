"""Oracle (test infrastructure): functional fp32 restatement of the reference
``model.Transformer`` forward.

Follows /root/reference/model/transformer.py:60-109 (wrapper, masks,
embedding * sqrt(E) + positional encoding), model/util/util.py:11-61 (mask
construction), model/component/positional_encoding.py:23-49 (sin/cos table)
and the post-LN / ReLU / eps=1e-5 arithmetic of ``torch.nn.Transformer`` that
transformer.py:40-45,82-87 delegates to.  Written with explicit matmuls so it
can be read line-by-line against the HIP kernels.

``sd`` is a dict of tensors keyed exactly like the reference ``state_dict()``.
"""
import math

import torch


# --------------------------------------------------------------------------
# model/component/positional_encoding.py:27-35
def positional_table(max_len, d_model, dtype=torch.float32):
    pe = torch.zeros(max_len, d_model, dtype=dtype)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(
        torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe  # [max_len, d_model]  (reference stores it as [max_len, 1, d_model])


# model/util/util.py:11-42  -> True where key j > query i (blocked)
def causal_mask(size):
    i = torch.arange(size).unsqueeze(1)
    j = torch.arange(size).unsqueeze(0)
    return j > i


# model/util/util.py:45-61  -> [B, len] bool, True at <pad>
def padding_mask(ids_len_first, pad_idx):
    return (ids_len_first == pad_idx).transpose(0, 1)


def dropout(x, p, masks, name):
    """Dropout with host-supplied keep masks (``masks[name]``: 0/1 tensor of
    x's shape).  ``masks is None`` or p == 0 -> identity (eval / parity mode).
    ``masks == "draw"``: fresh Bernoulli keep masks from torch's generator, what ``nn.Dropout`` does in the
    reference's train mode (used to time the CPU step like-for-like, not for parity)."""
    if masks is None or p == 0.0:
        return x
    if isinstance(masks, str):
        return x * torch.bernoulli(torch.full_like(x, 1.0 - p)) / (1.0 - p)
    return x * masks[name].to(x.dtype) / (1.0 - p)


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def mha(q_in, kv_in, w_in, b_in, w_out, b_out, H, attn_mask, key_pad, p, masks,
        name):
    """torch.nn.MultiheadAttention arithmetic, seq-first.
    q_in [T,B,E]; kv_in [S,B,E]; attn_mask [T,S] bool or None (True = blocked);
    key_pad [B,S] bool or None (True = ignored)."""
    T, B, E = q_in.shape
    S = kv_in.shape[0]
    dh = E // H
    q = q_in @ w_in[:E].T + b_in[:E]
    k = kv_in @ w_in[E:2 * E].T + b_in[E:2 * E]
    v = kv_in @ w_in[2 * E:].T + b_in[2 * E:]
    # [B,H,len,dh]
    q = q.reshape(T, B, H, dh).permute(1, 2, 0, 3)
    k = k.reshape(S, B, H, dh).permute(1, 2, 0, 3)
    v = v.reshape(S, B, H, dh).permute(1, 2, 0, 3)
    scores = (q @ k.transpose(-1, -2)) / math.sqrt(dh)  # [B,H,T,S]
    blocked = torch.zeros(B, 1, T, S, dtype=torch.bool)
    if attn_mask is not None:
        blocked = blocked | attn_mask.view(1, 1, T, S)
    if key_pad is not None:
        blocked = blocked | key_pad.view(B, 1, 1, S)
    scores = scores.masked_fill(blocked, float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    probs = dropout(probs, p, masks, name + ".attn")
    ctx = probs @ v  # [B,H,T,dh]
    ctx = ctx.permute(2, 0, 1, 3).reshape(T, B, E)
    return ctx @ w_out.T + b_out, probs


def encoder_layer(x, sd, pre, H, src_mask, src_pad, p, masks, taps=None):
    sa, _ = mha(x, x, sd[pre + "self_attn.in_proj_weight"],
                sd[pre + "self_attn.in_proj_bias"],
                sd[pre + "self_attn.out_proj.weight"],
                sd[pre + "self_attn.out_proj.bias"], H, src_mask, src_pad, p,
                masks, pre + "self_attn")
    x = layer_norm(x + dropout(sa, p, masks, pre + "dropout1"),
                   sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    h = torch.relu(x @ sd[pre + "linear1.weight"].T + sd[pre + "linear1.bias"])
    h = dropout(h, p, masks, pre + "dropout")
    ff = h @ sd[pre + "linear2.weight"].T + sd[pre + "linear2.bias"]
    x = layer_norm(x + dropout(ff, p, masks, pre + "dropout2"),
                   sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    return x


def decoder_layer(t, mem, sd, pre, H, tgt_mask, tgt_pad, p, masks):
    sa, _ = mha(t, t, sd[pre + "self_attn.in_proj_weight"],
                sd[pre + "self_attn.in_proj_bias"],
                sd[pre + "self_attn.out_proj.weight"],
                sd[pre + "self_attn.out_proj.bias"], H, tgt_mask, tgt_pad, p,
                masks, pre + "self_attn")
    t = layer_norm(t + dropout(sa, p, masks, pre + "dropout1"),
                   sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    # transformer.py:82-87 passes no memory_mask / memory_key_padding_mask.
    ca, _ = mha(t, mem, sd[pre + "multihead_attn.in_proj_weight"],
                sd[pre + "multihead_attn.in_proj_bias"],
                sd[pre + "multihead_attn.out_proj.weight"],
                sd[pre + "multihead_attn.out_proj.bias"], H, None, None, p,
                masks, pre + "multihead_attn")
    t = layer_norm(t + dropout(ca, p, masks, pre + "dropout2"),
                   sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    h = torch.relu(t @ sd[pre + "linear1.weight"].T + sd[pre + "linear1.bias"])
    h = dropout(h, p, masks, pre + "dropout")
    ff = h @ sd[pre + "linear2.weight"].T + sd[pre + "linear2.bias"]
    t = layer_norm(t + dropout(ff, p, masks, pre + "dropout3"),
                   sd[pre + "norm3.weight"], sd[pre + "norm3.bias"])
    return t


def forward(sd, X, y, *, num_heads, num_layers, pad_src=1, pad_tgt=1,
            p_drop=0.0, masks=None, taps=None):
    """X int64 [B,S] (batch_first=True as main.py:25 builds it), y int64 [B]
    -> log-probs float32 [B, V].  ``taps`` (dict) receives intermediates."""
    E = sd["src_embedding.weight"].shape[1]
    src = X.transpose(0, 1)            # transformer.py:64, adjust_batch_in
    tgt = y.unsqueeze(-1).transpose(0, 1)   # [1,B]
    S, T = src.shape[0], tgt.shape[0]
    src_mask = causal_mask(S)          # transformer.py:68 -- encoder IS causal
    tgt_mask = causal_mask(T)          # [[False]]
    src_pad = padding_mask(src, pad_src)
    tgt_pad = padding_mask(tgt, pad_tgt)

    pe = positional_table(max(S, T), E)
    x = sd["src_embedding.weight"][src] * math.sqrt(E) + pe[:S].unsqueeze(1)
    x = dropout(x, p_drop, masks, "src_pos_encoding.dropout")
    t = sd["tgt_embedding.weight"][tgt] * math.sqrt(E) + pe[:T].unsqueeze(1)
    t = dropout(t, p_drop, masks, "tgt_pos_encoding.dropout")
    if taps is not None:
        taps["src_embed"] = x
        taps["tgt_embed"] = t

    for i in range(num_layers):
        x = encoder_layer(x, sd, f"transformer.encoder.layers.{i}.", num_heads,
                          src_mask, src_pad, p_drop, masks)
        if taps is not None:
            taps[f"enc{i}"] = x
    mem = layer_norm(x, sd["transformer.encoder.norm.weight"],
                     sd["transformer.encoder.norm.bias"])
    if taps is not None:
        taps["memory"] = mem
    for i in range(num_layers):
        t = decoder_layer(t, mem, sd, f"transformer.decoder.layers.{i}.",
                          num_heads, tgt_mask, tgt_pad, p_drop, masks)
        if taps is not None:
            taps[f"dec{i}"] = t
    t = layer_norm(t, sd["transformer.decoder.norm.weight"],
                   sd["transformer.decoder.norm.bias"])
    logits = t @ sd["linear.weight"].T + sd["linear.bias"]  # [1,B,V]
    if taps is not None:
        taps["logits"] = logits
    logp = torch.log_softmax(logits, dim=-1)
    return logp.squeeze(0)             # transformer.py:101-104


def param_shapes(E, H, N, F, Vs, Vt):
    """Parameter names and shapes in the reference's ``state_dict()`` order
    (buffers ``*_pos_encoding.pe`` excluded)."""
    out = [("src_embedding.weight", (Vs, E)), ("tgt_embedding.weight", (Vt, E))]

    def attn(pre):
        return [(pre + "in_proj_weight", (3 * E, E)), (pre + "in_proj_bias", (3 * E,)),
                (pre + "out_proj.weight", (E, E)), (pre + "out_proj.bias", (E,))]

    def ffn(pre):
        return [(pre + "linear1.weight", (F, E)), (pre + "linear1.bias", (F,)),
                (pre + "linear2.weight", (E, F)), (pre + "linear2.bias", (E,))]

    def norms(pre, n):
        o = []
        for k in range(1, n + 1):
            o += [(pre + f"norm{k}.weight", (E,)), (pre + f"norm{k}.bias", (E,))]
        return o

    for i in range(N):
        pre = f"transformer.encoder.layers.{i}."
        out += attn(pre + "self_attn.") + ffn(pre) + norms(pre, 2)
    out += [("transformer.encoder.norm.weight", (E,)), ("transformer.encoder.norm.bias", (E,))]
    for i in range(N):
        pre = f"transformer.decoder.layers.{i}."
        out += attn(pre + "self_attn.") + attn(pre + "multihead_attn.") + ffn(pre) + norms(pre, 3)
    out += [("transformer.decoder.norm.weight", (E,)), ("transformer.decoder.norm.bias", (E,))]
    out += [("linear.weight", (Vt, E)), ("linear.bias", (Vt,))]
    return out
