"""Thin torch-tensor front-ends of the granular C-ABI kernels (used by the
parity tests and by the RNN host code).  Every function launches on torch's
current stream and fails loudly without a GPU / without the built library."""
import ctypes as C

import torch

from . import _lib
from ._lib import GemmArgs, check, load, ptr, stream_ptr


def make_rng(seed=0, step=0, device="cuda"):
    """Dropout state {seed, step} as int64[2] on the device (read as uint64)."""
    return torch.tensor([seed, step], dtype=torch.int64, device=device)


def gemm(A, B, *, M, N, K, a_kmajor=True, b_kmajor=True, lda=None, ldb=None, out=None, ldc=None,
         bias=None, relu=False, gate=None, gate_scale=1.0, gate_mode=0, drop_p=0.0, drop_site=0, rng=None,
         resid=None, rowsum_a=None, precision=3):
    _lib.require_gpu()
    lda = lda if lda is not None else (K if a_kmajor else M)
    ldb = ldb if ldb is not None else (K if b_kmajor else N)
    ldc = ldc if ldc is not None else N
    if out is None:
        out = torch.empty(M, ldc, dtype=torch.float32, device=A.device)
    a = GemmArgs()
    a.A, a.lda, a.a_kmajor = ptr(A), lda, int(a_kmajor)
    a.B, a.ldb, a.b_kmajor = ptr(B), ldb, int(b_kmajor)
    a.C, a.ldc, a.M, a.N, a.K = ptr(out), ldc, M, N, K
    a.bias, a.relu = ptr(bias), int(relu)
    a.gate, a.ldg, a.gate_scale = ptr(gate), (gate.stride(0) if gate is not None else 0), gate_scale
    a.drop_p, a.drop_site, a.rng = drop_p, drop_site, ptr(rng)
    a.resid, a.ldr = ptr(resid), (resid.stride(0) if resid is not None else 0)
    a.rowsum_a, a.precision, a.gate_mode = ptr(rowsum_a), precision, gate_mode
    check(load().slnlp_gemm(C.byref(a), stream_ptr()), "gemm")
    return out


def pad64(n):
    return (n + 63) // 64 * 64


def split_planes(x, want_lo=True):
    """fp32 [R,C] -> (hi, lo) bf16 planes (int16 storage) zero-padded to multiples of 64."""
    _lib.require_gpu()
    R, Cc = x.shape
    hi = torch.zeros(pad64(R), pad64(Cc), dtype=torch.int16, device=x.device)
    lo = torch.zeros_like(hi) if want_lo else None
    check(load().slnlp_split_planes(ptr(x), x.stride(0), R, Cc, ptr(hi), ptr(lo), hi.stride(0), stream_ptr()), "split_planes")
    return hi, lo


def pad128(n):
    return (n + 127) // 128 * 128


def quant_rows_fp8(x):
    """fp32 [R,K] -> (q uint8 [pad64(R), pad128(K)] OCP e4m3, scale fp32 [R]): q[r] = e4m3(x[r] / scale[r]), scale[r] = max|x[r]| / 448."""
    _lib.require_gpu()
    R, K = x.shape
    q = torch.zeros(pad64(R), pad128(K), dtype=torch.uint8, device=x.device)
    scale = torch.empty(R, dtype=torch.float32, device=x.device)
    check(load().slnlp_quant_rows_fp8(ptr(x), x.stride(0), R, K, ptr(q), q.stride(0), ptr(scale), stream_ptr()), "quant_rows_fp8")
    return q, scale


def gemm_fp8(Aq, Bq, *, M, N, K, col_scale=None, bias=None, relu=False, resid=None, out=None, want_q8=False):
    """C = (A8 B8^T) * col_scale (+ bias, relu, resid) over e4m3 byte planes (both k-major) on the fp8 MFMA (precision 8)."""
    _lib.require_gpu()
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=Aq.device)
    a = GemmArgs()
    a.C, a.ldc, a.M, a.N, a.K = ptr(out), out.stride(0), M, N, K
    a.a_kmajor, a.b_kmajor, a.precision = 1, 1, 8
    a.A_hi, a.lda_p, a.B_hi, a.ldb_p = ptr(Aq), Aq.stride(0), ptr(Bq), Bq.stride(0)
    a.col_scale, a.bias, a.relu = ptr(col_scale), ptr(bias), int(relu)
    a.resid, a.ldr = ptr(resid), (resid.stride(0) if resid is not None else 0)
    cq = None
    if want_q8:
        cq = torch.zeros(pad64(M), pad128(N), dtype=torch.uint8, device=out.device)
        a.C_q8, a.ldc_p = ptr(cq), cq.stride(0)
    check(load().slnlp_gemm(C.byref(a), stream_ptr()), "gemm_fp8")
    return (out, cq) if want_q8 else out


def plane_job(Ap, Bp, *, M, N, K, a_kmajor=True, b_kmajor=True, out=None, precision=3, rowsum_a=None, bias=None,
              relu=False, resid=None):
    """GemmArgs of one pre-split GEMM (for gemm_group); returns (args, out)."""
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=Ap[0].device)
    a = GemmArgs()
    a.C, a.ldc, a.M, a.N, a.K = ptr(out), out.stride(0), M, N, K
    a.a_kmajor, a.b_kmajor, a.precision = int(a_kmajor), int(b_kmajor), precision
    a.A_hi, a.A_lo, a.lda_p = ptr(Ap[0]), ptr(Ap[1]), Ap[0].stride(0)
    a.B_hi, a.B_lo, a.ldb_p = ptr(Bp[0]), ptr(Bp[1]), Bp[0].stride(0)
    a.bias, a.relu, a.rowsum_a = ptr(bias), int(relu), ptr(rowsum_a)
    a.resid, a.ldr = ptr(resid), (resid.stride(0) if resid is not None else 0)
    return a, out


def gemm_group(jobs, split_k=None, scratch=None):
    """ONE launch for up to 4 plane GEMMs (list of GemmArgs); split_k[i] > 1 = deterministic split-K."""
    _lib.require_gpu()
    n = len(jobs)
    arr = (GemmArgs * n)(*jobs)
    sk = (C.c_int32 * n)(*(split_k or [1] * n))
    if scratch is None:
        nbytes = int(load().slnlp_gemm_group_scratch_bytes(arr, sk, n))
        scratch = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    check(load().slnlp_gemm_group(arr, sk, n, ptr(scratch), scratch.numel(), stream_ptr()), "gemm_group")
    return scratch


def gemm_wd_plan(jw, jd):
    """(split, separate, geometry of the wgrad launch, of the dgrad launch) the library picks for this gradient pair."""
    v = [C.c_int32(0) for _ in range(4)]
    check(load().slnlp_gemm_wd_plan(C.byref(jw), C.byref(jd), *[C.byref(x) for x in v]), "gemm_wd_plan")
    return tuple(int(x.value) for x in v)


def gemm_wd(jw, jd, scratch=None):
    """The gradient pair of one dY (plane_job wgrad + dgrad) launched as the training plans launch it (slnlp_gemm_wd)."""
    _lib.require_gpu()
    if scratch is None:
        arr = (GemmArgs * 2)(jw, jd)
        sk = (C.c_int32 * 2)(8, 1)
        scratch = torch.zeros(int(load().slnlp_gemm_group_scratch_bytes(arr, sk, 2)), dtype=torch.uint8, device="cuda")
    check(load().slnlp_gemm_wd(C.byref(jw), C.byref(jd), ptr(scratch), scratch.numel(), stream_ptr()), "gemm_wd")
    return scratch


def gemm_planes(Ap, Bp, *, M, N, K, a_kmajor=True, b_kmajor=True, out=None, precision=3, rowsum_a=None, bias=None,
                relu=False, resid=None, want_planes=False):
    """C = A B^T over pre-split operands Ap = (hi, lo), Bp = (hi, lo) (see split_planes)."""
    _lib.require_gpu()
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=Ap[0].device)
    a = GemmArgs()
    a.C, a.ldc, a.M, a.N, a.K = ptr(out), out.stride(0), M, N, K
    a.a_kmajor, a.b_kmajor, a.precision = int(a_kmajor), int(b_kmajor), precision
    a.A_hi, a.A_lo, a.lda_p = ptr(Ap[0]), ptr(Ap[1]), Ap[0].stride(0)
    a.B_hi, a.B_lo, a.ldb_p = ptr(Bp[0]), ptr(Bp[1]), Bp[0].stride(0)
    a.bias, a.relu, a.rowsum_a = ptr(bias), int(relu), ptr(rowsum_a)
    a.resid, a.ldr = ptr(resid), (resid.stride(0) if resid is not None else 0)
    cp = None
    if want_planes:
        cp = (torch.zeros(pad64(M), pad64(N), dtype=torch.int16, device=out.device),
              torch.zeros(pad64(M), pad64(N), dtype=torch.int16, device=out.device))
        a.C_hi, a.C_lo, a.ldc_p = ptr(cp[0]), ptr(cp[1]), cp[0].stride(0)
    check(load().slnlp_gemm(C.byref(a), stream_ptr()), "gemm_planes")
    return (out, cp) if want_planes else out


def gemm_rows(Ap, W, *, M, N, K, out=None, precision=3, bias=None, relu=0, gate=None, gate_scale=1.0, gate_mode=0, drop_p=0.0,
              drop_site=0, rng=None, drop_head_dim=0, resid=None, want_planes=False):
    """C = A W^T for a few rows (the decoder's products): A as k-major planes Ap = (hi, lo), the weight W [N, K] as fp32 (the kernel
    splits it in registers -- the same bits as split_planes(W)): slnlp_gemm_rows."""
    _lib.require_gpu()
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=Ap[0].device)
    a = GemmArgs()
    a.C, a.ldc, a.M, a.N, a.K = ptr(out), out.stride(0), M, N, K
    a.a_kmajor, a.b_kmajor, a.precision = 1, 1, precision
    a.A_hi, a.A_lo, a.lda_p = ptr(Ap[0]), ptr(Ap[1]), Ap[0].stride(0)
    a.B, a.ldb = ptr(W), W.stride(0)
    a.bias, a.relu = ptr(bias), int(relu)
    a.gate, a.ldg, a.gate_scale, a.gate_mode = ptr(gate), (gate.stride(0) if gate is not None else 0), gate_scale, gate_mode
    a.drop_p, a.drop_site, a.rng, a.drop_head_dim = drop_p, drop_site, ptr(rng), drop_head_dim
    a.resid, a.ldr = ptr(resid), (resid.stride(0) if resid is not None else 0)
    cp = None
    if want_planes:
        cp = (torch.zeros(pad64(M), pad64(N), dtype=torch.int16, device=out.device),
              torch.zeros(pad64(M), pad64(N), dtype=torch.int16, device=out.device))
        a.C_hi, a.C_lo, a.ldc_p = ptr(cp[0]), ptr(cp[1]), cp[0].stride(0)
    check(load().slnlp_gemm_rows(C.byref(a), stream_ptr()), "gemm_rows")
    return (out, cp) if want_planes else out


def gemm_rows_bwd(dYp, W, Xp, *, B, Nout, Kin, precision=3, gate=None, gate_scale=1.0, gate_mode=0, drop_p=0.0, drop_site=0, rng=None,
                  drop_head_dim=0, resid=None, want_planes=False, want_db=True):
    """dX = dY W (+ epilogue), dW = dY^T x, db = colsum(dY) in one launch (slnlp_gemm_rows_bwd): dYp [B, Nout] and Xp [B, Kin] as
    (hi, lo) planes, the weight W [Nout, Kin] as fp32."""
    _lib.require_gpu()
    dev = dYp[0].device
    dX = torch.empty(B, Kin, dtype=torch.float32, device=dev)
    dW = torch.empty(Nout, Kin, dtype=torch.float32, device=dev)
    db = torch.empty(Nout, dtype=torch.float32, device=dev) if want_db else None
    d, w = GemmArgs(), GemmArgs()
    d.C, d.ldc, d.M, d.N, d.K = ptr(dX), dX.stride(0), B, Kin, Nout
    d.a_kmajor, d.b_kmajor, d.precision = 1, 0, precision
    d.A_hi, d.A_lo, d.lda_p = ptr(dYp[0]), ptr(dYp[1]), dYp[0].stride(0)
    d.B, d.ldb = ptr(W), W.stride(0)
    d.gate, d.ldg, d.gate_scale, d.gate_mode = ptr(gate), (gate.stride(0) if gate is not None else 0), gate_scale, gate_mode
    d.drop_p, d.drop_site, d.rng, d.drop_head_dim = drop_p, drop_site, ptr(rng), drop_head_dim
    d.resid, d.ldr = ptr(resid), (resid.stride(0) if resid is not None else 0)
    cp = None
    if want_planes:
        cp = (torch.zeros(pad64(B), pad64(Kin), dtype=torch.int16, device=dev), torch.zeros(pad64(B), pad64(Kin), dtype=torch.int16, device=dev))
        d.C_hi, d.C_lo, d.ldc_p = ptr(cp[0]), ptr(cp[1]), cp[0].stride(0)
    w.C, w.ldc, w.M, w.N, w.K = ptr(dW), dW.stride(0), Nout, Kin, B
    w.a_kmajor, w.b_kmajor, w.precision = 0, 0, precision
    w.A_hi, w.A_lo, w.lda_p = ptr(dYp[0]), ptr(dYp[1]), dYp[0].stride(0)
    w.B_hi, w.B_lo, w.ldb_p = ptr(Xp[0]), ptr(Xp[1]), Xp[0].stride(0)
    w.rowsum_a = ptr(db)
    check(load().slnlp_gemm_rows_bwd(C.byref(d), C.byref(w), stream_ptr()), "gemm_rows_bwd")
    return (dX, dW, db, cp) if want_planes else (dX, dW, db)


def embed_fwd(ids, table, pe, *, B, S, scale=None, drop_p=0.0, drop_site=0, rng=None, nan_idx=-1):
    _lib.require_gpu()
    V, E = table.shape
    out = torch.empty(S * B, E, dtype=torch.float32, device=table.device)
    check(load().slnlp_embed_fwd(ptr(ids), ids.stride(0) if ids.ndim == 2 else 1, B, S, E, V, ptr(table), ptr(pe),
                                 ptr(out), float(E ** 0.5 if scale is None else scale), drop_p, drop_site, ptr(rng),
                                 nan_idx, stream_ptr()), "embed_fwd")
    return out


def embed_bwd(ids, dx, *, B, S, V, scale=None, zero_row=-1, drop_p=0.0, drop_site=0, rng=None):
    _lib.require_gpu()
    E = dx.shape[1]
    dt = torch.empty(V, E, dtype=torch.float32, device=dx.device)
    scratch = torch.empty(int(load().slnlp_embed_bwd_scratch_bytes(B, S, E)), dtype=torch.uint8, device=dx.device)
    check(load().slnlp_embed_bwd(ptr(ids), ids.stride(0) if ids.ndim == 2 else 1, B, S, E, V, ptr(dx), ptr(dt),
                                 float(E ** 0.5 if scale is None else scale), zero_row, drop_p, drop_site, ptr(rng),
                                 ptr(scratch), stream_ptr()), "embed_bwd")
    return dt


def attn_self_fwd(qkv, ids, pad_idx, *, B, S, H, dh, causal=True, drop_p=0.0, drop_site=0, rng=None):
    _lib.require_gpu()
    E = H * dh
    ctx = torch.empty(S * B, E, dtype=torch.float32, device=qkv.device)
    probs = torch.empty(B, H, S, S, dtype=torch.float32, device=qkv.device)
    check(load().slnlp_attn_self_fwd(ptr(qkv), ptr(ids), ids.stride(0) if ids is not None else 0, pad_idx,
                                     int(causal), B, S, H, dh, ptr(ctx), ptr(probs), drop_p, drop_site, ptr(rng),
                                     stream_ptr()), "attn_self_fwd")
    return ctx, probs


def attn_self_bwd(qkv, probs, dctx, *, B, S, H, dh, drop_p=0.0, drop_site=0, rng=None):
    _lib.require_gpu()
    dqkv = torch.empty_like(qkv)
    check(load().slnlp_attn_self_bwd(ptr(qkv), ptr(probs), ptr(dctx), B, S, H, dh, ptr(dqkv), drop_p, drop_site,
                                     ptr(rng), stream_ptr()), "attn_self_bwd")
    return dqkv


def attn_cross_fwd(q, kv, *, B, S, H, dh, drop_p=0.0, drop_site=0, rng=None):
    _lib.require_gpu()
    E = H * dh
    ctx = torch.empty(B, E, dtype=torch.float32, device=q.device)
    probs = torch.empty(B, H, S, dtype=torch.float32, device=q.device)
    check(load().slnlp_attn_cross_fwd(ptr(q), ptr(kv), kv.stride(0), B, S, H, dh, ptr(ctx), ptr(probs), drop_p,
                                      drop_site, ptr(rng), stream_ptr()), "attn_cross_fwd")
    return ctx, probs


def attn_cross_bwd(q, kv, probs, dctx, *, B, S, H, dh, drop_p=0.0, drop_site=0, rng=None):
    _lib.require_gpu()
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    check(load().slnlp_attn_cross_bwd(ptr(q), ptr(kv), kv.stride(0), ptr(probs), ptr(dctx), B, S, H, dh, ptr(dq),
                                      ptr(dkv), dkv.stride(0), drop_p, drop_site, ptr(rng), stream_ptr()),
          "attn_cross_bwd")
    return dq, dkv


def layernorm_fwd(x, gamma, beta, eps=1e-5):
    _lib.require_gpu()
    rows, E = x.shape
    y = torch.empty_like(x)
    stats = torch.empty(rows, 2, dtype=torch.float32, device=x.device)
    check(load().slnlp_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), rows, E, eps, ptr(y), ptr(stats), stream_ptr()),
          "layernorm_fwd")
    return y, stats


def layernorm_bwd(dy, x, gamma, stats, *, add_to_dx=None, want_drop=False, drop_p=0.0, drop_site=0, rng=None):
    """-> (dx, dx_drop | None, dgamma, dbeta)"""
    _lib.require_gpu()
    rows, E = x.shape
    dx = torch.empty_like(x)
    dxd = torch.empty_like(x) if want_drop else None
    partial = torch.empty(1024, 2, E, dtype=torch.float32, device=x.device)     # SLNLP_LN_MAX_PARTIALS chunks
    nblk = C.c_int32(0)
    check(load().slnlp_layernorm_bwd(ptr(dy), ptr(x), ptr(gamma), ptr(stats), rows, E, ptr(add_to_dx), ptr(dx),
                                     ptr(dxd), drop_p, drop_site, ptr(rng), ptr(partial), C.byref(nblk),
                                     stream_ptr()), "layernorm_bwd")
    dg = torch.empty(E, dtype=torch.float32, device=x.device)
    db = torch.empty(E, dtype=torch.float32, device=x.device)
    ent = _lib.LnReduceEntry(ptr(partial), ptr(dg), ptr(db), nblk.value, E)
    table = torch.frombuffer(bytearray(bytes(ent)), dtype=torch.uint8).to(x.device)
    check(load().slnlp_ln_param_reduce(ptr(table), 1, E, stream_ptr()), "ln_param_reduce")
    torch.cuda.current_stream().synchronize()  # `table` must outlive the launch
    return dx, dxd, dg, db


def lsm_nll(logits, y, ignore_index, *, want_grad=True):
    """-> (logp [B,V], loss [1], dlogits [B,V] | None)"""
    _lib.require_gpu()
    B, V = logits.shape
    logp = torch.empty(B, V, dtype=torch.float32, device=logits.device)
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    dl = torch.empty(B, V, dtype=torch.float32, device=logits.device) if want_grad else None
    rows = torch.empty(B, dtype=torch.float32, device=logits.device)
    check(load().slnlp_lsm_nll(ptr(logits), logits.stride(0), ptr(y), B, V, ignore_index, ptr(logp), ptr(loss),
                               ptr(dl), V, ptr(rows), stream_ptr()), "lsm_nll")
    return logp, loss, dl


def lsm_bwd(logp, dlogp):
    _lib.require_gpu()
    B, V = logp.shape
    out = torch.empty_like(logp)
    check(load().slnlp_lsm_bwd(ptr(logp), ptr(dlogp), B, V, ptr(out), V, stream_ptr()), "lsm_bwd")
    return out


def clip_sgd_step(params, grads, buf, lr_dev, *, momentum=0.9, max_norm=0.5, rng=None):
    """In-place update of flat fp32 arenas; returns the pre-clip norm tensor [1]."""
    _lib.require_gpu()
    partials = torch.empty(1024, dtype=torch.float32, device=params.device)
    norm = torch.empty(1, dtype=torch.float32, device=params.device)
    check(load().slnlp_clip_sgd_step(ptr(params), ptr(grads), ptr(buf), params.numel(), ptr(lr_dev), momentum,
                                     max_norm, ptr(partials), ptr(norm), ptr(rng), stream_ptr()), "clip_sgd_step")
    return norm


def clip_adam_step(params, grads, exp_avg, exp_avg_sq, lr_dev, step_count, *, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.5):
    """In-place clip + Adam on flat fp32 arenas; ``step_count`` [1] float (device) is advanced.  Returns the pre-clip norm [1]."""
    _lib.require_gpu()
    partials = torch.empty(1024, dtype=torch.float32, device=params.device)
    norm = torch.empty(1, dtype=torch.float32, device=params.device)
    check(load().slnlp_clip_adam_step(ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), params.numel(), ptr(lr_dev), betas[0], betas[1],
                                      eps, weight_decay, max_norm, ptr(partials), ptr(norm), ptr(step_count), stream_ptr()), "clip_adam_step")
    return norm


def dropout_mask(R, C_, p, site, rng):
    _lib.require_gpu()
    out = torch.empty(R, C_, dtype=torch.float32, device=rng.device)
    check(load().slnlp_dropout_mask(ptr(out), R, C_, p, site, ptr(rng), stream_ptr()), "dropout_mask")
    return out
