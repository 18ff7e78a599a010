"""GPU parity of each HIP kernel, called through the C ABI (slnlp.ops ->
libslnlp.so), against fp64 / plain-torch fp32 references on the same seeded
inputs.  Tolerances: precision 3 (split-bf16 MFMA) 5e-5 of the tensor scale;
precision 1 (single bf16 pass) 2e-2; exact-fp32 kernels 2e-5."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from slnlp import ops as o
    return o


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


TOL = {3: 5e-5, 1: 2e-2}

GEMM_SHAPES = [(2400, 512, 512), (50, 202, 512), (50, 512, 202), (64, 64, 32), (7, 5, 3), (130, 70, 100),
               (2400, 1536, 512), (202, 512, 50)]


@pytest.mark.parametrize("prec", [3, 1])
@pytest.mark.parametrize("layout", ["fwd", "dgrad", "wgrad"])
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_layouts(ops, layout, M, N, K, prec):
    # logical: C[m,n] = sum_k A(m,k) B(n,k)
    Al, Bl = rnd(M, K, seed=1), rnd(N, K, seed=2)
    ref = (Al.double() @ Bl.double().T)
    a_k = layout in ("fwd", "dgrad")
    b_k = layout == "fwd"
    A = (Al if a_k else Al.T.contiguous()).cuda()
    B = (Bl if b_k else Bl.T.contiguous()).cuda()
    out = ops.gemm(A, B, M=M, N=N, K=K, a_kmajor=a_k, b_kmajor=b_k, precision=prec)
    assert rel(out, ref) < TOL[prec] * max(1.0, math.sqrt(K / 64))


def test_gemm_thread_groups_do_not_change_the_bits(ops):
    """The fp32-operand GEMM defines its K sum as (tiles [0, T)) + (tiles [T, ktiles)), T = ceil(ktiles / 2); one thread group walks
    both halves (KS = 1: merged lockstep launches) or two groups take one each (KS = 2: a solo fit's latency-bound launches) --
    slnlp_set_gemm_ks -- and every output, the fused bias-gradient row sums included, must come out bit for bit the same: all
    layouts, narrow and wide tiles, one / odd / even numbers of K tiles, ragged edges, the epilogue chain."""
    from slnlp._lib import load, check
    rng = ops.make_rng(seed=5, step=2)
    def run():
        outs = []
        for (M, N, K) in [(50, 512, 512), (50, 200, 64), (50, 96, 448), (300, 200, 320), (130, 72, 1000), (64, 64, 128)]:
            A, B, bias, R = rnd(M, K, seed=1).cuda(), rnd(N, K, seed=2).cuda(), rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
            outs.append(ops.gemm(A, B, M=M, N=N, K=K, bias=bias, relu=True, drop_p=0.2, drop_site=4, rng=rng, resid=R).clone())
            outs.append(ops.gemm(A, B.T.contiguous(), M=M, N=N, K=K, b_kmajor=False, resid=R).clone())                       # dgrad layout
            rs = torch.empty(M, device="cuda")
            outs.append(ops.gemm(A.T.contiguous(), B.T.contiguous(), M=M, N=N, K=K, a_kmajor=False, b_kmajor=False, rowsum_a=rs).clone())
            outs.append(rs.clone())
        torch.cuda.synchronize()
        return outs
    try:
        check(load().slnlp_set_gemm_ks(1), "set_gemm_ks")
        one = run()
        check(load().slnlp_set_gemm_ks(2), "set_gemm_ks")
        two = run()
    finally:
        load().slnlp_set_gemm_ks(0)
    for i, (a, b) in enumerate(zip(one, two)):
        assert torch.equal(a, b), f"output {i} differs between one and two thread groups"
    A, B = rnd(50, 512, seed=1), rnd(512, 512, seed=2)
    assert rel(one[0][:1] * 0 + ops.gemm(A.cuda(), B.cuda(), M=50, N=512, K=512)[:1], (A.double() @ B.double().T)[:1]) < 1e-4


def test_gemm_padded_ld_and_views(ops):
    # generator backward shapes: dlogits [B, Vp=204] with V=202 valid columns, garbage (NaN) in the pad
    Bt, V, Vp, E = 50, 202, 204, 512
    dl = torch.full((Bt, Vp), float("nan"))
    dl[:, :V] = rnd(Bt, V, seed=3)
    W, x = rnd(V, E, seed=4), rnd(Bt, E, seed=5)
    dlc, Wc, xc = dl.cuda(), W.cuda(), x.cuda()
    dx = ops.gemm(dlc, Wc, M=Bt, N=E, K=V, a_kmajor=True, b_kmajor=False, lda=Vp, ldb=E)
    assert rel(dx, dl[:, :V].double() @ W.double()) < 1e-4
    db = torch.empty(V, device="cuda")
    dW = ops.gemm(dlc, xc, M=V, N=E, K=Bt, a_kmajor=False, b_kmajor=False, lda=Vp, ldb=E, rowsum_a=db)
    assert rel(dW, dl[:, :V].double().T @ x.double()) < 1e-4
    assert rel(db, dl[:, :V].double().sum(0)) < 1e-4
    # weight-row slice as B operand (decoder v-projection: rows 2E..3E of in_proj)
    Win, t = rnd(3 * E, E, seed=6).cuda(), rnd(Bt, E, seed=7).cuda()
    v = ops.gemm(t, Win[2 * E:], M=Bt, N=E, K=E)
    assert rel(v, t.double().cpu() @ Win[2 * E:].double().cpu().T) < 1e-4


def test_gemm_epilogues(ops):
    M, N, K = 300, 200, 96
    A, B, bias, R, G = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, seed=5)
    base = A.double() @ B.double().T
    Ac, Bc = A.cuda(), B.cuda()
    out = ops.gemm(Ac, Bc, M=M, N=N, K=K, bias=bias.cuda(), relu=True, resid=R.cuda())
    assert rel(out, torch.relu(base + bias.double()) + R.double()) < 1e-4
    out = ops.gemm(Ac, Bc, M=M, N=N, K=K, gate=G.cuda(), gate_scale=1.25)
    assert rel(out, base * (G.double() > 0) * 1.25) < 1e-4
    # in-place residual (resid aliases C), as the dmem accumulation uses it
    Cbuf = R.cuda().clone()
    ops.gemm(Ac, Bc, M=M, N=N, K=K, out=Cbuf, resid=Cbuf)
    assert rel(Cbuf, base + R.double()) < 1e-4
    # dropout epilogue reproduces the mask the dump kernel reports
    rng = ops.make_rng(seed=1234, step=7)
    p = 0.3
    mask = ops.dropout_mask(M, N, p, 5, rng).cpu().double()
    out = ops.gemm(Ac, Bc, M=M, N=N, K=K, bias=bias.cuda(), drop_p=p, drop_site=5, rng=rng, resid=R.cuda())
    assert rel(out, (base + bias.double()) * mask / (1 - p) + R.double()) < 1e-4
    assert abs(float(mask.mean()) - (1 - p)) < 0.01


def test_dropout_mask_statistics(ops):
    rng = ops.make_rng(seed=99, step=0)
    m = ops.dropout_mask(2048, 512, 0.1, 3, rng)
    assert abs(float(m.mean()) - 0.9) < 2e-3
    m2 = ops.dropout_mask(2048, 512, 0.1, 4, rng)          # different site -> different mask
    assert float((m != m2).float().mean()) > 0.1
    rng2 = ops.make_rng(seed=99, step=1)                   # next step -> different mask
    assert float((m != ops.dropout_mask(2048, 512, 0.1, 3, rng2)).float().mean()) > 0.1
    assert torch.equal(m, ops.dropout_mask(2048, 512, 0.1, 3, rng))   # deterministic


@pytest.mark.parametrize("R,C,p,site,seed,step", [(64, 64, 0.1, 3, 99, 0), (50, 202, 0.5, 17, 1234, 7), (7, 48, 0.25, 65, 2**40 + 5, 2**33),
                                                   (300, 33, 0.1, 1, 0, 0), (129, 160, 0.9, 200, 3, 1)])
def test_dropout_mask_is_the_threefry_restatement_bit_for_bit(ops, R, C, p, site, seed, step):
    """slnlp_dropout_mask == tests/threefry_ref.keep_mask (Threefry4x32-12 pinned to the published known-answer vectors by
    tests/test_dropout_cpu.py): counter / key layout, 16-bit lots, the (c, c + 16) column pairing and the threshold."""
    import threefry_ref as tf
    rng = ops.make_rng(seed=seed, step=step)
    m = ops.dropout_mask(R, C, p, site, rng).cpu().numpy()
    assert np.array_equal(m, tf.keep_mask(R, C, p, site, seed, step))


def _mha_ref(qkv, ids, pad, B, S, H, dh, causal, mask=None, p=0.0):
    E = H * dh
    x = qkv.view(S, B, 3, H, dh)
    q, k, v = [x[:, :, i].permute(1, 2, 0, 3) for i in range(3)]           # [B,H,S,dh]
    sc = q @ k.transpose(-1, -2) / math.sqrt(dh)
    blocked = torch.zeros(B, 1, S, S, dtype=torch.bool)
    if causal:
        blocked = blocked | torch.triu(torch.ones(S, S, dtype=torch.bool), 1)
    if ids is not None:
        blocked = blocked | (ids == pad).view(B, 1, 1, S)
    pr = torch.softmax(sc.masked_fill(blocked, float("-inf")), -1)
    pd = pr if mask is None else pr * mask / (1 - p)
    ctx = (pd @ v).permute(2, 0, 1, 3).reshape(S * B, E)
    return ctx, pr


@pytest.mark.parametrize("B,S,H,dh", [(50, 48, 8, 64), (4, 12, 4, 8), (50, 48, 4, 32), (9, 64, 4, 256), (50, 48, 8, 16),
                                      (3, 37, 2, 128), (5, 37, 3, 32), (6, 20, 2, 64), (2, 64, 2, 48)])
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_attn_self(ops, B, S, H, dh, p):
    E = H * dh
    qkv = rnd(S * B, 3 * E, seed=1).double().requires_grad_(True)
    lengths = torch.randint(max(1, S // 6), S + 1, (B,), generator=torch.Generator().manual_seed(3))
    ids = torch.full((B, S), 5, dtype=torch.long)
    ids[torch.arange(S)[None, :] >= lengths[:, None]] = 1
    rng = ops.make_rng(seed=5, step=2)
    mask = ops.dropout_mask(B * H * S, S, p, 17, rng).cpu().double().view(B, H, S, S) if p > 0 else None
    ctx_ref, pr_ref = _mha_ref(qkv, ids, 1, B, S, H, dh, True, mask, p)
    dctx = rnd(S * B, E, seed=2)
    ctx_ref.backward(dctx.double())
    qc = qkv.detach().float().cuda()
    ctx, probs = ops.attn_self_fwd(qc, ids.cuda(), 1, B=B, S=S, H=H, dh=dh, causal=True, drop_p=p, drop_site=17, rng=rng)
    # split-bf16 MFMA (3 passes, ~2^-16 per product); head dims above 64 in chunks of 64
    e_p, e_c = rel(probs, pr_ref), rel(ctx, ctx_ref)
    dqkv = ops.attn_self_bwd(qc, probs, dctx.cuda(), B=B, S=S, H=H, dh=dh, drop_p=p, drop_site=17, rng=rng)
    e_g = rel(dqkv, qkv.grad)
    print(f"attn_self B{B} S{S} H{H} dh{dh} p{p}: probs {e_p:.2e} ctx {e_c:.2e} dqkv {e_g:.2e}")
    assert e_p < 5e-5 and e_c < 5e-5 and e_g < 1e-4


def test_attn_self_fully_masked_row_is_nan(ops):
    # a sequence whose first key is <pad>: query 0 has no visible key -> NaN like torch
    B, S, H, dh = 2, 8, 2, 8
    qkv = rnd(S * B, 3 * H * dh, seed=1).cuda()
    ids = torch.full((B, S), 5, dtype=torch.long)
    ids[1, 0] = 1
    ctx, probs = ops.attn_self_fwd(qkv, ids.cuda(), 1, B=B, S=S, H=H, dh=dh)
    assert torch.isnan(probs[1, :, 0]).all() and not torch.isnan(probs[0]).any()


@pytest.mark.parametrize("B,S,H,dh", [(50, 48, 8, 64), (4, 12, 4, 8), (7, 64, 4, 256), (50, 48, 8, 16)])
@pytest.mark.parametrize("p", [0.0, 0.25])
def test_attn_cross(ops, B, S, H, dh, p):
    E = H * dh
    q = rnd(B, E, seed=1).double().requires_grad_(True)
    kv = rnd(S * B, 2 * E, seed=2).double().requires_grad_(True)
    rng = ops.make_rng(seed=8, step=1)
    mask = ops.dropout_mask(B * H, S, p, 9, rng).cpu().double().view(B, H, 1, S) if p > 0 else None
    qh = q.view(B, H, 1, dh)
    k = kv[:, :E].reshape(S, B, H, dh).permute(1, 2, 0, 3)
    v = kv[:, E:].reshape(S, B, H, dh).permute(1, 2, 0, 3)
    pr = torch.softmax(qh @ k.transpose(-1, -2) / math.sqrt(dh), -1)       # [B,H,1,S]
    pd = pr if mask is None else pr * mask / (1 - p)
    ctx_ref = (pd @ v).reshape(B, E)
    dctx = rnd(B, E, seed=3)
    ctx_ref.backward(dctx.double())
    qc, kvc = q.detach().float().cuda(), kv.detach().float().cuda()
    ctx, probs = ops.attn_cross_fwd(qc, kvc, B=B, S=S, H=H, dh=dh, drop_p=p, drop_site=9, rng=rng)
    assert rel(probs.view(B, H, 1, S), pr) < 2e-5
    assert rel(ctx, ctx_ref) < 2e-5
    dq, dkv = ops.attn_cross_bwd(qc, kvc, probs, dctx.cuda(), B=B, S=S, H=H, dh=dh, drop_p=p, drop_site=9, rng=rng)
    assert rel(dq, q.grad) < 5e-5
    assert rel(dkv, kv.grad) < 5e-5


@pytest.mark.parametrize("rows,E", [(2400, 512), (50, 128), (4, 32), (2400, 1024), (257, 260), (3203, 512), (16384, 1024), (1030, 200)])
def test_layernorm(ops, rows, E):
    x = rnd(rows, E, seed=1, scale=3.0).double().requires_grad_(True)
    g = (1 + 0.1 * rnd(E, seed=2)).double().requires_grad_(True)
    b = (0.1 * rnd(E, seed=3)).double().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(x, (E,), g, b, 1e-5)
    dy, add = rnd(rows, E, seed=4), rnd(rows, E, seed=5)
    y_ref.backward(dy.double())
    y, stats = ops.layernorm_fwd(x.detach().float().cuda(), g.detach().float().cuda(), b.detach().float().cuda())
    assert rel(y, y_ref) < 2e-5
    rng = ops.make_rng(seed=3, step=3)
    p = 0.2
    dx, dxd, dg, db = ops.layernorm_bwd(dy.cuda(), x.detach().float().cuda(), g.detach().float().cuda(), stats,
                                        add_to_dx=add.cuda(), want_drop=True, drop_p=p, drop_site=11, rng=rng)
    assert rel(dx, x.grad + add.double()) < 5e-5
    assert rel(dg, g.grad) < 5e-5 and rel(db, b.grad) < 5e-5
    mask = ops.dropout_mask(rows, E, p, 11, rng).cpu().double()
    assert rel(dxd, (x.grad + add.double()) * mask / (1 - p)) < 5e-5


def test_embed(ops):
    B, S, E, V = 50, 48, 128, 300
    from slnlp.tf_engine import positional_table
    table = rnd(V, E, seed=1).double().requires_grad_(True)
    pe = positional_table(64, E)
    ids = torch.randint(0, 40, (B, S), generator=torch.Generator().manual_seed(2))   # many duplicates
    ref = table[ids.T] * math.sqrt(E) + pe[:S].double().unsqueeze(1)                 # [S,B,E]
    dx = rnd(S * B, E, seed=3)
    ref.reshape(S * B, E).backward(dx.double())
    out = ops.embed_fwd(ids.cuda(), table.detach().float().cuda(), pe.cuda(), B=B, S=S)
    assert rel(out, ref.reshape(S * B, E)) < 1e-6
    dt = ops.embed_bwd(ids.cuda(), dx.cuda(), B=B, S=S, V=V)
    assert rel(dt, table.grad) < 1e-5
    assert float(dt[40:].abs().max()) == 0.0                # untouched rows are exactly zero
    dt2 = ops.embed_bwd(ids.cuda(), dx.cuda(), B=B, S=S, V=V)
    assert torch.equal(dt, dt2)                             # deterministic (no atomics)
    # dropout path + <pad>-target NaN row
    rng = ops.make_rng(seed=4, step=0)
    p = 0.1
    mask = ops.dropout_mask(S * B, E, p, 1, rng).cpu().double()
    out = ops.embed_fwd(ids.cuda(), table.detach().float().cuda(), pe.cuda(), B=B, S=S, drop_p=p, drop_site=1, rng=rng)
    assert rel(out, ref.reshape(S * B, E).detach() * mask / (1 - p)) < 1e-6
    y = torch.tensor([5, 1, 7])
    t0 = ops.embed_fwd(y.cuda(), table.detach().float().cuda(), pe.cuda(), B=3, S=1, nan_idx=1)
    assert torch.isnan(t0[1]).all() and not torch.isnan(t0[[0, 2]]).any()


@pytest.mark.parametrize("B,S,V,E,hi", [(50, 1, 202, 512, 30), (64, 1, 40, 128, 40), (7, 3, 300, 64, 5), (1, 1, 9, 256, 9), (33, 1, 20, 1024, 3)])
def test_embed_bwd_single_launch_for_small_batches(ops, B, S, V, E, hi):
    """At most 64 tokens (the target side: one token per sequence) take the one-launch scatter-add: against autograd in fp64,
    rows nobody indexes exactly zero (over a NaN-filled table), duplicates summed, the zero_row cleared, the dropout path equal
    to the masked reference, run-to-run identical."""
    g = torch.Generator().manual_seed(B + V)
    ids = torch.randint(0, hi, (B, S), generator=g)
    table = rnd(V, E, seed=1).double().requires_grad_(True)
    dx = rnd(S * B, E, seed=3)
    (table[ids.T] * math.sqrt(E)).reshape(S * B, E).backward(dx.double())
    dt = ops.embed_bwd(ids.cuda(), dx.cuda(), B=B, S=S, V=V)
    assert rel(dt, table.grad) < 1e-5
    unused = torch.ones(V, dtype=torch.bool)
    unused[ids.flatten()] = False
    assert float(dt.cpu()[unused].abs().max() if unused.any() else 0.0) == 0.0
    assert torch.equal(dt, ops.embed_bwd(ids.cuda(), dx.cuda(), B=B, S=S, V=V))
    z = int(ids[0, 0])
    dz = ops.embed_bwd(ids.cuda(), dx.cuda(), B=B, S=S, V=V, zero_row=z)
    assert float(dz[z].abs().max()) == 0.0
    rng = ops.make_rng(seed=4, step=0)
    p = 0.2
    mask = ops.dropout_mask(S * B, E, p, 7, rng).cpu().double()
    table.grad = None
    (table[ids.T] * math.sqrt(E)).reshape(S * B, E).backward(dx.double() * mask / (1 - p))
    dd = ops.embed_bwd(ids.cuda(), dx.cuda(), B=B, S=S, V=V, drop_p=p, drop_site=7, rng=rng)
    assert rel(dd, table.grad) < 1e-5


def test_lsm_nll(ops):
    from oracle import train_ref
    B, V = 50, 202
    logits = rnd(B, V, seed=1, scale=2.0).double().requires_grad_(True)
    y = torch.randint(2, V, (B,), generator=torch.Generator().manual_seed(2))
    y[3] = 1
    y[17] = 1                                                # ignored targets (== pad)
    logp_ref = torch.log_softmax(logits, -1)
    loss_ref = train_ref.cross_entropy_on_logprobs(logp_ref, y, 1)
    loss_ref.backward()
    logp, loss, dl = ops.lsm_nll(logits.detach().float().cuda(), y.cuda(), 1)
    assert rel(logp, logp_ref) < 1e-6
    assert abs(float(loss) - float(loss_ref)) < 1e-6 * abs(float(loss_ref))
    assert rel(dl, logits.grad) < 1e-5
    # external-criterion path: d logits from d logp
    dlogp = rnd(B, V, seed=5)
    lp = logp_ref.detach().clone().requires_grad_(True)
    lg = logits.detach().clone().requires_grad_(True)
    torch.log_softmax(lg, -1).backward(dlogp.double())
    assert rel(ops.lsm_bwd(logp, dlogp.cuda()), lg.grad) < 1e-5


def test_clip_sgd(ops):
    from oracle import train_ref
    n = 1 << 20
    p0, g, buf0 = rnd(n, seed=1), rnd(n, seed=2, scale=1e-3), rnd(n, seed=3, scale=1e-3)
    for max_norm in (0.5, 1e9, 0.0):
        P, G, Bf = p0.cuda().clone(), g.cuda().clone(), buf0.cuda().clone()
        lr = torch.tensor([0.01], device="cuda")
        rng = ops.make_rng(1, 41)
        norm = ops.clip_sgd_step(P, G, Bf, lr, momentum=0.9, max_norm=max_norm, rng=rng)
        gg = [g.clone()]
        total = torch.sqrt((g.double() ** 2).sum()).float()
        if max_norm > 0:
            train_ref.clip_grad_norm(gg, max_norm)
        params, bufs = {"w": p0.clone()}, {"w": buf0.clone()}
        train_ref.sgd_momentum_step(params, {"w": gg[0]}, bufs, 0.01, 0.9)
        assert abs(float(norm) - float(total)) < 1e-5 * float(total)
        assert rel(P, params["w"]) < 1e-6 and rel(Bf, bufs["w"]) < 1e-6
        assert int(rng[1]) == 42                            # dropout step counter advanced


def _e4m3_decode(q):
    """OCP e4m3fn bytes -> float64 (no infinities; 0x7F / 0xFF are NaN)."""
    q = q.to(torch.int32)
    sign = torch.where((q & 0x80) != 0, -1.0, 1.0).double()
    e, m = (q >> 3) & 0xF, (q & 7).double()
    val = torch.where(e == 0, m / 8.0 * 2.0 ** -6, (1.0 + m / 8.0) * torch.pow(torch.tensor(2.0, dtype=torch.float64), (e - 7).double()))
    return sign * val


@pytest.mark.parametrize("M,N,K", [(64, 64, 128), (2400, 512, 512), (130, 192, 1024), (50, 64, 256)])
def test_fp8_quantiser_and_gemm(ops, M, N, K):
    """precision 8: (1) the row quantiser is round-to-nearest e4m3 of x / scale with scale = max|row| / 448; (2) the fp8 MFMA
    GEMM equals the fp64 product of the DEQUANTISED operands to ~3e-5 of the tensor max (measured 2.7e-5: the fp8 matrix pipe
    aligns the 32 products of an instruction to a common exponent before adding them, coarser than an fp32 FMA chain);
    (3) against the unquantised product the error is the format's: a few percent, reported not asserted tight."""
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05)
    bias = rnd(N, seed=3)
    Aq, sa = ops.quant_rows_fp8(A.cuda())
    Wq, sw = ops.quant_rows_fp8(W.cuda())
    Ad, Wd = _e4m3_decode(Aq[:M, :K].cpu()), _e4m3_decode(Wq[:N, :K].cpu())
    assert torch.allclose(sa.cpu().double(), A.abs().amax(1).double() / 448.0, rtol=1e-6)
    assert float((Ad * sa.cpu().double()[:, None] - A.double()).abs().max() / A.abs().max()) < 2 ** -4      # 3 mantissa bits
    assert float(Ad.abs().max()) <= 448.0 and not torch.isnan(Ad).any()
    got = ops.gemm_fp8(Aq, Wq, M=M, N=N, K=K, col_scale=sw, bias=bias.cuda(), relu=True).cpu().double()
    want = torch.relu((Ad @ Wd.T) * sw.cpu().double()[None, :] + bias.double())          # A rows are NOT rescaled (activation scale 1)
    assert rel(got, want) < 2e-4
    exact = torch.relu(((A.double() / sa.cpu().double()[:, None]) @ W.double().T) + bias.double())
    print(f"[{M}x{N}x{K}] fp8 vs unquantised product: rel err {rel(got, exact):.3e}")
    assert rel(got, exact) < 0.1


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_clip_adam_vs_torch(ops, wd):
    """Fused clip + Adam against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam on the same gradients, 5 steps (bias
    corrections from the device-side step count)."""
    n = 1 << 18
    p0 = rnd(n, seed=1)
    P, M1, M2 = p0.cuda().clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    lr, cnt = torch.tensor([3e-3], device="cuda"), torch.zeros(1, device="cuda")
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.Adam([ref], lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    for step in range(5):
        g = rnd(n, seed=10 + step, scale=1e-2 if step % 2 else 1e-4)
        norm = ops.clip_adam_step(P, g.cuda(), M1, M2, lr, cnt, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd, max_norm=0.5)
        ref.grad = g.clone().double()
        total = torch.nn.utils.clip_grad_norm_([ref], 0.5)
        opt.step()
        assert abs(float(norm) - float(total)) < 1e-5 * float(total)
        # |w| < 8: one fp32 ulp (9.5e-7) per step; with weight decay an element whose g + wd w nearly cancels has an
        # ill-conditioned m/(sqrt(v)+eps), so the fp32 kernel and the fp64 torch run may differ there by a fraction of lr
        assert float((P.cpu().double() - ref.detach()).abs().max()) < 1e-6 * (step + 1), step
    assert float(cnt) == 5.0
    st = opt.state[ref]
    assert rel(M1, st["exp_avg"].float()) < 1e-5 and rel(M2, st["exp_avg_sq"].float()) < 1e-5


@pytest.mark.parametrize("prec", [3, 1])
@pytest.mark.parametrize("layout", ["fwd", "dgrad", "wgrad"])
@pytest.mark.parametrize("M,N,K", [(2400, 512, 512), (50, 202, 512), (50, 512, 202), (64, 64, 64), (130, 70, 100),
                                   (512, 512, 2400), (202, 512, 50),
                                   (1024, 512, 320), (512, 1024, 1000), (2048, 256, 64)])
def test_gemm_planes_layouts(ops, layout, M, N, K, prec):
    """LDS-DMA GEMM over pre-split, zero-padded bf16 planes (gemm_planes.hip) vs fp64."""
    Al, Bl = rnd(M, K, seed=1), rnd(N, K, seed=2)
    ref = Al.double() @ Bl.double().T
    a_k = layout in ("fwd", "dgrad")
    b_k = layout == "fwd"
    Ap = ops.split_planes((Al if a_k else Al.T.contiguous()).cuda() if (K % 4 == 0 or not a_k) and (M % 4 == 0 or a_k) else None) \
        if False else None
    # split_planes needs column counts that are multiples of 4: pad the fp32 source like a producer would
    def planes(x):
        R, Cc = x.shape
        xp = torch.zeros(R, (Cc + 3) // 4 * 4)
        xp[:, :Cc] = x
        return ops.split_planes(xp.cuda())
    Ap = planes(Al if a_k else Al.T.contiguous())
    Bp = planes(Bl if b_k else Bl.T.contiguous())
    rs = torch.empty(M, device="cuda") if layout == "wgrad" else None
    out = ops.gemm_planes(Ap, Bp, M=M, N=N, K=K, a_kmajor=a_k, b_kmajor=b_k, precision=prec, rowsum_a=rs)
    assert rel(out, ref) < TOL[prec] * max(1.0, math.sqrt(K / 64))
    if rs is not None:
        assert rel(rs, Al.double().sum(1)) < (1e-4 if prec == 3 else 2e-2)


def test_gemm_planes_epilogue_and_output_planes(ops):
    M, N, K = 300, 200, 128
    A, B, bias, R = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    ref = torch.relu(A.double() @ B.double().T + bias.double()) + R.double()
    out, (hi, lo) = ops.gemm_planes(ops.split_planes(A.cuda()), ops.split_planes(B.cuda()), M=M, N=N, K=K,
                                    bias=bias.cuda(), relu=True, resid=R.cuda(), want_planes=True)
    assert rel(out, ref) < 1e-4
    back = (hi.view(torch.bfloat16).float() + lo.view(torch.bfloat16).float())[:M, :N]
    assert rel(back, ref) < 1e-4                                          # emitted planes reconstruct the result
    assert float(hi[M:].abs().max()) == 0 and float(hi[:, N:].abs().max()) == 0   # padding untouched


@pytest.mark.parametrize("prec", [3, 1])
@pytest.mark.parametrize("M,N,K", [(50, 512, 512), (50, 202, 512), (7, 64, 64), (130, 96, 192), (64, 512, 1024), (256, 1024, 512),
                                   (50, 512, 256), (33, 48, 128), (50, 200, 320)])
def test_gemm_rows_vs_fp64(ops, M, N, K, prec):
    """The decoder's B-row products, register-direct (gemm_rows.hip; x as k-major planes, the weight as fp32 split in registers):
    odd / even numbers of 64-k tiles, a single tile (seven waves idle), more than one block of 64 rows, N off the 16-column tile."""
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    out = ops.gemm_rows(ops.split_planes(A.cuda()), B.cuda(), M=M, N=N, K=K, precision=prec)
    assert rel(out, A.double() @ B.double().T) < TOL[prec] * max(1.0, math.sqrt(K / 64))


def test_gemm_rows_tiles_return_the_same_bits(ops):
    """The B-row kernel defines its K sum per 64-k tile (partials added in tile order), so the three workgroup tiles -- 16 x 16 for a
    solo fit's launch, 64 x 16 / 64 x 32 for merged lockstep launches (slnlp_set_rows_tile) -- must agree bit for bit: that is what
    lets a fit in lockstep keep the bits of its solo run."""
    from slnlp._lib import load, check
    rng = ops.make_rng(seed=5, step=2)
    def run():
        outs = []
        for (M, N, K) in [(50, 512, 512), (50, 202, 512), (7, 64, 64), (130, 96, 192), (64, 512, 1024), (50, 200, 320)]:
            A, B, bias, R = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
            Ap, Bp = ops.split_planes(A.cuda()), B.cuda()
            out, (hi, lo) = ops.gemm_rows(Ap, Bp, M=M, N=N, K=K, bias=bias, relu=1, drop_p=0.2, drop_site=4, rng=rng, resid=R, want_planes=True)
            outs += [out.clone(), hi.clone(), lo.clone()]
        torch.cuda.synchronize()
        return outs
    try:
        res = []
        for tile in (0, 1, 2):
            check(load().slnlp_set_rows_tile(tile), "set_rows_tile")
            res.append(run())
    finally:
        load().slnlp_set_rows_tile(-1)
    for other in res[1:]:
        for i, (a, b) in enumerate(zip(res[0], other)):
            assert torch.equal(a, b), f"output {i} differs between workgroup tiles"


@pytest.mark.parametrize("prec", [3, 1])
@pytest.mark.parametrize("B,Nout,Kin", [(50, 512, 512), (50, 512, 256), (7, 64, 64), (130, 192, 96), (64, 1024, 512), (50, 128, 200)])
def test_gemm_rows_bwd_vs_fp64(ops, B, Nout, Kin, prec):
    """dX = dY W, dW = dY^T x, db = column sums of dY in ONE launch (slnlp_gemm_rows_bwd): W (fp32) and both wgrad operands (planes)
    m-major through the waves' LDS images, at every workgroup tile (same bits), batches over 64 rows (several K tiles in the weight gradient)."""
    from slnlp._lib import load, check
    dY, W, X = rnd(B, Nout, seed=1), rnd(Nout, Kin, seed=2), rnd(B, Kin, seed=3)
    pad = lambda x: torch.nn.functional.pad(x, (0, (-x.shape[1]) % 4))
    dYp, Wp, Xp = ops.split_planes(pad(dY).cuda()), pad(W).cuda(), ops.split_planes(pad(X).cuda())
    res = []
    try:
        for tile in (0, 1, 2):
            check(load().slnlp_set_rows_tile(tile), "set_rows_tile")
            res.append([t.clone() for t in ops.gemm_rows_bwd(dYp, Wp, Xp, B=B, Nout=Nout, Kin=Kin, precision=prec)])
    finally:
        load().slnlp_set_rows_tile(-1)
    dX, dW, db = res[0]
    tol = TOL[prec] * max(1.0, math.sqrt(max(Nout, B) / 64))
    assert rel(dX, dY.double() @ W.double()) < tol
    assert rel(dW, dY.double().T @ X.double()) < tol
    assert rel(db, dY.double().sum(0)) < (1e-4 if prec == 3 else 2e-2)
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


def test_gemm_rows_bwd_epilogue(ops):
    """The data gradient's epilogue: ReLU-with-scale gate (the FFN's backward), residual, per-(row, head) dropout, planes out."""
    B, Nout, Kin, dh = 50, 512, 512, 64
    dY, W, X, G, R = rnd(B, Nout, seed=1), rnd(Nout, Kin, seed=2), rnd(B, Kin, seed=3), rnd(B, Kin, seed=4), rnd(B, Kin, seed=5)
    dYp, Wp, Xp = ops.split_planes(dY.cuda()), W.cuda(), ops.split_planes(X.cuda())
    base = dY.double() @ W.double()
    dX, dW, db, (hi, lo) = ops.gemm_rows_bwd(dYp, Wp, Xp, B=B, Nout=Nout, Kin=Kin, gate=G.cuda(), gate_scale=1.25, resid=R.cuda(), want_planes=True)
    assert rel(dX, base * (G.double() > 0) * 1.25 + R.double()) < 1e-4
    h2, l2 = ops.split_planes(dX)
    assert torch.equal(hi[:B, :Kin], h2[:B, :Kin]) and torch.equal(lo[:B, :Kin], l2[:B, :Kin])
    rng = ops.make_rng(seed=1234, step=7)
    p = 0.3
    mh = ops.dropout_mask(B * (Kin // dh), 1, p, 9, rng).cpu().double().view(B, Kin // dh).repeat_interleave(dh, dim=1)
    dX, dW, db = ops.gemm_rows_bwd(dYp, Wp, Xp, B=B, Nout=Nout, Kin=Kin, drop_p=p, drop_site=9, rng=rng, drop_head_dim=dh, want_db=False)
    assert rel(dX, base * mh / (1 - p)) < 1e-4 and db is None
    assert rel(dW, dY.double().T @ X.double()) < 1e-4


def test_gemm_rows_epilogue_dropout_and_output_planes(ops):
    """bias -> ReLU -> dropout -> residual with the masks slnlp_dropout_mask reports (per element and per (row, head)), the
    tanh / ReLU gates, an in-place residual, and the emitted planes equal to the split of the fp32 result."""
    M, N, K, dh = 50, 512, 512, 64
    A, B, bias, R, G = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, seed=5)
    Ap, Bp = ops.split_planes(A.cuda()), B.cuda()
    base = A.double() @ B.double().T
    rng = ops.make_rng(seed=1234, step=7)
    p = 0.3
    mask = ops.dropout_mask(M, N, p, 5, rng).cpu().double()
    out, (hi, lo) = ops.gemm_rows(Ap, Bp, M=M, N=N, K=K, bias=bias.cuda(), relu=1, drop_p=p, drop_site=5, rng=rng, resid=R.cuda(),
                                  want_planes=True)
    ref = torch.relu(base + bias.double()) * mask / (1 - p) + R.double()
    assert rel(out, ref) < 1e-4
    h2, l2 = ops.split_planes(out)
    assert torch.equal(hi[:M, :N], h2[:M, :N]) and torch.equal(lo[:M, :N], l2[:M, :N])
    assert float(hi[M:].abs().max()) == 0
    # per (row, head) dropout: one decision per head_dim columns (decoder self-attention over a single key)
    mh = ops.dropout_mask(M * (N // dh), 1, p, 9, rng).cpu().double().view(M, N // dh).repeat_interleave(dh, dim=1)
    out = ops.gemm_rows(Ap, Bp, M=M, N=N, K=K, bias=bias.cuda(), drop_p=p, drop_site=9, rng=rng, drop_head_dim=dh)
    assert rel(out, (base + bias.double()) * mh / (1 - p)) < 1e-4
    # gates (ReLU-with-scale, tanh') and an in-place residual
    out = ops.gemm_rows(Ap, Bp, M=M, N=N, K=K, gate=G.cuda(), gate_scale=1.25)
    assert rel(out, base * (G.double() > 0) * 1.25) < 1e-4
    out = ops.gemm_rows(Ap, Bp, M=M, N=N, K=K, gate=torch.tanh(G).cuda(), gate_mode=1)
    assert rel(out, base * (1 - torch.tanh(G).double() ** 2)) < 1e-4
    Cbuf = R.cuda().clone()
    ops.gemm_rows(Ap, Bp, M=M, N=N, K=K, out=Cbuf, resid=Cbuf)
    assert rel(Cbuf, base + R.double()) < 1e-4
    # the K sum is two halves of the 64-k tiles, each in tile order: the same bits as two half-K launches added up
    h1 = ops.gemm_rows((Ap[0][:, :256].contiguous(), Ap[1][:, :256].contiguous()), Bp[:, :256].contiguous(),
                       M=M, N=N, K=256)
    # (a half-K launch splits its own 4 tiles 2 + 2, so only the structure is checked here, not bits)
    assert rel(h1, A[:, :256].double() @ B[:, :256].double().T) < 1e-4


@pytest.mark.parametrize("Mtok,Nout,Kin", [(2400, 512, 512), (16384, 3072, 1024), (16384, 2048, 1024), (16300, 1024, 512)])
def test_gradient_pair_as_the_plans_launch_it(ops, Mtok, Nout, Kin):
    """slnlp_gemm_wd: the weight and data gradient of one dY in ONE grouped launch, or -- both large -- a launch each with the weight
    gradient's K-slices added by a launch of their own (plane_splitk_reduce_kernel).  Whatever the library picks, the results are the
    grouped launch's with the same split, bit for bit (the K partition and the slice order define the sums), and close to fp64."""
    dY, X, W = rnd(Mtok, Nout, seed=1).cuda(), rnd(Mtok, Kin, seed=2).cuda(), rnd(Nout, Kin, seed=3).cuda()
    dYp, Xp, Wp = ops.split_planes(dY), ops.split_planes(X), ops.split_planes(W)
    rs = torch.empty(Nout, device="cuda")
    jw, dW = ops.plane_job(dYp, Xp, M=Nout, N=Kin, K=Mtok, a_kmajor=False, b_kmajor=False, rowsum_a=rs, precision=2)
    jd, dX = ops.plane_job(dYp, Wp, M=Mtok, N=Kin, K=Nout, a_kmajor=True, b_kmajor=False, precision=2)
    split, separate, _, _ = ops.gemm_wd_plan(jw, jd)
    assert separate == int(Nout >= 2048)      # (the rule: gemm_planes_wd_plan -- both products large)
    ops.gemm_wd(jw, jd)
    torch.cuda.synchronize()
    got = [dW.clone(), rs.clone(), dX.clone()]
    for t in (dW, rs, dX): t.fill_(float("nan"))
    ops.gemm_group([jw, jd], [split, 1])
    torch.cuda.synchronize()
    for a, b, name in zip(got, (dW, rs, dX), ("dW", "db", "dX")):
        assert torch.equal(a, b), f"{name}: slnlp_gemm_wd differs from the grouped launch with split {split}"
    # two-pass products: dY enters with its bf16 head -- held against the fp64 product of bf16(dY) (on the GPU: these are 100-GFLOP products)
    dYh = dY.bfloat16().double()
    relg = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    tol = 1e-4 * max(1.0, math.sqrt(max(Mtok, Nout) / 2400))
    assert relg(dW, dYh.T @ X.double()) < tol and relg(dX, dYh @ W.double()) < tol and relg(rs, dYh.sum(0)) < tol


@pytest.mark.parametrize("knob", [64, 128, 12832, 256])
def test_plane_epilogue_is_the_per_element_composition_bit_for_bit(ops, knob):
    """The plane GEMM's epilogue takes two routes through an LDS image of the tile -- bias / ReLU / gate / residual row-major for
    jobs without dropout, everything up to the dropout in the accumulator layout for the others -- and both must be the plain
    per-element chain  +bias -> ReLU -> gate -> dropout -> +residual  in fp32, one rounding per operation: held bit for bit
    against that chain applied by torch to the kernel's own raw product, at every tile geometry, on vector and scalar stores
    (N % 4 != 0), ragged edges, an in-place residual, and with the output planes equal to the split of the fp32 result."""
    from slnlp._lib import load, check
    import numpy as np
    rng = ops.make_rng(seed=99, step=3)
    p = 0.25
    scale_d = torch.tensor(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))      # the kernel's 1 / (1 - p) in fp32
    try:
        check(load().slnlp_set_plane_tile(knob), "set_plane_tile")
        for (M, N, K) in [(300, 200, 128), (700, 384, 192), (130, 70, 100), (520, 264, 64)]:
            A, B, bias, R, G = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, seed=5)
            pad = lambda x: torch.nn.functional.pad(x, (0, (-x.shape[1]) % 4))
            Ap, Bp = ops.split_planes(pad(A).cuda()), ops.split_planes(pad(B).cuda())
            raw = ops.gemm_planes(Ap, Bp, M=M, N=N, K=K).clone()
            biasc, Rc, Gc = bias.cuda(), R.cuda(), G.cuda()
            mask = ops.dropout_mask(M, N, p, 6, rng) > 0
            zero = torch.zeros((), device="cuda")
            def run(**kw):
                j, out = ops.plane_job(Ap, Bp, M=M, N=N, K=K, bias=kw.get("bias"), relu=kw.get("relu", False), resid=kw.get("resid"),
                                       out=kw.get("out"))
                if kw.get("gate") is not None:
                    j.gate, j.ldg, j.gate_mode, j.gate_scale = kw["gate"].data_ptr(), kw["gate"].stride(0), kw.get("gate_mode", 0), 1.25
                if kw.get("drop"):
                    j.drop_p, j.drop_site, j.rng = p, 6, rng.data_ptr()
                planes = None
                if kw.get("planes") and N % 4 == 0:
                    planes = ops.split_planes(torch.zeros(M, N).cuda())
                    j.C_hi, j.C_lo, j.ldc_p = planes[0].data_ptr(), planes[1].data_ptr(), planes[0].stride(0)
                ops.gemm_group([j], [1])
                torch.cuda.synchronize()
                return out, planes
            # without dropout: bias -> ReLU -> gate (both modes) -> residual
            out, planes = run(bias=biasc, relu=True, gate=Gc, resid=Rc, planes=True)
            want = torch.where(Gc > 0, torch.relu(raw + biasc) * 1.25, zero) + Rc
            assert torch.equal(out, want), (knob, M, N, K, "late route")
            if planes is not None:
                hi, lo = ops.split_planes(want)
                assert torch.equal(planes[0][:M, :N], hi[:M, :N]) and torch.equal(planes[1][:M, :N], lo[:M, :N])
            out, _ = run(gate=Gc, gate_mode=1, resid=Rc)
            g_two_roundings, g_fused = 1.0 - Gc * Gc, (1.0 - Gc.double() * Gc.double()).float()      # 1 - g^2 with or without an FMA
            assert torch.equal(out, raw * g_two_roundings + Rc) or torch.equal(out, raw * g_fused + Rc), (knob, M, N, K, "tanh gate")
            # with dropout: the same chain, the mask of the dump kernel, 1 / (1 - p) in fp32
            out, _ = run(bias=biasc, relu=True, gate=Gc, drop=True, resid=Rc)
            want = torch.where(mask, torch.where(Gc > 0, torch.relu(raw + biasc) * 1.25, zero) * scale_d.cuda(), zero) + Rc
            assert torch.equal(out, want), (knob, M, N, K, "dropout route")
            # in-place residual: C += product
            buf = Rc.clone()
            run(bias=biasc, resid=buf, out=buf)
            assert torch.equal(buf, (raw + biasc) + Rc)
    finally:
        load().slnlp_set_plane_tile(0)


@pytest.mark.parametrize("split", [1, 4, 7])
def test_gemm_group_dgrad_wgrad_one_launch(ops, split):
    """Data- and weight-gradient of one dY in ONE launch (gemm_planes.hip group kernel); the weight gradient's long
    K loop (= tokens) is split over `split` workgroups per tile and reduced in a fixed order -> bit-reproducible."""
    Mtok, Nout, Kin = 2400, 192, 320
    dY, X, W = rnd(Mtok, Nout, seed=1), rnd(Mtok, Kin, seed=2), rnd(Nout, Kin, seed=3)
    dYp, Xp, Wp = ops.split_planes(dY.cuda()), ops.split_planes(X.cuda()), ops.split_planes(W.cuda())
    rs = torch.empty(Nout, device="cuda")
    jw, dW = ops.plane_job(dYp, Xp, M=Nout, N=Kin, K=Mtok, a_kmajor=False, b_kmajor=False, rowsum_a=rs)   # dW = dY^T X, db = colsum dY
    jd, dX = ops.plane_job(dYp, Wp, M=Mtok, N=Kin, K=Nout, a_kmajor=True, b_kmajor=False)                 # dX = dY W
    scratch = ops.gemm_group([jw, jd], [split, 1])
    assert rel(dW, dY.double().T @ X.double()) < 2e-4
    assert rel(dX, dY.double() @ W.double()) < 1e-4
    assert rel(rs, dY.double().sum(0)) < 1e-4
    if split > 1:
        assert int(scratch[:16384].view(torch.int32).abs().max()) == 0          # arrival counters back to zero
        dW1, rs1 = dW.clone(), rs.clone()
        dW.zero_(); rs.zero_()
        ops.gemm_group([jw, jd], [split, 1], scratch)                           # same scratch, no re-zeroing
        assert torch.equal(dW, dW1) and torch.equal(rs, rs1)                    # deterministic
    # three jobs, ragged shapes, split larger than the number of K tiles (clamped)
    A2, B2 = rnd(70, 130, seed=5), rnd(50, 130, seed=6)
    def planes(x):
        xp = torch.zeros(x.shape[0], (x.shape[1] + 3) // 4 * 4); xp[:, :x.shape[1]] = x
        return ops.split_planes(xp.cuda())
    A2p, B2p = planes(A2), planes(B2)                                           # jobs hold raw pointers: keep the planes alive
    j3, o3 = ops.plane_job(A2p, B2p, M=70, N=50, K=130)
    ops.gemm_group([jw, j3, jd], [split, 9, 1])
    assert rel(o3, A2.double() @ B2.double().T) < 1e-4 and rel(dW, dY.double().T @ X.double()) < 2e-4


def test_gemm_group_large_shapes_split_k_and_epilogue(ops):
    """Plane GEMM at configs[4]-sized operands: grouped dgrad + wgrad with deterministic 8-way split-K and fused
    bias-gradient row sums, and the epilogue (bias, ReLU, residual, output planes)."""
    Mtok, Nout, Kin = 4096, 1024, 512
    dY, X, W = rnd(Mtok, Nout, seed=1), rnd(Mtok, Kin, seed=2), rnd(Nout, Kin, seed=3)
    dYp, Xp, Wp = ops.split_planes(dY.cuda()), ops.split_planes(X.cuda()), ops.split_planes(W.cuda())
    rs = torch.empty(Nout, device="cuda")
    jw, dW = ops.plane_job(dYp, Xp, M=Nout, N=Kin, K=Mtok, a_kmajor=False, b_kmajor=False, rowsum_a=rs)
    jd, dX = ops.plane_job(dYp, Wp, M=Mtok, N=Kin, K=Nout, a_kmajor=True, b_kmajor=False)
    scratch = ops.gemm_group([jw, jd], [8, 1])
    assert rel(dW, dY.double().T @ X.double()) < 3e-4 and rel(dX, dY.double() @ W.double()) < 2e-4
    assert rel(rs, dY.double().sum(0)) < 1e-4
    dW1 = dW.clone(); dW.zero_()
    ops.gemm_group([jw, jd], [8, 1], scratch)
    assert torch.equal(dW, dW1) and int(scratch[:16384].view(torch.int32).abs().max()) == 0
    M, N, K = 1024, 512, 256
    A, B, bias, R = rnd(M, K, seed=4), rnd(N, K, seed=5), rnd(N, seed=6), rnd(M, N, seed=7)
    ref = torch.relu(A.double() @ B.double().T + bias.double()) + R.double()
    out, (hi, lo) = ops.gemm_planes(ops.split_planes(A.cuda()), ops.split_planes(B.cuda()), M=M, N=N, K=K,
                                    bias=bias.cuda(), relu=True, resid=R.cuda(), want_planes=True)
    assert rel(out, ref) < 1e-4
    assert rel((hi.view(torch.bfloat16).float() + lo.view(torch.bfloat16).float())[:M, :N], ref) < 1e-4


@pytest.mark.parametrize("knob", [128, 12832, 256])
def test_plane_tile_geometries_return_the_bits_of_the_64_tile(ops, knob):
    """Every tile geometry of the plane GEMM (slnlp_set_plane_tile: 128 x 128 with 64-k or 32-k stages) accumulates
    every output element in the order of the 64 x 64 tile -- the K partition is the same -- so a launch may take whichever is
    fastest (merged lockstep launches do) without changing a bit: all three layouts, ragged edges, bias / ReLU / residual /
    output planes, split-K with the fused bias-gradient row sums."""
    from slnlp._lib import load, check
    def planes(x):
        xp = torch.zeros(x.shape[0], (x.shape[1] + 3) // 4 * 4); xp[:, :x.shape[1]] = x
        return ops.split_planes(xp.cuda())
    def run():
        outs = []
        for (M, N, K) in [(300, 200, 128), (2400, 512, 512), (130, 70, 100), (700, 384, 1000)]:
            A, B, bias, R = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
            Ap, Bp = planes(A), planes(B)
            if N % 4 == 0:
                out, (hi, lo) = ops.gemm_planes(Ap, Bp, M=M, N=N, K=K, bias=bias.cuda(), relu=True, resid=R.cuda(), want_planes=True)
                outs += [out.clone(), hi.clone(), lo.clone()]
            else:
                outs.append(ops.gemm_planes(Ap, Bp, M=M, N=N, K=K, bias=bias.cuda()).clone())
            outs.append(ops.gemm_planes(Ap, planes(B.T.contiguous()), M=M, N=N, K=K, b_kmajor=False).clone())           # dgrad layout
            rs = torch.empty(M, device="cuda")
            outs.append(ops.gemm_planes(planes(A.T.contiguous()), planes(B.T.contiguous()), M=M, N=N, K=K, a_kmajor=False, b_kmajor=False,
                                        rowsum_a=rs).clone())                                                         # wgrad layout
            outs.append(rs.clone())
        Mtok, Nout, Kin = 2400, 192, 320
        dY, X, W = rnd(Mtok, Nout, seed=1), rnd(Mtok, Kin, seed=2), rnd(Nout, Kin, seed=3)
        dYp, Xp, Wp = ops.split_planes(dY.cuda()), ops.split_planes(X.cuda()), ops.split_planes(W.cuda())
        rs = torch.empty(Nout, device="cuda")
        jw, dW = ops.plane_job(dYp, Xp, M=Nout, N=Kin, K=Mtok, a_kmajor=False, b_kmajor=False, rowsum_a=rs)
        jd, dX = ops.plane_job(dYp, Wp, M=Mtok, N=Kin, K=Nout, a_kmajor=True, b_kmajor=False)
        scratch = ops.gemm_group([jw, jd], [5, 1])
        assert int(scratch[:16384].view(torch.int32).abs().max()) == 0
        outs += [dW.clone(), rs.clone(), dX.clone()]
        assert rel(dW, dY.double().T @ X.double()) < 2e-4 and rel(dX, dY.double() @ W.double()) < 1e-4 and rel(rs, dY.double().sum(0)) < 1e-4
        torch.cuda.synchronize()
        return outs
    try:
        check(load().slnlp_set_plane_tile(64), "set_plane_tile")
        ref = run()
        check(load().slnlp_set_plane_tile(knob), "set_plane_tile")
        got = run()
    finally:
        load().slnlp_set_plane_tile(0)
    assert len(ref) == len(got)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), f"output {i} differs between the 64 x 64 tile and geometry {knob}"


def test_two_pass_gradient_products_equal_the_product_with_the_bf16_head_of_dY(ops):
    """precision 2 (slnlp_set_backward_passes: the default for weight gradients): dY enters with its bf16 head only,
    A_hi (B_hi + B_lo).  Held against the fp64 product of bf16(dY) with the other operand (1e-4: the split of B is still
    there), against the full product at bf16's rounding (2^-9 of dY, random), and bit for bit across the tile geometries;
    a 2-pass and a 3-pass job share one grouped launch, with split-K and the fused bias-gradient row sums."""
    from slnlp._lib import load, check
    Mtok, Nout, Kin = 2400, 192, 320
    dY, X, W = rnd(Mtok, Nout, seed=1), rnd(Mtok, Kin, seed=2), rnd(Nout, Kin, seed=3)
    dYh = dY.bfloat16().double()                                               # the hi plane: bf16 rounded to nearest
    dYp, Xp, Wp = ops.split_planes(dY.cuda()), ops.split_planes(X.cuda()), ops.split_planes(W.cuda())
    def run():
        rs = torch.empty(Nout, device="cuda")
        jw, dW = ops.plane_job(dYp, Xp, M=Nout, N=Kin, K=Mtok, a_kmajor=False, b_kmajor=False, rowsum_a=rs, precision=2)
        jd, dX = ops.plane_job(dYp, Wp, M=Mtok, N=Kin, K=Nout, a_kmajor=True, b_kmajor=False, precision=2)
        jd3, dX3 = ops.plane_job(dYp, Wp, M=Mtok, N=Kin, K=Nout, a_kmajor=True, b_kmajor=False, precision=3)
        ops.gemm_group([jw, jd, jd3], [5, 1, 1])
        torch.cuda.synchronize()
        return [dW.clone(), rs.clone(), dX.clone(), dX3.clone()]
    try:
        outs = {}
        for knob in (64, 128, 12832, 256):
            check(load().slnlp_set_plane_tile(knob), "set_plane_tile")
            outs[knob] = run()
    finally:
        load().slnlp_set_plane_tile(0)
    dW, rs, dX, dX3 = outs[64]
    assert rel(dW, dYh.T @ X.double()) < 1e-4 and rel(dX, dYh @ W.double()) < 1e-4 and rel(rs, dYh.sum(0)) < 1e-4
    assert rel(dX3, dY.double() @ W.double()) < 1e-4
    assert 1e-5 < rel(dW, dY.double().T @ X.double()) < 2e-3 and rel(dX, dY.double() @ W.double()) < 2e-3
    for knob in (128, 12832, 256):
        for a, b in zip(outs[64], outs[knob]):
            assert torch.equal(a, b), f"geometry {knob} differs from the 64 x 64 tile"


@pytest.mark.parametrize("knob", [128])
def test_fp8_tile_geometries(ops, knob):
    """precision 8 on the block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, every block scale 2^0) at each tile geometry
    (slnlp_set_fp8_tile): equals the fp64 product of the dequantised operands, the epilogue (per-column weight scale, bias, ReLU,
    residual, bf16 + fp8 output planes) included, on aligned and ragged shapes; all geometries agree bit for bit."""
    from slnlp._lib import load, check
    def run():
        outs = []
        for (M, N, K) in [(300, 200, 256), (1000, 512, 1024), (130, 70, 384)]:
            A, W, bias, R = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3), rnd(M, N, seed=4)
            Aq, sa = ops.quant_rows_fp8(A.cuda())
            Wq, sw = ops.quant_rows_fp8(W.cuda())
            Ad, Wd = _e4m3_decode(Aq[:M, :K].cpu()), _e4m3_decode(Wq[:N, :K].cpu())
            want = torch.relu((Ad @ Wd.T) * sw.cpu().double()[None, :] + bias.double()) + R.double()
            got, cq = ops.gemm_fp8(Aq, Wq, M=M, N=N, K=K, col_scale=sw, bias=bias.cuda(), relu=True, resid=R.cuda(), want_q8=True)
            assert rel(got.cpu().double(), want) < 2e-4, (M, N, K)
            back = _e4m3_decode(cq[:M, :N].cpu())
            assert float((back - want.clamp(-448, 448)).abs().max() / want.abs().max()) < 2 ** -4
            assert int(cq[M:].to(torch.int32).abs().max()) == 0                     # padding untouched
            outs += [got.clone(), cq.clone()]
        return outs
    try:
        check(load().slnlp_set_fp8_tile(64), "set_fp8_tile")
        ref = run()
        check(load().slnlp_set_fp8_tile(knob), "set_fp8_tile")
        got = run()
    finally:
        load().slnlp_set_fp8_tile(0)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), f"output {i} differs between fp8 geometries 64 and {knob}"
