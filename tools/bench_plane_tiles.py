"""Plane-GEMM launches at the 64 x 64 and the 128 x 128 tile (slnlp_set_plane_tile): device time per launch, algorithmic
TFLOP/s, and that both tiles return the same bits (same K partition => same accumulation order).

    python tools/bench_plane_tiles.py [quick]
"""
import sys, torch
sys.path.insert(0, "sign-language-nlp_amd")
from slnlp import ops
from slnlp._lib import load, check

def timeit(fn, n=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def case(name, Mtok, Nout, Kin, split, copies=1, check_ref=False):
    """dgrad + wgrad of `copies` independent dY [Mtok, Nout] against W [Nout, Kin] / X [Mtok, Kin] in ONE launch (<= 4 jobs)."""
    g = torch.Generator().manual_seed(0)
    jobs, splits, outs, refs = [], [], [], []
    for c in range(copies):
        dY, X, W = [torch.randn(*s, generator=g).cuda() for s in ((Mtok, Nout), (Mtok, Kin), (Nout, Kin))]
        dYp, Xp, Wp = ops.split_planes(dY), ops.split_planes(X), ops.split_planes(W)
        rs = torch.empty(Nout, device="cuda")
        jw, dW = ops.plane_job(dYp, Xp, M=Nout, N=Kin, K=Mtok, a_kmajor=False, b_kmajor=False, rowsum_a=rs, precision=WPASS)
        jd, dX = ops.plane_job(dYp, Wp, M=Mtok, N=Kin, K=Nout, a_kmajor=True, b_kmajor=False, precision=DPASS)
        jobs += [jw, jd]; splits += [split, 1]; outs += [dW, rs, dX]
        if check_ref and c == 0:
            refs = [dY.double().T @ X.double(), dY.double().sum(0), dY.double() @ W.double()]
        keep.append((dYp, Xp, Wp))
    scr = ops.gemm_group(jobs, splits)
    flops = copies * 2 * 2.0 * Mtok * Nout * Kin
    res = {}
    for tile in TILES:
        check(load().slnlp_set_plane_tile(tile), "set_plane_tile")
        for o in outs: o.fill_(float("nan"))
        ops.gemm_group(jobs, splits, scr)
        torch.cuda.synchronize()
        res[tile] = [o.clone() for o in outs]
        t = timeit(lambda: ops.gemm_group(jobs, splits, scr))
        err = ""
        if refs:
            e = max(float((a.double() - b).abs().max() / b.abs().max()) for a, b in zip(res[tile][:3], refs))
            err = f"  rel err vs fp64 {e:.1e}"
        cd = lambda a, b: (a + b - 1) // b
        bm, bn = DIMS[tile]
        units = copies * (cd(Nout, bm) * cd(Kin, bn) * split + cd(Mtok, bm) * cd(Kin, bn))
        print(f"{name:34s} tile {tile:6d}: {units:5d} workgroups {t:8.1f} us  {flops / t / 1e6:7.1f} TFLOP/s{err}", flush=True)
    same = all(torch.equal(a, b) for tile in TILES[1:] for a, b in zip(res[TILES[0]], res[tile]))
    print(f"{'':34s} tiles bit-identical: {same}", flush=True)
    check(load().slnlp_set_plane_tile(0), "set_plane_tile")
    return same

def fwd_case(name, Mtok, Nout, Kin, planes=False):
    """Y = X W^T alone (the forward launches); planes=True: with bias, residual and the bf16 planes of Y, as in a train step."""
    g = torch.Generator().manual_seed(0)
    X, W = [torch.randn(*s, generator=g).cuda() for s in ((Mtok, Kin), (Nout, Kin))]
    Xp, Wp = ops.split_planes(X), ops.split_planes(W)
    kw = {}
    if planes:
        kw = dict(bias=torch.randn(Nout, generator=g).cuda(), resid=torch.randn(Mtok, Nout, generator=g).cuda())
    j, Y = ops.plane_job(Xp, Wp, M=Mtok, N=Nout, K=Kin, a_kmajor=True, b_kmajor=True, **kw)
    if planes:
        Yp = ops.split_planes(torch.zeros(Mtok, Nout).cuda())
        j.C_hi, j.C_lo, j.ldc_p = Yp[0].data_ptr(), Yp[1].data_ptr(), Yp[0].stride(0)
        keep.append((Yp, kw))
    scr = ops.gemm_group([j], [1])
    res = {}
    for tile in TILES:
        check(load().slnlp_set_plane_tile(tile), "set_plane_tile")
        Y.fill_(float("nan")); ops.gemm_group([j], [1], scr); torch.cuda.synchronize()
        res[tile] = Y.clone()
        t = timeit(lambda: ops.gemm_group([j], [1], scr))
        print(f"{name:34s} tile {tile:6d}: {t:8.1f} us  {2.0 * Mtok * Nout * Kin / t / 1e6:7.1f} TFLOP/s", flush=True)
    same = all(torch.equal(res[TILES[0]], res[t]) for t in TILES[1:])
    print(f"{'':34s} tiles bit-identical: {same}", flush=True)
    check(load().slnlp_set_plane_tile(0), "set_plane_tile")
    return same

keep = []
import os
WPASS, DPASS = [int(v) for v in os.environ.get("PLANE_PASSES", "3,3").split(",")]   # split-bf16 passes of the wgrad / dgrad jobs
print(f"passes: wgrad {WPASS}, dgrad {DPASS} (FLOP/s are algorithmic: 2 m n k per product whatever the passes)")
# slnlp_set_plane_tile knobs: 64 x 64, 128 x 128 (64-k x 2 stages, 128 KiB), 128 x 128 (32-k x 2, 64 KiB), 256 x 256 (32-k x 2, 128 KiB)
TILES = (64, 128, 12832, 256)
DIMS = {64: (64, 64), 128: (128, 128), 12832: (128, 128), 256: (256, 256)}
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
big_only = len(sys.argv) > 1 and sys.argv[1] == "big"
ok = True
if len(sys.argv) > 1 and sys.argv[1] == "fwd":      # forward launches (no split-K): which ring pays
    for pl in (False, True):
        tag = " +planes" if pl else ""
        ok &= fwd_case("cfg2 in_proj 2400x1536x512" + tag, 2400, 1536, 512, pl)
        ok &= fwd_case("15 fits 36000x512x512" + tag, 36000, 512, 512, pl)
        ok &= fwd_case("15 fits in_proj 36000x1536x512" + tag, 36000, 1536, 512, pl)
        ok &= fwd_case("configs[4] FFN 16384x512x1024" + tag, 16384, 512, 1024, pl)
        ok &= fwd_case("configs[4] in_proj 16384x3072x1024" + tag, 16384, 3072, 1024, pl)
        ok &= fwd_case("configs[4] FFN2 16384x1024x3072" + tag, 16384, 1024, 3072, pl)
        ok &= fwd_case("ragged 1000x328x192" + tag, 1000, 328, 192, pl)
    print("ALL TILES BIT-IDENTICAL" if ok else "TILE MISMATCH")
    sys.exit(0 if ok else 1)
if not big_only:
    ok &= case("cfg2 dgrad+wgrad E512", 2400, 512, 512, 3, check_ref=True)
    ok &= case("cfg2 x2 fits (one launch)", 2400, 512, 512, 3, copies=2)
    ok &= case("cfg2 in_proj grads 1536", 2400, 1536, 512, 3)
    ok &= case("ragged 1000 x 320 x 192", 1000, 320, 192, 2, check_ref=True)
    ok &= case("ragged 300 x 72 x 200 (K 300)", 300, 72, 200, 2, check_ref=True)
if not quick:
    ok &= case("tokens x4 (9600) E512", 9600, 512, 512, 8)
    ok &= case("tokens x16 (38400) E512", 38400, 512, 512, 8)
    ok &= case("cfg5 FFN grads 16384x1024x512", 16384, 1024, 512, 8)
    ok &= case("cfg5 in_proj grads 16384x3072x1024", 16384, 3072, 1024, 6)
print("ALL TILES BIT-IDENTICAL" if ok else "TILE MISMATCH")
sys.exit(0 if ok else 1)
