"""How much of a plane-GEMM launch is the memory side: the same launch with its operand planes ALIASED (row stride 0: every tile
streams the same few KB per K-step, all of it L2-resident) against the real operands.  Results are garbage in the aliased run; the
time is what the kernel would take with a perfect L2.   python tools/bench_plane_alias.py [tokens n_out k_in split]"""
import sys, torch
sys.path.insert(0, "sign-language-nlp_amd")
from slnlp import ops
from slnlp._lib import load, check
import ctypes as C

def timeit(fn, n=60, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def run(Mtok, Nout, Kin, split, wp, dp):
    g = torch.Generator().manual_seed(0)
    dY, X, W = [torch.randn(*s, generator=g).cuda() for s in ((Mtok, Nout), (Mtok, Kin), (Nout, Kin))]
    dYp, Xp, Wp = ops.split_planes(dY), ops.split_planes(X), ops.split_planes(W)
    rs = torch.empty(Nout, device="cuda")
    res = {}
    for alias in (False, True):
        jw, dW = ops.plane_job(dYp, Xp, M=Nout, N=Kin, K=Mtok, a_kmajor=False, b_kmajor=False, rowsum_a=rs, precision=wp)
        jd, dX = ops.plane_job(dYp, Wp, M=Mtok, N=Kin, K=Nout, a_kmajor=True, b_kmajor=False, precision=dp)
        if alias:
            for j in (jw, jd): j.lda_p = 0; j.ldb_p = 0
        scr = ops.gemm_group([jw, jd], [split, 1])
        res[alias] = timeit(lambda: ops.gemm_group([jw, jd], [split, 1], scr))
    fl = 2 * 2.0 * Mtok * Nout * Kin
    print(f"grads {Mtok}x{Nout}x{Kin} split {split} passes {wp},{dp}: real operands {res[False]:7.1f} us ({fl / res[False] / 1e6:6.1f} TFLOP/s)   "
          f"aliased (perfect L2) {res[True]:7.1f} us ({fl / res[True] / 1e6:6.1f} TFLOP/s)   memory side costs {100 * (1 - res[True] / res[False]):.0f} %", flush=True)

def fwd(Mtok, Nout, Kin):
    g = torch.Generator().manual_seed(0)
    X, W = [torch.randn(*s, generator=g).cuda() for s in ((Mtok, Kin), (Nout, Kin))]
    Xp, Wp = ops.split_planes(X), ops.split_planes(W)
    res = {}
    for alias in (False, True):
        j, Y = ops.plane_job(Xp, Wp, M=Mtok, N=Nout, K=Kin, a_kmajor=True, b_kmajor=True)
        if alias: j.lda_p = 0; j.ldb_p = 0
        scr = ops.gemm_group([j], [1])
        res[alias] = timeit(lambda: ops.gemm_group([j], [1], scr))
    fl = 2.0 * Mtok * Nout * Kin
    print(f"fwd   {Mtok}x{Nout}x{Kin}: real operands {res[False]:7.1f} us ({fl / res[False] / 1e6:6.1f} TFLOP/s)   aliased {res[True]:7.1f} us "
          f"({fl / res[True] / 1e6:6.1f} TFLOP/s)   memory side costs {100 * (1 - res[True] / res[False]):.0f} %", flush=True)

if len(sys.argv) == 2 and sys.argv[1] in ("rounds", "half"):
    pass
elif len(sys.argv) > 4:
    run(*[int(v) for v in sys.argv[1:5]], 2, 2)
else:
    for passes in ((3, 3), (2, 2)):
        run(16384, 3072, 1024, 6, *passes)
        run(16384, 1024, 512, 8, *passes)
        run(38400, 512, 512, 8, *passes)
    fwd(16384, 3072, 1024)
    fwd(36000, 512, 512)
    fwd(36000, 1536, 512)
if len(sys.argv) == 2 and sys.argv[1] == "rounds":       # does the memory-side cost grow with the number of rounds (K-phase drift)?
    for M in (8192, 16384, 32768, 65536):
        fwd(M, 1024, 3072)
    for M in (8192, 16384, 32768, 65536):
        fwd(M, 1024, 1024)
if len(sys.argv) == 2 and sys.argv[1] == "half":         # only ONE operand aliased: about half the L2 -> LDS bytes per FLOP, what a 256-wide tile would move
    for (M, N, K) in ((16384, 3072, 1024), (36000, 512, 512), (36000, 1536, 512), (65536, 1024, 3072)):
        g = torch.Generator().manual_seed(0)
        X, W = [torch.randn(*s, generator=g).cuda() for s in ((M, K), (N, K))]
        Xp, Wp = ops.split_planes(X), ops.split_planes(W)
        res = {}
        for mode in ("real", "B aliased", "A aliased", "both"):
            j, Y = ops.plane_job(Xp, Wp, M=M, N=N, K=K, a_kmajor=True, b_kmajor=True)
            if mode in ("A aliased", "both"): j.lda_p = 0
            if mode in ("B aliased", "both"): j.ldb_p = 0
            scr = ops.gemm_group([j], [1])
            res[mode] = timeit(lambda: ops.gemm_group([j], [1], scr))
        fl = 2.0 * M * N * K
        print(f"fwd {M}x{N}x{K}: " + "  ".join(f"{m} {t:.1f} us ({fl / t / 1e6:.0f} TF)" for m, t in res.items()), flush=True)
