"""The decoder's products at one shape: the B-row kernels (gemm_rows.hip) at every workgroup tile, beside the kernels that could run
the same products -- the fp32-operand kernel (gemm.hip) and the plane GEMM's gradient pair (slnlp_gemm_wd) -- back to back, us per launch.

    python tools/bench_rows_shapes.py [B,Nout,Kin ...]        (default: the configs[1] and configs[4] decoder shapes)
"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sign-language-nlp_amd")]
import torch
from slnlp import ops
from slnlp._lib import load, check

shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]] or [(50, 512, 512), (50, 1024, 512), (256, 1024, 1024), (256, 512, 1024), (256, 1024, 512)]
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).cuda()


def timed(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, Nout, Kin) in shapes:
    x, W, dY, bias = rnd(B, Kin), rnd(Nout, Kin), rnd(B, Nout), rnd(Nout)
    xp, dYp, Wp = ops.split_planes(x), ops.split_planes(dY), ops.split_planes(W)
    out = torch.empty(B, Nout, device="cuda")
    line = [f"B {B} n_out {Nout} k_in {Kin}:"]
    for tile in (-1, 0, 1, 2):
        load().slnlp_set_rows_tile(tile)
        try:
            f = timed(lambda: ops.gemm_rows(xp, W, M=B, N=Nout, K=Kin, bias=bias, out=out))
            b = timed(lambda: ops.gemm_rows_bwd(dYp, W, xp, B=B, Nout=Nout, Kin=Kin))
            line.append(f"tile {tile if tile >= 0 else 'auto'}: fwd {f:.1f} bwd {b:.1f} |")
        except RuntimeError as e:
            line.append(f"tile {tile}: {str(e)[:40]} |")
    load().slnlp_set_rows_tile(-1)
    f32 = timed(lambda: ops.gemm(x, W, M=B, N=Nout, K=Kin, bias=bias, out=out))
    fpl = timed(lambda: ops.gemm_planes(xp, Wp, M=B, N=Nout, K=Kin, bias=bias, out=out))
    rs = torch.empty(Nout, device="cuda")
    jw, dW = ops.plane_job(dYp, xp, M=Nout, N=Kin, K=B, a_kmajor=False, b_kmajor=False, rowsum_a=rs, precision=3)
    jd, dX = ops.plane_job(dYp, Wp, M=B, N=Kin, K=Nout, a_kmajor=True, b_kmajor=False, precision=3)
    scr = ops.gemm_wd(jw, jd)
    pw = timed(lambda: ops.gemm_wd(jw, jd, scr))
    line.append(f"fp32-operand fwd {f32:.1f} | plane GEMM fwd {fpl:.1f} pair {pw:.1f}  (us per launch)")
    print(" ".join(line), flush=True)
