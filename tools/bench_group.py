"""Device time of the grouped (wgrad + dgrad) plane-GEMM launch at cfg2 shapes, per split factor."""
import sys, torch
sys.path.insert(0, "sign-language-nlp_amd")
from slnlp import ops
from slnlp._lib import load
Mtok, Nout = 2400, int(sys.argv[1]) if len(sys.argv) > 1 else 512
Kin = int(sys.argv[2]) if len(sys.argv) > 2 else 512
g = torch.Generator().manual_seed(0)
dY, X, W = [torch.randn(*s, generator=g) for s in ((Mtok, Nout), (Mtok, Kin), (Nout, Kin))]
dYp, Xp, Wp = ops.split_planes(dY.cuda()), ops.split_planes(X.cuda()), ops.split_planes(W.cuda())
rs = torch.empty(Nout, device="cuda")
import ctypes as _C
_w, _d = _C.c_int32(3), _C.c_int32(3)
load().slnlp_get_backward_passes(_C.byref(_w), _C.byref(_d))   # the passes the plans use (default 2, 2)
jw, dW = ops.plane_job(dYp, Xp, M=Nout, N=Kin, K=Mtok, a_kmajor=False, b_kmajor=False, rowsum_a=rs, precision=_w.value)
jd, dX = ops.plane_job(dYp, Wp, M=Mtok, N=Kin, K=Nout, a_kmajor=True, b_kmajor=False, precision=_d.value)
scr = ops.gemm_group([jw, jd], [8, 1])
def timeit(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"shapes: dY[{Mtok},{Nout}] X[{Mtok},{Kin}] W[{Nout},{Kin}]")
print(f"dgrad alone            {timeit(lambda: ops.gemm_group([jd], [1], scr)):7.1f} us")
quick = len(sys.argv) > 3
for n in ((3,) if quick else (1, 2, 3, 4, 6, 8)):
    print(f"wgrad alone split {n}    {timeit(lambda: ops.gemm_group([jw], [n], scr)):7.1f} us")
for n in ((3,) if quick else (1, 2, 3, 4, 6, 8)):
    t = timeit(lambda: ops.gemm_group([jw, jd], [n, 1], scr))
    print(f"group       split {n}    {t:7.1f} us   {2 * 2 * Mtok * Nout * Kin / t / 1e6:7.1f} TFLOP/s")
