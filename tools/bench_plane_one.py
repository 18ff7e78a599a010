"""ONE plane-GEMM launch shape at ONE tile geometry, a few launches -- the thing to run under rocprofv3 --pmc.
    python tools/bench_plane_one.py <tokens> <n_out> <k_in> <split> <tile knob> [launches]
split 0 = the gradient pair as the training plans launch it (slnlp_gemm_wd); ONLY=dgrad / wgrad in the environment = one job alone."""
import sys, torch
sys.path.insert(0, "sign-language-nlp_amd")
from slnlp import ops
from slnlp._lib import load, check
Mtok, Nout, Kin, split, tile = [int(v) for v in sys.argv[1:6]]
n = int(sys.argv[6]) if len(sys.argv) > 6 else 5
g = torch.Generator().manual_seed(0)
dY, X, W = [torch.randn(*s, generator=g).cuda() for s in ((Mtok, Nout), (Mtok, Kin), (Nout, Kin))]
dYp, Xp, Wp = ops.split_planes(dY), ops.split_planes(X), ops.split_planes(W)
rs = torch.empty(Nout, device="cuda")
import ctypes as _C
_w, _d = _C.c_int32(3), _C.c_int32(3)
load().slnlp_get_backward_passes(_C.byref(_w), _C.byref(_d))   # the passes the plans use (default 2, 2)
jw, dW = ops.plane_job(dYp, Xp, M=Nout, N=Kin, K=Mtok, a_kmajor=False, b_kmajor=False, rowsum_a=rs, precision=_w.value)
jd, dX = ops.plane_job(dYp, Wp, M=Mtok, N=Kin, K=Nout, a_kmajor=True, b_kmajor=False, precision=_d.value)
check(load().slnlp_set_plane_tile(tile), "set_plane_tile")
import os
only = os.environ.get("ONLY", "")                      # ONLY=dgrad / wgrad: one of the two jobs alone (which one misses the L2?)
if only:
    jobs, splits = ([jd], [1]) if only == "dgrad" else ([jw], [split])
    scr = ops.gemm_group(jobs, splits)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.gemm_group(jobs, splits, scr)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n * 1e3
    print(f"{only} alone: tokens {Mtok} n_out {Nout} k_in {Kin} split {split} tile {tile}: {t:.1f} us/launch, {2.0 * Mtok * Nout * Kin / t / 1e6:.1f} TFLOP/s")
    sys.exit(0)
if split == 0:      # split 0: the pair as the training plans launch it (slnlp_gemm_wd: the library picks split, tile, one launch or two)
    print("library plan (split, separate, geometry wgrad, geometry dgrad):", ops.gemm_wd_plan(jw, jd))
    launch = lambda scr=None: ops.gemm_wd(jw, jd, scr)
else:
    launch = lambda scr=None: ops.gemm_group([jw, jd], [split, 1], scr)
scr = launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): launch(scr)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / n * 1e3
print(f"tokens {Mtok} n_out {Nout} k_in {Kin} split {split} tile {tile}: {t:.1f} us/launch, {2 * 2.0 * Mtok * Nout * Kin / t / 1e6:.1f} TFLOP/s")
