"""The dgrad + wgrad group of one dY at several split-K factors of the weight gradient and several tile geometries: what the
round structure of a launch (units on 512 / 256 workgroup slots) costs.  Interleaved rounds in ONE process (guide rule 24).

    python tools/bench_plane_split.py [<tokens> <n_out> <k_in>] [--splits 4,6,8,12] [--tiles 12832,256] [--rounds 3]
"""
import argparse, ctypes as C, sys, torch
sys.path.insert(0, "sign-language-nlp_amd")
from slnlp import ops
from slnlp._lib import load, check

ap = argparse.ArgumentParser()
ap.add_argument("shape", nargs="*", type=int, default=[16384, 3072, 1024])
ap.add_argument("--splits", default="4,6,8,12")
ap.add_argument("--tiles", default="12832,256")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--launches", type=int, default=30)
a = ap.parse_args()
Mtok, Nout, Kin = a.shape
splits = [int(v) for v in a.splits.split(",")]
tiles = [int(v) for v in a.tiles.split(",")]
g = torch.Generator().manual_seed(0)
dY, X, W = [torch.randn(*s, generator=g).cuda() for s in ((Mtok, Nout), (Mtok, Kin), (Nout, Kin))]
dYp, Xp, Wp = ops.split_planes(dY), ops.split_planes(X), ops.split_planes(W)
rs = torch.empty(Nout, device="cuda")
w, d = C.c_int32(3), C.c_int32(3)
load().slnlp_get_backward_passes(C.byref(w), C.byref(d))
jw, dW = ops.plane_job(dYp, Xp, M=Nout, N=Kin, K=Mtok, a_kmajor=False, b_kmajor=False, rowsum_a=rs, precision=w.value)
jd, dX = ops.plane_job(dYp, Wp, M=Mtok, N=Kin, K=Nout, a_kmajor=True, b_kmajor=False, precision=d.value)
scr = {s: ops.gemm_group([jw, jd], [s, 1]) for s in splits}
flops = 2 * 2.0 * Mtok * Nout * Kin


def timeit(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = {}
for r in range(a.rounds):
    for tile in tiles:
        check(load().slnlp_set_plane_tile(tile), "set_plane_tile")
        for s in splits:
            res.setdefault((tile, s), []).append(timeit(lambda: ops.gemm_group([jw, jd], [s, 1], scr[s]), a.launches))
check(load().slnlp_set_plane_tile(0), "set_plane_tile")
print(f"dgrad + wgrad of dY [{Mtok} x {Nout}], W [{Nout} x {Kin}]; passes wgrad {w.value} dgrad {d.value}; us per launch (min / median of {a.rounds} rounds), TFLOP/s at the min")
for (tile, s), ts in sorted(res.items()):
    ts = sorted(ts)
    print(f"  tile {tile:6d} split {s:3d}: {ts[0]:8.1f} / {ts[len(ts) // 2]:8.1f} us   {flops / ts[0] / 1e6:7.1f} TFLOP/s", flush=True)
