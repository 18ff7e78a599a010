"""Where a plane-GEMM workgroup spends its time: per-workgroup timestamps (100 MHz) of kernel entry, first K-step landed, K loop
done, split-K meeting done, epilogue done -- from the timeline probe build (make PROBE=128 in sign-language-nlp_amd/).

    SLNLP_PROBE_LIB=128 python tools/probes/probe_tile_timeline.py
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sign-language-nlp_amd")]
os.environ.setdefault("SLNLP_PROBE_LIB", "128")
import numpy as np, torch
from slnlp import ops
from slnlp._lib import load, check
lib = load()
lib.slnlp_probe_ts.restype = C.c_int
lib.slnlp_probe_ts.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((1 << 16, 8), dtype=np.uint64)   # ring of the last 65536 workgroups; words: probe build, gemm_planes.hip TS_W

def read():
    n = lib.slnlp_probe_ts(buf.ctypes.data, buf.shape[0])
    assert n >= 0
    return buf[:n].astype(np.int64).copy()

def report(name, launch, tile):
    check(lib.slnlp_set_plane_tile(tile), "tile")
    import time
    for _ in range(3): launch()
    torch.cuda.synchronize()
    # the clock the chip HOLDS under this kernel: >= 2 s of back-to-back launches on random data first (MI355X guide, DVFS item 6),
    # then the stamps of the workgroups still in the ring: shader cycles (s_memtime) / 100 MHz ticks (s_memrealtime) over the K loop
    t_end = time.time() + SUSTAIN
    n_l = 0
    while time.time() < t_end:
        for _ in range(20): launch()
        torch.cuda.synchronize(); n_l += 20
    hot = read()
    dt = (hot[:, 2] - hot[:, 1]).astype(np.float64)
    ok = dt > 50                                             # K loops of at least 0.5 us
    ghz = (hot[ok, 7] - hot[ok, 6]) / dt[ok] * 0.1
    launch()
    t = read()
    if ok.any():
        print(f"{name} tile {tile}: held shader clock over the K loop after {SUSTAIN:.0f} s ({n_l} launches): "
              f"p10/p50/p90 {np.percentile(ghz, 10):.3f} {np.percentile(ghz, 50):.3f} {np.percentile(ghz, 90):.3f} GHz ({ok.sum()} workgroups)")
    t0 = t[:, 0].min()
    us = lambda a: a / 100.0
    fill, loop, meet, epi, tot = [us(t[:, i + 1] - t[:, i]) for i in range(4)] + [us(t[:, 4] - t[:, 0])]
    start = us(t[:, 0] - t0)
    q = lambda a: "%6.2f %6.2f %6.2f" % tuple(np.percentile(a, [10, 50, 90]))
    print(f"{name} tile {tile}: {len(t)} workgroups, launch span {us(t[:, 4].max() - t0):7.2f} us")
    print(f"    p10/p50/p90 us: first K-step landed {q(fill)} | K loop {q(loop)} | drain+meeting {q(meet)} | epilogue {q(epi)} | total {q(tot)}")
    print(f"    workgroup start times: p10/p50/p90 {q(start)}; started within 2 us of the first: {(start < 2).sum()}")
    if COLD:
        # the same launch behind evicted caches (a pass over 1 GiB, other kernels through the instruction cache), as inside a train step
        spans, rows = [], []
        for _ in range(6):
            evict(); read(); launch(); c = read()
            spans.append(us(c[:, 4].max() - c[:, 0].min())); rows.append(c)
        c = np.concatenate(rows)
        ph = [us(c[:, i + 1] - c[:, i]) for i in range(4)] + [us(c[:, 4] - c[:, 0])]
        ph[2] = ph[2][np.abs(ph[2]) < 1e6]; ph[3] = ph[3][np.abs(ph[3]) < 1e6]          # (workgroups that left at the split-K meeting carry no mark 3)
        print(f"    COLD, 6 launches: span {np.mean(spans):7.2f} us | first K-step landed {q(ph[0])} | K loop {q(ph[1])} | drain+meeting {q(ph[2])} | epilogue {q(ph[3])} | total {q(ph[4])}")
    check(lib.slnlp_set_plane_tile(0), "tile")

SUSTAIN = float(os.environ.get("PROBE_SUSTAIN_S", "2"))
COLD = os.environ.get("PROBE_COLD", "0") == "1"
_big = torch.empty(256 << 20, dtype=torch.float32, device="cuda") if COLD else None      # 1 GiB
_other = torch.randn(4096, 512, device="cuda") if COLD else None
def evict():
    _big.add_(1.0)
    torch.nn.functional.layer_norm(_other, (512,)); torch.softmax(_other, -1); _other @ _other.T
    torch.cuda.synchronize()
g = torch.Generator().manual_seed(0)
def fwd(M, N, K):
    X, W = [torch.randn(*s, generator=g).cuda() for s in ((M, K), (N, K))]
    Xp, Wp = ops.split_planes(X), ops.split_planes(W)
    j, Y = ops.plane_job(Xp, Wp, M=M, N=N, K=K, a_kmajor=True, b_kmajor=True)
    scr = ops.gemm_group([j], [1])
    return lambda: ops.gemm_group([j], [1], scr)

def grads(M, N, K, split):
    dY, X, W = [torch.randn(*s, generator=g).cuda() for s in ((M, N), (M, K), (N, K))]
    dYp, Xp, Wp = ops.split_planes(dY), ops.split_planes(X), ops.split_planes(W)
    rs = torch.empty(N, device="cuda")
    jw, dW = ops.plane_job(dYp, Xp, M=N, N=K, K=M, a_kmajor=False, b_kmajor=False, rowsum_a=rs)
    jd, dX = ops.plane_job(dYp, Wp, M=M, N=K, K=N, a_kmajor=True, b_kmajor=False)
    scr = ops.gemm_group([jw, jd], [split, 1])
    return lambda: ops.gemm_group([jw, jd], [split, 1], scr)

def one_grad(M, N, K, split, which):
    dY, X, W = [torch.randn(*s, generator=g).cuda() for s in ((M, N), (M, K), (N, K))]
    dYp, Xp, Wp = ops.split_planes(dY), ops.split_planes(X), ops.split_planes(W)
    rs = torch.empty(N, device="cuda")
    if which == "wgrad":
        j, _ = ops.plane_job(dYp, Xp, M=N, N=K, K=M, a_kmajor=False, b_kmajor=False, rowsum_a=rs, precision=2)
    else:
        j, _ = ops.plane_job(dYp, Wp, M=M, N=K, K=N, a_kmajor=True, b_kmajor=False, precision=2)
    scr = ops.gemm_group([j], [split])
    return lambda: ops.gemm_group([j], [split], scr)

if os.environ.get("PROBE_PAIR"):      # the configs[4] in_proj gradient products ALONE, as slnlp_gemm_wd launches them (two-pass, 256 x 256)
    report("configs[4] in_proj dgrad alone          ", one_grad(16384, 3072, 1024, 1, "dgrad"), 256)
    for sp in (5, 4, 3):
        report(f"configs[4] in_proj wgrad alone, split {sp}", one_grad(16384, 3072, 1024, sp, "wgrad"), 256)
    sys.exit(0)
report("cfg2 forward 2400x512x512        ", fwd(2400, 512, 512), 64)
report("cfg2 dgrad+wgrad (split 3)       ", grads(2400, 512, 512, 3), 64)
report("15 fits' forward 36000x512x512   ", fwd(36000, 512, 512), 128)
report("15 fits' forward 36000x512x512   ", fwd(36000, 512, 512), 64)
report("15 fits' forward 36000x512x512   ", fwd(36000, 512, 512), 12832)
if os.environ.get("PROBE_EXTRA"):
    report("15 fits' forward 36000x512x512   ", fwd(36000, 512, 512), 256)
    report("15 fits' in_proj 36000x1536x512  ", fwd(36000, 1536, 512), 12832)
    report("15 fits' in_proj 36000x1536x512  ", fwd(36000, 1536, 512), 256)
report("configs[4] in_proj 16384x3072x1024", fwd(16384, 3072, 1024), 128)
report("configs[4] in_proj 16384x3072x1024", fwd(16384, 3072, 1024), 12832)
report("configs[4] in_proj 16384x3072x1024", fwd(16384, 3072, 1024), 256)
report("configs[4] in_proj grads 16384x3072x1024 split 6", grads(16384, 3072, 1024, 6), 12832)
