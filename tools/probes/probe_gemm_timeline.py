"""Where a workgroup of the fp32-operand GEMM (the decoder's 50-row products) spends its time: per-workgroup timestamps (100 MHz)
of kernel entry, first K tile in LDS, K loop done, image written, epilogue done -- from the probe build (make PROBE=256 in
sign-language-nlp_amd/).  Each shape is timed WARM (back-to-back launches of the same kernel) and COLD (behind a pass over 1 GiB
that evicts L2 / MALL and other kernels that evict the instruction cache), as it runs inside a train step.

    SLNLP_PROBE_LIB=256 python tools/probes/probe_gemm_timeline.py
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sign-language-nlp_amd")]
os.environ.setdefault("SLNLP_PROBE_LIB", "256")
import numpy as np, torch
from slnlp import ops
from slnlp._lib import load
lib = load()
lib.slnlp_probe_gemm_ts.restype = C.c_int
lib.slnlp_probe_gemm_ts.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((1 << 14, 6), dtype=np.uint64)

lib.slnlp_probe_rows_ts.restype = C.c_int
lib.slnlp_probe_rows_ts.argtypes = [C.c_void_p, C.c_int]
READ = [lib.slnlp_probe_gemm_ts]        # which recorder report() reads: the fp32-operand kernel's or the B-row plane kernel's (gemm_rows.hip)

def read():
    n = READ[0](buf.ctypes.data, buf.shape[0])
    assert n >= 0
    return buf[:n].astype(np.int64).copy()

big = torch.empty(256 << 20, dtype=torch.float32, device="cuda")      # 1 GiB
other_x = torch.randn(4096, 512, device="cuda")
def evict():
    big.add_(1.0)                                                       # L2 / MALL
    torch.nn.functional.layer_norm(other_x, (512,)); torch.softmax(other_x, -1); other_x @ other_x.T    # other code through the I-cache
    torch.cuda.synchronize()

def report(name, launch):
    for _ in range(20): launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): launch()
    e1.record(); torch.cuda.synchronize()
    b2b = e0.elapsed_time(e1) / 50 * 1e3
    read()
    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
    print(f"{name:44s} events: {b2b:6.2f} us per launch back to back, {e0.elapsed_time(e1) * 1e3:6.2f} us one launch alone")
    warm = read()
    cold = []
    for _ in range(8):
        evict(); read(); launch(); cold.append(read())
    cold = np.concatenate(cold)
    groups = []
    for tag, t in (("warm", warm), ("cold", cold)):       # (backward pairs: the weight-gradient units stamp with bit 31 of the block word, bit 30 = + row sums)
        kind = (t[:, 5] >> 30) & 3
        groups.append((tag, t[kind == 0]))
        if (kind != 0).any():
            groups += [(tag + " wgrad", t[kind == 2]), (tag + " wgrad+rowsum", t[kind == 3])]
    for tag, t in groups:
        if not len(t): continue
        us = lambda a: a / 100.0
        ph = [us(t[:, i + 1] - t[:, i]) for i in range(4)]
        q = lambda a: "%5.2f %5.2f %5.2f" % tuple(np.percentile(a, [10, 50, 90]))
        span = us(t[:, 4].max() - t[:, 0].min()) if tag.startswith("warm") else float("nan")
        print(f"{name:44s} {tag}: {len(t):4d} wg | entry->first tile {q(ph[0])} | K loop {q(ph[1])} | reduce+image {q(ph[2])} | epilogue {q(ph[3])} | "
              f"total {q(us(t[:, 4] - t[:, 0]))} | span {span:5.2f}", flush=True)

g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).cuda()
rng = ops.make_rng(seed=1, step=1)
B, E, F = 50, 512, 512
x, W, bias, R = rnd(B, E), rnd(E, E), rnd(E), rnd(B, E)
out = torch.empty(B, E, device="cuda")
report("linear [50x512]x[512x512] bias",            lambda: ops.gemm(x, W, M=B, N=E, K=E, bias=bias, out=out))
report("linear ... + dropout + residual",           lambda: ops.gemm(x, W, M=B, N=E, K=E, bias=bias, drop_p=0.1, drop_site=3, rng=rng, resid=R, out=out))
report("linear ... + relu + dropout (FFN1)",        lambda: ops.gemm(x, W, M=B, N=E, K=E, bias=bias, relu=True, drop_p=0.1, drop_site=3, rng=rng, out=out))
Wt = W.T.contiguous()
report("dgrad [50x512]x[512x512] (W m-major) +res", lambda: ops.gemm(x, Wt, M=B, N=E, K=E, b_kmajor=False, resid=R, out=out))
W3 = rnd(3 * E, E); out3 = torch.empty(B, 3 * E, device="cuda")
report("linear [50x512]x[512x1536]",                lambda: ops.gemm(x, W3, M=B, N=3 * E, K=E, out=out3))

# the same products on plane operands, register-direct (gemm_rows.hip): entry -> first tile's operands landed -> partials stored ->
# K sum done -> epilogue done
READ[0] = lib.slnlp_probe_rows_ts
xp, Wp, W3p = ops.split_planes(x), W, W3        # (the weight goes in as fp32: the kernel splits it in registers)
report("rows linear [50x512]x[512x512] bias",       lambda: ops.gemm_rows(xp, Wp, M=B, N=E, K=E, bias=bias, out=out))
report("rows linear ... + dropout + residual",      lambda: ops.gemm_rows(xp, Wp, M=B, N=E, K=E, bias=bias, drop_p=0.1, drop_site=3, rng=rng, resid=R, out=out))
report("rows linear [50x512]x[512x1536]",           lambda: ops.gemm_rows(xp, W3p, M=B, N=3 * E, K=E, out=out3))
# the backward pair (data-gradient tiles; weight-gradient units: entry -> images landed -> first tile's MFMAs -> K loop + row sums -> end) and
# the configs[4] decoder shape, at each workgroup tile
dYp = ops.split_planes(rnd(B, E))
report("rows bwd pair [50x512]",                    lambda: ops.gemm_rows_bwd(dYp, W, xp, B=B, Nout=E, Kin=E))
B5, E5 = 256, 1024
x5p, dY5p, W5, out5 = ops.split_planes(rnd(B5, E5)), ops.split_planes(rnd(B5, E5)), rnd(E5, E5), torch.empty(B5, E5, device="cuda")
for tile in (0, 1):
    lib.slnlp_set_rows_tile(tile)
    report(f"rows linear [256x1024]x[1024x1024] tile {tile}", lambda: ops.gemm_rows(x5p, W5, M=B5, N=E5, K=E5, out=out5))
    report(f"rows bwd pair [256x1024] tile {tile}",         lambda: ops.gemm_rows_bwd(dY5p, W5, x5p, B=B5, Nout=E5, Kin=E5))
lib.slnlp_set_rows_tile(-1)
