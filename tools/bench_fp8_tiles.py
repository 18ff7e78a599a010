"""precision-8 GEMM (block-scaled fp8 MFMA) per tile geometry (slnlp_set_fp8_tile) at the configs[4] forward shapes."""
import sys, torch
sys.path.insert(0, "sign-language-nlp_amd")
from slnlp import ops
from slnlp._lib import load, check
def timeit(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
for name, M, N, K in [("cfg5 in_proj 16384x3072x1024", 16384, 3072, 1024), ("cfg5 FFN1 16384x512x1024", 16384, 512, 1024),
                      ("cfg2 in_proj 2400x1536x512", 2400, 1536, 512), ("cfg2 x16 fits 38400x1536x512", 38400, 1536, 512)]:
    X, W = torch.randn(M, K, generator=g).cuda(), (torch.randn(N, K, generator=g) * 0.05).cuda()
    Xq, _ = ops.quant_rows_fp8(X)
    Wq, sw = ops.quant_rows_fp8(W)
    out = torch.empty(M, N, device="cuda")
    ref = None
    for knob in (64, 128):
        check(load().slnlp_set_fp8_tile(knob), "set_fp8_tile")
        ops.gemm_fp8(Xq, Wq, M=M, N=N, K=K, col_scale=sw, out=out)
        torch.cuda.synchronize()
        same = "" if ref is None else f"  == tile 64: {torch.equal(ref, out)}"
        if ref is None: ref = out.clone()
        t = timeit(lambda: ops.gemm_fp8(Xq, Wq, M=M, N=N, K=K, col_scale=sw, out=out))
        print(f"{name:32s} tile {knob:6d}: {t:8.1f} us  {2.0 * M * N * K / t / 1e6:8.1f} TFLOP/s{same}", flush=True)
    load().slnlp_set_fp8_tile(0)
