"""Which NEIGHBOUR makes a packed-fp32 kernel compute wrongly?  The victim is the round-2 LayerNorm backward
(library of commit fcccc0d built WITH packed fp32: make PROBE=900 there, copy lib/libslnlp_probe900.so here) on constant inputs,
compared with its first result; ONE kind of library kernel at a time runs in a loop on two other streams.

    SLNLP_PROBE_LIB=900 python tools/probes/probe_aggressor.py [seconds per aggressor]
"""
import os, sys, threading, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
warnings.filterwarnings("ignore")
import torch
from slnlp import ops, _lib
dev = torch.device("cuda", 0)
_lib.load().slnlp_set_stream_policy(0)
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
M, E = 2400, 512
dy, x, gamma = rnd(M, E), rnd(M, E), rnd(E)
_, stats = ops.layernorm_fwd(x, gamma, rnd(E))
rng = ops.make_rng(3, 0)
victim = lambda: ops.layernorm_bwd(dy, x, gamma, stats, want_drop=True, drop_p=0.1, drop_site=5, rng=rng)[:2]

def make_aggressors():
    out = {}
    dY, X, W = rnd(M, E), rnd(M, E), rnd(E, E)
    dYp, Xp, Wp = ops.split_planes(dY), ops.split_planes(X), ops.split_planes(W)
    rs = torch.empty(E, device=dev)
    jw, _ = ops.plane_job(dYp, Xp, M=E, N=E, K=M, a_kmajor=False, b_kmajor=False, rowsum_a=rs)
    jd, _ = ops.plane_job(dYp, Wp, M=M, N=E, K=E, a_kmajor=True, b_kmajor=False)
    jf, _ = ops.plane_job(Xp, Wp, M=M, N=E, K=E, a_kmajor=True, b_kmajor=True)
    scr = ops.gemm_group([jw, jd], [3, 1])
    scr1 = ops.gemm_group([jf], [1])
    out["plane GEMM group (split-K 3: LDS-DMA, MFMA, sc1 hand-off)"] = lambda: ops.gemm_group([jw, jd], [3, 1], scr)
    out["plane GEMM forward (LDS-DMA, MFMA)"] = lambda: ops.gemm_group([jf], [1], scr1)
    A50, W50, o50 = rnd(50, E), rnd(E, E), torch.empty(50, E, device=dev)
    out["50-row fp32-operand GEMM (k-major, KS=2)"] = lambda: ops.gemm(A50, W50, M=50, N=E, K=E, out=o50)
    Wt, o51 = rnd(E, E), torch.empty(50, E, device=dev)
    out["50-row dgrad GEMM (ds_read_b64_tr_b16)"] = lambda: ops.gemm(A50, Wt, M=50, N=E, K=E, b_kmajor=False, out=o51)
    Awg, Bwg, owg = rnd(50, E), rnd(50, E), torch.empty(E, E, device=dev)
    out["wgrad GEMM 512x512x50 (both m-major)"] = lambda: ops.gemm(Awg, Bwg, M=E, N=E, K=50, a_kmajor=False, b_kmajor=False, out=owg)
    xl, gl = rnd(M, E), rnd(E)
    out["layernorm_fwd"] = lambda: ops.layernorm_fwd(xl, gl, gl)
    B, S, H, dh = 50, 48, 8, 64
    qkv = rnd(S * B, 3 * E)
    ids = torch.randint(2, 100, (B, S), generator=g).to(dev)
    out["attn_self_fwd (MFMA out of LDS images)"] = lambda: ops.attn_self_fwd(qkv, ids, 1, B=B, S=S, H=H, dh=dh)
    _, probs = ops.attn_self_fwd(qkv, ids, 1, B=B, S=S, H=H, dh=dh)[:2]
    dctx = rnd(S * B, E)
    out["attn_self_bwd"] = lambda: ops.attn_self_bwd(qkv, probs, dctx, B=B, S=S, H=H, dh=dh)
    n = 27_000_000
    p_, g_, b_ = torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    lr = torch.full((1,), 0.01, device=dev)
    out["clip + SGD over 27 M parameters (streaming)"] = lambda: ops.clip_sgd_step(p_, g_, b_, lr)
    out["torch matmul 2400x512x512 (rocBLAS)"] = lambda: torch.matmul(dY, W)
    return out

aggs = make_aggressors()
torch.cuda.synchronize()
print(f"victim: round-2 layernorm_bwd [2400 x 512], constant inputs; {SECS:.0f} s per aggressor; library {os.environ.get('SLNLP_PROBE_LIB', 'product')}", flush=True)
for name, fn in [("(none)", None)] + list(aggs.items()):
    stop = False
    def loop(st):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(st):
            while not stop:
                for _ in range(20): fn()
                st.synchronize()
    ths = [threading.Thread(target=loop, args=(torch.cuda.Stream(),)) for _ in range(2)] if fn else []
    [t.start() for t in ths]
    sv = torch.cuda.Stream()
    runs = diff = 0
    with torch.cuda.stream(sv):
        ref = [t.clone() for t in victim()]
        sv.synchronize()
        t0 = time.time()
        while time.time() - t0 < SECS:
            outs = [victim() for _ in range(20)]
            sv.synchronize()
            for o in outs:
                runs += 1
                diff += any(not torch.equal(a, b) for a, b in zip(ref, o))
    stop = True
    [t.join() for t in ths]
    torch.cuda.synchronize()
    print(f"  beside {name:62s}: {runs:6d} runs, {diff:6d} differ", flush=True)
