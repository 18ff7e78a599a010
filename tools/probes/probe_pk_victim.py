"""The synthetic packed-fp32 victim (tools/probes/pk_victim.hip: the library kernel's load / v_pk_* / 16-byte-store sequence, checked
against scalar instructions by a second kernel) beside ONE kind of LIBRARY kernel at a time -- does the victim side reproduce
outside the library?    python tools/probes/probe_pk_victim.py [seconds per aggressor]"""
import ctypes as C, os, sys, threading, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
warnings.filterwarnings("ignore")
import torch
from slnlp import ops, _lib
dev = torch.device("cuda", 0)
_lib.load().slnlp_set_stream_policy(0)
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
torch.zeros(1, device=dev)
pk = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpkvictim.so"))
pk.pkv_run.restype = C.c_uint
pk.pkv_run.argtypes = [C.c_void_p, C.c_int]
assert pk.pkv_init() == 0
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
M, E, B, S, H, dh = 2400, 512, 50, 48, 8, 64
dY, X, W = rnd(M, E), rnd(M, E), rnd(E, E)
dYp, Xp, Wp = ops.split_planes(dY), ops.split_planes(X), ops.split_planes(W)
rs = torch.empty(E, device=dev)
jw, _ = ops.plane_job(dYp, Xp, M=E, N=E, K=M, a_kmajor=False, b_kmajor=False, rowsum_a=rs)
jd, _ = ops.plane_job(dYp, Wp, M=M, N=E, K=E, a_kmajor=True, b_kmajor=False)
scr = ops.gemm_group([jw, jd], [3, 1])
qkv, dctx = rnd(S * B, 3 * E), rnd(S * B, E)
ids = torch.randint(2, 100, (B, S), generator=g).to(dev)
probs = ops.attn_self_fwd(qkv, ids, 1, B=B, S=S, H=H, dh=dh)[1]
Awg, Bwg, owg = rnd(50, E), rnd(50, E), torch.empty(E, E, device=dev)
aggs = {"(none)": None,
        "plane GEMM group (split-K 3)": lambda: ops.gemm_group([jw, jd], [3, 1], scr),
        "attn_self_bwd": lambda: ops.attn_self_bwd(qkv, probs, dctx, B=B, S=S, H=H, dh=dh),
        "attn_self_fwd": lambda: ops.attn_self_fwd(qkv, ids, 1, B=B, S=S, H=H, dh=dh),
        "wgrad GEMM 512x512x50 (both m-major)": lambda: ops.gemm(Awg, Bwg, M=E, N=E, K=50, a_kmajor=False, b_kmajor=False, out=owg)}
torch.cuda.synchronize()
total = 0
for name, fn in aggs.items():
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
    t0, n = time.time(), 0
    while time.time() - t0 < SECS:
        now = pk.pkv_run(C.c_void_p(sv.cuda_stream), 50)
        n += 50
    stop = True
    [t.join() for t in ths]
    print(f"  synthetic victim beside {name:40s}: {n:6d} launches, {now - total:6d} float4 differ", flush=True)
    total = now
