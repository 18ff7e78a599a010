"""The instrumented LayerNorm-backward copy (tools/probes/lnb_probe.hip) as the victim beside two whole fits: when a dx row
differs from the solo reference, do the checksums of what the wave loaded, the sums it computed and what it read back differ?"""
import ctypes as C, os, sys, threading, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
warnings.filterwarnings("ignore")
import torch
import bench
from slnlp import ops, synth, tf_engine as te, _lib
dev = torch.device("cuda", 0)
_lib.load().slnlp_set_stream_policy(0)
lnb = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblnb.so"))
lnb.lnb_launch.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p]
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
g = torch.Generator().manual_seed(0)
M, E = 2400, 512
dy, x, gamma = [torch.randn(*s, generator=g).to(dev) for s in ((M, E), (M, E), (E,))]
_, stats = ops.layernorm_fwd(x, gamma, gamma)
stV = torch.cuda.Stream()
def run():
    dx = torch.empty(M, E, device=dev)
    diag = torch.zeros(M, 8, dtype=torch.int32, device=dev)
    lnb.lnb_launch(stV.cuda_stream, dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), stats.data_ptr(), M, dx.data_ptr(), diag.data_ptr())
    return dx, diag
with torch.cuda.stream(stV):
    ref_dx, ref_diag = run()
    for _ in range(200):
        a, b = run()
        assert torch.equal(a, ref_dx) and torch.equal(b, ref_diag), "not deterministic alone"
    stV.synchronize()
print("alone: 200 runs identical", flush=True)
c = dict(E=512, H=8, N=2, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.1, precision=3)
engs = []
for s in (2, 3):
    cfg, sd = bench.build_sd(c, seed=s)
    Xn, _, yn = synth.make_batch(c["B"], c["S"], c["Vs"], c["Vt"], seed=s)
    e = te.TransformerEngine(cfg, device=dev, seed=s); e.load_state(sd)
    engs.append((e, torch.from_numpy(Xn).to(dev), torch.from_numpy(yn).to(dev), torch.cuda.Stream()))
torch.cuda.synchronize()
stop = False
def aggress(k):
    torch.cuda.set_device(dev)
    e, Xd, yd, st = engs[k]
    with torch.cuda.stream(st):
        while not stop:
            for _ in range(4):
                e.forward(Xd, yd, train=True); e.backward()
            st.synchronize()
ths = [threading.Thread(target=aggress, args=(k,)) for k in (0, 1)]
[t.start() for t in ths]
n = nbad = shown = 0
kinds = {}
try:
    with torch.cuda.stream(stV):
        t0 = time.time()
        while time.time() - t0 < SECS:
            dx, diag = run()
            stV.synchronize()
            n += 1
            rows = torch.nonzero((dx != ref_dx).any(1)).flatten().tolist()
            drows = torch.nonzero((diag != ref_diag).any(1)).flatten().tolist()
            if not rows and not drows:
                continue
            nbad += 1
            for r in sorted(set(rows) | set(drows)):
                dd = (diag[r] != ref_diag[r]).tolist()
                nel = int((dx[r] != ref_dx[r]).sum())
                key = (nel > 0, dd[0], dd[1], dd[2] or dd[3], dd[4])
                kinds[key] = kinds.get(key, 0) + 1
                if shown < 12:
                    shown += 1
                    cols = torch.nonzero(dx[r] != ref_dx[r]).flatten()
                    print(f"  run {n} row {r}: dx differs in {nel} elements (cols {cols[:1].tolist()}..{cols[-1:].tolist()}); loaded-dy checksum differs {dd[0]}, loaded-x {dd[1]}, "
                          f"sums {dd[2] or dd[3]}, read-back {dd[4]}", flush=True)
finally:
    stop = True
    [t.join() for t in ths]
print(f"beside 2 fits: {n} runs, {nbad} with a difference")
for k, v in sorted(kinds.items(), key=lambda kv: -kv[1]):
    print(f"   {v:6d} rows: dx differs {k[0]}, dy-load checksum differs {k[1]}, x-load checksum differs {k[2]}, sums differ {k[3]}, read-back checksum differs {k[4]}")
