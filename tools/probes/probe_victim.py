"""Which single kernel is a victim?  Stream V runs ONE kernel of the library over and over on CONSTANT inputs and compares
every output with its first result; two other streams run whole fits of the library (slnlp_set_stream_policy(0)).  A kernel
whose output changes although its inputs never do is corrupted on the consumer side (no producer involved); if none does,
the corruption needs a producer -> consumer chain.   python tools/probes/probe_victim.py [seconds per victim]"""
import os, sys, threading, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
warnings.filterwarnings("ignore")
import torch
import bench
from slnlp import ops, synth, tf_engine as te, _lib
dev = torch.device("cuda", 0)
_lib.load().slnlp_set_stream_policy(0)
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
M, E = 2400, 512
# ---- victims: closures returning a tuple of output tensors
dy, x, gamma = rnd(M, E), rnd(M, E), rnd(E)
_, stats = ops.layernorm_fwd(x, gamma, rnd(E))
rng = ops.make_rng(3, 0)
def v_ln_bwd():
    return ops.layernorm_bwd(dy, x, gamma, stats, want_drop=True, drop_p=0.1, drop_site=5, rng=rng)[:2]
def v_ln_bwd_nodrop():
    return ops.layernorm_bwd(dy, x, gamma, stats)[:1]
def v_ln_bwd_params():
    return ops.layernorm_bwd(dy, x, gamma, stats)[2:4]
def v_ln_fwd():
    return ops.layernorm_fwd(x, gamma, gamma)
dY, X, W = rnd(M, E), rnd(M, E), rnd(E, E)
dYp, Xp, Wp = ops.split_planes(dY), ops.split_planes(X), ops.split_planes(W)
rs = torch.empty(E, device=dev)
jw, dW = ops.plane_job(dYp, Xp, M=E, N=E, K=M, a_kmajor=False, b_kmajor=False, rowsum_a=rs)
jd, dX = ops.plane_job(dYp, Wp, M=M, N=E, K=E, a_kmajor=True, b_kmajor=False)
scr = ops.gemm_group([jw, jd], [3, 1])
def v_plane_group():
    ops.gemm_group([jw, jd], [3, 1], scr)
    return dW.clone(), dX.clone(), rs.clone()
A50, W50 = rnd(50, E), rnd(E, E)
def v_brow_gemm():
    return (ops.gemm(A50, W50, M=50, N=E, K=E),)
def v_chain():                     # LayerNorm -> plane GEMM (fresh planes of the LN output) -> LayerNorm backward of the product
    y, st = ops.layernorm_fwd(x, gamma, gamma)
    yp = ops.split_planes(y)
    out = ops.gemm_planes(yp, Wp, M=M, N=E, K=E)
    return ops.layernorm_bwd(out, x, gamma, st)[:1]
Ba, Sa, Ha, dha = 50, 48, 8, 64
qkv_a, dctx_a = rnd(Sa * Ba, 3 * E), rnd(Sa * Ba, E)
ids_a = torch.randint(2, 100, (Ba, Sa), generator=g).to(dev)
probs_a = ops.attn_self_fwd(qkv_a, ids_a, 1, B=Ba, S=Sa, H=Ha, dh=dha)[1]
def v_attn_fwd():          # (the two kernels of the library that still contain v_pk_mov_b32 ... op_sel)
    return ops.attn_self_fwd(qkv_a, ids_a, 1, B=Ba, S=Sa, H=Ha, dh=dha)[:2]
def v_attn_bwd():
    r = ops.attn_self_bwd(qkv_a, probs_a, dctx_a, B=Ba, S=Sa, H=Ha, dh=dha)
    return r if isinstance(r, (tuple, list)) else (r,)
VICTIMS = [("attn_self_fwd", v_attn_fwd), ("attn_self_bwd", v_attn_bwd), ("layernorm_bwd (constant inputs)", v_ln_bwd), ("layernorm_bwd, no dropout: dx", v_ln_bwd_nodrop),
           ("layernorm_bwd: dgamma, dbeta", v_ln_bwd_params), ("layernorm_fwd", v_ln_fwd), ("plane GEMM group, split-K 3", v_plane_group),
           ("50-row fp32-operand GEMM", v_brow_gemm), ("chain LN -> split -> plane GEMM -> LN bwd", v_chain)]
# ---- aggressors: whole fits
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
stV = torch.cuda.Stream()
REF = {}
def run_victim(fn, secs, name=None, describe=False):
    with torch.cuda.stream(stV):
        if name not in REF:
            REF[name] = [t.clone() for t in fn()]            # (the reference is taken while the victim runs ALONE)
        ref = REF[name]
        bad = torch.zeros(1, dtype=torch.int64, device=dev)
        n, t0, shown = 0, time.time(), 0
        while time.time() - t0 < secs:
            for _ in range(20):
                out = fn()
                for k, (a, b) in enumerate(zip(ref, out)):
                    ne = a != b
                    bad += ne.any()
                    if describe and shown < 3 and bool(ne.any()):
                        shown += 1
                        idx = torch.nonzero(ne.reshape(-1)).flatten()
                        w = a.shape[-1]
                        rows = torch.unique(idx // w)
                        d = (a.reshape(-1)[idx] - b.reshape(-1)[idx]).abs()
                        print(f"      output {k}: {idx.numel()} elements in {rows.numel()} rows (first rows {rows[:8].tolist()}), max |d| {float(d.max()):.3e}, "
                              f"|ref| max {float(a.abs().max()):.3e}; first: ref {a.reshape(-1)[idx[:3]].tolist()} got {b.reshape(-1)[idx[:3]].tolist()}", flush=True)
                n += 1
            stV.synchronize()
        return n, int(bad)
INPUTS = {"dy": dy, "x": x, "gamma": gamma, "stats": stats, "rng": rng, "dY planes hi": dYp[0], "W planes hi": Wp[0], "A50": A50, "W50": W50}
torch.cuda.synchronize()
before = {k: v.clone() for k, v in INPUTS.items()}
def check_inputs(tag):
    torch.cuda.synchronize()
    for k, v in INPUTS.items():
        if not torch.equal(v, before[k]):
            a, b = before[k].reshape(-1), v.reshape(-1)
            idx = torch.nonzero(a != b).flatten()
            print(f"   !! INPUT `{k}` CHANGED {tag}: {idx.numel()} of {a.numel()} elements, first at {idx[:6].tolist()}, was {a[idx[:3]].tolist()} now {b[idx[:3]].tolist()}", flush=True)
for name, fn in VICTIMS:
    n, bad = run_victim(fn, min(SECS, 3.0), name)
    print(f"alone        {name:44s}: {n:6d} runs, {bad} differ", flush=True)
check_inputs("after the solo runs")
ths = [threading.Thread(target=aggress, args=(k,)) for k in (0, 1)]
[t.start() for t in ths]
try:
    for name, fn in VICTIMS:
        n, bad = run_victim(fn, SECS, name, describe=True)
        print(f"beside 2 fits {name:44s}: {n:6d} runs, {bad} differ", flush=True)
        check_inputs(f"while `{name}` ran beside the fits")
finally:
    stop = True
    [t.join() for t in ths]
check_inputs("at the end")
