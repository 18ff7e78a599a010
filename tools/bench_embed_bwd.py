import sys, torch
sys.path[:0] = [".", "sign-language-nlp_amd"]
from slnlp import ops, synth
B, S, E, V = 50, 48, 512, 3000
Xn, Ln, yn = synth.make_batch(B, S, V, 202, seed=1)
ids = torch.from_numpy(Xn).cuda()
dx = torch.randn(S * B, E, device="cuda")
rng = ops.make_rng(seed=1, step=0)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("embed_bwd p=0.1:", round(t(lambda: ops.embed_bwd(ids, dx, B=B, S=S, V=V, drop_p=0.1, drop_site=1, rng=rng)), 1), "us")
print("embed_bwd p=0  :", round(t(lambda: ops.embed_bwd(ids, dx, B=B, S=S, V=V)), 1), "us")
