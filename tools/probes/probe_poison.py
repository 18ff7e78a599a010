"""ONE fit (forward + backward, cfg2-like, 2 layers) next to a foreign LDS-poisoning load on another stream (tools/probes/
poison.hip): does the fit's result depend on what other kernels leave in LDS?  Prints which workspace buffers differ from the
solo run, like probe_concurrent5.py.   python tools/probes/probe_poison.py [pattern hex] [reps]"""
import bisect, ctypes as C, os, sys, threading, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "sign-language-nlp_amd")):
    sys.path.insert(0, p)
warnings.filterwarnings("ignore")
import torch
import bench
from slnlp import synth, tf_engine as te, _lib
dev = torch.device("cuda", 0)
_lib.load().slnlp_set_stream_policy(0)
pattern = int(sys.argv[1], 16) if len(sys.argv) > 1 else 0x7FC00000
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
poison = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpoison.so"))
poison.poison_launch.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.c_int, C.c_void_p]
c = dict(E=512, H=8, N=2, F=512, Vs=3000, Vt=202, B=50, S=48, dropout=0.1, precision=3)
cfg, sd = bench.build_sd(c, seed=1)
Xn, _, yn = synth.make_batch(c["B"], c["S"], c["Vs"], c["Vt"], seed=1)
X, y = torch.from_numpy(Xn).to(dev), torch.from_numpy(yn).to(dev)
e = te.TransformerEngine(cfg, device=dev, seed=1)
buf = C.create_string_buffer(1 << 16)
_lib.check(_lib.load().slnlp_tf_debug_layout(C.byref(e.cfg), buf, len(buf)), "layout")
lay = [(l.split()[0], int(l.split()[1])) for l in buf.value.decode().strip().split("\n")]
names, offs = [n for n, _ in lay], [o for _, o in lay]
act_end = dict(lay)["wp.hi"]
st_fit, st_poison = torch.cuda.Stream(), torch.cuda.Stream()
sink = torch.zeros(4, dtype=torch.int32, device=dev)

def reset():
    e.load_state(sd); e.grads.zero_(); e.momentum.zero_(); e.rng[1] = 0; e.workspace[:act_end].zero_()
    _lib.load().slnlp_tf_params_changed(e.handle)
    torch.cuda.synchronize()

def fit():
    with torch.cuda.stream(st_fit):
        e.forward(X, y, train=True); e.backward()
        st_fit.synchronize()
    return e.workspace[:act_end].clone(), e.grads.clone()

reset(); solo = fit()
reset(); solo2 = fit()
print("solo repeat identical:", torch.equal(solo[0], solo2[0]) and torch.equal(solo[1], solo2[1]), flush=True)
stop = False
def hammer():
    torch.cuda.set_device(dev)
    while not stop:
        for _ in range(32):
            poison.poison_launch(st_poison.cuda_stream, pattern, 512, 400, sink.data_ptr())
        st_poison.synchronize()
th = threading.Thread(target=hammer); th.start()
try:
    for rep in range(REPS):
        reset(); got = fit()
        ne = solo[0] != got[0]
        if not int(ne.sum()) and torch.equal(solo[1], got[1]):
            print(f"rep {rep}: identical to solo next to the poison load (pattern {pattern:#x})", flush=True)
            continue
        idx = torch.nonzero(solo[0].view(torch.int32) != got[0].view(torch.int32)).flatten().tolist()
        seen = {}
        for i in idx:
            k = bisect.bisect_right(offs, i * 4) - 1
            seen.setdefault(names[k], 0); seen[names[k]] += 1
        print(f"rep {rep}: DIFFERS, grads equal {torch.equal(solo[1], got[1])}: " + ", ".join(f"{n} {seen[n]}" for n in names if n in seen), flush=True)
finally:
    stop = True; th.join()
