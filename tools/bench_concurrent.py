"""How much does running K independent fits concurrently on ONE GPU (one stream + one hipGraph each)
raise aggregate train throughput?  (grid-search workload: SURVEY.md section 7.5)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sign-language-nlp_amd")]
import torch
import bench
from slnlp import synth, tf_engine as te

def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    steps = 40
    c = dict(bench.WORKLOADS[wl], precision=3)
    B, S = c["B"], c["S"]
    for K in (1, 2, 3, 4):
        engs, streams, data = [], [], []
        for k in range(K):
            cfg, sd = bench.build_sd(c, seed=1 + k)
            e = te.TransformerEngine(cfg, seed=1 + k); e.load_state(sd); e.set_lr(0.01)
            Xn, Ln, yn = synth.make_batch(20 * B, S, c["Vs"], c["Vt"], seed=1 + k)
            engs.append(e); streams.append(torch.cuda.Stream()); data.append((torch.from_numpy(Xn).cuda(), torch.from_numpy(yn).cuda()))
        import threading
        for mode in ("graph", "eager", "eager-threads"):
            def one(e, s, X, y, n):
                with torch.cuda.stream(s):
                    for i in range(n):
                        j = (i % 20) * B
                        (e.train_step_graph if mode == "graph" else e.train_step)(X[j:j + B], y[j:j + B], 0.9, 0.5)
            def run(n):
                if mode == "eager-threads":      # one host thread per fit (ctypes releases the GIL during the C call)
                    th = [threading.Thread(target=one, args=(e, s, X, y, n)) for e, s, (X, y) in zip(engs, streams, data)]
                    [t.start() for t in th]; [t.join() for t in th]
                    return
                for i in range(n):
                    j = (i % 20) * B
                    for e, s, (X, y) in zip(engs, streams, data):
                        with torch.cuda.stream(s):
                            (e.train_step_graph if mode == "graph" else e.train_step)(X[j:j + B], y[j:j + B], 0.9, 0.5)
            run(5); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f"{wl}: {K} concurrent fits [{mode:13s}]: {K * steps * B / dt:9.1f} seq/s aggregate  ({dt / steps * 1e3:.2f} ms per round of {K} steps)", flush=True)
        del engs, streams, data
        torch.cuda.empty_cache()

main()
