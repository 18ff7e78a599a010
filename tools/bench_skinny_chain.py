"""A dependent chain of the decoder's 50-row GEMMs (y_i = relu(y_{i-1} W_i^T + b_i) + y_{i-1}, 12 distinct weights): device time per
launch when every launch waits for the previous one -- what these launches cost inside a train step.

    [SLNLP_PROBE_LIB=k] python tools/bench_skinny_chain.py [rows] [E]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sign-language-nlp_amd")]
import torch
from slnlp import ops
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50
E = int(sys.argv[2]) if len(sys.argv) > 2 else 512
g = torch.Generator().manual_seed(0)
Ws = [(torch.randn(E, E, generator=g) / E ** 0.5).cuda() for _ in range(12)]
bs = [torch.zeros(E).cuda() for _ in range(12)]
ys = [torch.randn(rows, E, generator=g).cuda() for _ in range(13)]

def chain():
    for i in range(12):
        ops.gemm(ys[i], Ws[i], M=rows, N=E, K=E, a_kmajor=True, b_kmajor=True, out=ys[i + 1], bias=bs[i], resid=ys[i])

for _ in range(5): chain()
torch.cuda.synchronize()
st = torch.cuda.Stream()
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(st):
    chain()
    st.synchronize()
    with torch.cuda.graph(graph, stream=st):       # host-side argument marshalling would dominate: replay a captured chain
        for _ in range(10): chain()
    for _ in range(3): graph.replay()
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20): graph.replay()
    e1.record(st); st.synchronize()
print(f"lib {os.environ.get('SLNLP_PROBE_LIB', 'product'):8s} rows {rows} E {E}: {e0.elapsed_time(e1) / 2400 * 1e3:6.2f} us per dependent launch", flush=True)
