"""One GEMM shape, few launches -- target for rocprofv3 --pmc runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sign-language-nlp_amd")]
import torch
from slnlp import ops
layout, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
prec = int(sys.argv[5]) if len(sys.argv) > 5 else 3
a_k = layout in ("fwd", "dgrad"); b_k = layout == "fwd"
A = torch.randn((M, K) if a_k else (K, M), device="cuda"); B = torch.randn((N, K) if b_k else (K, N), device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(20):
    ops.gemm(A, B, M=M, N=N, K=K, a_kmajor=a_k, b_kmajor=b_k, out=out, precision=prec)
torch.cuda.synchronize()
