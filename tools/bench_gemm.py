"""GPU micro-benchmark of the GEMM shapes one cfg2 train step issues (avg us per launch, back-to-back)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sign-language-nlp_amd")]
import torch
from slnlp import ops

SHAPES = [  # (label, layout, M, N, K)
    ("enc in_proj fwd", "fwd", 2400, 1536, 512), ("enc out/ffn fwd", "fwd", 2400, 512, 512),
    ("dec kv fwd", "fwd", 2400, 1024, 512), ("dec small fwd", "fwd", 50, 512, 512), ("generator fwd", "fwd", 50, 202, 512),
    ("enc in_proj dgrad", "dgrad", 2400, 512, 1536), ("enc dgrad", "dgrad", 2400, 512, 512),
    ("dec small dgrad", "dgrad", 50, 512, 512),
    ("enc in_proj wgrad", "wgrad", 1536, 512, 2400), ("enc wgrad", "wgrad", 512, 512, 2400),
    ("dec kv wgrad", "wgrad", 1024, 512, 2400), ("dec small wgrad", "wgrad", 512, 512, 50),
]

def main():
    prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    planes = len(sys.argv) > 2 and sys.argv[2] == "planes"
    for label, layout, M, N, K in SHAPES:
        a_k = layout in ("fwd", "dgrad"); b_k = layout == "fwd"
        A = torch.randn((M, K) if a_k else (K, M), device="cuda")
        B = torch.randn((N, K) if b_k else (K, N), device="cuda")
        out = torch.empty(M, N, device="cuda")
        rs = torch.empty(M, device="cuda") if layout == "wgrad" else None
        if planes:
            Ap, Bp = ops.split_planes(A), ops.split_planes(B)
            f = lambda: ops.gemm_planes(Ap, Bp, M=M, N=N, K=K, a_kmajor=a_k, b_kmajor=b_k, out=out, rowsum_a=rs, precision=prec)
        else:
            f = lambda: ops.gemm(A, B, M=M, N=N, K=K, a_kmajor=a_k, b_kmajor=b_k, out=out, rowsum_a=rs, precision=prec)
        for _ in range(10): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        R = 200
        e0.record()
        for _ in range(R): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / R * 1e3
        print(f"{label:22s} {layout:6s} M{M:5d} N{N:5d} K{K:5d}  {us:8.1f} us  {2*M*N*K/us/1e6:8.1f} TFLOP/s(alg)")

if __name__ == "__main__":
    main()
