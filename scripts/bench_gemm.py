"""GPU micro-benchmark of the bf16 NT GEMM on the train step's three shapes (run with DS2_GEMM_TILE=128|glds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops
dev = torch.device("cuda:0")
shapes = [("fwd Gx   ", 32064, 6144, 1024), ("dXn      ", 32064, 1024, 6144), ("dW_ih    ", 6144, 1024, 32064), ("dW_hh rz ", 2048, 1024, 32000)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
    for sk in ([0] if K < 8000 else [1, 2, 4, 8]):
        out = ops.gemm_bf16_nt(A, B, splitk=sk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): ops.gemm_bf16_nt(A, B, out=out, splitk=sk)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{name} M={M} N={N} K={K} splitk={sk}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TF/s", flush=True)
