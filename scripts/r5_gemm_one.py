"""one NT GEMM shape, a few launches (for PMC passes): python scripts/r5_gemm_one.py M N K [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops
M, N, K = (int(a) for a in sys.argv[1:4]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
C = ops.gemm_bf16_nt(A, B)
for _ in range(reps): ops.gemm_bf16_nt(A, B, out=C)
torch.cuda.synchronize()
