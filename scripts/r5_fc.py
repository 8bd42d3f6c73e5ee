"""fc block's logits GEMM (M = T*B, N = 29, K = 1024, fp32): time with the split-K rule of ops.gemm_raw.  python scripts/r5_fc.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops
dev = torch.device("cuda:0")
M, N, K = 32064, 29, 1024
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev)
out = torch.empty(M, N, device=dev)
for sk in (1, 2, 4, 8, 0):
    f = lambda: ops.gemm_raw(False, True, M, N, K, A.data_ptr(), K, 0, W.data_ptr(), K, 0, out.data_ptr(), N, 0, dev, splitk=sk)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ref = A.double() @ W.double().t()
    print(f"splitk={sk}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  err {((out.double() - ref).norm() / ref.norm()).item():.2e}")
