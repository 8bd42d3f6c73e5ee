"""Persistent (one workgroup per CU, several tiles each) against one-workgroup-per-tile form of the 256 x 256 NT GEMM: run with DS2_GEMM_PERS=1|0."""
import os, sys, torch
sys.path.insert(0, os.environ.get("REPO", "."))
from asr_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, M, N, K in [("fwd Gx", 32064, 6144, 1024), ("dXn", 32064, 1024, 6144), ("dXn layer 0", 32064, 1312, 6144)]:
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, device=dev) if N % 4 == 0 else None
    out = ops.gemm_bf16_nt(A, B, bias=bias, splitk=1)
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): ops.gemm_bf16_nt(A, B, bias=bias, out=out, splitk=1)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    print(f"{name} M={M} N={N} K={K}: {best*1e3:.1f} us  {2*M*N*K/best/1e9:.0f} TF/s", flush=True)
