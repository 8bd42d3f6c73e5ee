import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from asr_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in [(8192, 8192, 8192), (4096, 4096, 4096), (6144, 1024, 4096), (6144, 1024, 32064)]:
    A = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16(); B = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    if os.environ.get("GEMM_ZERO"): A.zero_(); B.zero_()
    for sk in ([1] if K <= 8192 else [1, 8]):
        out = ops.gemm_bf16_nt(A, B, splitk=sk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): ops.gemm_bf16_nt(A, B, out=out, splitk=sk)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"M={M} N={N} K={K} splitk={sk}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.0f} TF/s", flush=True)
