"""Race / correctness diagnosis of one NT GEMM variant (DS2_GEMM_RING from the environment): run-to-run identity and where the differences sit.
python scripts/r5_w4_diag.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops
dev = torch.device("cuda:0")
SHAPES = [("fwd K=1024", 32064, 6144, 1024, True), ("dX  K=6144", 32064, 1024, 6144, False), ("c2 fwd", 16032, 4608, 768, True), ("small", 2048, 2048, 256, True),
          ("k128", 8192, 4096, 128, False), ("k192", 8192, 4096, 192, False)]
print("variant", os.environ.get("DS2_GEMM_RING"))
for name, M, N, K, hb in SHAPES:
    g = torch.Generator(device=dev); g.manual_seed(1234 + M + N + K)
    A = torch.randn(M, K, device=dev, generator=g).bfloat16(); B = torch.randn(N, K, device=dev, generator=g).bfloat16()
    bias = torch.randn(N, device=dev, generator=g) if hb else None
    rows = torch.randint(0, M, (256,), device=dev, generator=g)
    ref = A[rows].double() @ B.double().T + (bias.double() if hb else 0)
    C = ops.gemm_bf16_nt(A, B, bias=bias)
    err = ((C[rows].double() - ref).abs().max() / ref.abs().max()).item()
    nbad = 0
    for rep in range(6):
        C2 = ops.gemm_bf16_nt(A, B, bias=bias)
        d = (C != C2)
        n = int(d.sum().item())
        if n:
            nbad += 1
            idx = d.nonzero()
            tiles = torch.unique(torch.stack([idx[:, 0] // 256, idx[:, 1] // 256], 1), dim=0)
            r16 = torch.unique((idx[:, 0] % 256) // 16); c16 = torch.unique((idx[:, 1] % 256) // 16)
            print(f"  {name}: rep {rep}: {n} elements differ in {len(tiles)} tiles (first {tiles[:6].tolist()}), row blocks {r16.tolist()[:16]}, col blocks {c16.tolist()[:16]}, max |d| {(C - C2).abs().max().item():.3e}")
            C = C2
    print(f"{name:12s} M={M} N={N} K={K}: spot err {err:.2e}; runs differing {nbad}/6", flush=True)
