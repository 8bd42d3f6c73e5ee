"""fc logits product (M = T*B rows, 29 classes): the skinny kernel against the tile kernel (DS2_GEMM_SKINNY=0), same box."""
import os, subprocess, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def child():
    from asr_amd import ops
    out = {}
    for name, M, N, K in (("c3 fc", 32064, 29, 1024), ("c2 fc", 16000, 29, 768), ("c4 fc", 24000, 29, 1280)):
        torch.manual_seed(M); A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05
        C = ops.gemm(A, W, transB=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): ops.gemm(A, W, transB=True, out=C)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        out[name] = (us, M * K * 4 / us / 1e6, C.double().sum().item())
        dl = torch.randn(M, N, device="cuda") * 0.01; Wk = torch.randn(N, K, device="cuda")
        D = ops.gemm(dl, Wk)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): ops.gemm(dl, Wk, out=D)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        out[name + " dX"] = (us, M * K * 4 / us / 1e6, D.double().sum().item())
        G = ops.gemm(dl, A, transA=True)                       # dW = dLogits^T Xn (split-K + ordered reduce)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): ops.gemm(dl, A, transA=True, out=G)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        out[name + " dW"] = (us, M * K * 4 / us / 1e6, G.double().sum().item())
    print(repr(out))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(); sys.exit(0)
res = {}
for rep in range(2):
    for v in ("0", "1"):
        env = dict(os.environ, DS2_EXPERIMENTAL="1", DS2_GEMM_SKINNY=v)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        if r.returncode: print(r.stderr[-2000:]); sys.exit(1)
        res[v] = eval(r.stdout.strip().splitlines()[-1])
for k in res["0"]:
    a, b = res["0"][k], res["1"][k]
    print(f"{k:8s} tile kernel {a[0]:7.1f} us ({a[1]:5.2f} TB/s of A | C)   skinny {b[0]:7.1f} us ({b[1]:5.2f} TB/s)   same checksum: {a[2] == b[2]}")
