"""Round 5: the ring NT GEMM (gemm_nt_ring.h, DS2_GEMM_RING=1) against the double-buffered persistent kernel (DS2_GEMM_RING=0):
bit-identity on the train step's shapes + edge shapes, speed interleaved across processes.  Usage: python scripts/r5_ring.py [check|time]"""
import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [("fwd K=1024", 32064, 6144, 1024, True), ("fwd K=1312", 32064, 6144, 1312, True), ("dX  K=6144", 32064, 1024, 6144, False),
          ("edge K=160", 20000, 5004, 160, True), ("c2 fwd", 16032, 4608, 768, True), ("c4 fwd", 24032, 10240, 1280, True)]

def child(mode):
    from asr_amd import ops
    dev = torch.device("cuda:0")
    out = {}
    for name, M, N, K, hb in SHAPES:
        g = torch.Generator(device=dev); g.manual_seed(1234 + M + N + K)
        A = torch.randn(M, K, device=dev, generator=g).bfloat16(); B = torch.randn(N, K, device=dev, generator=g).bfloat16()
        bias = torch.randn(N, device=dev, generator=g) if hb else None
        C = ops.gemm_bf16_nt(A, B, bias=bias)
        if mode == "check":
            # fp64 spot check of rows spread over the matrix (incl. the last rows) + a checksum of everything
            rows = torch.cat([torch.tensor([0, 1, 15, 16, 127, 128, 255, 256, M // 2, M - 257, M - 2, M - 1], device=dev),
                              torch.randint(0, M, (116,), device=dev, generator=g)])
            ref = A[rows].double() @ B.double().T + (bias.double() if hb else 0)
            err = ((C[rows].double() - ref).norm() / ref.norm()).item()
            torch.cuda.synchronize()
            out[name] = (err, C.double().sum().item(), C.view(torch.int32).to(torch.int64).sum().item())
            for rep in range(3):      # run-to-run identity (a race would show here)
                C2 = ops.gemm_bf16_nt(A, B, bias=bias)
                assert torch.equal(C, C2), f"{name}: run {rep} differs"
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3): ops.gemm_bf16_nt(A, B, bias=bias, out=C)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): ops.gemm_bf16_nt(A, B, bias=bias, out=C)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            out[name] = (ms * 1e3, 2.0 * M * N * K / ms / 1e9)
        del A, B, C
    print(repr(out))

if len(sys.argv) > 2 and sys.argv[1] == "child":
    child(sys.argv[2]); sys.exit(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "check"
def run(ring):
    env = dict(os.environ, DS2_GEMM_RING=ring, DS2_EXPERIMENTAL="1")
    r = subprocess.run([sys.executable, __file__, "child", mode], env=env, capture_output=True, text=True)
    if r.returncode != 0: print(r.stdout[-2000:], r.stderr[-3000:]); sys.exit(1)
    return eval(r.stdout.strip().splitlines()[-1])
VARIANTS = os.environ.get("RING_VARIANTS", "0 p q").split()
if mode == "check":
    res = {v: run(v) for v in VARIANTS}
    a = res[VARIANTS[0]]
    ok = True
    for v in VARIANTS[1:]:
        b = res[v]
        for k in a:
            same = a[k][2] == b[k][2] and a[k][1] == b[k][1]
            ok &= b[k][0] < 2e-6 and abs(a[k][1] - b[k][1]) <= 1e-6 * max(1.0, abs(a[k][1])) * 1e3
            print(f"{k:12s} RING={VARIANTS[0]} err {a[k][0]:.2e} RING={v} err {b[k][0]:.2e}  bit-identical checksum: {same}")
    print("RING CHECK", "PASS" if ok else "FAIL")
else:
    res = {v: [] for v in VARIANTS}
    for rep in range(3):
        for v in VARIANTS: res[v].append(run(v))
    for k in res[VARIANTS[0]][0]:
        sh = [s for s in SHAPES if s[0] == k][0]; fl = 2.0 * sh[1] * sh[2] * sh[3]
        line = f"{k:12s}"
        base = sorted(r[k][0] for r in res[VARIANTS[0]])[1]
        for v in VARIANTS:
            t = sorted(r[k][0] for r in res[v])
            line += f"  RING={v} {t[1]:7.1f} us ({fl/t[1]/1e6:5.0f} TF/s, x{t[1]/base:.3f})"
        print(line)
