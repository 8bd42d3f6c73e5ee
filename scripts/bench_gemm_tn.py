"""TN-form bf16 GEMM (C = A^T B, reduction index on the rows of both operands): parity against torch and speed against the
NT kernel fed with pre-transposed operands.  GPU only.   python scripts/bench_gemm_tn.py [--quick]"""
import sys
import torch
from asr_amd import ops


def check(K, M, N, lda=None, ldb=None, splitk=0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    lda = lda or M
    ldb = ldb or N
    A = torch.randn(K, lda, device="cuda", generator=g).bfloat16()
    B = torch.randn(K, ldb, device="cuda", generator=g).bfloat16()
    Av, Bv = A[:, :M], B[:, :N]
    out = ops.gemm_bf16_tn(Av, Bv, splitk=splitk)
    ref = Av.float().t() @ Bv.float()
    err = (out - ref).abs().max().item()
    scale = ref.abs().max().item()
    nt = ops.gemm_bf16_nt(Av.t().contiguous(), Bv.t().contiguous(), splitk=1 if splitk == 0 else splitk)
    same = (out - nt).abs().max().item()
    ok = err <= 2e-3 * scale + 1e-3
    print(f"K={K} M={M} N={N} lda={lda} ldb={ldb} splitk={splitk}: max err {err:.3e} (scale {scale:.1f}) vs NT {same:.3e} {'ok' if ok else 'FAIL'}")
    return ok


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    ok = True
    for args in [(64, 256, 256), (128, 256, 512), (200, 264, 328), (72, 8, 8), (640, 96, 192), (1000, 520, 1312), (4096, 512, 256, 600, 304),
                 (5000, 768, 256, 1024, 512, 3)]:
        ok &= check(*args)
    # pair with shifted rows / column blocks, as dW_hh uses it
    T, B, H, G = 20, 16, 64, 3
    M = T * B
    g = torch.Generator(device="cuda").manual_seed(1)
    dgx = torch.randn(M, 2 * G * H, device="cuda", generator=g).bfloat16()
    h = torch.randn(M, 2 * H, device="cuda", generator=g).bfloat16()
    out = torch.empty(2, 2 * H, H, device="cuda")
    ops.gemm_bf16_tn_pair(dgx[B:, 0:2 * H], dgx[:M - B, G * H:G * H + 2 * H], h[:M - B, 0:H], h[B:, H:2 * H], out)
    r0 = dgx[B:, 0:2 * H].float().t() @ h[:M - B, 0:H].float()
    r1 = dgx[:M - B, G * H:G * H + 2 * H].float().t() @ h[B:, H:2 * H].float()
    e = max((out[0] - r0).abs().max().item(), (out[1] - r1).abs().max().item())
    print(f"pair: max err {e:.3e} {'ok' if e < 1e-2 else 'FAIL'}")
    ok &= e < 1e-2
    if "--quick" not in sys.argv:
        for (K, Mm, N) in [(32064, 6144, 2048), (32064, 6144, 1312), (32000, 2048, 1024)]:
            A = torch.randn(K, Mm, device="cuda").bfloat16()
            Bm = torch.randn(K, N, device="cuda").bfloat16()
            At, Bt = A.t().contiguous(), Bm.t().contiguous()
            o1 = torch.empty(Mm, N, device="cuda")
            o2 = torch.empty(Mm, N, device="cuda")
            t_tn = timeit(lambda: ops.gemm_bf16_tn(A, Bm, out=o1))
            t_nt = timeit(lambda: ops.gemm_bf16_nt(At, Bt, out=o2))
            fl = 2.0 * K * Mm * N
            print(f"K={K} M={Mm} N={N}: TN {t_tn * 1e3:.0f} us ({fl / t_tn / 1e9:.0f} TF/s)   NT {t_nt * 1e3:.0f} us ({fl / t_nt / 1e9:.0f} TF/s)   "
                  f"max diff {(o1 - o2).abs().max().item():.3e}")
            zA, zB = torch.zeros_like(A), torch.zeros_like(Bm)
            print(f"   zero operands: TN {timeit(lambda: ops.gemm_bf16_tn(zA, zB, out=o1)) * 1e3:.0f} us")
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
