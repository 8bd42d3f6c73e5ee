"""Reference point for the hand-written bf16 GEMMs: the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the same shapes.
GPU only.   PYTHONPATH=. python scripts/bench_gemm_lib.py"""
import torch
from asr_amd import ops


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    M = 32064
    for name, (m, n, k) in {"fwd Gx (K=1024)": (M, 6144, 1024), "fwd Gx (K=1312)": (M, 6144, 1312), "dX (K=6144)": (M, 1024, 6144),
                            "dW_ih (K=T*B)": (6144, 1024, M), "dW_hh (K=T*B)": (2048, 1024, M - 64)}.items():
        A = torch.randn(m, k, device="cuda").bfloat16()
        B = torch.randn(n, k, device="cuda").bfloat16()
        Bt = B.t().contiguous()
        out = torch.empty(m, n, device="cuda")
        t_nt = timeit(lambda: ops.gemm_bf16_nt(A, B, out=out))
        t_lib_nt = timeit(lambda: torch.matmul(A, B.t()))            # bf16 output
        t_lib_nn = timeit(lambda: torch.matmul(A, Bt))
        o32 = torch.empty(m, n, device="cuda", dtype=torch.float32)
        fl = 2.0 * m * n * k
        line = f"{name:18s} M={m} N={n} K={k}: ours NT(fp32 out) {t_nt * 1e3:6.0f} us {fl / t_nt / 1e9:5.0f} TF/s | lib NT(bf16 out) {t_lib_nt * 1e3:6.0f} us {fl / t_lib_nt / 1e9:5.0f} TF/s | lib NN {t_lib_nn * 1e3:6.0f} us {fl / t_lib_nn / 1e9:5.0f} TF/s"
        if k >= 16000:
            At, Bt2 = A.t().contiguous(), B.t().contiguous()     # (k, m), (k, n): the TN form
            t_tn = timeit(lambda: ops.gemm_bf16_tn(At, Bt2, out=out))
            t_lib_tn = timeit(lambda: torch.matmul(At.t(), Bt2))
            line += f" | ours TN {t_tn * 1e3:6.0f} us {fl / t_tn / 1e9:5.0f} TF/s | lib TN {t_lib_tn * 1e3:6.0f} us {fl / t_lib_tn / 1e9:5.0f} TF/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
