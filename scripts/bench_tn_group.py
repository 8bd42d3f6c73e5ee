"""Grouped co-resident TN GEMM (ds2_gemm_bf16_tn_group) on one c3 layer's weight-gradient problem list: stand-alone time against round 3's
three launches (256 x 256 TN + split-K reduce), and the time of a K-split backward recurrence with / without the grouped kernel running
beside it on a second stream.   python scripts/bench_tn_group.py [H] [B] [T]"""
import sys
import torch
from asr_amd import ops


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
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 501
    G, I = 3, H
    M = T * B
    g = torch.Generator(device="cuda").manual_seed(0)
    dgx = (torch.randn(M, 2 * G * H, device="cuda", generator=g) * 0.1).bfloat16()
    dhn = (torch.randn(M, 2 * H, device="cuda", generator=g) * 0.1).bfloat16()
    h = torch.randn(M, 2 * H, device="cuda", generator=g).bfloat16()
    xn = torch.randn(M, I, device="cuda", generator=g).bfloat16()
    dwih = torch.empty(2 * G * H, I, device="cuda")
    dwhh = torch.empty(2, G * H, H, device="cuda")
    rows = 2 * H
    probs = [(dgx, xn, dwih), (dgx[B:, 0:rows], h[:M - B, 0:H], dwhh[0, :rows]), (dgx[:M - B, G * H:G * H + rows], h[B:, H:2 * H], dwhh[1, :rows]),
             (dhn[B:, 0:H], h[:M - B, 0:H], dwhh[0, rows:]), (dhn[:M - B, H:2 * H], h[B:, H:2 * H], dwhh[1, rows:])]
    fl = sum(2.0 * a.shape[0] * a.shape[1] * b.shape[1] for a, b, _ in probs)

    def old():
        ops.gemm_bf16_tn_pair(dgx[B:, 0:rows], dgx[:M - B, G * H:G * H + rows], h[:M - B, 0:H], h[B:, H:2 * H], dwhh[:, :rows])
        ops.gemm_bf16_tn_pair(dhn[B:, 0:H], dhn[:M - B, H:2 * H], h[:M - B, 0:H], h[B:, H:2 * H], dwhh[:, rows:])
        ops.gemm_bf16_tn(dgx, xn, out=dwih)

    t_old = timeit(old)
    ref_ih, ref_hh = dwih.clone(), dwhh.clone()
    t_new = timeit(lambda: ops.gemm_bf16_tn_group(probs))
    print(f"H={H} B={B} T={T}: {fl / 1e12:.2f} TFLOP per layer;  round-3 launches {t_old * 1e3:.0f} us ({fl / t_old / 1e9:.0f} TF/s)   "
          f"grouped co-resident {t_new * 1e3:.0f} us ({fl / t_new / 1e9:.0f} TF/s)   max diff ih {(dwih - ref_ih).abs().max().item():.2e} "
          f"hh {(dwhh - ref_hh).abs().max().item():.2e}")
    for wg in (64, 128, 192):
        t = timeit(lambda: ops.gemm_bf16_tn_group(probs, max_workgroups=wg), 4)
        print(f"   {wg} workgroups: {t * 1e3:.0f} us")


if __name__ == "__main__":
    main()
