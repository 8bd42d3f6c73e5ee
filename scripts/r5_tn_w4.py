"""Four-wave grouped TN GEMM (gemm_tn_w4.h, default) against the 8-wave kernel (DS2_GEMM_W4=0): bit-identity on one c3 / c2 / c4 layer's weight-
gradient problem lists + ragged shapes, then speed (interleaved processes).   python scripts/r5_tn_w4.py [check|time]"""
import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def layer_problems(G, H, B, T, I, g):
    M = T * B
    dgx = (torch.randn(M, 2 * G * H, device="cuda", generator=g) * 0.1).bfloat16()
    dhn = (torch.randn(M, 2 * H, device="cuda", generator=g) * 0.1).bfloat16()
    h = torch.randn(M, 2 * H, device="cuda", generator=g).bfloat16()
    xn = torch.randn(M, I, device="cuda", generator=g).bfloat16()
    dwih = torch.empty(2 * G * H, I, device="cuda")
    dwhh = torch.empty(2, G * H, H, device="cuda")
    if G == 3:
        rows = 2 * H
        probs = [(dgx, xn, dwih), (dgx[B:, 0:rows], h[:M - B, 0:H], dwhh[0, :rows]), (dgx[:M - B, G * H:G * H + rows], h[B:, H:2 * H], dwhh[1, :rows]),
                 (dhn[B:, 0:H], h[:M - B, 0:H], dwhh[0, rows:]), (dhn[:M - B, H:2 * H], h[B:, H:2 * H], dwhh[1, rows:])]
    else:
        probs = [(dgx, xn, dwih), (dgx[B:, 0:G * H], h[:M - B, 0:H], dwhh[0]), (dgx[:M - B, G * H:], h[B:, H:2 * H], dwhh[1])]
    return probs, (dwih, dwhh)

CASES = [("c3 layer", 3, 1024, 64, 501, 1024, 4), ("c3 layer 0", 3, 1024, 64, 501, 1312, 4), ("c2 layer", 3, 768, 32, 500, 768, 4), ("c4 layer", 4, 1280, 32, 750, 1280, 4),
         ("ragged", 3, 264, 16, 400, 200, 3)]

def child(mode):
    from asr_amd import ops
    out = {}
    for name, G, H, B, T, I, sk in CASES:
        g = torch.Generator(device="cuda"); g.manual_seed(7 + H + T)
        probs, (dwih, dwhh) = layer_problems(G, H, B, T, I, g)
        fl = sum(2.0 * a.shape[0] * a.shape[1] * b.shape[1] for a, b, _ in probs)
        ops.gemm_bf16_tn_splitk_group(probs, splitk=sk)
        torch.cuda.synchronize()
        if mode == "check":
            a = probs[0][0].float().t() @ probs[0][1].float()
            err = ((dwih - a).norm() / a.norm()).item()
            s1 = (dwih.view(torch.int32).to(torch.int64).sum().item(), dwhh.view(torch.int32).to(torch.int64).sum().item())
            for rep in range(3):
                ops.gemm_bf16_tn_splitk_group(probs, splitk=sk)
                s2 = (dwih.view(torch.int32).to(torch.int64).sum().item(), dwhh.view(torch.int32).to(torch.int64).sum().item())
                assert s1 == s2, f"{name}: run {rep} differs"
            out[name] = (err, s1)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3): ops.gemm_bf16_tn_splitk_group(probs, splitk=sk)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10): ops.gemm_bf16_tn_splitk_group(probs, splitk=sk)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            out[name] = (ms * 1e3, fl / ms / 1e9)
        del probs, dwih, dwhh
    print(repr(out))

def main():
    global mode
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2]); sys.exit(0)
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    def run(w4, order="1", child_mode=None):
        env = dict(os.environ, DS2_GEMM_W4=w4, DS2_TN_ORDER=order, DS2_EXPERIMENTAL="1")
        r = subprocess.run([sys.executable, __file__, "child", child_mode or mode], env=env, capture_output=True, text=True)
        if r.returncode != 0: print(r.stdout[-2000:], r.stderr[-3000:]); sys.exit(1)
        return eval(r.stdout.strip().splitlines()[-1])
    if mode == "check":
        a, b = run("0"), run("1")
        ok = True
        for k in a:
            same = a[k][1] == b[k][1]
            ok &= same and b[k][0] < 1e-5
            print(f"{k:12s} 8-wave err {a[k][0]:.2e}  4-wave err {b[k][0]:.2e}  bit-identical checksums: {same}")
        print("TN W4 CHECK", "PASS" if ok else "FAIL")
    elif mode == "order":
        # item order of the four-wave kernel: XCD runs of 32 consecutive items (default) against DS2_TN_ORDER=0 (a contiguous run of every slice per XCD)
        a, b = run("1", "0", "check"), run("1", "1", "check")
        print("bit-identical checksums across orders:", all(a[k][1] == b[k][1] for k in a))
        res = {"0": [], "1": []}
        for rep in range(3):
            for v in ("0", "1"): res[v].append(run("1", v, "time"))
        for k in res["0"][0]:
            t0 = sorted(r[k][0] for r in res["0"])[1]; t1 = sorted(r[k][0] for r in res["1"])[1]
            print(f"{k:12s} per-slice order {t0:7.1f} us   XCD runs of 32 {t1:7.1f} us  x{t1 / t0:.3f}   (GEMM + reduce launch)")
    else:
        res = {"0": [], "1": []}
        for rep in range(3):
            for v in ("0", "1"): res[v].append(run(v))
        for k in res["0"][0]:
            t0 = sorted(r[k][0] for r in res["0"])[1]; t1 = sorted(r[k][0] for r in res["1"])[1]
            f0 = sorted(r[k][1] for r in res["0"])[1]; f1 = sorted(r[k][1] for r in res["1"])[1]
            print(f"{k:12s} 8-wave {t0:7.1f} us ({f0:5.0f} TF/s)   4-wave {t1:7.1f} us ({f1:5.0f} TF/s)  x{t1 / t0:.3f}   (GEMM + reduce launch)")

if __name__ == "__main__":
    main()
