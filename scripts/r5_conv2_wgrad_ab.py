"""conv2 weight gradient (channels-last operands) at the c3 shape: this tree's library against another build (DS2_LIB_PATH), same box.
usage: python scripts/r5_conv2_wgrad_ab.py [other_lib.so]     (child mode: python ... child)"""
import os, subprocess, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child():
    from asr_amd import ops
    B, D1, D2, T = 64, 81, 41, 501
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    a1 = torch.randn(B, D1, T, 32, device="cuda", generator=g).bfloat16()
    dy = (torch.randn(B, D2, T, 32, device="cuda", generator=g) * 0.1).bfloat16()
    lens = torch.randint(300, T + 1, (B,), device="cuda", generator=g, dtype=torch.int32); lens[0] = T
    for b in range(B):                      # the operands the step hands over are zero past each utterance's length
        a1[b, :, lens[b]:] = 0; dy[b, :, lens[b]:] = 0
    dW = torch.empty(32, 32, 21, 11, device="cuda")
    ops.conv2_wgrad_nhwc_bf16(a1, dy, lens, dW); torch.cuda.synchronize()
    first = dW.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.conv2_wgrad_nhwc_bf16(a1, dy, lens, dW)
    e1.record(); torch.cuda.synchronize()
    # fp32 reference of a few taps: dW[co][ci][kd][kt] = sum dY[b,o,t][co] A1[b,2o+kd-10,t+kt-5][ci]
    err = 0.0
    a1f, dyf = a1.float(), dy.float()
    for kd, kt in ((0, 0), (10, 5), (20, 10), (7, 3)):
        acc = torch.zeros(32, 32, device="cuda", dtype=torch.float64)
        for o in range(D2):
            f = 2 * o + kd - 10
            if f < 0 or f >= D1: continue
            lo, hi = max(0, 5 - kt), min(T, T + 5 - kt)
            x = a1f[:, f, lo + kt - 5:hi + kt - 5].double().reshape(-1, 32); y = dyf[:, o, lo:hi].double().reshape(-1, 32)
            acc += y.t() @ x
        err = max(err, ((dW[:, :, kd, kt].double() - acc).abs().max() / acc.abs().max()).item())
    print(repr({"us": e0.elapsed_time(e1) * 50, "same_run_to_run": bool((first == dW).all()), "rel_err_vs_fp64": err,
                "checksum": dW.double().sum().item()}))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(); sys.exit(0)
other = sys.argv[1] if len(sys.argv) > 1 else None
for rep in range(3):
    for tag, lib in (("other", other), ("this", None)):
        if tag == "other" and not other: continue
        env = dict(os.environ); env.pop("DS2_LIB_PATH", None)
        if lib: env["DS2_LIB_PATH"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(tag, r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-1500:])
