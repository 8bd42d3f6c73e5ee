"""K-split persistent backward recurrence (default) against the all-gather persistent kernel (ds2_debug_flags 128) and the one-launch-per-step
kernels (64): agreement on one saved forward state, error of each against the fp64 oracle at a small T, and us per time step at T = 501."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from asr_amd import ops, _lib
from oracle import ds2_oracle as O
lib = _lib.load()
dev = torch.device("cuda:0")

def rel(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

def setup(G, H, B, T, seed=0, ragged=True):
    torch.manual_seed(seed)
    gx = torch.randn(T * B, 2 * G * H, device=dev) * 0.5
    whh = (torch.rand(2, G * H, H, device=dev) * 2 - 1) / H ** 0.5
    bhh = torch.randn(2, G * H, device=dev) * 0.1
    lens = torch.randint(max(1, T // 3), T + 1, (B,), dtype=torch.int32, device=dev) if ragged else torch.full((B,), T, dtype=torch.int32, device=dev)
    lens = torch.sort(lens, descending=True).values.contiguous(); lens[0] = T
    dy = torch.randn(T * B, H, device=dev)
    return gx, whh, bhh, lens, dy

def bwd(G, H, B, T, fw, wpb, lens, dy, flags):
    hb, aux0, rec = fw
    ops.debug_flags(flags)
    side = torch.empty(T * B, 2 * G * H, dtype=torch.bfloat16, device=dev)
    aux = aux0.clone()
    dhn = torch.empty(T * B, 2 * H, dtype=torch.bfloat16, device=dev) if G == 3 else None
    bp = torch.zeros(B, 2, 4, H, device=dev)
    ops.rnn_bwd(G, dy, None, aux, hb, wpb, lens, T, B, H, bf16=True, dgx_bf16=side, gates_bf16=rec, dhn_bf16=dhn, bias_part=bp)
    path = ops.rnn_last_path()
    torch.cuda.synchronize()
    ops.debug_flags(0)
    ops.rnn_persistent_check()
    return side, (dhn if G == 3 else aux), bp, path

def check(kind, H, B, T, oracle=True):
    G = 3 if kind == "gru" else 4
    gx, whh, bhh, lens, dy = setup(G, H, B, T)
    wpf, wpb = ops.rnn_pack(G, whh, bf16=True)
    fw = ops.rnn_fwd(G, gx.clone(), wpf, bhh, lens, T, B, H, bf16=True, packed_gates=True)
    res = {f: bwd(G, H, B, T, fw, wpb, lens, dy, f) for f in (0, 128, 64)}
    again = bwd(G, H, B, T, fw, wpb, lens, dy, 0)
    det = torch.equal(again[0].view(torch.int16), res[0][0].view(torch.int16)) and torch.equal(again[1].view(torch.int16) if G == 3 else again[1], res[0][1].view(torch.int16) if G == 3 else res[0][1])
    line = f"{kind} H={H} B={B} T={T}: paths ks={res[0][3]} ag={res[128][3]} st={res[64][3]}  rerun-identical {det}  ag==st {torch.equal(res[128][0].view(torch.int16), res[64][0].view(torch.int16))}"
    line += f"  dGx ks-vs-st {rel(res[0][0], res[64][0]):.2e}"
    if G == 3:
        line += f" dhn(vs ag) {rel(res[0][1], res[128][1]):.2e}"
    line += f" bias {rel(res[0][2], res[128][2]):.2e}"
    if oracle:
        lens_c = lens.cpu()
        gxd = gx.double().cpu().view(T, B, 2, G * H).requires_grad_(True)
        wd, bd = whh.double().cpu().requires_grad_(True), bhh.double().cpu()
        step = O.gru_direction if kind == "gru" else O.lstm_direction
        y = step(gxd[:, :, 0], wd[0], bd[0], lens_c, False) + step(gxd[:, :, 1], wd[1], bd[1], lens_c, True)
        (y * dy.double().cpu().view(T, B, H)).sum().backward()
        ref = gxd.grad.reshape(T * B, 2 * G * H)
        line += "  vs fp64 oracle: " + " ".join(f"{n} {rel(res[f][0], ref):.3e}" for n, f in (("ks", 0), ("ag", 128), ("st", 64)))
    print(line, flush=True)

def timeit(name, G, H, B, T=501):
    gx, whh, bhh, lens, dy = setup(G, H, B, T, ragged=False)
    wpf, wpb = ops.rnn_pack(G, whh, bf16=True)
    fbest = 1e9
    for _ in range(4):
        h_bf = torch.empty(T * B, 2 * H, dtype=torch.bfloat16, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g2 = gx.clone()
        torch.cuda.synchronize(); e0.record()
        fw = ops.rnn_fwd(G, g2, wpf, bhh, lens, T, B, H, bf16=True, packed_gates=True, h_bf16=h_bf)
        e1.record(); torch.cuda.synchronize()
        fbest = min(fbest, e0.elapsed_time(e1) * 1e3 / T)
    fpath = ops.rnn_last_path() & 1
    out = []
    for flags in (0, 128):
        best = 1e9
        for _ in range(4):
            ops.debug_flags(flags)
            side = torch.empty(T * B, 2 * G * H, dtype=torch.bfloat16, device=dev)
            dhn = torch.empty(T * B, 2 * H, dtype=torch.bfloat16, device=dev) if G == 3 else None
            bp = torch.empty(B, 2, 4, H, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            ops.rnn_bwd(G, dy, None, fw[1], fw[0], wpb, lens, T, B, H, bf16=True, dgx_bf16=side, gates_bf16=fw[2], dhn_bf16=dhn, bias_part=bp)
            e1.record(); torch.cuda.synchronize()
            path = ops.rnn_last_path()
            ops.debug_flags(0)
            best = min(best, e0.elapsed_time(e1) * 1e3 / T)
        ops.rnn_persistent_check()
        out.append(f"{'k-split' if flags == 0 else 'all-gather'} (path {path}) {best:5.2f}")
    print(f"{name}: fwd (persistent {fpath}) {fbest:5.2f} us/step   bwd us/step  " + "  |  ".join(out), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "check"):
        check("gru", 1024, 64, 6)
        check("gru", 1024, 61, 9)
        check("lstm", 1024, 33, 5)
        check("gru", 768, 32, 7)
        check("lstm", 768, 40, 4)
        check("gru", 256, 4, 12)
        check("gru", 512, 17, 8)
        check("lstm", 1280, 32, 5)
        check("gru", 1024, 64, 120, oracle=False)
    if what in ("all", "time"):
        for (name, G, H, B) in [("c3", 3, 1024, 64), ("c2", 3, 768, 32), ("c4", 4, 1280, 32), ("c5/32", 3, 1024, 32), ("c1", 3, 256, 4)]:
            timeit(name, G, H, B)
