"""persistent vs step-kernel forward recurrence at the bench shape: bit-identity of h in both gate-saving modes (debug helper)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops, _lib
lib = _lib.load()
G, H = 3, 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[1]) if len(sys.argv) > 1 else 101
dev = torch.device("cuda:0")
torch.manual_seed(0)
gx = torch.randn(T * B, 2 * G * H, device=dev) * 0.5
whh = (torch.rand(2, G * H, H, device=dev) * 2 - 1) / H ** 0.5
bhh = torch.randn(2, G * H, device=dev) * 0.1
lens = torch.randint(T // 4, T + 1, (B,), dtype=torch.int32, device=dev); lens[0] = T
wpf, wpb = ops.rnn_pack(G, whh, bf16=True)
def run(packed, flags):
    ops.debug_flags(flags)          # any non-zero flag selects the step kernels (64 is unused by them)
    out = ops.rnn_fwd(G, gx.clone(), wpf, bhh, lens, T, B, H, bf16=True, packed_gates=packed)
    torch.cuda.synchronize()
    ops.debug_flags(0)
    return out
ref = run(False, 64)
for packed in (False, True):
    for it in range(3):
        o = run(packed, 0)
        d = (o[0] - ref[0]).abs()
        print(f"persistent packed={packed} it={it}: h equal to step kernels: {torch.equal(o[0], ref[0])}  max diff {float(d.max()):.3e}  first bad row {int((d.amax(1) > 0).nonzero()[0]) if float(d.max()) > 0 else -1}", flush=True)

# ---- backward: persistent vs step kernels on the same saved forward state
hb, aux0, rec = run(True, 0)
dy = torch.randn(T * B, H, device=dev)
def runb(flags):
    ops.debug_flags(flags)
    side = torch.empty(T * B, 2 * G * H, dtype=torch.bfloat16, device=dev)
    aux = aux0.clone()
    ops.rnn_bwd(G, dy, None, aux, hb, wpb, lens, T, B, H, bf16=True, dgx_bf16=side, gates_bf16=rec)
    torch.cuda.synchronize()
    ops.debug_flags(0)
    return side, aux
sref, aref = runb(64)
for it in range(3):
    sd, ax = runb(0)
    d = (sd.float() - sref.float()).abs()
    bad_rows = (d.amax(1) > 0).nonzero().flatten()
    print(f"persistent bwd it={it}: dGx equal {torch.equal(sd, sref)} aux equal {torch.equal(ax, aref)} finite {bool(torch.isfinite(sd.float()).all())} "
          f"max diff {float(d.max()):.3e} bad rows {bad_rows.numel()} first {bad_rows[:4].tolist()} (t = row // B)", flush=True)
if not torch.equal(sd, sref):
    bad = torch.isnan(sd.float()) | (sd.float() != sref.float())
    r0 = int(bad.any(1).nonzero()[0])
    cols = bad[r0].nonzero().flatten()
    print("first bad row", r0, "(t", r0 // B, "b", r0 % B, ") bad cols:", cols.numel(), "min", int(cols.min()), "max", int(cols.max()), "first", cols[:12].tolist())
    # per direction / gate / 32-unit block summary over all rows
    bb = bad.view(T, B, 2, G, H)
    for d in range(2):
        for g in range(G):
            blk = bb[:, :, d, g].view(T, B, H // 32, 32).any(3)            # (T, B, nblk)
            print(f"dir {d} gate {g}: bad (t,b,block) count {int(blk.sum())}; blocks hit {blk.any(0).any(0).nonzero().flatten()[:16].tolist()}; b hit {blk.any(2).any(0).nonzero().flatten()[:20].tolist()}; t hit {blk.any(2).any(1).nonzero().flatten()[:10].tolist()}")
