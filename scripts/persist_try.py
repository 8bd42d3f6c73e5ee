"""persistent vs step-kernel forward recurrence at the bench shape: bit-identity of h in both gate-saving modes (debug helper)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import ops, _lib
lib = _lib.load()
G, H, B, T = 3, 1024, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 101
dev = torch.device("cuda:0")
torch.manual_seed(0)
gx = torch.randn(T * B, 2 * G * H, device=dev) * 0.5
whh = (torch.rand(2, G * H, H, device=dev) * 2 - 1) / H ** 0.5
bhh = torch.randn(2, G * H, device=dev) * 0.1
lens = torch.randint(T // 4, T + 1, (B,), dtype=torch.int32, device=dev); lens[0] = T
wpf, wpb = ops.rnn_pack(G, whh, bf16=True)
def run(packed, flags):
    lib.ds2_debug_flags(flags)          # any non-zero flag selects the step kernels (64 is unused by them)
    out = ops.rnn_fwd(G, gx.clone(), wpf, bhh, lens, T, B, H, bf16=True, packed_gates=packed)
    torch.cuda.synchronize()
    lib.ds2_debug_flags(0)
    return out
ref = run(False, 64)
for packed in (False, True):
    for it in range(3):
        o = run(packed, 0)
        d = (o[0] - ref[0]).abs()
        print(f"persistent packed={packed} it={it}: h equal to step kernels: {torch.equal(o[0], ref[0])}  max diff {float(d.max()):.3e}  first bad row {int((d.amax(1) > 0).nonzero()[0]) if float(d.max()) > 0 else -1}", flush=True)
