import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from asr_amd import ops, _lib
dev = torch.device("cuda:0")
G, H, B, T = 3, 1024, 64, 501
M = T * B
gx = torch.randn(M, 2 * G * H, device=dev) * 0.5
whh = (torch.rand(2, G * H, H, device=dev) * 2 - 1) / H ** 0.5
bhh = torch.zeros(2, G * H, device=dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
wpf, _ = ops.rnn_pack(G, whh, bf16=True)
lib = _lib.load()
for flags in (0, 4, 8, 12, 0, 4, 12):
    lib.ds2_debug_flags(flags)
    ops.rnn_fwd(G, gx.clone(), wpf, bhh, lens, T, B, H, bf16=True)
    g2 = gx.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    ops.rnn_fwd(G, g2, wpf, bhh, lens, T, B, H, bf16=True)
    e1.record(); torch.cuda.synchronize()
    print(f"flags={flags:2d}: {e0.elapsed_time(e1) * 1e3 / T:.3f} us/step", flush=True)
lib.ds2_debug_flags(0)
