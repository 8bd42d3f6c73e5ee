import os, sys, torch
sys.path.insert(0, "/root/repo")
from asr_amd import ops
dev = torch.device("cuda:0")
def run(G, H, B, T=501):
    M = T * B
    gx0 = torch.randn(M, 2 * G * H, device=dev) * 0.5
    whh = (torch.rand(2, G * H, H, device=dev) * 2 - 1) / H ** 0.5
    bhh = torch.zeros(2, G * H, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    wpf, wpb = ops.rnn_pack(G, whh, bf16=False)
    dy = torch.randn(M, H, device=dev)
    best = [1e9, 1e9]
    for _ in range(3):
        gx = gx0.clone()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        torch.cuda.synchronize(); e[0].record()
        hb, aux = ops.rnn_fwd(G, gx, wpf, bhh, lens, T, B, H, bf16=False)
        e[1].record()
        ops.rnn_bwd(G, dy, gx, aux, hb, wpb, lens, T, B, H, bf16=False)
        e[2].record(); torch.cuda.synchronize()
        best = [min(best[0], e[0].elapsed_time(e[1]) * 1e3 / T), min(best[1], e[1].elapsed_time(e[2]) * 1e3 / T)]
    return best
for (name, G, H, B) in [("c2", 3, 768, 32), ("c3", 3, 1024, 64)]:
    f, b = run(G, H, B)
    print(f"{name} fp32 fwd {f:6.2f} bwd {b:6.2f} us/step", flush=True)
ops.rnn_persistent_check()
