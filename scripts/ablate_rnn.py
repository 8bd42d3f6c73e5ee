"""GPU micro-benchmark of the recurrent step kernels with ablations (ds2_debug_flags)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import _lib, ops
lib = _lib.load()
dev = torch.device("cuda:0")
def run(G, H, B, T, bwd, flags, bf=False):
    ops.debug_flags(flags)
    M = T * B
    gx = torch.randn(M, 2 * G * H, device=dev) * 0.5
    whh = (torch.rand(2, G * H, H, device=dev) * 2 - 1) / H ** 0.5
    bhh = torch.zeros(2, G * H, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    wpf, wpb = ops.rnn_pack(G, whh, bf16=bf)
    hbuf, aux = ops.rnn_fwd(G, gx, wpf, bhh, lens, T, B, H, bf16=bf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if not bwd:
        gx2 = torch.randn(M, 2 * G * H, device=dev) * 0.5
        torch.cuda.synchronize(); e0.record()
        ops.rnn_fwd(G, gx2, wpf, bhh, lens, T, B, H, bf16=bf)
        e1.record(); torch.cuda.synchronize()
    else:
        dy = torch.randn(M, H, device=dev)
        torch.cuda.synchronize(); e0.record()
        ops.rnn_bwd(G, dy, gx, aux, hbuf, wpb, lens, T, B, H, bf16=bf)
        e1.record(); torch.cuda.synchronize()
    ops.debug_flags(0)
    return e0.elapsed_time(e1) * 1e3 / T
for (name, G, H, B) in ([] if os.environ.get("ABLATE_SKIP") else [("c2", 3, 768, 32), ("c3", 3, 1024, 64), ("c4-lstm", 4, 1280, 32)]):
    for bwd in (False, True):
      for bf in (False, True):
        r = [run(G, H, B, 501, bwd, f, bf) for f in (0, 1, 2, 3)]
        print(f"{name:8s} {'bf16' if bf else 'fp32'} {'bwd' if bwd else 'fwd'}  us/step: full {r[0]:6.2f} | no-gemm {r[1]:6.2f} | no-epilogue {r[2]:6.2f} | neither {r[3]:6.2f}", flush=True)
