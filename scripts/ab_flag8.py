"""Same-box A/B of ds2_debug_flags tile-shape switches of the recurrent step kernels.
   bit 8 : backward keeps 32 rows x 16 units per workgroup (default: 16 x 32 where the forward takes 32 x 16)
   bit 16: forward takes 16 rows x 32 units per workgroup (default: 32 x 16)"""
import os, sys
os.environ["ABLATE_SKIP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ablate_rnn import run
for (name, G, H, B) in [("c3", 3, 1024, 64), ("c4-lstm", 4, 1280, 32), ("c5", 3, 1024, 32)]:
    for bf in (False, True):
        r = [min(run(G, H, B, 501, True, f, bf) for _ in range(3)) for f in (0, 8)]
        print(f"{name:8s} {'bf16' if bf else 'fp32'} bwd  us/step: 16x32 {r[0]:6.2f} | 32x16 {r[1]:6.2f}", flush=True)
        r = [min(run(G, H, B, 501, False, f, bf) for _ in range(3)) for f in (0, 16)]
        print(f"{name:8s} {'bf16' if bf else 'fp32'} fwd  us/step: 32x16 {r[0]:6.2f} | 16x32 {r[1]:6.2f}", flush=True)
