#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# same-box A/B of two builds of the library (asr_amd/lib/libds2hip.so vs $1, default libds2hip_prev.so): persistent-recurrence micro-benchmark
# (bf16 training mode, c3 / c2 / c5-at-B=32 layer shapes) and the c3 bench
cd "$(dirname "$0")/.."
OTHER=${1:-libds2hip_prev.so}
for rep in 1 2; do
for lib in libds2hip.so $OTHER; do
  echo "== $lib (rep $rep)"
  DS2_LIB_PATH=$PWD/asr_amd/lib/$lib timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, torch
sys.path.insert(0, ".")
from asr_amd import ops
dev = torch.device("cuda:0")
def run(G, H, B, T=501):
    M = T * B
    gx = torch.randn(M, 2 * G * H, device=dev) * 0.5
    whh = (torch.rand(2, G * H, H, device=dev) * 2 - 1) / H ** 0.5
    bhh = torch.zeros(2, G * H, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    wpf, wpb = ops.rnn_pack(G, whh, bf16=True)
    dy = torch.randn(M, H, device=dev)
    best = [1e9, 1e9]
    for _ in range(4):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        g = gx.clone()
        torch.cuda.synchronize(); e[0].record()
        hb, aux, rec = ops.rnn_fwd(G, g, wpf, bhh, lens, T, B, H, bf16=True, packed_gates=True)
        e[1].record()
        side = torch.empty(M, 2 * G * H, dtype=torch.bfloat16, device=dev)
        ops.rnn_bwd(G, dy, None, aux, hb, wpb, lens, T, B, H, bf16=True, dgx_bf16=side, gates_bf16=rec)
        e[2].record(); torch.cuda.synchronize()
        best = [min(best[0], e[0].elapsed_time(e[1]) * 1e3 / T), min(best[1], e[1].elapsed_time(e[2]) * 1e3 / T)]
    ops.rnn_persistent_check()
    return best
for (name, G, H, B) in [("c3", 3, 1024, 64), ("c2", 3, 768, 32), ("c5/32", 3, 1024, 32)]:
    f, b = run(G, H, B)
    print(f"{name} bf16 fwd {f:6.2f} bwd {b:6.2f} us/step", flush=True)
PY
  DS2_LIB_PATH=$PWD/asr_amd/lib/$lib timeout 300 python bench.py --workload c3 --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done; done
