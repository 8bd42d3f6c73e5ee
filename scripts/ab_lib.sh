#!/bin/bash
# same-box A/B of two builds of the library (asr_amd/lib/libds2hip.so vs libds2hip_b.so) on the recurrent micro-benchmark and c3
cd "$(dirname "$0")/.."
for rep in 1 2; do
for lib in libds2hip.so libds2hip_b.so; do
  echo "== $lib (rep $rep)"
  DS2_LIB_PATH=$PWD/asr_amd/lib/$lib ABLATE_SKIP=1 timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys
sys.path.insert(0, "scripts")
from ablate_rnn import run
for (name, G, H, B) in [("c3", 3, 1024, 64), ("c4", 4, 1280, 32)]:
    f = min(run(G, H, B, 501, False, 0, True) for _ in range(3)); b = min(run(G, H, B, 501, True, 0, True) for _ in range(3))
    print(f"{name} bf16 fwd {f:6.2f} bwd {b:6.2f} us/step", flush=True)
PY
  DS2_LIB_PATH=$PWD/asr_amd/lib/$lib timeout 300 python bench.py --workload c3 --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done; done
