#!/bin/bash
# same-box A/B: persistent forward recurrence (default) vs one launch per step (DS2_RNN_PERSISTENT=0)
cd "$(dirname "$0")/.."
for rep in 1 2; do for p in 1 0; do
  echo "== DS2_RNN_PERSISTENT=$p (rep $rep)"
  DS2_RNN_PERSISTENT=$p ABLATE_SKIP=1 timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys
sys.path.insert(0, "scripts")
from ablate_rnn import run
for (name, G, H, B) in [("c3", 3, 1024, 64), ("c2", 3, 768, 32), ("c4", 4, 1280, 32), ("c5", 3, 1024, 32)]:
    f = min(run(G, H, B, 501, False, 0, True) for _ in range(3))
    print(f"{name} bf16 fwd {f:6.2f} us/step", flush=True)
PY
  DS2_RNN_PERSISTENT=$p timeout 300 python bench.py --workload c3 --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done; done
timeout 300 python scripts/det_check.py 40 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -3
