#!/bin/bash
export DS2_EXPERIMENTAL=1
# c3 train step with the NT GEMM variants (same box, interleaved): DS2_GEMM_RING = 0 (production) / w / x
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for rep in 1 2; do for v in ${RING_VARIANTS:-0 w x}; do
  echo -n "RING=$v: "; DS2_GEMM_RING=$v timeout 600 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms', d.get('loss'))"
done; done
