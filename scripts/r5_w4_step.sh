#!/bin/bash
export DS2_EXPERIMENTAL=1
# c3 train step with / without the four-wave GEMM kernels (same box, interleaved): DS2_GEMM_W4 = 0 (8-wave kernels) / 1 (default)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for rep in 1 2 3; do for v in 0 1; do
  echo -n "W4=$v: "; DS2_GEMM_W4=$v timeout 600 python bench.py --workload ${WL:-c3} --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms', d.get('loss'))"
done; done
