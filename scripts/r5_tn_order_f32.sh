#!/bin/bash
# the 8-wave grouped TN kernel's item order (the fp32 configurations take it: their split operands have K = 3 x T*B, not a multiple of 64 per slice)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp DS2_EXPERIMENTAL=1
for wl in c3 c4 c2; do for rep in 1 2; do for v in 0 1; do
  echo -n "$wl f32 DS2_TN_ORDER=$v: "; DS2_TN_ORDER=$v timeout 600 python bench.py --workload $wl --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms', d.get('loss'))"
done; done; done
