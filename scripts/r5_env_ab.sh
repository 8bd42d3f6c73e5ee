#!/bin/bash
# the c3 train step with one experimental switch off / on, same box, interleaved:   bash scripts/r5_env_ab.sh DS2_TN_ORDER 0 1      (value - = unset)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp DS2_EXPERIMENTAL=1
for rep in 1 2 3; do
  for v in $2 $3; do
    if [ "$v" = "-" ]; then E="DS2_NOP=1"; else E="$1=$v"; fi         # "-": the variable stays unset
    echo -n "$1=$v: "; env $E timeout 600 python bench.py --workload c3 --steps 12 --warmup 3 --no-cpu-baseline --no-other-workloads 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms', d.get('loss'))"
  done
done
