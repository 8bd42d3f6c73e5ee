#!/bin/bash
# the c3 train step of this tree against the tree the round started from (a git worktree built under ab_old/), same box, interleaved
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for rep in 1 2 3; do
  for d in ab_old .; do
    ( cd $d; echo -n "$( [ $d = . ] && echo HEAD || echo round-start ): "; timeout 600 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms', d.get('loss'))" )
  done
done
