#!/bin/bash
# the c3 step on the plain path and on the data-parallel path with a forced single-rank RCCL all-reduce ("conv" schedule), same box, interleaved
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for rep in 1 2 3; do
  echo -n "plain: "; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads 2>&1 | grep -v amdgpu.ids | grep "^{" | tail -1 | grep -o "\"ms_per_step\": [0-9.]*" | head -1
  echo -n "dp conv (1 rank, RCCL): "; MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29700 + rep)) RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 DS2_FORCE_ALLREDUCE=1 DS2_DP_MODE=conv timeout 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads 2>&1 | grep -v amdgpu.ids | grep "^{" | tail -1 | grep -o "\"ms_per_step\": [0-9.]*" | head -1
done
