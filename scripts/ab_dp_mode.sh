#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# the three data-parallel schedules with a single-rank RCCL all-reduce forced (exercises the collective + persistent-kernel interplay on one GPU)
cd "$(dirname "$0")/.."
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 DS2_FORCE_ALLREDUCE=1
for mode in conv serial overlap; do
  echo "== DS2_DP_MODE=$mode (single-rank RCCL all-reduce forced)"
  DS2_DP_MODE=$mode timeout 300 python bench.py --gpus 1 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/dp_$mode.log 2>&1; echo "rc=$?"
  grep -v amdgpu.ids gpurun_out/dp_$mode.log | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"loss": [0-9.]*\|"persistent_starved_steps": [0-9]*'
done
