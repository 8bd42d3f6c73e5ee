#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# every bench workload in both precisions (the DESIGN.md §5 table) -> gpurun_out/all_workloads.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/all_workloads.log
for wl in c3 c2 c4 c5 c1; do for dt in bf16 f32; do
  echo "== $wl $dt" >> gpurun_out/all_workloads.log
  timeout 600 python bench.py --workload $wl --dtype $dt --steps 8 --no-cpu-baseline --no-other-workloads 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-700 >> gpurun_out/all_workloads.log
done; done
grep -o '== .*\|"ms_per_step": [0-9.]*\|"value": [0-9.]*' gpurun_out/all_workloads.log | paste - - - 
