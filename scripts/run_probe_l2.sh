#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# runs the L2 residency / fresh-data probe over its configurations (on the GPU box) -> gpurun_out/probe_l2_residency.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
P=scripts/build/probe_l2_residency; K=scripts/build/probe_l2_kernel.hsaco
{
  echo "== read-only operand only (96 KB slice per workgroup)"; timeout 60 $P $K 96 0
  for kb in 1 16 64; do echo "== + ${kb} KB fresh per workgroup, all loads in flight at once (mode 0)"; timeout 60 $P $K 96 1 $kb 0; done
  echo "== + 64 KB fresh, one load per thread in flight (serialised round trips, mode 8)"; timeout 60 $P $K 96 1 64 8
  for m in 1 3; do echo "== + 64 KB fresh, mode $m"; timeout 60 $P $K 96 1 64 $m; done
  echo "== 192 KB slice + 96 KB fresh (backward geometry 16 rows x 32 units)"; timeout 60 $P $K 192 1 64 0
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/probe_l2_residency.txt
