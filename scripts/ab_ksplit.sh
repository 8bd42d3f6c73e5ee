#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# K-split backward recurrence: correctness vs the other two kernel families + the oracle, per-step time, c3 bench A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python scripts/ab_ksplit.py ${1:-all} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_ksplit.log
for k in 1 0; do
  echo "== bench c3 DS2_RNN_KSPLIT=$k"
  DS2_RNN_KSPLIT=$k timeout 600 python bench.py --workload c3 --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"roofline": {[^}]*}' | tee -a gpurun_out/ab_ksplit.log
done
