#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# round 4: weight-gradient launches on the side stream beside a persistent backward recurrence that leaves CUs idle (DS2_WGRAD_IDLE), same box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -q --tb=short -p no:cacheprovider -x -k "leaves_cus_idle" > gpurun_out/r4_idle_tests.log 2>&1; echo "idle tests rc=$?"; tail -5 gpurun_out/r4_idle_tests.log
for rep in 1 2; do for wd in "c2 bf16" "c2 f32" "c4 bf16" "c4 f32" "c1 bf16" "c1 f32" "c3 bf16"; do set -- $wd; for mode in 0 1; do
  DS2_WGRAD_IDLE=$mode timeout 600 python bench.py --workload $1 --dtype $2 --steps 10 --no-cpu-baseline --no-other-workloads > gpurun_out/r4_idle_$1_$2_$mode.log 2>&1
  echo "$1 $2 DS2_WGRAD_IDLE=$mode rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_idle_$1_$2_$mode.log) $(grep -o '"loss": [0-9.]*' gpurun_out/r4_idle_$1_$2_$mode.log | head -1) starved $(grep -o '"persistent_starved_steps": [0-9]*' gpurun_out/r4_idle_$1_$2_$mode.log | head -1 | grep -o '[0-9]*$') us/step $(grep -o '"us_per_time_step": [0-9.]*' gpurun_out/r4_idle_$1_$2_$mode.log | head -2 | grep -o '[0-9.]*$' | tr '\n' ' ')"
done; done; done
