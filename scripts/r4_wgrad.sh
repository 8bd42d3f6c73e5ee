#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# round 4: grouped co-resident weight-gradient kernel: tests, stand-alone bench, whole-step A/B of the three schedules
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r4_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r4_tests.log
PYTHONPATH=. timeout 300 python scripts/bench_tn_group.py > gpurun_out/r4_tn_group.log 2>&1; echo "bench_tn_group rc=$?"; cat gpurun_out/r4_tn_group.log | tail -8
timeout 900 python bench.py --gpus 2 --workload c1 --steps 3 --warmup 1 --ranks-on-one-gpu --no-cpu-baseline > gpurun_out/r4_bench_2rank.log 2>&1; echo "2-rank bench rc=$?"; tail -3 gpurun_out/r4_bench_2rank.log | cut -c1-1500
for mode in 0 main 1 0 main 1; do
  DS2_WGRAD_SIDE=$mode timeout 600 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4_bench_side_$mode.log 2>&1
  echo "DS2_WGRAD_SIDE=$mode rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_bench_side_$mode.log) $(grep -o '"persistent_starved_steps": [0-9]*' gpurun_out/r4_bench_side_$mode.log) $(grep -o '"us_per_time_step": [0-9.]*' gpurun_out/r4_bench_side_$mode.log | tr '\n' ' ')"
done
