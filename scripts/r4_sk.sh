#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# round 4: grouped split-K weight-gradient launch (DS2_WGRAD_SIDE=sk) vs round 3's three launches (0): tests + same-box A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -q --tb=short -p no:cacheprovider -x -k "grouped or side" > gpurun_out/r4_sk_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4_sk_tests.log
for wl in c3 c2 c4 c5; do for mode in 0 sk 0 sk; do
  DS2_WGRAD_SIDE=$mode timeout 600 python bench.py --workload $wl --dtype bf16 --steps 10 --no-cpu-baseline --no-other-workloads > gpurun_out/r4_sk_${wl}_$mode.log 2>&1
  echo "$wl DS2_WGRAD_SIDE=$mode rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_sk_${wl}_$mode.log) $(grep -o '"loss": [0-9.]*' gpurun_out/r4_sk_${wl}_$mode.log | head -1)"
done; done
