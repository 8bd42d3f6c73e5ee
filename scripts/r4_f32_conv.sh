#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# round 4: fp32 mode, conv2 through three-term split products on the bf16 conv kernels: tests, then fp32 A/B per workload (same box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -x -k "fp32 or f32 or golden or conv" > gpurun_out/r4_conv_tests.log 2>&1; echo "conv tests rc=$?"; tail -8 gpurun_out/r4_conv_tests.log
for wl in c2 c4; do for mode in f32 split; do
  DS2_F32_CONV=$mode timeout 600 python bench.py --workload $wl --dtype f32 --steps 6 --no-cpu-baseline --no-other-workloads > gpurun_out/r4_f32conv_${wl}_$mode.log 2>&1
  echo "$wl DS2_F32_CONV=$mode rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_f32conv_${wl}_$mode.log) $(grep -o '"loss": [0-9.]*' gpurun_out/r4_f32conv_${wl}_$mode.log | head -1)"
done; done
