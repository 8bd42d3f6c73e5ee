#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# round 4: fp32 mode with three-term split-bf16 GEMMs: tests, then c2 / c4 fp32 A/B against the fp32-MFMA kernels (same box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -q --tb=short -p no:cacheprovider -x -k "split or bench_two" > gpurun_out/r4_split_tests.log 2>&1; echo "split tests rc=$?"; grep -E "split-bf16|passed|failed" gpurun_out/r4_split_tests.log | tail -8
for wl in c2 c4; do for mode in f32 split; do
  DS2_F32_GEMM=$mode timeout 600 python bench.py --workload $wl --dtype f32 --steps 6 --no-cpu-baseline --no-other-workloads > gpurun_out/r4_f32_${wl}_$mode.log 2>&1
  echo "$wl DS2_F32_GEMM=$mode rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_f32_${wl}_$mode.log) $(grep -o '"loss": [0-9.]*' gpurun_out/r4_f32_${wl}_$mode.log | head -1)"
done; done
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider -s > gpurun_out/full_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/full_gpu.log
