#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# same-box A/B of two builds of the library: asr_amd/lib/libds2hip_old.so (built from an earlier commit) vs the current one (DS2_LIB_PATH)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x -k "ksplit or rnn" > gpurun_out/r4_ab_tests.log 2>&1; echo "rnn kernel tests rc=$?"; tail -3 gpurun_out/r4_ab_tests.log
for wl in ${WLS:-c3 c2 c5}; do for which in old new old new; do
  lib=$PWD/asr_amd/lib/libds2hip.so; [ $which = old ] && lib=$PWD/asr_amd/lib/libds2hip_old.so
  DS2_LIB_PATH=$lib timeout 600 python bench.py --workload $wl --dtype bf16 --steps 10 --no-cpu-baseline --no-other-workloads > gpurun_out/r4_ab_${wl}_$which.log 2>&1
  echo "$wl $which rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_ab_${wl}_$which.log) $(grep -o '"us_per_time_step": [0-9.]*' gpurun_out/r4_ab_${wl}_$which.log | tr '\n' ' ') $(grep -o '"loss": [0-9.]*' gpurun_out/r4_ab_${wl}_$which.log | head -1)"
done; done
