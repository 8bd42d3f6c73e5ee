#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# quick GPU loop: rnn kernel tests + model tests + bench (c2, c3)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "${KSEL:-rnn or gemm}" > gpurun_out/k.log 2>&1; echo "kernels rc=$?"; tail -4 gpurun_out/k.log
if [ -z "$SKIP_MODEL" ]; then
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/model.log 2>&1; echo "model rc=$?"; tail -4 gpurun_out/model.log
fi
for wl in ${WLS:-c2 c3}; do
  timeout 900 python bench.py --workload $wl --steps 5 --warmup 2 --breakdown --no-cpu-baseline --no-other-workloads $BENCHARGS > gpurun_out/bench_$wl.log 2>&1; echo "bench $wl rc=$?"
  grep -vE "amdgpu.ids" gpurun_out/bench_$wl.log | tail -9 | cut -c1-600
done
