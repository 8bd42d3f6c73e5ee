#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# Runs on the GPU box via gpurun: per-family kernel parity tests (separate processes so one fault
# does not hide the others), whole-model tests, smoke and a short bench.  Logs -> gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name"; timeout ${TMO:-600} "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n ${TAILN:-15} gpurun_out/$name.log; }
rocminfo 2>/dev/null | grep -E "gfx|Compute Unit" | head -4
for fam in gemm bn1d bn2d transposes conv rnn ctc softmax; do
  TAILN=12 run k_$fam python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "$fam"
done
if [ "$1" != "kernels" ]; then
  TAILN=25 run model python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short -p no:cacheprovider
  run smoke python __graft_entry__.py smoke
  TMO=900 run bench_c2 python bench.py --workload c2 --steps 3 --warmup 1 --breakdown --no-cpu-baseline
fi
