#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# round 5: conv2 weight gradient from channels-last operands (DS2_CONV2_WGRAD=nhwc) vs the padded-copy kernel (=pad): kernel tests + same-box c3 A/B
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 800 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q --tb=short -p no:cacheprovider -k "conv" 2>&1 | tail -6
O=gpurun_out/r05_conv_ab.txt
{
echo "# scripts/r5_conv.sh: c3 bf16 (8 timed steps), DS2_CONV2_WGRAD=pad (round-2 kernel) vs nhwc (round 5), same box, interleaved"
for rep in 1 2 3; do for v in ${VARIANTS:-pad nhwc}; do
  echo "== DS2_CONV2_WGRAD=$v"
  DS2_CONV2_WGRAD=$v python bench.py --workload c3 --steps 8 --warmup 3 --breakdown --no-cpu-baseline --no-other-workloads 2>&1 | grep -E "forward:|backward:|\"ms_per_step\"" | sed -E 's/.*("ms_per_step": [0-9.]+).*("loss": [0-9.]+).*/\1 \2/' | cut -c1-120
done; done
} > $O 2>&1
cat $O
