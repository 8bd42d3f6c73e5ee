#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# round 5, VERDICT item 2 priced: the 128 / 128 CU split that a 32-row K-split backward kernel would create at c3, measured with the EXISTING
# kernels at B = 32 (c3's layer shape, half the batch: the recurrence then holds 128 of the 256 CUs and the layer above's grouped
# weight-gradient launch runs on the other 128) — what the recurrence pays per time step for the GEMM beside it, and what the step gains.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05_idle_split_proxy.txt
{
echo "# scripts/r5_idle_proxy.sh: c3 at B = 32 (recurrence on 128 CUs), DS2_WGRAD_IDLE=0 vs 1, same box, 3 repetitions"
for rep in 1 2 3; do for idle in 0 1; do
  echo "== DS2_WGRAD_IDLE=$idle"
  DS2_WGRAD_IDLE=$idle python bench.py --workload c3 --batch 32 --steps 8 --warmup 3 --breakdown --no-cpu-baseline --no-other-workloads 2>&1 | grep -E "forward:|backward:|rnn_fwd=|\"ms_per_step\"" | sed -E 's/.*("ms_per_step": [0-9.]+).*("us_per_time_step": [0-9.]+).*/\1 \2/' | cut -c1-200
done; done
} > $O 2>&1
cat $O
