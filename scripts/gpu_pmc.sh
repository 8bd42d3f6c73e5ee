#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc_$ctr -- python $GRAFT_REPO_ROOT/scripts/pmc_rnn.py bf16 > $GRAFT_REPO_ROOT/gpurun_out/pmc/log_$ctr.txt 2>&1
  echo "$ctr rc=$?"
done
cd $GRAFT_REPO_ROOT
ls gpurun_out/pmc | head; f=$(ls gpurun_out/pmc/*FETCH_SIZE*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && head -3 "$f"
