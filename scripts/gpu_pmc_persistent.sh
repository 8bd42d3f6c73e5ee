#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# HBM-side bytes of the persistent recurrence kernels (C3 layer shape, bf16 training mode): two separate rocprofv3 PMC passes
# (--kernel-trace only beside --pmc), reduced to gpurun_out/pmc_r06/summary.json by scripts/pmc_summarize.py.  Copy to profiles/ to commit.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/pmc_r06
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_r06 -o pmc_$ctr -- python $R/scripts/pmc_rnn.py bf16 ${PMC_T:-501} > $R/gpurun_out/pmc_r06/log_$ctr.txt 2>&1
  echo "$ctr rc=$?"
done
cd $R
# rocprofv3 nests its output under <dir>/<host>/<pid>_..., flatten
for ctr in FETCH_SIZE WRITE_SIZE; do
  f=$(find gpurun_out/pmc_r06 -name "*pmc_${ctr}_counter_collection.csv" | head -1)
  [ -n "$f" ] && [ "$f" != "gpurun_out/pmc_r06/pmc_${ctr}_counter_collection.csv" ] && cp "$f" gpurun_out/pmc_r06/pmc_${ctr}_counter_collection.csv
done
python scripts/pmc_summarize.py gpurun_out/pmc_r06 ${PMC_T:-501} gpurun_out/pmc_r06/summary.json "GRU H=1024 B=64 bf16 operands, packed gate records, T=${PMC_T:-501}" | tail -40
# keep the merge small: drop the raw per-dispatch CSVs of torch's own kernels
find gpurun_out/pmc_r06 -name "*kernel_trace.csv" -delete; find gpurun_out/pmc_r06 -name "*agent_info.csv" -delete
