#!/bin/bash
# HBM-side bytes of the persistent recurrence kernels (C3 layer shape, bf16 training mode): two separate rocprofv3 PMC passes
# (--kernel-trace only beside --pmc), reduced to gpurun_out/pmc_r03/summary.json by scripts/pmc_summarize.py.  Copy to profiles/ to commit.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/pmc_r03
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_r03 -o pmc_$ctr -- python $R/scripts/pmc_rnn.py bf16 > $R/gpurun_out/pmc_r03/log_$ctr.txt 2>&1
  echo "$ctr rc=$?"
done
cd $R
# rocprofv3 nests its output under <dir>/<host>/<pid>_..., flatten
for ctr in FETCH_SIZE WRITE_SIZE; do
  f=$(find gpurun_out/pmc_r03 -name "*pmc_${ctr}_counter_collection.csv" | head -1)
  [ -n "$f" ] && [ "$f" != "gpurun_out/pmc_r03/pmc_${ctr}_counter_collection.csv" ] && cp "$f" gpurun_out/pmc_r03/pmc_${ctr}_counter_collection.csv
done
python scripts/pmc_summarize.py gpurun_out/pmc_r03 101 gpurun_out/pmc_r03/summary.json "GRU H=1024 B=64 bf16 operands, packed gate records, T=101" | tail -40
# keep the merge small: drop the raw per-dispatch CSVs of torch's own kernels
find gpurun_out/pmc_r03 -name "*kernel_trace.csv" -delete; find gpurun_out/pmc_r03 -name "*agent_info.csv" -delete
