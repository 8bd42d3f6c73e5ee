#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# SQ counters of the bf16 GEMM kernels on the train step's shapes (scripts/bench_gemm.py) -> gpurun_out/pmc_gemm/
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_gemm
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm -o sq -- python $GRAFT_REPO_ROOT/scripts/bench_gemm.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_gemm/log.txt 2>&1
echo "rc=$?"
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/pmc_gemm/*counter_collection.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = (r["Kernel_Name"][:70], r.get("Grid_Size", ""))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in agg.items():
    if "gemm" not in k[0]: continue
    n = max(cnt[k], 1)
    print(k, "launches", n)
    for c, x in sorted(v.items()): print(f"   {c:28s} {x / n:16.0f}")
PY
tail -12 gpurun_out/pmc_gemm/log.txt
