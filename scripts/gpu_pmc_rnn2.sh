#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# memory-path counters of the recurrent forward step kernel (two passes) -> gpurun_out/pmc_rnn2/
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_rnn2
cd /tmp
# (at most 4 counters of one block per pass: more and rocprofv3 aborts with "exceeds the capabilities of the hardware" and then hangs)
i=0
for P in "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TOTAL_READ" \
         "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
         "TCC_HIT TCC_MISS TCC_READ TCC_TAG_STALL"; do      # (a TA_* pass hangs rocprofv3 on this box)
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_rnn2 -o p$i -- python $GRAFT_REPO_ROOT/scripts/pmc_rnn.py bf16 > $GRAFT_REPO_ROOT/gpurun_out/pmc_rnn2/log$i.txt 2>&1
  echo "pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_rnn2/*counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in rows:
        if "rnn_fwd" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for c in sorted(agg): print(f"{c:40s} per launch {agg[c] / max(n[c], 1):14.1f}   ({n[c]} launches)")
PY
