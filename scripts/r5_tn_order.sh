#!/bin/bash
# Item order of the grouped four-wave TN kernel: time + L2 counters, DS2_TN_ORDER=0 (old) against the default.
export DS2_EXPERIMENTAL=1
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python scripts/r5_tn_w4.py order 2>&1 | grep -v amdgpu.ids | tail -8
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_tn_order; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/tn_one.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "scripts"))
from asr_amd import ops
import r5_tn_w4 as R
g = torch.Generator(device="cuda"); g.manual_seed(1)
probs, _ = R.layer_problems(3, 1024, 64, 501, 1024, g)
for _ in range(4): ops.gemm_bf16_tn_splitk_group(probs, splitk=4)
torch.cuda.synchronize()
PY
cd /tmp
for o in 0 1; do
  DS2_TN_ORDER=$o rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT -o ord$o -- python /tmp/tn_one.py > $OUT/log$o.txt 2>&1
  DS2_TN_ORDER=$o rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -o ordm$o -- python /tmp/tn_one.py > $OUT/logm$o.txt 2>&1
done
python3 - $OUT <<'PY'
import csv, sys, glob, collections
for o in "01":
    for pre in ("ord", "ordm"):
        f = glob.glob(f"{sys.argv[1]}/**/{pre}{o}_counter_collection.csv", recursive=True)
        if not f: print("order", o, "no counters"); continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f[0])):
            if "tn_w4" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("DS2_TN_ORDER=" + o, "  ".join(f"{c} {sorted(x)[len(x) // 2]:.4g}" for c, x in sorted(agg.items())))
PY
