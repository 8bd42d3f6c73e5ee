#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# L2 counters of the NT GEMM (RING variant in $RING, default q) on fwd and dX shapes
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_pmc_gemm; rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for shape in "32064 1024 6144" "32064 6144 1024"; do
  i=$((i+1))
  DS2_GEMM_RING=${RING:-q} rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT -o a$i -- python $GRAFT_REPO_ROOT/scripts/r5_gemm_one.py $shape > $OUT/log_a$i.txt 2>&1
  DS2_GEMM_RING=${RING:-q} rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $OUT -o b$i -- python $GRAFT_REPO_ROOT/scripts/r5_gemm_one.py $shape > $OUT/log_b$i.txt 2>&1
  DS2_GEMM_RING=${RING:-q} rocprofv3 --kernel-trace --pmc WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT -o c$i -- python $GRAFT_REPO_ROOT/scripts/r5_gemm_one.py $shape > $OUT/log_c$i.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for tag in ("a1", "b1", "c1", "a2", "b2", "c2"):
    f = glob.glob(f"{out}/**/{tag}_counter_collection.csv", recursive=True)
    t = glob.glob(f"{out}/**/{tag}_kernel_trace.csv", recursive=True)
    if not f: print(tag, "no counter file"); import os; print(open(f"{out}/log_{tag}.txt").read()[-600:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "gemm" in r["Kernel_Name"]: agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    if t:
        for r in csv.DictReader(open(t[0])):
            if "gemm" in r["Kernel_Name"]: dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in agg.items():
        d = sorted(dur.get(k, [0])); print(tag, k, "launches", len(d), "median us", d[len(d) // 2])
        for c, x in sorted(v.items()): print(f"   {c:28s} {sorted(x)[len(x) // 2]:16.0f}")
PY
