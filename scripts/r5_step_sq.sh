#!/bin/bash
# SQ counters of EVERY kernel of the c3 bf16 train step: matrix-pipe busy share, LDS bank conflicts, VALU / LDS instruction counts.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_step_sq; rm -rf $OUT; mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
cd /tmp
B="python $ROOT/bench.py --workload ${WL:-c3} --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -o a -- $B > $OUT/log_a.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $OUT -o b -- $B > $OUT/log_b.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT -o c -- $B > $OUT/log_c.txt 2>&1
cd $ROOT
python3 - $OUT <<'PY'
import csv, sys, glob, collections, re
out = sys.argv[1]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:46]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
for tag in "abc":
    f = glob.glob(f"{out}/**/{tag}_counter_collection.csv", recursive=True)
    if not f: print(tag, "no counters:", open(f"{out}/log_{tag}.txt").read()[-300:]); continue
    disp = set()
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if tag == "a" and r["Dispatch_Id"] not in disp: disp.add(r["Dispatch_Id"]); n[k] += 1
    if tag == "a":
        t = glob.glob(f"{out}/**/a_kernel_trace.csv", recursive=True)
        for r in csv.DictReader(open(t[0])): dur[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"{'kernel':48s} launches  us/launch  MFMA busy   wave-cycles/launch  LDS conflict cyc / LDS active cyc   wait-any / wave-cyc   wait-LDS / wave-cyc")
for k in sorted(agg, key=lambda k: -dur[k])[:26]:
    c = agg[k]; m = max(1, n[k])
    cyc = c["GRBM_GUI_ACTIVE"] / 8          # per XCD
    print(f"{k:48s} {m:5d}  {dur[k] / m:9.1f}  {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / max(1, cyc):7.1f} %   {c['SQ_WAVE_CYCLES'] / m:12.4g}"
          f"     {c['SQ_LDS_BANK_CONFLICT'] / m:10.3g} / {c['SQ_LDS_IDX_ACTIVE'] / m:10.3g}      {c['SQ_WAIT_INST_ANY'] / max(1, c['SQ_WAVE_CYCLES']):6.2f}      {c['SQ_WAIT_INST_LDS'] / max(1, c['SQ_WAVE_CYCLES']):6.2f}")
PY
