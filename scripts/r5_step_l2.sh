#!/bin/bash
# L2 (TCC) request / hit / miss counts of EVERY kernel of the c3 bf16 train step: misses x 128 B = bytes that left the XCD's L2 per launch.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_step_l2; rm -rf $OUT; mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT -o l2 -- python $ROOT/bench.py --workload ${WL:-c3} --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > $OUT/log.txt 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT -o ea -- python $ROOT/bench.py --workload ${WL:-c3} --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > $OUT/log_ea.txt 2>&1
cd $ROOT
python3 - $OUT <<'PY'
import csv, sys, glob, collections, re
out = sys.argv[1]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:52]
for tag in ("l2", "ea"):
    f = glob.glob(f"{out}/**/{tag}_counter_collection.csv", recursive=True)
    if not f: print(tag, "no counters:", open(f"{out}/log{'_ea' if tag == 'ea' else ''}.txt").read()[-400:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    disp = set()
    for r in csv.DictReader(open(f[0])):
        k = short(r["Kernel_Name"]); agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in disp: disp.add(r["Dispatch_Id"]); n[k] += 1
    rows = []
    for k, c in agg.items():
        if tag == "l2":
            rows.append((c["TCC_MISS_sum"] * 128 / n[k] / 1e6, f"{k:54s} x{n[k]:3d}  req {c['TCC_REQ_sum'] / n[k]:10.3g}  hit {100 * c['TCC_HIT_sum'] / max(1, c['TCC_REQ_sum']):5.1f} %  miss MB/launch {c['TCC_MISS_sum'] * 128 / n[k] / 1e6:8.1f}"))
        else:
            rd = (c["TCC_EA0_RDREQ_sum"] - c["TCC_EA0_RDREQ_32B_sum"]) * 128 + c["TCC_EA0_RDREQ_32B_sum"] * 32   # gfx950: a full read request is 128 B (the x2 of FETCH_SIZE, MI355X_MICROARCH.md)
            wr = c["TCC_EA0_WRREQ_64B_sum"] * 64 + (c["TCC_EA0_WRREQ_sum"] - c["TCC_EA0_WRREQ_64B_sum"]) * 32
            rows.append((rd / n[k], f"{k:54s} x{n[k]:3d}  EA read MB/launch {rd / n[k] / 1e6:8.1f}  EA write MB/launch {wr / n[k] / 1e6:8.1f}"))
    print("==", tag)
    for _, s in sorted(rows, reverse=True)[:34]: print(s)
PY
