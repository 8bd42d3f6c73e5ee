#!/bin/bash
# L2 / fabric traffic of the four-wave NT kernel on the step's two shapes (forward projection: N = 6144, K = 1024; dX: N = 1024, K = 6144)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_nt_l2; rm -rf $OUT; mkdir -p $OUT
cd /tmp
for shp in "32064 6144 1024" "32064 1024 6144" "32064 6144 1344"; do
  tag=$(echo $shp | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT -o l2_$tag -- python $GRAFT_REPO_ROOT/scripts/r5_gemm_one.py $shp > $OUT/log_$tag.txt 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT -o ea_$tag -- python $GRAFT_REPO_ROOT/scripts/r5_gemm_one.py $shp >> $OUT/log_$tag.txt 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for f in sorted(glob.glob(f"{out}/**/*_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sorted(v)[len(v) // 2] for k, v in agg.items()}
    name = f.split("/")[-1].replace("_counter_collection.csv", "")
    if "TCC_REQ_sum" in m: print(name, f"req {m['TCC_REQ_sum']:.3g} hit {100 * m['TCC_HIT_sum'] / m['TCC_REQ_sum']:.1f} %")
    else:
        rd = (m["TCC_EA0_RDREQ_sum"] - m["TCC_EA0_RDREQ_32B_sum"]) * 128 + m["TCC_EA0_RDREQ_32B_sum"] * 32
        wr = m["TCC_EA0_WRREQ_64B_sum"] * 64 + (m["TCC_EA0_WRREQ_sum"] - m["TCC_EA0_WRREQ_64B_sum"]) * 32
        print(name, f"EA read {rd / 1e6:.0f} MB  write {wr / 1e6:.0f} MB per launch")
PY
