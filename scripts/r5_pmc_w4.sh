#!/bin/bash
export DS2_EXPERIMENTAL=1
# SQ / L2 counters of the four-wave NT GEMM (default) and the 8-wave one (DS2_GEMM_W4=0) on the dX shape, and of the four-wave grouped TN kernel on a
# c3 layer: matrix-pipe busy share, LDS bank conflicts, L2 hit rate.  Separate --pmc passes, --kernel-trace only (no other trace domain).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_pmc_w4; rm -rf $OUT; mkdir -p $OUT
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
run() {  # tag, env assignment, command...
  tag=$1; shift; envs=$1; shift
  env $envs rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -o ${tag}_a -- "$@" > $OUT/log_${tag}_a.txt 2>&1
  env $envs rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT -o ${tag}_b -- "$@" > $OUT/log_${tag}_b.txt 2>&1
  env $envs rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT -o ${tag}_c -- "$@" > $OUT/log_${tag}_c.txt 2>&1
}
run nt_w4 DS2_GEMM_W4=1 python $GRAFT_REPO_ROOT/scripts/r5_gemm_one.py 32064 1024 6144
run nt_8w DS2_GEMM_W4=0 python $GRAFT_REPO_ROOT/scripts/r5_gemm_one.py 32064 1024 6144
run tn_w4 DS2_GEMM_W4=1 python /tmp/tn_one.py
run tn_8w DS2_GEMM_W4=0 python /tmp/tn_one.py
cd $GRAFT_REPO_ROOT
python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for tag in ("nt_w4", "nt_8w", "tn_w4", "tn_8w"):
    print("==", tag)
    for p in "abc":
        f = glob.glob(f"{out}/**/{tag}_{p}_counter_collection.csv", recursive=True)
        t = glob.glob(f"{out}/**/{tag}_{p}_kernel_trace.csv", recursive=True)
        if not f:
            print("  pass", p, "no counter file:", open(f"{out}/log_{tag}_{p}.txt").read()[-300:].replace("\n", " | ")); continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f[0])):
            if "gemm_bf16" in r["Kernel_Name"]: agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur = collections.defaultdict(list)
        if t:
            for r in csv.DictReader(open(t[0])):
                if "gemm_bf16" in r["Kernel_Name"]: dur[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in agg.items():
            d = sorted(dur.get(k, [0]))
            print(f"  {k}  launches {len(d)} median {d[len(d) // 2]:.0f} us  " + "  ".join(f"{c} {sorted(x)[len(x) // 2]:.4g}" for c, x in sorted(v.items())))
PY
