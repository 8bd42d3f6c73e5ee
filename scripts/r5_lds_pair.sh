#!/bin/bash
cd "$(dirname "$0")"
mkdir -p build /tmp/pl; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 probe_lds_pair.hip -o /tmp/pl/probe_lds_pair 2>/dev/null
/tmp/pl/probe_lds_pair
cd /tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d /tmp/pl/out -o p -- /tmp/pl/probe_lds_pair > /tmp/pl/log.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/pl/out/**/p_counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.OrderedDict()
for r in rows: agg.setdefault((r["Dispatch_Id"], r["Kernel_Name"][:40]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, k), c in agg.items(): print(d, k, {x: f"{v:.3g}" for x, v in c.items()})
PY
