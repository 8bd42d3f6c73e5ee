"""Reduce the counter_collection CSVs of separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over scripts/pmc_rnn.py to one
JSON file: per persistent recurrence kernel, HBM-side bytes per launch and per time step (FETCH_SIZE doubled: the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md §HBM; counters report KB).  bench.py reads the committed copy (profiles/*.json) for `roofline.traffic`.

    python scripts/pmc_summarize.py <dir with pmc_FETCH_SIZE_counter_collection.csv, pmc_WRITE_SIZE_...> <T> <out.json> [shape note]
"""
import csv, hashlib, json, os, re, sys


def kernel_source_sha256():
    """sha256 over the sources the recurrence kernels are built from: rnn.hip, the headers it includes and the Makefile (the same function is
    in bench.py, which refuses a summary whose hash differs from the tree it runs in: the counters then belong to another build of the kernels)"""
    CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "asr_amd", "csrc")
    import hashlib, re
    h = hashlib.sha256()
    seen, todo = [], ["rnn.hip"]
    while todo:                                        # rnn.hip and every local header it (transitively) includes, in discovery order
        name = todo.pop(0)
        if name in seen:
            continue
        seen.append(name)
        src = open(os.path.join(CSRC, name), "rb").read()
        h.update(src)
        todo += [m for m in re.findall(r'#include "([^"/]+)"', src.decode("utf-8", "replace")) if os.path.exists(os.path.join(CSRC, m))]
    h.update(open(os.path.join(CSRC, "Makefile"), "rb").read())
    return h.hexdigest()


d, T, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
note = sys.argv[4] if len(sys.argv) > 4 else ""
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    path = os.path.join(d, f"pmc_{ctr}_counter_collection.csv")
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "persistent_kernel" not in name and "_step_kernel" not in name and "ksplit_kernel" not in name:
                continue
            m = re.search(r"(rnn_\w+_kernel)(<[^>]*>)?", name)
            key, short = m.group(1), m.group(0)
            e = res.setdefault(key, {"kernel": short, "launches": {}, "grid": int(row["Grid_Size"]), "vgpr": int(row["VGPR_Count"])})
            e["launches"].setdefault(ctr, []).append(float(row["Counter_Value"]))
summary = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over scripts/pmc_rnn.py", "time_steps_per_launch": T, "shape": note,
           "corrections": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B); counters in KB -> bytes x1024",
           "kernel_source_sha256": kernel_source_sha256(), "kernels": {}}
for key, e in res.items():
    f = e["launches"].get("FETCH_SIZE", [])
    w = e["launches"].get("WRITE_SIZE", [])
    persistent = "persistent" in key or "ksplit" in key          # one launch covers all T time steps
    n = max(len(f), len(w), 1)
    fetch_b = (sum(f) / max(len(f), 1)) * 1024 * 2
    write_b = (sum(w) / max(len(w), 1)) * 1024
    per_launch = fetch_b + write_b
    summary["kernels"][key] = {"kernel": e["kernel"], "launches_seen": n, "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
                               "hbm_bytes_per_launch": per_launch, "hbm_bytes_per_time_step": per_launch / T if persistent else per_launch,
                               "grid_threads": e["grid"], "vgprs": e["vgpr"]}
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps(summary, indent=1))
