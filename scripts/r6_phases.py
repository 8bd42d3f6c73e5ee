"""profiles/r06_probe_persist_timeline.txt (scripts/probe_persist_timeline.hip: the product's own recurrence kernels built with -DDS2_RNN_TRACE at the
c3 layer shape) -> profiles/r06_persist_phases.json: per kernel the mean over the 8 waves of every phase of a time step, and the LATENCY FLOOR of
the step = the phases that are waiting, not working (exchange: publish -> visible -> gathered; the workgroup barrier; the loop-top store drain) —
what bench.py reports as roofline.latency_floor_us_per_time_step beside the MFMA fraction."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "profiles", "r06_probe_persist_timeline.txt")
out = {"source": "profiles/r06_probe_persist_timeline.txt", "shape": "GRU H=1024 B=64 T=501 bf16 training mode (c3 layer)", "kernels": {}}
LAT = {"fwd": ("publish -> loop top", "gather (poll)", "barrier"),
       "bwd": ("store issue -> loop top", "gather (poll) + reduce-scatter", "dGh -> LDS + barrier")}
cur = None
for line in open(src):
    m = re.match(r"(FORWARD|BACKWARD): ([0-9.]+) us/step", line)
    if m:
        cur = "fwd" if m.group(1) == "FORWARD" else "bwd"
        out["kernels"][cur] = {"us_per_time_step_traced_build": float(m.group(2)), "phases_us": {}}
        continue
    m = re.match(r"  (.+?)\s{2,}((?:\s*[0-9.]+){8})\s+us \(waves", line)
    if m and cur:
        vals = [float(v) for v in m.group(2).split()]
        if m.group(1).strip() != "-":
            out["kernels"][cur]["phases_us"][m.group(1).strip()] = round(sum(vals) / len(vals), 3)
for k, d in out["kernels"].items():
    d["latency_phases"] = list(LAT[k])
    d["latency_floor_us"] = round(sum(d["phases_us"][p] for p in LAT[k]), 3)
json.dump(out, open(os.path.join(ROOT, "profiles", "r06_persist_phases.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
