#!/bin/bash
# same-box A/B of one engine switch: r6_ab.sh VAR "A B" [workload] [steps]   (interleaved, three repetitions)
export DS2_EXPERIMENTAL=1
cd "$(dirname "$0")/.."
VAR=$1; VALS=${2:-"0 1"}; WL=${3:-c3}; ST=${4:-10}
mkdir -p gpurun_out
for rep in 1 2 3; do
  for v in $VALS; do
    line=$(env $VAR=$v timeout 600 python bench.py --workload $WL --steps $ST --warmup 3 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1)
    echo "$VAR=$v rep $rep: $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.3f ms/step  loss %.6f  bwd %.3f us/step  fwd %.3f us/step  starved %d" % (d["ms_per_step"], d["loss"], r["us_per_time_step"], r["second_kernel"]["us_per_time_step"], d["persistent_starved_steps"]))')"
  done
done | tee -a gpurun_out/ab_$VAR.txt
