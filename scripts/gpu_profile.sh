#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# rocprofv3 kernel trace + stats of the bench command (summary copied to profiles/ by the caller)
cd "$(dirname "$0")/.."
WL=${1:-c2}; TAG=${2:-r01}
mkdir -p gpurun_out/prof_$TAG
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads $BENCHARGS > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/bench.log 2>&1
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$TAG -name "*stats*" | head
f=$(find gpurun_out/prof_$TAG -name "*kernel_stats*.csv" | head -1)
[ -n "$f" ] && head -30 "$f"
find gpurun_out/prof_$TAG -name "*kernel_trace*.csv" -size +20M -delete
tail -2 gpurun_out/prof_$TAG/bench.log
