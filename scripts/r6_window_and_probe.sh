#!/bin/bash
# round 6: (a) the data-parallel window priced with the RCCL stand-in, conv and serial schedules; (b) phase timeline of the persistent kernels
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/probe_hog.hip -o /tmp/libhog.so || exit 1
[ -n "$SKIP_CONV" ] || { DS2_DP_MODE=conv timeout 900 python scripts/r6_dp_window.py > gpurun_out/r06_dp_window_conv.txt 2> gpurun_out/r06_dp_window_conv.err; echo "conv rc=$?"; }
MASTER_PORT=29735 DS2_DP_MODE=serial timeout 900 python scripts/r6_dp_window.py > gpurun_out/r06_dp_window_serial.txt 2> gpurun_out/r06_dp_window_serial.err; echo "serial rc=$?"
tail -15 gpurun_out/r06_dp_window_conv.txt; tail -4 gpurun_out/r06_dp_window_conv.err
cd scripts && mkdir -p build && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../include -I../asr_amd/csrc -DDS2_RNN_TRACE -mllvm -amdgpu-kernarg-preload-count=9 -x hip probe_persist_timeline.hip ../asr_amd/csrc/api.cpp -o build/probe_persist_timeline 2>&1 | grep -E "error" | head
cd ..; timeout 300 scripts/build/probe_persist_timeline > gpurun_out/r06_probe_persist_timeline.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r06_probe_persist_timeline.txt | head -30
