#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# K-split backward: MFMA / publish schedule variants (DS2_KS_PIPE) through the timeline probe (builds rnn.hip itself)
cd "$(dirname "$0")"
mkdir -p build
for v in ${@:-0 1 2 3 4}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../include -I../asr_amd/csrc -DDS2_RNN_TRACE -DDS2_KS_PIPE=$v -mllvm -amdgpu-kernarg-preload-count=9 -x hip probe_persist_timeline.hip ../asr_amd/csrc/api.cpp -o build/probe_ks_pipe_$v 2>&1 | grep error
done
