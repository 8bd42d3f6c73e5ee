#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# build rnn.hip with different waves-per-block and run the ablation (GPU box)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for nw in 4 8 16; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DDS2_RNN_NW=$nw -x hip -c asr_amd/csrc/rnn.hip -o asr_amd/csrc/build/rnn.hip.o 2>&1 | grep -E "error" 
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC asr_amd/csrc/build/*.o -o asr_amd/lib/libds2hip.so
  echo "=== NW=$nw"
  timeout 300 python scripts/ablate_rnn.py 2>&1 | grep -E "c3 .*bf16|c2 .*fp32"
done
