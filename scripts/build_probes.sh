#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# builds the measurement probes (not part of the product library) into scripts/build/ (git-ignored, travels with gpurun)
cd "$(dirname "$0")"
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 probe_step_exchange.hip -o build/probe_step_exchange
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../include -I../asr_amd/csrc -DDS2_RNN_TRACE -mllvm -amdgpu-kernarg-preload-count=9 -x hip probe_rnn_timeline.hip ../asr_amd/csrc/api.cpp -o build/probe_rnn_timeline
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 --cuda-device-only probe_l2_kernel.hip -o build/probe_l2_kernel.bundle
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=build/probe_l2_kernel.bundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=build/probe_l2_kernel.hsaco
/opt/rocm/bin/hipcc -O2 -std=c++17 probe_l2_residency.cpp -o build/probe_l2_residency -L/opt/rocm/lib -lhsa-runtime64
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 probe_xcd_exchange.hip -o build/probe_xcd_exchange
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 probe_xcd_local.hip -o build/probe_xcd_local
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../include -I../asr_amd/csrc -DDS2_RNN_TRACE -mllvm -amdgpu-kernarg-preload-count=9 -x hip probe_persist_timeline.hip ../asr_amd/csrc/api.cpp -o build/probe_persist_timeline
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -Wno-unused-result probe_tr_read.hip -o build/probe_tr_read
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 probe_ksplit_reduce.hip -o build/probe_ksplit_reduce
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../include -I../asr_amd/csrc -x hip probe_conv1.hip ../asr_amd/csrc/api.cpp -o build/probe_conv1
