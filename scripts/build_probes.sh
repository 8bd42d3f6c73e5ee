#!/bin/bash
# builds the measurement probes (not part of the product library) into scripts/build/ (git-ignored, travels with gpurun)
cd "$(dirname "$0")"
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 probe_step_exchange.hip -o build/probe_step_exchange
