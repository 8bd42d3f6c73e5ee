#!/bin/bash
cd "$(dirname "$0")"; mkdir -p /tmp/mg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 probe_memset_gap.hip -o /tmp/mg/probe 2>/dev/null && /tmp/mg/probe
