#!/bin/bash
export DS2_EXPERIMENTAL=1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/r5_tn_w4.py check 2>&1 | grep -v amdgpu.ids | tail -8
timeout 900 python scripts/r5_tn_w4.py time 2>&1 | grep -v amdgpu.ids | tail -8
