#!/bin/bash
export DS2_EXPERIMENTAL=1
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in ${RING_VARIANTS:-w wp}; do DS2_GEMM_RING=$v timeout 300 python scripts/r5_w4_diag.py 2>&1 | grep -v amdgpu.ids; done
