#!/bin/bash
export DS2_EXPERIMENTAL=1
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for rep in 1 2; do for d in 18 22 82 146 26; do echo "== RING=q DS2_RING_DBG=$d"; DS2_GEMM_RING=q DS2_RING_DBG=$d timeout 200 python scripts/bench_gemm.py 2>&1 | grep -E "fwd|dXn"; done; done
