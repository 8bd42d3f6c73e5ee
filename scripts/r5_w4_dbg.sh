#!/bin/bash
export DS2_EXPERIMENTAL=1
# timing ablations of the four-wave NT GEMM (WRONG RESULTS by construction): 1 no steady-state DMA, 2 L2-resident operand stream, 4 no MFMA, 8 no stores
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for rep in 1 2; do for d in ${W4_DBGS:-0 1 2 4 5 6 8 10}; do echo "== RING=w DS2_W4_DBG=$d"; DS2_GEMM_RING=w DS2_W4_DBG=$d timeout 200 python scripts/bench_gemm.py 2>&1 | grep -E "fwd|dXn"; done; done
