#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "DS2_GEMM_RING=0" "DS2_GEMM_WAVES=pp" "DS2_GEMM_PERS=0" "DS2_GEMM_RING=0" "DS2_GEMM_WAVES=pp" "DS2_GEMM_PERS=0"; do echo "== $v"; env $v python scripts/bench_gemm_square.py 2>&1 | grep -E "^M=(8192|4096)"; done
