#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for v in "DS2_GEMM_RING=0" "DS2_GEMM_WAVES=pp" "DS2_GEMM_PERS=0" "DS2_GEMM_RING=0" "DS2_GEMM_WAVES=pp" "DS2_GEMM_PERS=0"; do echo "== $v"; env $v python scripts/bench_gemm_square.py 2>&1 | grep -E "^M=(8192|4096)"; done
