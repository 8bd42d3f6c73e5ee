#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
RING_VARIANTS="0 q" python scripts/r5_ring.py check | tail -8
for rep in 1 2; do for v in "DS2_GEMM_RING=0 DS2_GEMM_SR=0" "DS2_GEMM_RING=0 DS2_GEMM_SR=4" "DS2_GEMM_RING=q DS2_GEMM_SR=4" "DS2_GEMM_RING=1 DS2_GEMM_SR=4"; do echo "== $v"; env $v timeout 200 python scripts/bench_gemm.py 2>&1 | grep -E "fwd|dXn"; done; done
