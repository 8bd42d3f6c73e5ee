#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# four-wave NT GEMM (gemm_nt_w4.h, DS2_GEMM_RING=w) against the production kernel: correctness on the step's shapes + edge shapes, then speed
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
export RING_VARIANTS="${RING_VARIANTS:-0 w}"
timeout 600 python scripts/r5_ring.py check > gpurun_out/r5_w4_check.log 2>&1; echo "check rc=$?"; tail -12 gpurun_out/r5_w4_check.log
timeout 900 python scripts/r5_ring.py time > gpurun_out/r5_w4_time.log 2>&1; echo "time rc=$?"; tail -12 gpurun_out/r5_w4_time.log
