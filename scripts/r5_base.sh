#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# round 5 baseline: GEMM micro-benchmarks + c3 bench on one box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
python scripts/bench_gemm.py > gpurun_out/r5_gemm_base.log 2>&1; tail -12 gpurun_out/r5_gemm_base.log
python scripts/bench_gemm_tn.py > gpurun_out/r5_gemm_tn_base.log 2>&1; tail -12 gpurun_out/r5_gemm_tn_base.log
timeout 900 python bench.py --workload c3 --steps 8 --warmup 3 --breakdown --no-cpu-baseline --no-other-workloads > gpurun_out/r5_bench_c3_base.log 2>&1; echo "bench rc=$?"
grep -vE "amdgpu.ids" gpurun_out/r5_bench_c3_base.log | tail -12 | cut -c1-900
