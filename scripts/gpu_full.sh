#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# whole GPU suite + smoke + default bench (what the driver runs at round end)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/full_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/full_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/default_bench.log 2>&1; echo "bench rc=$?"; grep -v amdgpu.ids gpurun_out/default_bench.log | tail -2 | cut -c1-1500
