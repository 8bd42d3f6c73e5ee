#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# same-box A/B of two builds of the library (asr_amd/lib/libds2hip.so vs libds2hip_b.so): the c3 bench's step time and its in-region recurrence
# times; TESTS=1 also runs the recurrence parity tests on the B build
cd "$(dirname "$0")/.."
for rep in 1 2; do
for lib in libds2hip.so libds2hip_b.so; do
  echo "== $lib (rep $rep)"
  DS2_LIB_PATH=$PWD/asr_amd/lib/$lib timeout 300 python bench.py --workload ${WL:-c3} --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; f,b=(r,r['second_kernel']) if 'fwd' in r['kernel'] else (r['second_kernel'],r)
print('ms_per_step', round(d['ms_per_step'],2), 'fwd us/step', round(f['us_per_time_step'],3), 'bwd us/step', round(b['us_per_time_step'],3), 'starved', d.get('persistent_starved_steps'))"
done; done
if [ -n "$TESTS" ]; then DS2_LIB_PATH=$PWD/asr_amd/lib/libds2hip_b.so timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -x -k "rnn_fwd_bwd or persistent or ksplit" 2>&1 | grep -v amdgpu.ids | tail -3; fi
