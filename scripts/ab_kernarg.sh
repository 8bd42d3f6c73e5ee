export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
for cfg in "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; do
  echo "== [$cfg]"; env $cfg timeout 300 python bench.py --workload c3 --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
  env $cfg timeout 120 scripts/build/probe_rnn_timeline 2>&1 | grep -E "period|entry|MFMAs"
done
