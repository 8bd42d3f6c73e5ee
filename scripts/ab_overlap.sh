export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
for cfg in "" "DS2_OVERLAP=1" "DS2_GEMM_TILE=128" "DS2_OVERLAP=1 DS2_GEMM_TILE=128"; do
  echo "== [$cfg]"; env $cfg timeout 300 python bench.py --workload c3 --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
