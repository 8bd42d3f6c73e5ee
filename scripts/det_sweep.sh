export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
cd "$(dirname "$0")/.."
N=${N:-40}
for cfg in "c3 bf16" "c3 fp32" "c2 bf16" "c4 bf16" "c4 fp32" "c5 bf16" "c1 bf16"; do
  set -- $cfg
  echo "== $1 $2"; WL=$1 PREC=$2 timeout 600 python scripts/det_check.py $N 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -3
done
