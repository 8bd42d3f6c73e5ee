export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
for z in "" 1; do for d in 0 1 ; do echo "== zero=[$z] DBG=$d"; GEMM_ZERO=$z DS2_GEMM_DBG=$d timeout 200 python scripts/bench_gemm_square.py 2>&1 | grep TF | head -2; done; done
