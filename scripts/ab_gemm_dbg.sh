export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
for d in ${DBGS:-0 32}; do for nt in 1 0; do echo "== DS2_GEMM_DBG=$d NT=$nt"; DS2_GEMM_NT=$nt DS2_GEMM_DBG=$d timeout 200 python scripts/bench_gemm.py 2>&1 | grep -E "fwd"; done; done
