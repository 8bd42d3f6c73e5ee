export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm" 2>&1 | tail -5
timeout 200 python scripts/bench_gemm_square.py 2>&1 | grep TF; timeout 200 python scripts/bench_gemm.py 2>&1 | grep -E "TF"
echo "== 8 waves forced"; DS2_GEMM_WAVES=8 timeout 200 python scripts/bench_gemm.py 2>&1 | grep -E "fwd"
