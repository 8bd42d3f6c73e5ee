cd "$(dirname "$0")/.."
N=${N:-40}
echo "== default"; timeout 600 python scripts/det_check.py $N 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -4
echo "== DS2_GEMM_TILE=128"; DS2_GEMM_TILE=128 timeout 600 python scripts/det_check.py $N 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -4
echo "== DS2_GEMM_WAVES=8"; DS2_GEMM_WAVES=8 timeout 600 python scripts/det_check.py $N 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -4
echo "== DS2_GEMM_WAVES=16"; DS2_GEMM_WAVES=16 timeout 600 python scripts/det_check.py $N 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -4
echo "== DS2_GEMM_WAVES=pp"; DS2_GEMM_WAVES=pp timeout 600 python scripts/det_check.py $N 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -4
