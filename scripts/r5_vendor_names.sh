#!/bin/bash
# rocprofv3 kernel trace of the vendor library (torch.matmul) on the step's GEMM shapes: which kernels does it pick (tile, wave layout)?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/vendor
export TMPDIR=/tmp
cat > /tmp/vn.py <<'PY'
import torch
M = 32064
for name, (m, n, k) in {"fwd": (M, 6144, 1024), "dX": (M, 1024, 6144), "dWih": (6144, 1024, M)}.items():
    A = torch.randn(m, k, device="cuda").bfloat16(); B = torch.randn(n, k, device="cuda").bfloat16()
    for _ in range(5): C = torch.matmul(A, B.t())
    torch.cuda.synchronize()
PY
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/vendor -o vn -- python /tmp/vn.py > $GRAFT_REPO_ROOT/gpurun_out/vendor/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/vendor -name "*kernel_stats*.csv" | head -1); cat "$f" | cut -c1-600
find gpurun_out/vendor -name "*kernel_trace*.csv" -size +5M -delete
