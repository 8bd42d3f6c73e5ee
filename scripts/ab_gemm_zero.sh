for z in "" 1; do for d in 0 1 ; do echo "== zero=[$z] DBG=$d"; GEMM_ZERO=$z DS2_GEMM_DBG=$d timeout 200 python scripts/bench_gemm_square.py 2>&1 | grep TF | head -2; done; done
