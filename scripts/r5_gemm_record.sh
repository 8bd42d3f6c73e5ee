#!/bin/bash
export DS2_EXPERIMENTAL=1   # the A/B switches below are honoured only with this (asr_amd/engine.py::_tune, csrc/common.h::ds2_exp_getenv)
# round 5: the GEMM investigation in one same-box record -> gpurun_out/r05_gemm_ring_ab.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05_gemm_ring_ab.txt
{
echo "# scripts/r5_gemm_record.sh on one MI355X box.  N(0,1) bf16 operands, fp32 output + bias, median of 3 processes x 10 launches."
echo "# RING=0: production kernel (double-buffered 256x256x64, software-pipelined, persistent);  RING=1: four-slice ring, software-pipelined;"
echo "# RING=q: four-slice ring, ping-pong wave groups, v_mfma_f32_16x16x32_bf16, register-direct epilogue.  All three bit-identical."
echo "## tile walk in super-rows of 4 row tiles (DS2_GEMM_SR=4, the new default)"
DS2_GEMM_SR=4 RING_VARIANTS="0 1 q" python scripts/r5_ring.py time
echo "## row-major tile walk (DS2_GEMM_SR=0, rounds 1-4)"
DS2_GEMM_SR=0 RING_VARIANTS="0 q" python scripts/r5_ring.py time
echo "## timing ablations of RING=q on the fwd / dX shapes (WRONG RESULTS by construction): 1 no operand DMA behind the prologue, 2 no fragment reads,"
echo "## 3 neither (MFMA + barriers), 8 DMA re-reads one L2-resident 64 KB, 18 DMA stream + barriers only (no MFMA, no reads), 26 = 18 from the L2-resident 64 KB"
for d in 0 1 2 3 8 18 26; do echo "DS2_RING_DBG=$d"; DS2_GEMM_RING=q DS2_RING_DBG=$d python scripts/bench_gemm.py 2>&1 | grep -E "fwd|dXn"; done
echo "## scripts/probe_ingest.hip: what one CU takes in per second, by instruction form and source"
scripts/bin/probe_ingest
echo "## scripts/probe_mfma_power.hip: matrix pipe on random / zero operands by MFMA shape (power limit)"
scripts/bin/probe_mfma_power 0; scripts/bin/probe_mfma_power 1
} > $O 2>&1
tail -5 $O
