#!/bin/bash
# PMC comparison of GEMM tile/loop variants on one shape.  Usage: devtools/pmc_gemm.sh "<tiles>" [SHAPE]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
SHAPE=${2:-0,1,1024,2048,2048}
for T in $1; do
  for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_MISC"; do
    rm -rf /tmp/pmc_$T
    SHAPE=$SHAPE TILE=$T REPS=10 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmc_$T -o g -- python3 $R/devtools/one_gemm.py > /tmp/pmc_$T.log 2>&1 || tail -5 /tmp/pmc_$T.log
    echo "=== tile $T shape $SHAPE"
    python3 $R/devtools/prof_summary.py /tmp/pmc_$T/g_results.db 2>&1 | grep -A12 "^gemm_f32" | grep -v "^--"
  done
done
