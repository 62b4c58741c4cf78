#!/bin/bash
# Where do the waves of one split-fp16 product wait?  devtools/pmc_s16.sh "tA,tB,M,N,K" -> per-dispatch SQ wait / LDS / MFMA counters (three passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $R
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/og
  SHAPE=$1 TILE=0 REPS=10 rocprofv3 --kernel-trace --pmc $set -d /tmp/og -o og -- python3 devtools/one_gemm.py > /tmp/og.log 2>&1
  python3 devtools/prof_summary.py $(find /tmp/og -name "*.db" | head -1) 2>&1 | grep -A5 "^gemm_s16.*dispatches" | cut -c1-70,200-260 | grep -v '^--'
done
