#!/bin/bash
# the two L2 (TCC) passes on one product shape: devtools/r6_pmc_tcc.sh "tA,tB,M,N,K" -> stdout
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd $R
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/og
  SHAPE=$1 TILE=0 REPS=10 timeout 150 rocprofv3 --kernel-trace --pmc $set -d /tmp/og -o og -- python3 devtools/one_gemm.py > /tmp/og.log 2>&1
  echo "## pass: $set"
  python3 devtools/prof_summary.py $(find /tmp/og -name "*.db" | head -1) 2>&1 | grep -A6 "^gemm_s16.*dispatches" | cut -c1-110
done
