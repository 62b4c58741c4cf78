#!/bin/bash
# MFMA-busy counters of the LC-BLSTM (cfg3) step's kernels: devtools/pmc_lc.sh <tag>; output gpurun_out/<tag>_lc_pmc_mfma.txt
# (separate pass with --kernel-trace only, as MI355X_MICROARCH.md prescribes)
TAG=${1:-lc}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/lc_pmc
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/lc_pmc -o lc -- python3 $R/devtools/bench_lc.py 32 3 > /tmp/lc_pmc.log 2>&1
python3 $R/devtools/prof_summary.py $(find /tmp/lc_pmc -name "*.db" | head -1) > $O/${TAG}_lc_pmc_mfma.txt 2>&1
grep -n "lstm_seq" -A3 $O/${TAG}_lc_pmc_mfma.txt | tail -12
