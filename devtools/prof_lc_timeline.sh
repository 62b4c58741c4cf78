#!/bin/bash
# kernel timeline of ONE steady-state LC-BLSTM (cfg3) step (start offset, duration, idle gap before each kernel):
# devtools/prof_lc_timeline.sh [S]; output gpurun_out/lc_timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/lc_tl
rocprofv3 --kernel-trace -d /tmp/lc_tl -o lc -- python3 $R/devtools/bench_lc.py ${1:-32} 5 > /tmp/lc_tl.log 2>&1
python3 $R/devtools/prof_timeline.py $(find /tmp/lc_tl -name "*.db" | head -1) xent_rows_kernel 4 > $O/lc_timeline.txt 2>&1
tail -3 $O/lc_timeline.txt
