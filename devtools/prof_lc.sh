#!/bin/bash
# rocprofv3 kernel trace of the LC-BLSTM (cfg3) step: devtools/prof_lc.sh <tag> [S] [steps]; output gpurun_out/<tag>_lc_kernel_stats.txt
set -u
TAG=${1:-lc}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python3 $R/devtools/bench_lc.py ${2:-32} ${3:-20} > $O/${TAG}_lc_bench.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_lc_trace -o lc -- python3 $R/devtools/bench_lc.py ${2:-32} 5 > $O/${TAG}_lc_trace.log 2>&1
python3 $R/devtools/prof_summary.py $O/${TAG}_lc_trace/lc_results.db > $O/${TAG}_lc_kernel_stats.txt 2>&1
rm -rf $O/${TAG}_lc_trace
grep -v amdgpu.ids $O/${TAG}_lc_bench.txt | tail -2
head -24 $O/${TAG}_lc_kernel_stats.txt
