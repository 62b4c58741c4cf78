#!/bin/bash
# LC-BLSTM (cfg3) timing + kernel trace on the GPU box.  Usage: devtools/measure_lc.sh <tag> [S]
TAG=${1:-lc}
S=${2:-32}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python3 $R/devtools/bench_lc.py $S 20 2>&1 | grep 'ms/step' | tee $O/${TAG}_bench.txt
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace -o lc -- python3 $R/devtools/bench_lc.py $S 5 > $O/${TAG}_trace.log 2>&1
python3 $R/devtools/prof_summary.py $O/${TAG}_trace/lc_results.db > $O/${TAG}_kernel_stats.txt 2>&1
head -45 $O/${TAG}_kernel_stats.txt
rm -rf $O/${TAG}_trace
