#!/bin/bash
# rocprofv3 kernel trace of the cfg2 bench step (per-kernel, weight-gradient GEMMs in line): devtools/prof_cfg2.sh <tag>
set -u
TAG=${1:-c2}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --headline-only --no-update-overlap > $O/${TAG}_trace.log 2>&1
python3 $R/devtools/prof_summary.py $O/${TAG}_trace/bench_results.db > $O/${TAG}_kernel_stats.txt 2>&1
rm -rf $O/${TAG}_trace
head -24 $O/${TAG}_kernel_stats.txt
