#!/bin/bash
# rocprofv3 kernel trace of the CompactFsmn / RowConvolution steps (cfg5 swaps): devtools/prof_temporal.sh <tag>
# output gpurun_out/<tag>_temporal_kernel_stats.txt (+ the bench lines of devtools/bench_temporal.py)
set -u
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python3 $R/devtools/bench_temporal.py > $O/${TAG}_temporal_bench.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_temporal_trace -o t -- python3 $R/devtools/bench_temporal.py > $O/${TAG}_temporal_trace.log 2>&1
python3 $R/devtools/prof_summary.py $O/${TAG}_temporal_trace/t_results.db > $O/${TAG}_temporal_kernel_stats.txt 2>&1
rm -rf $O/${TAG}_temporal_trace
grep -v amdgpu.ids $O/${TAG}_temporal_bench.txt | tail -3
head -20 $O/${TAG}_temporal_kernel_stats.txt
