#!/bin/bash
# Runs on the GPU box (via gpurun): bench line, rocprofv3 kernel trace + stats, and two separate PMC
# passes (FETCH_SIZE, WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# The profiler passes run with --no-update-overlap: every kernel alone on the chip, so the per-kernel averages are the
# ones bench.py's roofline pass measures (the timed `value` pass overlaps the weight-gradient GEMMs on a side stream).
# Usage: devtools/measure.sh <tag>      outputs under gpurun_out/<tag>_*
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python3 $R/bench.py --steps 500 --warmup 50 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -c 3000 $O/${TAG}_bench.json
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --headline-only --no-update-overlap > $O/${TAG}_trace.log 2>&1
python3 $R/devtools/prof_summary.py $O/${TAG}_trace/bench_results.db > $O/${TAG}_kernel_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/${TAG}_pmc_fetch -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --headline-only --no-gemm-profile --no-update-overlap > $O/${TAG}_pmc_fetch.log 2>&1
python3 $R/devtools/prof_summary.py $O/${TAG}_pmc_fetch/bench_results.db > $O/${TAG}_pmc_fetch.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/${TAG}_pmc_write -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --headline-only --no-gemm-profile --no-update-overlap > $O/${TAG}_pmc_write.log 2>&1
python3 $R/devtools/prof_summary.py $O/${TAG}_pmc_write/bench_results.db > $O/${TAG}_pmc_write.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/${TAG}_pmc_mfma -o bench -- python3 $R/bench.py --steps 4 --warmup 2 --headline-only --no-gemm-profile --no-update-overlap > $O/${TAG}_pmc_mfma.log 2>&1
python3 $R/devtools/prof_summary.py $O/${TAG}_pmc_mfma/bench_results.db > $O/${TAG}_pmc_mfma.txt 2>&1
head -40 $O/${TAG}_kernel_stats.txt
rm -rf $O/${TAG}_trace $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_mfma   # keep the text summaries only (dbs are large)
