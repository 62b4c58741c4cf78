#!/bin/bash
# kernel timeline of ONE steady-state cfg2 step (bench.py's workload): devtools/prof_cfg2_timeline.sh; output gpurun_out/cfg2_timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/c2_tl
rocprofv3 --kernel-trace -d /tmp/c2_tl -o c2 -- python3 $R/bench.py --steps 20 --warmup 10 --headline-only ${1:-} > /tmp/c2_tl.log 2>&1
python3 $R/devtools/prof_timeline.py $(find /tmp/c2_tl -name "*.db" | head -1) xent_rows_kernel 12 > $O/cfg2_timeline.txt 2>&1
tail -3 $O/cfg2_timeline.txt
