#!/bin/bash
# kernel timeline of ONE steady-state cfg1 step (5x2048 sigmoid DNN, minibatch 256): devtools/prof_cfg1_timeline.sh; output gpurun_out/cfg1_timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/c1_tl
rocprofv3 --kernel-trace -d /tmp/c1_tl -o c1 -- python3 $R/devtools/bench_cfg1.py 256 60 > /tmp/c1_tl.log 2>&1
python3 $R/devtools/prof_timeline.py $(find /tmp/c1_tl -name "*.db" | head -1) xent_rows_kernel 40 > $O/cfg1_timeline.txt 2>&1
tail -3 $O/cfg1_timeline.txt
