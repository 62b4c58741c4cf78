#!/bin/bash
# kernel timeline of ONE steady-state step of the one-layer recurrent nets of devtools/bench_rnn.py: devtools/prof_rnn_timeline.sh; output gpurun_out/rnn_timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/rnn_tl
rocprofv3 --kernel-trace -d /tmp/rnn_tl -o rnn -- python3 $R/devtools/bench_rnn.py 32 60 > /tmp/rnn_tl.log 2>&1
python3 $R/devtools/prof_timeline.py $(find /tmp/rnn_tl -name "*.db" | head -1) xent_rows_kernel 12 > $R/gpurun_out/rnn_timeline.txt 2>&1
grep "ms/step" /tmp/rnn_tl.log
