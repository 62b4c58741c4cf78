R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/st_tl
rocprofv3 --kernel-trace -d /tmp/st_tl -o st -- python3 $R/devtools/bench_lstm_stack.py 32 60 5 > /tmp/st_tl.log 2>&1
python3 $R/devtools/prof_timeline.py $(find /tmp/st_tl -name "*.db" | head -1) xent_rows_kernel 10 > $R/gpurun_out/stack_timeline.txt 2>&1
