#!/bin/bash
# Kernel trace of one aslp-nnet-train-frame run (cfg2 from archives, devtools/bench_tool_e2e.py writes the inputs) and the
# GPU idle gaps in it.  Usage (on the GPU box): devtools/prof_tool_e2e.sh [frames]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FRAMES=${1:-600000}
python "$ROOT/devtools/bench_tool_e2e.py" /tmp/e2e "$FRAMES"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tool
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_tool -o tool -- "$ROOT/kaldi-aslp_amd/bin/aslp-nnet-train-frame" --print-args=false \
  --learn-rate=0.00001 --minibatch-size=1024 --randomizer-size=32768 ark:/tmp/e2e/feats.ark ark:/tmp/e2e/post.ark /tmp/e2e/nnet.init /tmp/e2e/nnet.out 2>&1 | grep -E "fps|AvgLoss" || true
DB=$(find /tmp/prof_tool -name "*.db" | head -1)
python "$ROOT/devtools/prof_tool_gaps.py" "$DB" 150
python "$ROOT/devtools/prof_tool_refill.py" "$DB" 10 > "$ROOT/gpurun_out/tool_refill.txt" 2>&1 || true
python "$ROOT/devtools/prof_tool_head.py" "$DB" 120 80 > "$ROOT/gpurun_out/tool_head_tail.txt" 2>&1 || true
python "$ROOT/devtools/prof_timeline.py" "$DB" xent_rows_kernel 200 > "$ROOT/gpurun_out/tool_step_timeline.txt" 2>&1 || true
tail -1 "$ROOT/gpurun_out/tool_step_timeline.txt"
