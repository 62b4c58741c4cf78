#!/bin/bash
mkdir -p gpurun_out/r3
L=gpurun_out/r3/run4_bench.log; : > $L
for d in 0 1200; do
  echo "=== HALF delay $d ns" >> $L
  ASLP_LSTM_HALF_DELAY_NS=$d SEQ_CENSUS=1 timeout 300 python devtools/bench_lc.py 32 50 2>&1 | grep -v "^LOG\|amdgpu.ids\|census" >> $L
done
cat $L
