#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/run12.log; : > $L
for v in 3 2 1 0 3 2 1 0; do
  echo "=== ASLP_LSTM_PLANES=$v" >> $L
  ASLP_LSTM_PLANES=$v timeout 300 python devtools/bench_lc.py 32 100 2>&1 | grep "ms/step" >> $L
done
for t in 308 311; do
  echo "=== PLANES=3 TILE=$t" >> $L
  ASLP_GEMM_SPLIT_F16_TILE=$t timeout 300 python devtools/bench_lc.py 32 100 2>&1 | grep "ms/step" >> $L
done
cat $L
