#!/bin/bash
# round 3, GPU run 2: phase offset between the two chains of a CU (half-chain LSTM kernels), fast activations
mkdir -p gpurun_out/r3
L=gpurun_out/r3/run2_bench.log; : > $L
for d in 0 600 1000 1400 1800 2400; do
  echo "=== HALF delay $d ns" >> $L
  ASLP_LSTM_HALF_DELAY_NS=$d SEQ_TIMING=1 timeout 300 python devtools/bench_lc.py 32 100 2>&1 | grep -v "^LOG\|amdgpu.ids" >> $L
done
echo "=== HALF delay 1200 FAST_ACT=1" >> $L
ASLP_LSTM_FAST_ACT=1 SEQ_TIMING=1 timeout 300 python devtools/bench_lc.py 32 100 2>&1 | grep -v "^LOG\|amdgpu.ids" >> $L
echo "=== chains of 8" >> $L
ASLP_LSTM_HALF_CHAINS=0 timeout 300 python devtools/bench_lc.py 32 100 2>&1 | grep -v "^LOG\|amdgpu.ids" >> $L
echo "=== parity with FAST_ACT=1" >> $L
ASLP_LSTM_FAST_ACT=1 timeout 600 python -m pytest tests/test_rnn_gpu.py tests/test_cfg3_step_gpu.py -x -q -m gpu 2>&1 | tail -5 >> $L
cat $L
