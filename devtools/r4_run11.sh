#!/bin/bash
mkdir -p gpurun_out/r4
L=gpurun_out/r4/run11.log; : > $L
timeout 1500 python -m pytest tests/test_rnn_gpu.py tests/test_cfg3_step_gpu.py tests/test_gradcheck_gpu.py tests/test_ref_blas_gpu.py tests/test_ab_switches_gpu.py -x -q -m gpu 2>&1 | tail -15 >> $L
for v in 1 0 1 0; do
  echo "=== ASLP_LSTM_PLANES=$v" >> $L
  ASLP_LSTM_PLANES=$v timeout 300 python devtools/bench_lc.py 32 100 2>&1 | grep -v "^LOG\|amdgpu.ids" >> $L
done
cat $L
