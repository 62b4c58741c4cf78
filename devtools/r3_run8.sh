#!/bin/bash
mkdir -p gpurun_out/r3
L=gpurun_out/r3/run8.log; : > $L
echo "=== parity (dual forward default)" >> $L
timeout 900 python -m pytest tests/test_rnn_gpu.py tests/test_cfg3_step_gpu.py tests/test_gradcheck_gpu.py tests/test_ref_blas_gpu.py -x -q -m gpu 2>&1 | tail -8 >> $L
for v in 1 0 1 0; do
  echo "=== DUAL=$v" >> $L
  ASLP_LSTM_DUAL=$v SEQ_TIMING=1 timeout 300 python devtools/bench_lc.py 32 100 2>&1 | grep -v "^LOG\|amdgpu.ids" >> $L
done
cat $L
